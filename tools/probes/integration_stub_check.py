"""Runs the ctypes stub of INTEGRATION.md section 2 verbatim against a direct fp32 computation (GPU box)."""
import sys; sys.path.insert(0, '/root/repo')
import ctypes as C, torch, os
lib = C.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scaleprotoseg_amd/libspx_hip.so"))
class SpxPlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_prototypes", "num_classes", "num_scales", "channels_per_scale",
                                         "kc", "npb", "ncb", "npanels")] + \
               [(n, C.c_int32 * 64) for n in ("panel_ch0", "panel_p0", "panel_np")]
def _ptr(t): return C.c_void_p(t.data_ptr()) if t is not None else None
class M: pass
self = M(); self.num_prototypes = 40; self.num_scales = 4; self.epsilon = 1e-4
self.scale_num_prototypes = {s: (10 * s, 10 * s + 10) for s in range(4)}
self.prototype_vectors = torch.rand(40, 16, 1, 1, device="cuda")
x = torch.rand(2, 64, 9, 11, device="cuda")
B, Cx, H, W = x.shape
P, S = self.num_prototypes, self.num_scales
lo = (C.c_int32 * S)(*[self.scale_num_prototypes[s][0] for s in range(S)])
hi = (C.c_int32 * S)(*[self.scale_num_prototypes[s][1] for s in range(S)])
plan = SpxPlan()
assert lib.spx_make_plan(P, 1, S, Cx // S, lo, hi, C.byref(plan)) == 0
lib.spx_packed_bank_bytes.restype = lib.spx_packed_p2_bytes.restype = C.c_size_t
bank = self.prototype_vectors.detach().reshape(P, -1).float().contiguous()
pk = torch.empty(lib.spx_packed_bank_bytes(C.byref(plan)), dtype=torch.uint8, device=x.device)
p2 = torch.empty(lib.spx_packed_p2_bytes(C.byref(plan)) // 4, dtype=torch.float32, device=x.device)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
assert lib.spx_pack_bank(C.byref(plan), _ptr(bank), _ptr(pk), None, _ptr(p2), s) == 0
d = torch.empty(B, P, H, W, dtype=torch.float32, device=x.device)
rc = lib.spx_dist_fwd(C.byref(plan), _ptr(x.contiguous()), 1, B, H * W, _ptr(pk), _ptr(p2), None, _ptr(d), None, None, C.c_float(self.epsilon), 1, s)
assert rc == 0
torch.cuda.synchronize()
xb = x.bfloat16().float(); pb = bank.bfloat16().float()
ref = torch.stack([((xb[:, 16*(p//10):16*(p//10)+16] - pb[p].view(1, 16, 1, 1)) ** 2).sum(1) for p in range(P)], 1)
print("INTEGRATION.md stub max err", (d - ref).abs().max().item())
