"""Extended randomised parity run (not collected by pytest): the bodies of test_random_configurations and
test_random_gather_and_tail over many more seeds than the suite carries.  Usage (GPU box):
    python tests/fuzz_extended.py FIRST LAST [LOGFILE]
    python tests/fuzz_extended.py kld|push|pushfused FIRST LAST [LOGFILE]
Prints one line per failing seed and a summary; progress goes to LOGFILE every 10 seeds."""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import test_gpu_parity as T  # noqa: E402


def kld_case(seed):
    """KLDLoss kernels (all slot-count instances, both walks, ragged grids, void / absent classes, tiny segments) against
    the fp64 torch form of the same loss."""
    import numpy as np
    import torch
    import scaleprotoseg_amd as spx

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9000 + seed)
    K = int(rng.integers(1, 24))
    S = int(rng.choice([1, 2, 4]))
    r = int(rng.integers(1, 5))                     # prototypes per (class, scale): J = S * r <= 16
    P = K * S * r
    B = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 120)), int(rng.integers(1, 400))
    ident = T.O.default_class_identity(P, K, S)
    ranges = T.O.default_scale_ranges(P, S)
    lay = T._layout(P, K, S, 16, ranges)
    keys, J, table = spx.class_gather_table(lay, ident, dev)
    gen = torch.Generator(device=dev).manual_seed(seed)
    ph, pw = int(rng.choice([1, 3, 16, 40])), int(rng.choice([1, 5, 64, 100]))
    t = torch.randint(0, K + 1, (B, -(-H // ph), -(-W // pw)), device=dev, generator=gen)
    t = t.repeat_interleave(ph, 1).repeat_interleave(pw, 2)[:, :H, :W].contiguous()
    if K > 2:
        t[t == 2] = 0                               # class 1 absent
    base = torch.rand(B, J, H * W, device=dev, generator=gen) * float(rng.choice([1.0, 30.0, 200.0]))
    lab = (t.reshape(B, -1) - 1).int()
    grid = (H, W) if seed % 3 else (1, H * W)
    v1 = base.clone().requires_grad_(True)
    l1 = spx.KLDLoss(ident, S, ranges)(spx.ClassDistances(v1, lab, table, grid), t)
    l1.backward()
    v2 = base.double().clone().requires_grad_(True)
    from oracle import loss_oracle as LO

    l2 = LO.kld_loss(spx.KLDLoss(ident, S, ranges), spx.ClassDistances(v2, lab, table, (H, W)), t)
    if v2.grad is None and l2.requires_grad:
        l2.backward()
    torch.cuda.synchronize()
    tag = f"B{B} K{K} S{S} r{r} J{J} {H}x{W} patch {ph}x{pw} grid {grid}"
    assert abs(l1.item() - l2.item()) <= 2e-5 * max(1.0, abs(l2.item())), (tag, l1.item(), l2.item())
    if v2.grad is not None:
        s = v2.grad.abs().max().item()
        e = (v1.grad.double() - v2.grad).abs().max().item()
        assert e <= 2e-4 * s + 1e-12, (tag, e, s)


def push_case(seed):
    """Push argmin (vector and scalar kernels) on random shapes, label ranges, void classes and tie-heavy maps: bit-exact."""
    import numpy as np
    import torch
    from scaleprotoseg_amd.functional import push_masked_argmin

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7000 + seed)
    B, P, K = int(rng.integers(1, 4)), int(rng.integers(1, 70)), int(rng.integers(1, 25))
    H, W = int(rng.integers(1, 90)), int(rng.integers(1, 200))
    if seed % 2:
        W = 4 * max(1, W // 4)                      # the 4-pixel x 8-prototype kernel
    void = None if seed % 3 == 0 else int(rng.integers(0, K + 1))
    g = torch.Generator().manual_seed(seed)
    q = float(rng.choice([1.0, 8.0, 1024.0]))
    dist = torch.floor(torch.rand(B, P, H, W, generator=g) * 64 * q) / q
    target = torch.randint(0, K if void is None else K + 1, (B, H, W), generator=g)
    ident = (torch.rand(P, K, generator=g) < 1.5 / K).float()      # 0, 1 or several classes per prototype
    ref_idx, ref_val = T.O.push_masked_argmin(dist, target, ident, K, void_class=void)
    idx, val = push_masked_argmin(dist.to(dev), target.to(dev), ident, void_class=void)
    tag = f"B{B} P{P} K{K} {H}x{W} void {void}"
    assert torch.equal(idx.cpu(), ref_idx), tag
    assert torch.equal(val.cpu(), ref_val), tag


def push_fused_case(seed):
    """Fused push (spx_dist_push_min: the minimum taken inside the distance kernel) on random banks, grids, label ranges and
    void classes against the two-step push on the map the same kernel writes, and the oracle's push on that map: bit-exact."""
    import numpy as np
    import torch
    from scaleprotoseg_amd.functional import proto_head_forward, push_masked_argmin, push_min_from_features

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7500 + seed)
    S = int(rng.choice([1, 2, 4]))
    Cs = int(rng.choice([16, 32, 48, 64, 128, 256]))
    K = int(rng.integers(1, 25))
    P = S * K * int(rng.integers(1, 5))
    if rng.random() < 0.2:
        P = S * int(rng.integers(1, 500))              # prototype counts that are no multiple of the classes (several panels)
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 70)), int(rng.integers(1, 140))
    void = None if seed % 3 == 0 else int(rng.integers(0, K + 1))
    xdt = torch.bfloat16 if seed % 2 else torch.float32
    g = torch.Generator().manual_seed(seed)
    conv = T.O.bf16_representable(torch.sigmoid(torch.randn(B, S * Cs, H, W, generator=g)))
    if H > 2:
        conv[:, :, 1] = conv[:, :, 0]                  # a repeated row of pixels: exact ties
    bank = T.O.bf16_representable(torch.rand(P, Cs, 1, 1, generator=g))
    ranges = T.O.default_scale_ranges(P, S)
    ident = torch.zeros(P, K)
    cls = torch.randint(0, K, (P,), generator=g)
    ident[torch.arange(P), cls] = 1.0
    ident[torch.rand(P, generator=g) < 0.1] = 0.0      # prototypes of no class
    target = torch.randint(0, K if void is None else K + 1, (B, H, W), generator=g)
    lay = T._layout(P, 1, S, Cs, ranges)
    x = conv.to(dev, xdt)
    _, dist, _ = proto_head_forward(x, bank.to(dev), None, lay)
    idx2, val2 = push_masked_argmin(dist, target.to(dev), ident, void_class=void)
    idx, val = push_min_from_features(x, bank.to(dev), lay, target.to(dev), ident, void_class=void)
    ref_idx, ref_val = T.O.push_masked_argmin(dist.cpu(), target, ident, K, void_class=void)
    tag = f"B{B} S{S} Cs{Cs} P{P} K{K} {H}x{W} void {void} {xdt}"
    assert torch.equal(idx, idx2) and torch.equal(val, val2), "fused vs two-step: " + tag
    assert torch.equal(idx.cpu(), ref_idx) and torch.equal(val.cpu(), ref_val), "fused vs oracle: " + tag


def grad_stats(first, last):
    """Peak err / max|ref| per gradient over the randomised configurations (no assertion): what the tolerances are set from."""
    peaks = {}

    def record(got, ref, what, tol=3e-3):
        got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
        ratio = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        key = what.split()[0]
        peaks[key] = max(peaks.get(key, 0.0), ratio)

    T._grad_close = record
    for seed in range(first, last):
        for fn in (T.test_random_configurations, T.test_random_gather_and_tail):
            fn(seed)
    print("peak err / max|ref|:", {k: f"{v:.2e}" for k, v in sorted(peaks.items())})
    return 0


def main():
    if sys.argv[1] == "gradstats":
        return grad_stats(int(sys.argv[2]), int(sys.argv[3]))
    if sys.argv[1] in ("kld", "push", "pushfused"):
        case = {"kld": kld_case, "push": push_case, "pushfused": push_fused_case}[sys.argv[1]]
        first, last = int(sys.argv[2]), int(sys.argv[3])
        log = open(sys.argv[4], "a") if len(sys.argv) > 4 else sys.stdout
        fails = []
        for seed in range(first, last):
            try:
                case(seed)
            except Exception as e:  # noqa: BLE001
                fails.append(seed)
                print(f"FAIL {case.__name__}({seed}): {type(e).__name__}: {str(e)[:300]}", file=log, flush=True)
                if not isinstance(e, AssertionError):
                    traceback.print_exc(file=log)
            if seed % 20 == 0:
                print(f"{sys.argv[1]} seed {seed} done, {len(fails)} failures so far", file=log, flush=True)
        print(f"{sys.argv[1]} fuzz {first}..{last}: {len(fails)} failures {fails}")
        return 1 if fails else 0
    first, last = int(sys.argv[1]), int(sys.argv[2])
    log = open(sys.argv[3], "a") if len(sys.argv) > 3 else sys.stdout
    fails = []
    for seed in range(first, last):
        for fn in (T.test_random_configurations, T.test_random_gather_and_tail):
            try:
                fn(seed)
            except Exception as e:  # noqa: BLE001
                fails.append((fn.__name__, seed))
                print(f"FAIL {fn.__name__}({seed}): {type(e).__name__}: {str(e)[:300]}", file=log, flush=True)
                if not isinstance(e, AssertionError):
                    traceback.print_exc(file=log)
        if seed % 10 == 0:
            print(f"seed {seed} done, {len(fails)} failures so far", file=log, flush=True)
    print(f"fuzz {first}..{last}: {len(fails)} failures {fails}", file=log, flush=True)
    print(f"fuzz {first}..{last}: {len(fails)} failures {fails}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
