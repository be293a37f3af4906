"""Diagnosis of the round-1 `capture_end` segfault (gpurun_out/seg.log): each variant runs ONCE in its own child
process with a native-backtrace SIGSEGV handler (tools/segv/libsegv_trace.so); the parent never touches the GPU.

Hypothesis under test: autograd's AccumulateGrad node of a leaf keeps the stream that was current when the node was
CREATED (torch/csrc/autograd/function.h "Function Streams"; the node is cached on the leaf for as long as some graph
references it).  If an eager step on the legacy default stream left its graph alive, a later captured `.backward()`
re-uses those nodes, the engine runs them on the DEFAULT stream behind an event of the capturing stream, the legacy
stream is pulled into the capture, and hipStreamEndCapture of this ROCm crashes (CUDA reports
cudaErrorStreamCaptureImplicit instead).  `torch_*` variants use stock torch ops only; `spx_*` the package's operators.

    python tools/probes/capture_repro.py            # all variants, one child each
    python tools/probes/capture_repro.py --variant torch_stale_accumulator
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = [
    "torch_fresh_accumulator",      # eager graph freed before the capture: expected ok
    "torch_stale_accumulator",      # eager graph (default stream) alive during a captured .backward(): the suspect
    "torch_stale_autograd_grad",    # same, but the captured backward uses autograd.grad (no AccumulateGrad runs)
    "spx_probe_form",               # round-1 probe: autograd.grad, nothing alive: expected ok
    "spx_stale_accumulator",        # package operators, eager graph alive, captured .backward()
    "spx_side_stream_discipline",   # package operators, warm-up + capture on ONE side stream, .backward(): the fix
]


def _torch_case(stale: bool, use_grad: bool):
    import torch

    dev = torch.device("cuda:0")
    x = torch.randn(1 << 16, device=dev, requires_grad=True)
    w = torch.randn(1 << 16, device=dev, requires_grad=True)
    keep = (x * w).sum()                 # eager, legacy default stream: creates AccumulateGrad(x), AccumulateGrad(w)
    if not stale:
        del keep                         # graph freed -> the accumulators die with it
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = (x * w * 3.0).sum()
        if use_grad:
            grads = torch.autograd.grad(y, [x, w])
        else:
            y.backward()
    g.replay()
    torch.cuda.synchronize()
    print("ok: captured + replayed", flush=True)


def _spx_case(kind: str):
    import torch

    sys.path.insert(0, ROOT)
    import scaleprotoseg_amd as spx
    from scaleprotoseg_amd.functional import proto_head_forward

    dev = torch.device("cuda:0")
    B, S, Cs, P, K, H, W = 2, 4, 64, 228, 19, 33, 33
    x = torch.sigmoid(torch.randn(B, S * Cs, H, W, device=dev)).bfloat16().requires_grad_(True)
    bank = torch.rand(P, Cs, 1, 1, device=dev).requires_grad_(True)
    head = (torch.randn(K, P, device=dev) * 0.1).requires_grad_(True)
    per = P // S
    lay = spx.BankLayout(P, K, S, Cs, tuple((s * per, (s + 1) * per) for s in range(S)))
    gl = torch.randn(B * H * W, K, device=dev) * 1e-3
    gd = torch.randn(B, P, H, W, device=dev) * 1e-3

    def fwd():
        logits, d, _ = proto_head_forward(x, bank, head, lay)
        return logits, d

    if kind == "probe":
        def step():
            logits, d = fwd()
            return torch.autograd.grad([logits, d], [x, bank, head], [gl, gd])
        ref = [t.clone() for t in step()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = step()
        g.replay()
        torch.cuda.synchronize()
        print("ok: identical", all(torch.equal(a, b) for a, b in zip(outs, ref)), flush=True)
        return

    def step_backward():
        for t in (x, bank, head):
            t.grad = None
        logits, d = fwd()
        torch.autograd.backward([logits, d], [gl, gd])

    step_backward()
    ref = [t.grad.clone() for t in (x, bank, head)]
    if kind == "stale":
        keep = fwd()                     # eager forward on the default stream, graph alive across the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step_backward()
    else:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step_backward()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step_backward()
    for t in (x, bank, head):
        t.grad.zero_()
    g.replay()
    torch.cuda.synchronize()
    print("ok: identical", all(torch.equal(t.grad, r) for t, r in zip((x, bank, head), ref)), flush=True)


def child(variant: str):
    ctypes.CDLL(os.path.join(ROOT, "tools", "segv", "libsegv_trace.so"))
    if variant == "torch_fresh_accumulator":
        _torch_case(stale=False, use_grad=False)
    elif variant == "torch_stale_accumulator":
        _torch_case(stale=True, use_grad=False)
    elif variant == "torch_stale_autograd_grad":
        _torch_case(stale=True, use_grad=True)
    elif variant == "spx_probe_form":
        _spx_case("probe")
    elif variant == "spx_stale_accumulator":
        _spx_case("stale")
    elif variant == "spx_side_stream_discipline":
        _spx_case("discipline")
    else:
        raise SystemExit(f"unknown variant {variant}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default=None)
    args = ap.parse_args()
    if args.variant:
        child(args.variant)
        return
    for v in VARIANTS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--variant", v], capture_output=True, text=True, timeout=300)
        tail = (r.stdout + r.stderr).strip().splitlines()
        print(f"##### {v}: rc={r.returncode}", flush=True)
        for line in tail[-45:]:
            print("    " + line, flush=True)


if __name__ == "__main__":
    main()
