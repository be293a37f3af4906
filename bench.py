#!/usr/bin/env python3
"""Benchmark of the prototype-distance hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward (logits + the fp32 distance map, the reference's forward contract) + backward
(incoming dLogits and dDistances -> dX, dPrototypes, dLastLayer) of ONE synthetic Cityscapes-shaped image per
GPU (1024x2048 latent pixels x 256 channels, 190 prototypes, 19 classes, bf16 features), followed — for
N > 1 — by one RCCL all-reduce of the flat gradient bucket.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (C, P, S, K, H, W)   latent grid H x W per image, one image per GPU
    "cityscapes_1024x2048_c256_p190_s1": (256, 190, 1, 19, 1024, 2048),   # north-star shape (SURVEY.md 8d primary)
    "cityscapes_native_129x257_p228_s4": (256, 228, 4, 19, 129, 257),     # scaleproto_cityscapes.gin full image
    "cityscapes_1024x2048_c256_p228_s4": (256, 228, 4, 19, 1024, 2048),   # the gin's 4-scale bank at the north-star grid
    "odd_1023x2047_c256_p190_s1": (256, 190, 1, 19, 1023, 2047),          # diagnostic: H*W odd -> rows of X are only 2-byte aligned
    "cityscapes_crops_10x65x65_p228_s4": (256, 228, 4, 19, 650, 65),       # diagnostic: the gin's training crops (10 x 65 x 65) as one 650 x 65 grid
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0


def algorithmic_bytes_per_px(C, P, K):
    """bf16 X, fp32 distances + logits, distances materialised."""
    fwd = 2 * C + 4 * P + 4 * K
    bwd = 4 * P + 4 * K + 2 * C + 2 * C
    return fwd, bwd


def cpu_baseline(C, P, S, K, sample_hw, reps=6):
    """The oracle (CPU restatement of the reference op sequence) timed on this host's cores."""
    from oracle import ppnet_oracle as O

    H, W = sample_hw
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("SPX_CPU_THREADS", "16"))))   # a 1-GPU box has a 16-core share
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(20220227)
    conv = O.bf16_representable(torch.sigmoid(torch.randn(1, C, H, W, generator=g)))
    bank = O.bf16_representable(torch.rand(P, C // S, 1, 1, generator=g))
    ident = O.default_class_identity(P, K, S)
    Wl = O.last_layer_init(ident)
    gl = torch.randn(1, H, W, K, generator=g) * 1e-3
    gd = torch.randn(1, P, H, W, generator=g) * 1e-3
    ranges = O.default_scale_ranges(P, S)
    O.fwd_bwd_reference(conv, bank, ranges, S, Wl, gl, gd)  # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        O.fwd_bwd_reference(conv, bank, ranges, S, Wl, gl, gd)
    dt = (time.perf_counter() - t0) / reps
    return {
        "value": round(H * W / dt / 1e6, 4),
        "unit": "Mpix/s",
        "cores": cores,
        "kind": "port",
        "sample": f"oracle fwd+bwd, 1x{C}x{H}x{W} latent px slice of the workload, P={P}, S={S}, fp32, "
                  f"{reps} reps after 1 warm-up, {dt:.2f} s/rep, torch {torch.__version__} CPU",
    }


def secondary_modes(dev, spx, F_, steps=5, warmup=2):
    """The driver-visible secondary measurements: each entry is timed in this same process run, after the
    headline, on synthetic inputs of the named shape: ms per forward or per fwd+bwd step, and the algorithmic GB/s and
    TFLOP/s that time corresponds to (SURVEY.md 8d byte / flop counts for the mode's own outputs)."""
    from scaleprotoseg_amd.graphs import capture_step

    out = {}

    def problem(C, P, S, K, H, W, B, seed=1):
        g = torch.Generator(device=dev).manual_seed(20220227 + seed)
        Cs = C // S
        x = torch.sigmoid(torch.randn(B, C, H, W, device=dev, generator=g)).to(torch.bfloat16).requires_grad_(True)
        per_scale = P // S
        layout = spx.BankLayout(P, K, S, Cs, tuple((s_ * per_scale, (s_ + 1) * per_scale) for s_ in range(S)))
        bank = torch.rand(P, Cs, 1, 1, device=dev, generator=g).to(torch.bfloat16).float().requires_grad_(True)
        ident = torch.zeros(P, K, device=dev)
        per_cs = max(1, P // K // S)
        for s_ in range(S):
            for k in range(K):
                ident[s_ * per_scale + k * per_cs: s_ * per_scale + (k + 1) * per_cs, k] = 1
        head = (ident.t() - 0.5 * (1 - ident.t())).contiguous().requires_grad_(True)
        return g, x, layout, bank, ident, head

    def timed(fn, n=steps, w=warmup):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def entry(ms, M, nbytes_px, flop_px, **extra):
        return {"ms": round(ms, 4), "Mpix_s": round(M / (ms * 1e-3) / 1e6, 1), "algorithmic_GBs": round(nbytes_px * M / (ms * 1e-3) / 1e9, 1),
                "algorithmic_TFLOPs": round(flop_px * M / (ms * 1e-3) / 1e12, 1), **extra}

    # (1) logits-only forward at the north-star shape (no fp32 distance map: eval_test.py:96-97 deletes it)
    C, P, S, K, H, W = WORKLOADS["cityscapes_1024x2048_c256_p190_s1"]
    M = H * W
    g, x, layout, bank, ident, head = problem(C, P, S, K, H, W, 1)

    def fwd_logits():
        with torch.no_grad():
            spx.proto_head_forward(x, bank, head, layout, want_distances=False)
    ms = timed(fwd_logits)
    out["logits_only_forward_1024x2048_p190_s1"] = entry(ms, M, 2 * C + 4 * K, 2 * P * (C // S), mfma_frac=round(2 * P * C * M / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4))

    # (2) / (3) class-gathered distances: fwd+bwd with a random gradient on the gathered planes, then with the
    # KLD loss (HIP kernels) producing it
    keys, J, table = spx.class_gather_table(layout, ident, dev)
    patches = torch.randint(0, K, (1, (H + 63) // 64, (W + 63) // 64), device=dev, generator=g, dtype=torch.int32)
    labels0 = patches.repeat_interleave(64, 1).repeat_interleave(64, 2)[:, :H, :W].reshape(1, H * W).contiguous()
    gather = spx.ClassGather(labels=labels0, keys=keys, width=J, table=table)
    g_logits = torch.randn(M, K, device=dev, generator=g) * 1e-3
    g_cls = torch.randn(1, J, M, device=dev, generator=g) * 1e-3
    kld_fn = spx.KLDLoss(ident, S, {s_: layout.scale_ranges[s_] for s_ in range(S)})
    target1 = (labels0 + 1).reshape(1, H, W)
    fwd_b = 2 * C + 4 * K + 4 * J + 4
    bwd_b = 4 * K + 2 * C + 2 * C + 4 * J + 4

    def gathered(kld):
        def f():
            x.grad = bank.grad = head.grad = None
            logits, dmap, _ = spx.proto_head_forward(x, bank, head, layout, want_distances=False, class_gather=gather)
            if kld:
                loss = kld_fn(spx.ClassDistances(dmap, gather.labels, gather.table, (H, W), target1, target1._version), target1)
                torch.autograd.backward([logits, loss], [g_logits, None])
            else:
                torch.autograd.backward([logits, dmap], [g_logits, g_cls])
        return f
    out["class_gathered_step_1024x2048_p190_s1"] = entry(timed(gathered(False)), M, fwd_b + bwd_b, 6 * P * (C // S))
    out["class_gathered_kld_step_1024x2048_p190_s1"] = entry(timed(gathered(True)), M, fwd_b + bwd_b, 6 * P * (C // S))
    # (3b) the prototype push's per-image reduction (push_multiscale_optimization.py:68-91) on one 1024x2048 latent grid:
    # fused into the distance kernel (spx_dist_push_min: features in, [P] minima out, no map) against the two-step form
    # (spx_dist_fwd writes the fp32 map, spx_push_argmin reduces it)
    push_target = target1.to(torch.int64)

    def push_fused():
        with torch.no_grad():
            F_.push_min_from_features(x, bank, layout, push_target, ident, void_class=0, keys=keys)

    def push_two_step():
        with torch.no_grad():
            _, dmap, _ = spx.proto_head_forward(x, bank, None, layout, want_distances=True)
            spx.push_masked_argmin(dmap, push_target, ident, void_class=0)
    out["push_min_fused_1024x2048_p190_s1"] = entry(timed(push_fused), M, 2 * C + 4, 2 * P * (C // S))
    out["push_min_two_step_1024x2048_p190_s1"] = entry(timed(push_two_step), M, 2 * C + 4, 2 * P * (C // S))
    del x, bank, head, g_logits, g_cls, gather, labels0, patches, target1, push_target
    torch.cuda.empty_cache()

    # (4) / (5) the reference's own shapes as HIP-graph replays of one fwd+bwd (eager is host-bound at these sizes)
    for name, (C, P, S, K, H, W, B) in (("native_129x257_p228_s4_graph_step", (256, 228, 4, 19, 129, 257, 1)),
                                        ("ade_2x65x65_p1800_s4_graph_step", (256, 1800, 4, 150, 65, 65, 2))):
        M = B * H * W
        g, x, layout, bank, ident, head = problem(C, P, S, K, H, W, B, seed=2)
        gl = torch.randn(M, K, device=dev, generator=g) * 1e-3
        gd = torch.randn(B, P, H, W, device=dev, generator=g) * 1e-3

        def f():
            x.grad = bank.grad = head.grad = None
            logits, dmap, _ = spx.proto_head_forward(x, bank, head, layout, want_distances=True)
            torch.autograd.backward([logits, dmap], [gl, gd])
        graph, _ = capture_step(f, warmup=2)
        ms = timed(graph.replay, n=20, w=3)
        fb, bb = algorithmic_bytes_per_px(C, P, K)
        out[name] = entry(ms, M, fb + bb, 6 * P * (C // S))
        del graph, x, bank, head, gl, gd
        torch.cuda.empty_cache()

    # (6) the reference's training crops with the losses on the path's output inside the step: class-gathered distances +
    # KLDLoss + logits gradient, 10 x 65 x 65 latent pixels, irregular label regions (argmax of smooth random fields), graph replay
    C, P, S, K, H, W, B = 256, 228, 4, 19, 65, 65, 10
    M = B * H * W
    g, x, layout, bank, ident, head = problem(C, P, S, K, H, W, B, seed=3)
    keys, J, table = spx.class_gather_table(layout, ident, dev)
    coarse = torch.randn(B, K + 1, 5, 5, device=dev, generator=g)
    target = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bicubic", align_corners=False).argmax(dim=1)
    gather = spx.ClassGather(labels=(target.reshape(B, -1) - 1).to(torch.int32).contiguous(), keys=keys, width=J, table=table)
    kld_fn = spx.KLDLoss(ident, S, {s_: layout.scale_ranges[s_] for s_ in range(S)})
    gl = torch.randn(M, K, device=dev, generator=g) * 1e-3

    def crop_step():
        x.grad = bank.grad = head.grad = None
        logits, dmap, _ = spx.proto_head_forward(x, bank, head, layout, want_distances=False, class_gather=gather)
        loss = kld_fn(spx.ClassDistances(dmap, gather.labels, gather.table, (H, W), target, target._version), target)
        torch.autograd.backward([logits, loss], [gl, None])
    graph, _ = capture_step(crop_step, warmup=2)
    ms = timed(graph.replay, n=20, w=3)
    out["crops_10x65x65_p228_s4_gathered_kld_graph_step"] = entry(ms, M, 2 * C + 4 * K + 4 * J + 4 + 4 * K + 2 * C + 2 * C + 4 * J + 4, 6 * P * (C // S))
    del graph, x, bank, head, gl
    torch.cuda.empty_cache()

    # (7) / (8) the grouping phase THROUGH the drop-in module (model_multiscale_group.py:404-452 and its autograd: dense group
    # projection, exp, last_layer_group; module_multiscale_group_train.py:224-262), as graph replays of one fwd+bwd step:
    # Cityscapes crops with the fused cross entropy, and the ADE crops whose 450 units run on the product kernels
    import torch.nn as nn
    from scaleprotoseg_amd.loss import PixelWiseCrossEntropyLoss
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

    class _Backbone(nn.Module):            # stand-in (out of scope): features are fed directly
        def __init__(self, ch):
            super().__init__()
            self.base = nn.Sequential(nn.Conv2d(3, ch, 1), nn.Conv2d(ch, ch, 1))

        def __repr__(self):
            return "MSC(standin)"

        def forward(self, x):
            return x

    for name, (P, K, B, ce) in (("group_crops_10x65x65_p228_s4_g3_ce_graph_step", (228, 19, 10, True)),
                                ("ade_group_2x65x65_p1800_s4_g3_graph_step", (1800, 150, 2, False))):
        C, S, H, W = 256, 4, 65, 65
        M = B * H * W
        torch.manual_seed(7)
        net = GroupNet(_Backbone(C), 64, (P, C // S, 1, 1), [], K, num_groups=3, add_on_layers_type="deeplab_simple",
                       patch_classification=True, num_scales=S).to(dev)
        net.add_on_layers = nn.Sequential()
        g = torch.Generator(device=dev).manual_seed(20220227 + 4)
        xg = torch.sigmoid(torch.randn(B, C, H, W, device=dev, generator=g)).to(torch.bfloat16).requires_grad_(True)
        glg = torch.randn(B, H, W, K, device=dev, generator=g) * 1e-3
        tgt = torch.randint(0, K + 1, (B, H, W), device=dev, generator=g)
        lossf = PixelWiseCrossEntropyLoss(ignore_index=-1)
        params = [p_ for p_ in net.parameters() if p_.requires_grad]

        def gstep():
            xg.grad = None
            for p_ in params:
                p_.grad = None
            if ce:
                logits, _ = net.forward_from_conv_features(xg, ce_target=tgt)
                lossf(logits, tgt).backward()
            else:
                logits, _ = net.forward_from_conv_features(xg)
                torch.autograd.backward([logits], [glg])
        graph, _ = capture_step(gstep, warmup=2)
        ms = timed(graph.replay, n=20, w=3)
        U = 3 * K
        fb, bb = algorithmic_bytes_per_px(C, P, K)
        out[name] = entry(ms, M, fb + bb, 6 * P * (C // S) + 6 * P * U + 6 * U * K, units=U)
        del graph, net, xg, glg, params
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cityscapes_1024x2048_c256_p190_s1", choices=sorted(WORKLOADS))
    ap.add_argument("--x-dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the secondary `modes` measurements (logits-only forward, class-gathered "
                    "steps, the reference's own shapes as graph replays)")
    ap.add_argument("--grads", default="logits,dist", help="diagnostic only: which output gradients flow back")
    ap.add_argument("--outputs", default="logits,dist", help="diagnostic only: 'logits' = logits-only forward (no fp32 distance map); "
                    "'logits,class_dist' = class-gathered distances (SURVEY 8f-1: what the fused KLD consumes) instead of the P-wide map")
    ap.add_argument("--kld", action="store_true", help="diagnostic only (with --outputs logits,class_dist): the gradient of the gathered "
                    "distances comes from KLDLoss (HIP kernels) inside the timed step instead of a fixed random tensor")
    ap.add_argument("--torch-profile", action="store_true", help="diagnostic only: print torch.profiler's top ops of one extra step to stderr")
    ap.add_argument("--group-tail", action="store_true", help="diagnostic only: group phase - the head is the dense [3K, P] grouping "
                    "matrix and exp + last_layer_group run fused in the kernels (spx_dist_fwd_group / spx_dist_bwd_group)")
    ap.add_argument("--freeze", default="", help="diagnostic only: comma list of x,bank,head to exclude from the backward")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` typed directly: start the N ranks ourselves (one process per GPU, the same
        # torch.distributed.run command line the driver uses) BEFORE anything touches the GPU in this process, relay
        # their output (rank 0 prints the JSON line) and exit with their status.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # one rank per GPU over RCCL ("nccl").  SPX_BENCH_BACKEND=gloo is a rehearsal switch for a one-GPU box: the
    # ranks then share the visible devices round-robin and the gradient bucket is reduced by gloo.
    backend = os.environ.get("SPX_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import scaleprotoseg_amd as spx
    from scaleprotoseg_amd import functional as F_
    from scaleprotoseg_amd.dp import FlatGradBucket

    spx.load_library()
    C, P, S, K, H, W = WORKLOADS[args.workload]
    Cs = C // S
    xdt = torch.bfloat16 if args.x_dtype == "bf16" else torch.float32

    # ---- synthetic inputs, resident in HBM (SURVEY.md 8d "Synthetic inputs") ----
    g = torch.Generator(device=dev).manual_seed(20220227 + rank)
    x = torch.sigmoid(torch.randn(1, C, H, W, device=dev, generator=g)).to(xdt).requires_grad_(True)
    per_scale = P // S
    layout = spx.BankLayout(P, K, S, Cs, tuple((s * per_scale, (s + 1) * per_scale) for s in range(S)))
    gp = torch.Generator(device=dev).manual_seed(20220227)           # identical parameters on every rank
    bank = torch.rand(P, Cs, 1, 1, device=dev, generator=gp).to(torch.bfloat16).float().requires_grad_(True)
    ident = torch.zeros(P, K, device=dev)
    per_cs = P // K // S
    for s in range(S):
        for k in range(K):
            ident[s * per_scale + k * per_cs : s * per_scale + (k + 1) * per_cs, k] = 1
    head = (ident.t() - 0.5 * (1 - ident.t())).contiguous().requires_grad_(True)
    tail = None
    if args.group_tail:
        G = 3                                            # scaleproto_*.gin: num_groups = 3
        U = G * K
        wd = torch.zeros(U, P, device=dev)
        for k in range(K):
            cols = ident[:, k].nonzero().flatten()
            wd[k * G:(k + 1) * G, cols] = torch.rand(G, cols.numel(), device=dev, generator=gp) / max(1, cols.numel())
        head = wd.requires_grad_(True)
        gci = torch.zeros(U, K, device=dev)
        for k in range(K):
            gci[k * G:(k + 1) * G, k] = 1
        tail = (gci.t() - 0.5 * (1 - gci.t())).contiguous().requires_grad_(True)
        layout = spx.BankLayout(P, U, S, Cs, layout.scale_ranges)
    g_logits = torch.randn(H * W, K, device=dev, generator=g) * 1e-3
    g_dist = torch.randn(1, P, H, W, device=dev, generator=g) * 1e-3
    for name in filter(None, args.freeze.split(",")):
        {"x": x, "bank": bank, "head": head}[name].requires_grad_(False)
    # the gradients live in ONE flat fp32 buffer (`.grad` = views of it): the N > 1 step all-reduces that buffer in place,
    # with no gather / scatter copies around the collective
    # (one GPU: no collective, the gradients stay ordinary `.grad` tensors)
    bucket = FlatGradBucket([p for p in (bank, head, tail) if p is not None and p.requires_grad] or [bank.requires_grad_(True)], attach=world > 1)

    gather = None
    if "class_dist" in args.outputs.split(","):
        keys, J, table = spx.class_gather_table(layout, ident, dev)
        # piecewise-constant label map (segmentation masks are): 64x64-px patches of one class each
        patches = torch.randint(0, K, (1, (H + 63) // 64, (W + 63) // 64), device=dev, generator=g, dtype=torch.int32)
        labels0 = patches.repeat_interleave(64, 1).repeat_interleave(64, 2)[:, :H, :W].reshape(1, H * W).contiguous()
        gather = spx.ClassGather(labels=labels0, keys=keys, width=J, table=table)
        g_cls = torch.randn(1, J, H * W, device=dev, generator=g) * 1e-3
        kld_fn = spx.KLDLoss(ident, S, {s_: layout.scale_ranges[s_] for s_ in range(S)})
        target1 = (labels0 + 1).reshape(1, H, W)

    def step():
        x.grad = None
        if world > 1:
            bucket.zero()                 # (not `.grad = None`: the views stay attached to the bucket)
        else:
            bank.grad = head.grad = None
            if tail is not None:
                tail.grad = None
        want_d = "dist" in args.outputs.split(",")
        if tail is not None:
            logits, dmap, _, _ = spx.proto_head_forward(x, bank, head, layout, want_distances=want_d, group_tail=tail)
        else:
            logits, dmap, _ = spx.proto_head_forward(x, bank, head, layout, want_distances=want_d, class_gather=gather)
        outs, gouts = [], []
        if "logits" in args.grads:
            outs.append(logits); gouts.append(g_logits)
        if "dist" in args.grads and gather is not None and args.kld:
            outs.append(kld_fn(spx.ClassDistances(dmap, gather.labels, gather.table, (H, W), target1, target1._version), target1)); gouts.append(None)
        elif "dist" in args.grads and gather is not None:
            outs.append(dmap); gouts.append(g_cls)
        elif "dist" in args.grads and want_d:
            outs.append(dmap); gouts.append(g_dist)
        torch.autograd.backward(outs, gouts)
        if world > 1:
            bucket.all_reduce()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if args.torch_profile and rank == 0:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as tp:
            step()
            torch.cuda.synchronize()
        print(tp.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60), file=sys.stderr)
    prof = []
    fence()
    F_.set_profile(prof)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    F_.set_profile(None)
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-operator durations from the HIP events of the timed steps ----
    per_op = {}
    for name, e0, e1 in prof:
        per_op.setdefault(name, []).append(e0.elapsed_time(e1))
    op_ms = {k: sum(v) / len(v) for k, v in per_op.items()}
    M = H * W
    fwd_b, bwd_b = algorithmic_bytes_per_px(C, P, K)
    if gather is not None:      # the P-wide fp32 map and its gradient are replaced by J entries per pixel
        fwd_b += 4 * gather.width + 4 - 4 * P
        bwd_b += 4 * gather.width + 4 - 4 * P
    elif "dist" not in args.outputs.split(","):
        fwd_b -= 4 * P
        bwd_b -= 4 * P
    nb = P * Cs  # sum_s Cs * Ps
    # Algorithmic bytes / flops per OPERATOR.  The backward is two C-ABI calls (pixel side spx_dist_bwd = kernel 1,
    # parameter side spx_bank_bwd = kernels 2 + 3) that only together produce (dX, dPrototypes, dLastLayer): its
    # algorithmic bytes are charged to the pair, against the SUM of their durations.
    op_bytes = {"spx_dist_fwd": fwd_b * M, "backward": bwd_b * M}
    op_flops = {"spx_dist_fwd": 2 * nb * M, "backward": 4 * nb * M}
    if "spx_dist_bwd" in op_ms:
        op_ms["backward"] = op_ms["spx_dist_bwd"] + op_ms.get("spx_bank_bwd", 0.0)
    dominant = max((k for k in ("spx_dist_fwd", "backward") if k in op_ms), key=op_ms.get, default=None)
    kernels = {
        k: {
            "ms": round(v, 4),
            **({"algorithmic_GBs": round(op_bytes[k] / (v * 1e-3) / 1e9, 1),
                "algorithmic_TFLOPs": round(op_flops[k] / (v * 1e-3) / 1e12, 1)} if k in op_bytes else {}),
        }
        for k, v in op_ms.items()
    }

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * M / (elapsed / args.steps) / 1e6
        out = {
            "metric": "Mpix/s fwd+bwd prototype-distance, Cityscapes 1024x2048",
            "value": round(value, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if args.x_dtype == "bf16" else "fp32->bf16",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "latent_px_per_gpu": M,
                "channels": C,
                "prototypes": P,
                "scales": S,
                "classes": K,
                "features_dtype": args.x_dtype,
                "outputs": ("logits + fp32 distance map (reference forward contract); grads dX, dPrototypes, dLastLayer"
                            if (args.outputs == "logits,dist" and not args.group_tail)
                            else f"DIAGNOSTIC outputs={args.outputs} grads={args.grads} group_tail={args.group_tail} kld={args.kld}"),
                "parallelism": f"dp{world}" if world > 1 else "single",
            },
            "kernels": kernels,
        }
        if dominant is not None:
            dom_ms = op_ms[dominant]
            ach = op_bytes[dominant] / (dom_ms * 1e-3) / 1e9
            # HBM bytes per launch from the PMC passes of tools/profile_cmd.sh (FETCH_SIZE x2 + WRITE_SIZE, the guide's
            # gfx950 correction): a RECORDED figure of the profiled commit named in the file, not measured in this run;
            # for the backward it is the sum over its kernels.  Only valid for the default workload / dtype.
            traffic, traffic_src = None, None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile) and args.workload == "cityscapes_1024x2048_c256_p190_s1" and args.x_dtype == "bf16" \
                    and args.outputs == "logits,dist" and args.grads == "logits,dist" and not args.freeze and not args.group_tail:
                try:
                    tj = json.load(open(tfile))
                    parts = ["spx_dist_fwd"] if dominant == "spx_dist_fwd" else ["spx_dist_bwd", "spx_bank_bwd", "spx_dw_reduce", "spx_bank_reduce"]
                    if all(p_ in tj for p_ in parts[:2 if dominant == "backward" else 1]):
                        traffic = int(sum(tj.get(p_, 0) for p_ in parts))
                        traffic_src = f"profiles/traffic.json ({tj.get('_source', 'rocprofv3 --pmc passes')})"
                except Exception:
                    traffic = None
            out["roofline"] = {
                "kernel": dominant if dominant != "backward" else "backward = spx_dist_bwd (pixel kernel) + spx_bank_bwd (parameter kernel + the two fixed-order reductions)",
                "bound": "hbm",
                "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
            }
            step_gbs = (fwd_b + bwd_b) * M / (ms_per_step * 1e-3) / 1e9
            out["roofline_step"] = {
                "bound": "hbm",
                "achieved": round(step_gbs, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(step_gbs / HBM_PEAK_GBS, 4),
                "mfma_frac": round(6 * nb * M / (ms_per_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                "algorithmic_bytes_per_px": fwd_b + bwd_b,
                "algorithmic_flop_per_px": 6 * nb,
            }
        default_run = (args.workload == "cityscapes_1024x2048_c256_p190_s1" and args.x_dtype == "bf16" and args.outputs == "logits,dist"
                       and args.grads == "logits,dist" and not args.freeze and not args.group_tail)
        if world == 1 and default_run and not args.no_modes:
            del x, g_logits, g_dist
            torch.cuda.empty_cache()
            try:
                out["modes"] = secondary_modes(dev, spx, F_)
            except Exception as exc:                     # a secondary measurement must never cost the headline line
                out["modes"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(C, P, S, K, (128, W) if H >= 128 else (H, W))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
