"""world_size = 2 on ONE MI355X (both ranks share cuda:0, gloo as the rehearsal backend): the data-parallel push driver
and optimizer step on the REAL kernels.  The exchange code is the one RCCL runs on a multi-GPU node (dp.py); only the
backend string differs.  Three processes touch the card (pytest + 2 ranks), inside the box's limit of 6."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = fn(rank, world)
        torch.save(res, os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _run(fn, tmp_path, world=2):
    mp.spawn(_worker, args=(world, _free_port(), fn, str(tmp_path)), nprocs=world, join=True)
    return [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]


def _push_setup():
    from test_gpu_modules import _push_problem

    return _push_problem(torch.device("cuda:0"), S=4, Cs=16, K=5, per=2, n_img=7, seed=5)


def _dp_push_case(rank, world):
    from scaleprotoseg_amd.push import push_prototypes_multiscale

    net, data, P = _push_setup()
    best, _, dup = push_prototypes_multiscale(data, net, log=lambda *_: None)
    return (net.prototype_vectors.detach().cpu(), best.cpu(), list(dup),
            {s: tuple(net.scale_num_prototypes[s]) for s in range(4)})


def test_data_parallel_push_on_the_kernels(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scaleprotoseg_amd.push import push_prototypes_multiscale

    res = _run(_dp_push_case, tmp_path)
    net, data, P = _push_setup()
    best, _, dup = push_prototypes_multiscale(data, net, log=lambda *_: None)          # single process
    assert 1 in dup
    for bank, gbest, gdup, ranges in res:
        assert torch.equal(bank, net.prototype_vectors.detach().cpu())                  # bit-identical bank
        assert torch.equal(gbest, best.cpu()) and gdup == list(dup)
        assert ranges == {s: tuple(net.scale_num_prototypes[s]) for s in range(4)}


def _dp_step_case(rank, world):
    """3 optimizer steps of the group phase on rank-local images: CE-like loss on the logits + a distance term."""
    from test_gpu_modules import _Backbone
    from scaleprotoseg_amd.dp import DataParallelStep
    from scaleprotoseg_amd.model_multiscale_group import PPNetMultiScale as GroupNet

    dev = torch.device("cuda:0")
    S, Cs, K, G, P = 4, 16, 5, 3, 40
    torch.manual_seed(21)                                                  # identical initial replicas
    net = GroupNet(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                   patch_classification=True, num_scales=S, num_groups=G).to(dev)
    params = [net.prototype_vectors] + [gp.weight for gp in net.group_projection] + [net.last_layer_group.weight]
    opt = torch.optim.Adam(params, lr=1e-3)
    stepper = DataParallelStep(net, opt, iter_size=1)
    g = torch.Generator().manual_seed(100 + rank)                          # rank-local data
    losses = []
    for it in range(3):
        conv = torch.sigmoid(torch.randn(2, S * Cs, 9, 11, generator=g)).to(dev)
        tgt = torch.randint(0, K, (2 * 9 * 11,), generator=g).to(dev)
        logits, dist_map = net.forward_from_conv_features(conv)
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, K), tgt) + 1e-3 * dist_map.mean()
        losses.append(float(loss))
        assert stepper.backward(loss)
    return [p.detach().cpu() for p in params], losses


def test_data_parallel_step_on_the_kernels(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    (p0, l0), (p1, l1) = _run(_dp_step_case, tmp_path)
    assert l0 != l1                                                        # the ranks really saw different images
    assert all(torch.equal(a, b) for a, b in zip(p0, p1))                  # replicas bit-identical after 3 steps
    for w in p0[1:-1]:                                                     # projections re-projected onto the simplex
        assert (w >= 0).all() and torch.allclose(w.sum(1), torch.ones(w.shape[0]), atol=1e-5)


def test_rccl_one_rank_allreduce():
    """RCCL itself on the one-GPU box: ONE fresh child process runs the "nccl" backend at world_size 1 (library load, communicator
    init, one all-reduce of the flat gradient bucket, barrier, destroy).  The 1 -> 8 GPU curve stays unmeasured until a
    multi-GPU node exists; this at least executes the collective path bench.py and dp.py take under WORLD_SIZE > 1."""
    import socket
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nccl_one_rank.py")
    r = subprocess.run([sys.executable, script, str(port)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
