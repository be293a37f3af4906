"""world_size = 2 over gloo (CPU): the data-parallel exchange logic of scaleprotoseg_amd.dp.
The same code runs over RCCL (backend "nccl") on the GPUs; kernels are not involved here."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = fn(rank, world)
        torch.save(res, os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _run(fn, tmp_path, world=2):
    mp.spawn(_worker, args=(world, _free_port(), fn, str(tmp_path)), nprocs=world, join=True)
    return [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]


def _grad_case(rank, world):
    from scaleprotoseg_amd.dp import FlatGradBucket

    torch.manual_seed(0)
    bank = torch.nn.Parameter(torch.rand(12, 8, 1, 1))
    head = torch.nn.Parameter(torch.rand(3, 12))
    frozen = torch.nn.Parameter(torch.rand(4), requires_grad=False)
    g = torch.Generator().manual_seed(100 + rank)
    bank.grad = torch.randn(bank.shape, generator=g)
    head.grad = torch.randn(head.shape, generator=g)
    bucket = FlatGradBucket([bank, head, frozen])
    assert bucket.flat.numel() == bank.numel() + head.numel()
    bucket.all_reduce(average=False)
    return bank.grad.clone(), head.grad.clone()


def test_flat_bucket_allreduce_sums_and_keeps_replicas_identical(tmp_path):
    res = _run(_grad_case, tmp_path)
    exp_bank = sum(torch.randn(12, 8, 1, 1, generator=torch.Generator().manual_seed(100 + r)) for r in range(2))
    # second tensor drawn from the same generator stream per rank
    exp_head = 0
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        torch.randn(12, 8, 1, 1, generator=g)
        exp_head = exp_head + torch.randn(3, 12, generator=g)
    for bank_g, head_g in res:
        assert torch.allclose(bank_g, exp_bank) and torch.allclose(head_g, exp_head)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])   # bit-identical replicas


N_IMG, P = 7, 10


def _push_tables():
    g = torch.Generator().manual_seed(5)
    vals = torch.floor(torch.rand(N_IMG, P, generator=g) * 6)       # many exact ties across images
    vals[:, 3] = 1e10                                                  # a prototype whose class never appears
    flats = torch.randint(0, 500, (N_IMG, P), generator=g)
    return vals, flats


def _push_case(rank, world):
    from scaleprotoseg_amd.dp import gather_push_patches, reduce_push_candidates, shard_range

    vals, flats = _push_tables()
    rng = shard_range(N_IMG, rank, world)
    local_v, local_f = vals[rng.start : rng.stop], flats[rng.start : rng.stop]
    best_local = local_v.argmin(dim=0)                                 # what the GPU reduction returns per shard
    ar = torch.arange(P)
    gimg, gval, gflat = reduce_push_candidates(best_local, local_v[best_local, ar], local_f[best_local, ar], rng.start)
    owner = (gimg >= rng.start) & (gimg < rng.stop)
    patches = torch.arange(P * 4, dtype=torch.float32).reshape(P, 4) * (rank + 1)
    full = gather_push_patches(patches, owner)
    return gimg, gval, gflat, owner, full


def test_sharded_push_matches_single_process_argmin(tmp_path):
    res = _run(_push_case, tmp_path)
    vals, flats = _push_tables()
    best = vals.argmin(dim=0)                                          # push_multiscale_optimization.py:137
    ar = torch.arange(P)
    for gimg, gval, gflat, owner, full in res:
        assert torch.equal(gimg, best)                                 # lowest image index on ties, as torch.argmin
        assert torch.equal(gval, vals[best, ar]) and torch.equal(gflat, flats[best, ar])
    owners = torch.stack([r[3] for r in res]).sum(0)
    assert (owners == 1).all()                                         # exactly one rank contributes each prototype
    base = torch.arange(P * 4, dtype=torch.float32).reshape(P, 4)
    exp = torch.where(res[0][3][:, None], base * 1, base * 2)
    assert torch.equal(res[0][4], exp) and torch.equal(res[1][4], exp)


def test_shard_range_covers_everything():
    from scaleprotoseg_amd.dp import shard_range

    for n in (0, 1, 7, 8, 2975):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in shard_range(n, r, w)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
