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


# ---------------------------------------------------------------------------------------------------------------
# the data-parallel push DRIVER and the data-parallel optimizer step (composition logic; the kernels are replaced by
# the CPU oracle here - the same drivers run on the real kernels in tests/test_gpu_dp.py)
# ---------------------------------------------------------------------------------------------------------------
class _Backbone(torch.nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.base = torch.nn.Sequential(torch.nn.Conv2d(3, ch, 1), torch.nn.Conv2d(ch, ch, 1))
        self.pool = torch.nn.AvgPool2d(4)

    def __repr__(self):
        return "MSC(standin)"

    def forward(self, x):
        return self.base(self.pool(x))


class _PushData:
    convert_targets = None

    def __init__(self, n, K, absent, seed):
        g = torch.Generator().manual_seed(seed)
        self.items = []
        for _ in range(n):
            img = torch.randn(3, 24, 32, generator=g)
            t = torch.randint(0, K + 1, (3, 4), generator=g).repeat_interleave(8, 0).repeat_interleave(8, 1)
            t[t == absent + 1] = 0
            self.items.append((img, t.numpy()))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _push_setup():
    import scaleprotoseg_amd as spx

    S, Cs, K, per = 2, 16, 3, 2
    P = S * K * per
    torch.manual_seed(11)
    net = spx.PPNetMultiScale(_Backbone(S * Cs), 64, (P, Cs, 1, 1), [], K, add_on_layers_type="deeplab_simple",
                              patch_classification=True, num_scales=S)
    with torch.no_grad():
        net.prototype_vectors[1].copy_(net.prototype_vectors[0])      # forced duplicate
    return net, _PushData(7, K, absent=2, seed=12), S, K


def _patch_kernels_with_oracle(S):
    """compute_distances / argmin_over_images on the CPU oracle (tests may use it; the product path never does)."""
    from oracle import ppnet_oracle as O
    from scaleprotoseg_amd import push as push_mod

    def compute_distances(ppnet, dataset, img, target, num_classes, max_dist=1e10, device=None, void_class=None):
        conv = ppnet.conv_features(img.unsqueeze(0))
        ranges = {s: tuple(ppnet.scale_num_prototypes[s]) for s in range(S)}
        d = O.scale_l2_convolution(conv, ppnet.prototype_vectors.detach(), ranges, S)
        lab = O.resize_label(target, (d.shape[3], d.shape[2])).unsqueeze(0)
        return O.push_masked_argmin(d, lab, ppnet.prototype_class_identity, num_classes, max_dist, void_class)

    push_mod.compute_distances = compute_distances
    push_mod.argmin_over_images = lambda v: v.argmin(dim=0)


def _dp_push_case(rank, world):
    from scaleprotoseg_amd.push import push_prototypes_multiscale

    net, data, S, K = _push_setup()
    _patch_kernels_with_oracle(S)
    best, _, dup = push_prototypes_multiscale(data, net, log=lambda *_: None, device="cpu")
    return (net.prototype_vectors.detach().clone(), best.clone(), list(dup),
            {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}, net.last_layer.weight.detach().clone())


def test_data_parallel_push_driver_equals_single_process(tmp_path):
    from scaleprotoseg_amd.push import push_prototypes_multiscale

    res = _run(_dp_push_case, tmp_path)
    net, data, S, K = _push_setup()
    _patch_kernels_with_oracle(S)
    best, _, dup = push_prototypes_multiscale(data, net, log=lambda *_: None, device="cpu")    # world = 1 here
    assert 1 in dup and len(dup) >= 2                    # forced duplicate + the absent class's prototypes per scale
    for bank, gbest, gdup, ranges, wl in res:
        assert torch.equal(bank, net.prototype_vectors.detach())        # bit-identical bank on every rank
        assert torch.equal(gbest, best) and gdup == list(dup)
        assert ranges == {s: tuple(net.scale_num_prototypes[s]) for s in range(S)}
        assert torch.equal(wl, net.last_layer.weight.detach())


class _TinyGroupModel(torch.nn.Module):
    """Parameters with the group phase's names and shapes; the 'loss' is a smooth function of them and of rank-local data."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.prototype_vectors = torch.nn.Parameter(torch.rand(12, 8, 1, 1))
        self.group_projection = torch.nn.ModuleList([torch.nn.Linear(4, 3, bias=False) for _ in range(3)])
        self.last_layer_group = torch.nn.Linear(9, 3, bias=False)

    def loss(self, x):
        act = torch.exp(-torch.cdist(x, self.prototype_vectors.flatten(1)))
        groups = [torch.exp(gp(act[:, 4 * i:4 * i + 4])) for i, gp in enumerate(self.group_projection)]
        return (self.last_layer_group(torch.cat(groups, 1)) ** 2).mean()


def _dp_step_case(rank, world):
    from scaleprotoseg_amd.dp import DataParallelStep

    m = _TinyGroupModel()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    stepper = DataParallelStep(m, opt, iter_size=2)
    g = torch.Generator().manual_seed(50 + rank)
    stepped = []
    for it in range(6):                                   # 3 optimizer steps of 2 accumulation micro-steps
        stepped.append(stepper.backward(m.loss(torch.rand(5, 8, generator=g))))
    return [p.detach().clone() for p in m.parameters()], stepped


def test_data_parallel_step_keeps_replicas_bit_identical(tmp_path):
    from scaleprotoseg_amd.utils import projection_simplex_sort

    res = _run(_dp_step_case, tmp_path)
    (p0, s0), (p1, s1) = res
    assert s0 == s1 == [False, True] * 3
    assert all(torch.equal(a, b) for a, b in zip(p0, p1))               # replicas bit-identical after 3 steps
    # single-process restatement: the gradient of the mean of the per-rank losses, same Adam, same re-projection
    m = _TinyGroupModel()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    gens = [torch.Generator().manual_seed(50 + r) for r in range(2)]
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        micro = [[torch.rand(5, 8, generator=g) for g in gens] for _ in range(2)]
        loss = sum(m.loss(x) for pair in micro for x in pair) / 4.0
        loss.backward()
        opt.step()
        for gp in m.group_projection:
            gp.weight.data = projection_simplex_sort(gp.weight.data)
    for a, b in zip(p0, m.parameters()):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-7)
    for gp_w in p0[1:4]:
        assert (gp_w >= 0).all() and torch.allclose(gp_w.sum(1), torch.ones(3), atol=1e-5)    # rows on the simplex


def _dp_hook_case(rank, world):
    """The reference's phase-1 group training masks last_layer_group's gradient before the optimizer step when
    incorrect_strength == 0 (module_multiscale_group_train.py:327-328) and steps the LR scheduler before the re-projection."""
    from scaleprotoseg_amd.dp import DataParallelStep

    m = _TinyGroupModel()
    mask = torch.zeros(3, 9)
    for k in range(3):
        mask[k, 3 * k:3 * k + 3] = 1.0                      # group_class_identity.T: own-class connections only
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)

    def hook():
        m.last_layer_group.weight.grad *= mask

    stepper = DataParallelStep(m, opt, iter_size=1, grad_hook=hook, scheduler=sched)
    w0 = m.last_layer_group.weight.detach().clone()
    g = torch.Generator().manual_seed(70 + rank)
    for it in range(3):
        stepper.backward(m.loss(torch.rand(5, 8, generator=g)))
    grads_are_views = all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(stepper.bucket.params, stepper.bucket.views))
    return m.last_layer_group.weight.detach().clone(), w0, opt.param_groups[0]["lr"], grads_are_views


def test_data_parallel_step_grad_hook_and_scheduler(tmp_path):
    res = _run(_dp_hook_case, tmp_path)
    (w_a, w0, lr_a, views_a), (w_b, _, lr_b, views_b) = res
    assert torch.equal(w_a, w_b) and lr_a == lr_b == 1e-2 * 0.5 ** 3
    mask = torch.zeros(3, 9)
    for k in range(3):
        mask[k, 3 * k:3 * k + 3] = 1.0
    assert torch.equal(w_a[mask == 0], w0[mask == 0])       # the masked ("incorrect") connections never moved
    assert not torch.equal(w_a[mask == 1], w0[mask == 1])
    assert views_a and views_b                               # gradients live in the flat bucket: no gather / scatter copies


def test_flat_bucket_views_survive_backward_and_zero():
    from scaleprotoseg_amd.dp import FlatGradBucket

    a = torch.nn.Parameter(torch.rand(4, 3))
    b = torch.nn.Parameter(torch.rand(5))
    bucket = FlatGradBucket([a, b], attach=True)
    ((a ** 2).sum() + (3 * b).sum()).backward()
    assert a.grad.data_ptr() == bucket.views[0].data_ptr() and b.grad.data_ptr() == bucket.views[1].data_ptr()
    assert torch.allclose(bucket.flat[:12].view(4, 3), 2 * a.detach()) and torch.allclose(bucket.flat[12:], torch.full((5,), 3.0))
    bucket.zero()
    ((a ** 2).sum()).backward()                              # accumulates into the zeroed views
    assert torch.allclose(a.grad, 2 * a.detach()) and float(b.grad.abs().sum()) == 0.0
    b.grad = None                                            # a caller that resets a gradient: copied in and re-attached
    (b.sum()).backward()
    bucket.all_reduce()
    assert b.grad.data_ptr() == bucket.views[1].data_ptr() and torch.allclose(b.grad, torch.ones(5))


def test_data_parallel_step_follows_a_parameter_unfrozen_later():
    """The warm-up -> joint phase switch unfreezes parameters the optimizer already holds: from that step on they must be in the
    bucket (all-reduced AND cleared), not accumulate outside it."""
    from scaleprotoseg_amd.dp import DataParallelStep

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.ones(3))
            self.b = torch.nn.Parameter(torch.ones(2), requires_grad=False)      # frozen during the warm-up phase

    net = Net()
    opt = torch.optim.SGD([net.a, net.b], lr=0.0)
    dp = DataParallelStep(net, opt)
    dp.backward((net.a * 2).sum() + (net.b * 3).sum())
    assert net.b.grad is None and len(dp.bucket.params) == 1
    net.b.requires_grad_(True)                                                    # joint phase
    for _ in range(3):
        assert dp.backward((net.a * 2).sum() + (net.b * 3).sum())
        assert len(dp.bucket.params) == 2
        assert float(net.b.grad.abs().sum()) == 0.0                               # cleared with the bucket: no accumulation across steps
    (net.b * 3).sum().backward()
    assert torch.allclose(net.b.grad, torch.full((2,), 3.0)) and net.b.grad.data_ptr() == dp.bucket.views[1].data_ptr()
