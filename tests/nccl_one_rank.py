"""Child process of tests/test_gpu_dp.py::test_rccl_one_rank_allreduce: RCCL ("nccl") load, init and one all-reduce of the
flat gradient bucket at world_size 1 on cuda:0 - the collective path of bench.py / scaleprotoseg_amd.dp, as far as a one-GPU
box can exercise it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
from scaleprotoseg_amd.dp import FlatGradBucket

bank = torch.nn.Parameter(torch.rand(190, 256, 1, 1, device=dev))
head = torch.nn.Parameter(torch.rand(19, 190, device=dev))
bucket = FlatGradBucket([bank, head], attach=True)
((bank ** 2).sum() + head.sum()).backward()
before = bucket.flat.clone()
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM)          # the collective bench.py / DataParallelStep issue per step
bucket.all_reduce(average=True)                             # world 1: the bucket logic around it
torch.cuda.synchronize()
assert torch.equal(bucket.flat, before) and bank.grad.data_ptr() == bucket.views[0].data_ptr()
t = torch.ones(1, device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank ok")
