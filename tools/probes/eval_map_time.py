import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import scaleprotoseg_amd as spx
dev = torch.device("cuda:0")
src = torch.rand(1, 228, 129, 257, device=dev) * 10
def t(f, name):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/5:.3f} ms")
t(lambda: spx.upsample_argext(src, (1024, 2048)), "fused upsample+argmin 228x129x257 -> 1024x2048")
t(lambda: torch.argmin(F.interpolate(src, size=(1024, 2048), mode="bilinear", align_corners=False)[0], dim=0), "torch interpolate + argmin (GPU)")
