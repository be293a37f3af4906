import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scaleprotoseg_amd as spx
from scaleprotoseg_amd.functional import proto_head_forward
dev = torch.device("cuda:0")
S, Cs, P, K = 1, 256, 190, 19
H, W = 256, 512
g = torch.Generator().manual_seed(6)
small = torch.sigmoid(torch.randn(1, Cs, 8, 16, generator=g)).to(dev, torch.bfloat16)
bank = torch.rand(P, Cs, 1, 1, generator=g).to(dev)
Wl = torch.randn(K, P, generator=g).to(dev)
lay = spx.BankLayout(P, K, S, Cs, ((0, P),))
big = small.repeat(1, 1, H // 8, W // 16).contiguous()
l_s, d_s, _ = proto_head_forward(small, bank, Wl, lay)
l_b, d_b, _ = proto_head_forward(big, bank, Wl, lay)
torch.cuda.synchronize()
ref = d_s.repeat(1, 1, H // 8, W // 16)
bad = (d_b != ref)
print("mismatches", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()
print(idx[:20].tolist())
if len(idx):
    pr = idx[:, 1]; px = idx[:, 2] * W + idx[:, 3]
    print("protos", torch.unique(pr)[:40].tolist())
    print("px%128", torch.unique(px % 128)[:64].tolist())
    print("tiles", torch.unique(px // 128)[:40].tolist(), len(torch.unique(px // 128)))
    i = idx[0]
    print(d_b[tuple(i)].item(), ref[tuple(i)].item())
    xf = big.float().view(Cs, -1); bf = bank.view(P, Cs).bfloat16().float()
    dd = ((xf[:, None, :] - bf.t()[:, :, None])[:, :4, :256] ** 2).sum(0)
    print(dd[0, :8], d_b.view(P, -1)[0, :8])
