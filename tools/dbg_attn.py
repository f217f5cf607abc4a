import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops
from oracle import dgq_oracle as orc
dev = torch.device("cuda:0")
def rel_l2(a, b): return ((a - b).norm() / b.norm()).item()
def run(D, T, S, H, mode, skip):
    g = torch.Generator().manual_seed(D * 1000 + T + mode)
    B, bits = 2, 8
    q = torch.randn(B, T, H * D, generator=g); k = torch.randn(B, S, H * D, generator=g); v = torch.randn(B, S, H * D, generator=g)
    scale = D ** -0.5
    qh, kh, vh = (x.view(B, -1, H, D).transpose(1, 2) for x in (q, k, v))
    p = torch.softmax(torch.matmul(qh, kh.transpose(-2, -1)) * scale, dim=-1)
    delta = None
    if mode == 1: pq = orc.log_quant(p[..., skip:], p[..., skip:].max(), bits)
    elif mode == 2:
        delta = torch.tensor([0.37 * float(p.max())]); pq = orc.log_quant(p[..., skip:], delta[0], bits)
    elif mode == 3:
        delta = torch.tensor([float(p.max()) / 255.0]); pq = orc.uaq(p[..., skip:], delta[0], torch.tensor(0.0), bits)
    pf = torch.cat([p[..., :skip], pq], dim=-1) if skip else pq
    ref = torch.matmul(pf, vh).transpose(1, 2).reshape(B, T, H * D)
    errs = []
    for _ in range(3):
        o = ops.attention_f32(q.to(dev), k.to(dev), v.to(dev), H, D, scale, mode, skip, delta.to(dev) if delta is not None else None, bits)
        torch.cuda.synchronize()
        errs.append(rel_l2(o.cpu(), ref))
    d = (o.cpu() - ref).abs().view(B, T, H, D).amax(-1)
    print(D, T, S, H, mode, skip, "errs", errs, "bad rows", (d > 1e-3 * ref.abs().max()).sum().item(), "of", d.numel())
for args in ((160, 256, 256, 8, 2, 0), (160, 256, 256, 8, 1, 0), (160, 256, 256, 8, 3, 1), (40, 200, 200, 2, 2, 0), (80, 257, 257, 2, 2, 0), (40, 4096, 4096, 8, 2, 0)):
    run(*args)
