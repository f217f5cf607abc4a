import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import ops
dev = torch.device("cuda:0")
D, T, S, H, mode = 40, 200, 200, 2, 1
g = torch.Generator().manual_seed(1)
B, bits, skip = 2, 8, 1
q, k, v = (torch.randn(B, n, H * D, generator=g).to(dev) for n in (T, S, S))
def table(n, lo):
    d = (torch.rand(n, generator=g) * 0.02 + lo).to(dev)
    z = torch.randint(100, 156, (n,), generator=g).float().to(dev)
    return d, z
fq_q = (1,) + table(T, 0.02) + (0, bits)
fq_k = (1,) + table(S - skip, 0.02) + (skip, bits)
fq_v = (2,) + table(D, 0.02) + (0, bits)
scale = D ** -0.5
for name, fqs in (("q", (fq_q, None, None)), ("k", (None, fq_k, None)), ("v", (None, None, fq_v)), ("all", (fq_q, fq_k, fq_v))):
    qq, kk, vv = q.clone(), k.clone(), v.clone()
    for ten, f, n in ((qq, fqs[0], T), (kk, fqs[1], S), (vv, fqs[2], S)):
        if f is not None:
            ops.fakequant_rows(ten.view(B * n, H * D), n, D, f[0], f[1], f[2], f[3], f[4])
    ref = ops.attention_f32(qq, kk, vv, H, D, scale, mode, skip, None, bits)
    out = ops.attention_f32(q, k, v, H, D, scale, mode, skip, None, bits, fq=fqs)
    ref2 = ops.attention_f32(qq, kk, vv, H, D, scale, mode, skip, None, bits)
    torch.cuda.synchronize()
    diff = (out - ref).abs()
    print(name, "maxdiff", diff.max().item(), "frac", (diff > 0).float().mean().item(), "ref repeat equal", torch.equal(ref, ref2),
          "rows differing", (diff.view(B, T, -1).amax(-1) > 0).sum().item())
