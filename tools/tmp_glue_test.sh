cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "timestep_embedding or cfg_ddim" 2>&1 | tail -15
python - <<'PY'
import torch, math
from dgq_amd import ops
dev=torch.device('cuda:0')
for dim,tv in ((320,[981]),(256,[1024.,0.,512.])):
    t=torch.tensor(tv, device=dev, dtype=torch.int64 if dim==320 else torch.float32)
    half=dim//2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=dev) / (half - 0.0))
    ang = t[:, None].float() * freqs[None, :]
    want = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    got=ops.timestep_embedding(t,dim)
    print(dim, 'equal', torch.equal(got,want), 'maxdiff', (got-want).abs().max().item(), 'n diff', (got!=want).sum().item())
PY
