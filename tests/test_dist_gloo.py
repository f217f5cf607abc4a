"""N>1 path on CPU: two gloo ranks shard the prompt list (no data-path collective), each runs the denoise loop
on its own prompts, and the union equals the single-process result.  The UNet here is the floating-point tiny
graph (plain torch) — the quantized kernels need a GPU; what is covered is the sharding / loop / timing logic
bench.py uses with RCCL."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dgq_amd import synth
from dgq_amd.runtime import denoise_loop, shard_prompts


def _unet():
    from dgq_amd.diffusers_rewrite import UNet2DConditionModel
    net = UNet2DConditionModel("tiny").eval()
    synth.load_synth_weights(net, "tiny", 0)
    return net


def _run_prompts(net, ids, steps=2):
    outs = {}
    for i in ids:
        lat = synth.named_randn("latent", (1, 4, 16, 16), 1000 + i)
        ctx = synth.named_randn("ctx", (2, 77, 64), 2000 + i)
        fn = lambda x, t, c: net(x, torch.tensor(t), encoder_hidden_states=c)[0]
        outs[i] = denoise_loop(fn, lat, ctx, 50, guidance=7.5, timesteps=[981, 961][:steps])
    return outs


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _unet()
    ids = shard_prompts(5, rank, world)
    dist.barrier()
    outs = _run_prompts(net, ids)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)           # bench.py's max-over-ranks timing reduction
    gathered = [None] * world
    dist.all_gather_object(gathered, {k: v.numpy() for k, v in outs.items()})
    if rank == 0:
        q.put((float(t.item()), gathered))
    dist.destroy_process_group()


def test_two_rank_sharded_denoise_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29731
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, gathered = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert tmax == 2.0
    merged = {}
    for d in gathered:
        assert not (set(d) & set(merged))               # disjoint shards
        merged.update(d)
    assert sorted(merged) == [0, 1, 2, 3, 4]
    torch.set_num_threads(2)
    ref = _run_prompts(_unet(), range(5))
    for i in range(5):
        assert torch.allclose(torch.from_numpy(merged[i]), ref[i], atol=1e-5), i
