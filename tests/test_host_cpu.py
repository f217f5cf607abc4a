"""CPU tests of the host logic: C-ABI export check, planner, loader side effects, FP graph, sharding.
No kernel is launched here (there is no GPU in the build container)."""
import json
import os
import re

import pytest
import torch

from dgq_amd import synth
from dgq_amd.plan import plan_act, natural_kperm, KCHUNK, KTILE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------ C ABI
def test_abi_library_loads_and_exports_every_declared_symbol():
    import ctypes
    from dgq_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "dgq_hip.h")).read()
    declared = set(re.findall(r"\b(dgq_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libdgq_hip.so does not export %s" % name
    assert set(_lib.SIGNATURES) | {"dgq_last_error"} == declared
    assert lib.dgq_version() == _lib.ABI_VERSION
    assert isinstance(_lib.last_error(), str)
    # argument validation happens before any launch: bad arguments -> DGQ_EINVAL and a message
    rc = lib.dgq_pack_w4(None, 0, 0, None, 0, 0, None, None)
    assert rc == -1 and "dgq_pack_w4" in _lib.last_error()


def test_shipped_library_is_not_a_diagnostic_build():
    """In-kernel clock stamps (csrc/diag.h) and the wrong-output timing switches of the 256-row kernel (BIG_STAMP, BIG_ABL) exist only
    behind -DDGQ_DIAG, which `make` / __graft_entry__.build() never set: the shipped library exports no dgq_diag_* symbol, and the
    sources refuse BIG_STAMP / BIG_ABL without DGQ_DIAG at compile time."""
    import ctypes
    from dgq_amd import _lib
    lib = _lib.load() if "diag" not in os.path.basename(_lib.LIB_PATH) else ctypes.CDLL(os.path.join(ROOT, "dgq_amd", "csrc", "libdgq_hip.so"))
    for name in ("dgq_diag_fetch_gemm", "dgq_diag_fetch_quant", "dgq_diag_fetch_panel", "dgq_diag_fetch_attn", "dgq_diag_clear_gemm"):
        assert not hasattr(lib, name), "the shipped libdgq_hip.so was built with -DDGQ_DIAG (%s)" % name
    mk = open(os.path.join(ROOT, "dgq_amd", "csrc", "Makefile")).read()
    all_rule = mk.split("diag:")[0]
    assert "-DDGQ_DIAG" not in all_rule.split("# Diagnostic build")[0], "the product build rule must not define DGQ_DIAG"
    big = open(os.path.join(ROOT, "dgq_amd", "csrc", "gemm_wxa8_big.hip")).read()
    assert "#if !defined(DGQ_DIAG) && (BIG_ABL != 0 || BIG_STAMP != 0)" in big and "#error" in big
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "DGQ_DIAG" not in entry and "make diag" not in entry and '"diag"' not in entry


def test_product_has_no_cpu_fallback():
    from dgq_amd.quant import QuantLayer, Scaler
    lin = torch.nn.Linear(32, 16)
    ql = QuantLayer(lin, {"bits": 4, "channel_wise": True, "scaler": Scaler.MINMAX},
                    {"bits": 8, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": True})
    ql.set_quant_state(True, True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ql(torch.randn(2, 4, 32))
    # the product never imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dgq_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


# ------------------------------------------------------------------------------------------ planner
def test_plan_perK_groups_are_chunk_aligned_permutation():
    C, taps, G = 32, 9, 8
    K = C * taps
    d, z = synth._group_params(K, G, 8, "plan-test", 0)
    lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, taps, 8)
    assert lay.mode == "perK" and lay.Kp % KTILE == 0 and lay.n_groups <= G
    real = lay.kperm[lay.kperm >= 0]
    assert sorted(real.tolist()) == list(range(K))                     # a permutation of the reference K order
    for c in range(lay.Kp // KCHUNK):                                  # one (δ,z) per 32-wide chunk
        ks = lay.kperm[c * KCHUNK:(c + 1) * KCHUNK]
        ks = ks[ks >= 0].long()
        assert torch.all(d[ks] == lay.cdelta[c]) and torch.all(z[ks] == lay.czp[c])
    # ksrc decodes to the same (c, tap) as kperm's reference index k = c*taps + tap
    m = lay.kperm >= 0
    cc, tt = lay.ksrc[m] & 0xFFFF, (lay.ksrc[m] >> 24) * 3 + ((lay.ksrc[m] >> 16) & 0xFF)
    assert torch.equal((cc * taps + tt).int(), lay.kperm[m])
    assert int(lay.cflush[-1]) == 1 and int(lay.cflush.sum()) >= lay.n_groups
    # every group's last chunk is flagged: chunks between flags share one scale
    start = 0
    for c in range(lay.Kp // KCHUNK):
        if lay.cflush[c]:
            assert torch.all(lay.cdelta[start:c + 1] == lay.cdelta[c])
            start = c + 1
    assert torch.allclose(lay.kcoef, (d.double() * (128.0 - z.double())))


def test_plan_layout_classification_and_natural_order():
    assert plan_act(torch.tensor(0.1), torch.tensor(3.0), "linear", 64, 1, 8).mode == "scalar"
    lay = plan_act(torch.rand(1, 24, 1) + 0.1, torch.zeros(1, 24, 1), "linear", 64, 1, 8)
    assert lay.mode == "perM" and lay.L == 24
    lay = plan_act(torch.rand(1, 1, 100) + 0.1, torch.zeros(1, 1, 100), "conv", 32, 9, 8)
    assert lay.mode == "perM" and lay.L == 100
    d6 = torch.rand(1, 1, 64) + 0.1
    lay6 = plan_act(d6, torch.full((1, 1, 64), 7.0), "linear", 64, 1, 6)
    assert lay6.mode == "perK" and torch.allclose(lay6.kcoef, d6.reshape(-1).double() * (32.0 - 7.0))
    p = natural_kperm(32, 9)
    assert p.numel() == 384 and p[0] == 0 and p[1] == 9 and p[32] == 1 and int(p[288]) == -1
    with pytest.raises(ValueError):
        plan_act(torch.rand(1, 1, 65) + 0.1, torch.zeros(1, 1, 65), "linear", 64, 1, 8)


# ------------------------------------------------------------------------------------------ FP graph + oracle structure
@pytest.mark.parametrize("arch", ["tiny"])
def test_fp_graph_matches_oracle_structure(arch):
    """Our table-driven UNet (FP) == the oracle's functional graph with quantisation off (tiny arch, CPU)."""
    from dgq_amd.diffusers_rewrite import UNet2DConditionModel
    from oracle import dgq_oracle as orc
    net = UNet2DConditionModel(arch).eval()
    synth.load_synth_weights(net, arch, 0)
    inp = synth.synth_inputs(arch, 2, 1, 16)
    with torch.no_grad():
        y = net(inp["sample"], torch.tensor(981), encoder_hidden_states=inp["encoder_hidden_states"])[0]
    w = synth.synth_weight_ckpt(arch, 4, 0)
    om = orc.OracleModel({"weight": w}, orc.OracleConfig(arch, use_wq=False, use_aq=False), synth.synth_state_dict(arch, 0))
    ref = om.forward(inp["sample"], 981, inp["encoder_hidden_states"])
    assert (y - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


def test_synthetic_ckpt_schema_matches_reference_schema(tmp_path):
    """F1: key set / shapes / dtypes of our writer == what the reference's own save->merge->load path accepted
    (tests/golden/f1_f6_schema_sd.json was produced from a ckpt loaded by the REAL load_cali_model)."""
    f = os.path.join(ROOT, "tests", "golden", "f1_f6_schema_sd.json")
    if not os.path.exists(f):
        pytest.skip("schema fixture missing")
    g = json.load(open(f))
    res = g["schema_res"]
    recs = synth.enumerate_act_quantizers("sd", 2, res)
    act = synth.synth_act_slot("sd", 8, 16, 0, 0, 2, res, start_peak=True, recs=recs)
    assert len(act) == g["n_act_keys"] == 752
    for k, (shape, dtype) in g["act0"].items():
        assert list(act[k].shape) == shape and str(act[k].dtype) == dtype, k
    # SD1.4: 1250 weight keys = 686 module params + 282 x 2 quantizer params
    assert g["n_weight_keys"] == 1250
    with torch.device("meta"):
        from dgq_amd.diffusers_rewrite import UNet2DConditionModel
        skel = UNet2DConditionModel("sd")
    n_q = sum(1 for m in skel.modules() if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)))
    assert n_q == 282 and len(skel.state_dict()) == 686


# ------------------------------------------------------------------------------------------ loader logic
def test_load_cali_model_side_effects_cpu(tmp_path):
    """load_cali_model host logic on the tiny arch (no forward, init_forward=False): weights/quantizer params
    land where the reference puts them (F6): per-channel wqtizer params from the ckpt, use_group_num flags,
    conv_in/conv_out floating point, time-aware tables for every slot, AdaRound α."""
    from dgq_amd.diffusers_rewrite import UNet2DConditionModel
    from dgq_amd.quant import QuantModel, load_cali_model, QuantLayer, Scaler, AdaRoundQuantizer
    from dgq_amd.runtime import quant_params
    arch, res = "tiny", 16
    path = str(tmp_path / "ck.pth")
    synth.write_cali_ckpt(path, arch, 4, 8, 8, num_slots=3, seed=0, batch=2, res=res, start_peak=True, adaround=True)
    ck = torch.load(path)
    net = UNet2DConditionModel(arch)
    synth.load_synth_weights(net, arch, 0)
    wq, aq, sm = quant_params(Scaler, 4, 8, True, True, True, True)
    qnn = QuantModel(net, wq, aq, sm).eval()
    load_cali_model(qnn, None, use_aq=True, path=path, time_aware_aqtizer=True, num_inference_steps=50, use_group=True,
                    init_forward=False)
    qnn.disable_out_quantization()
    m = qnn.model
    assert m.conv_in.use_wq is False and m.conv_in.disable_aq is True and m.conv_out.use_wq is False
    n_layers = n_grouped = 0
    for name, mod in qnn.named_modules():
        if isinstance(mod, QuantLayer) and name not in ("model.conv_in", "model.conv_out"):
            n_layers += 1
            assert isinstance(mod.wqtizer, AdaRoundQuantizer)
            assert torch.equal(mod.wqtizer.delta.data, ck["weight"][name + ".wqtizer.delta"])
            assert torch.equal(mod.wqtizer.alpha.data, ck["weight"][name + ".wqtizer.alpha"])
            assert torch.equal(mod.w.data, ck["weight"][name + ".w"])
            assert set(mod._act_tables) == {0, 1, 2}
            d0 = ck["act_0"][name + ".aqtizer.delta"]
            assert mod.use_group_num == (d0.dim() > 0 or any(ck["act_%d" % s][name + ".aqtizer.delta"].dim() > 0
                                                             for s in (1, 2)))
            n_grouped += int(mod.use_group_num)
            assert mod.use_wq and mod.use_aq
    assert n_layers > 30 and 0 < n_grouped < n_layers
    assert qnn.time_aware["slots"] == {0, 1, 2} and len(qnn.time_aware["attn"]) == 6 * 2 * 3   # 6 transformer blocks x 2 attn x qkv
    # slot formula of calibration.py:301-304 drives activate_slot
    qnn.activate_slot(2)
    q, tab = qnn.time_aware["attn"][0]
    assert torch.equal(q.delta.data, tab[2][0])
    with pytest.raises(KeyError):
        qnn.activate_slot(7)
    # calibration-time API is implemented since round 2: flags propagate like quant_model.py:135-149
    from dgq_amd.quant import quant_block
    qnn.set_group_num(16)
    assert all(m.aqtizer.group_num == 16 and m.use_group_num for m in qnn.modules() if isinstance(m, QuantLayer))
    assert quant_block.FUSION is False
    qnn.done_group_num(16, "minmax")                       # nothing recorded: every quantizer leaves calibration mode
    assert all(m.aqtizer.group_num == -1 for m in qnn.modules() if isinstance(m, QuantLayer)) and quant_block.FUSION is True


# ------------------------------------------------------------------------------------------ sharding / scheduler
def test_shard_prompts_and_ddim():
    from dgq_amd.runtime import shard_prompts, slot_for_timestep, DDIMScheduler
    assert shard_prompts(64, 3, 8) == list(range(24, 32))
    allp = sum((shard_prompts(10, r, 4) for r in range(4)), [])
    assert allp == list(range(10))
    sch = DDIMScheduler(50)
    assert sch.timesteps[:3] == [981, 961, 941] and sch.timesteps[-1] == 1
    assert [slot_for_timestep(t, 50) for t in sch.timesteps] == list(range(50))
    from oracle import dgq_oracle as orc
    o = orc.DDIM(50)
    x, e = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    for t in (981, 501, 1):
        assert torch.allclose(sch.step(e, t, x), o.step(e, t, x), atol=1e-6)


def test_as_f32_cache_lives_on_the_tensor():
    """ops.as_f32: fp32 copies of half parameters are cached on the Parameter object itself (never by address), are
    refreshed after an in-place update, and fp32 parameters are passed through without a copy."""
    import torch
    from dgq_amd import ops
    p = torch.nn.Parameter(torch.randn(8).half())
    a, b = ops.as_f32(p), ops.as_f32(p)
    assert a is b and a.dtype == torch.float32
    with torch.no_grad():
        p.mul_(2)
    c = ops.as_f32(p)
    assert c is not a and torch.allclose(c, p.detach().float())
    q = torch.nn.Parameter(torch.randn(8))
    assert ops.as_f32(q).data_ptr() == q.data_ptr()


# ---------------------------------------------------------------------------------------------- bench.py launcher
def test_bench_gpus_n_fails_loudly_without_n_devices():
    """`python bench.py --gpus 2` must start 2 ranks or fail — never report a 1-GPU run as n_gpus 2.
    In this container no GPU is visible, so the parent refuses before touching the GPU."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2 requested" in p.stderr and "n_gpus" not in p.stdout


def test_bench_world_size_must_equal_gpus(monkeypatch):
    import bench
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert bench.main(["--gpus", "2"]) == 2            # a torchrun world that disagrees with --gpus is refused


def test_bench_launch_command_is_one_rank_per_gpu():
    import bench
    a = bench.parse_args(["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "c5"])
    cmd = bench.launch_command(a, ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "c5"], 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-8:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "c5"]
    assert bench.CONFIGS["c5"]["prompts"] == 8 and bench.CONFIGS["c5"]["cfg"]["G"] == 1 and bench.CONFIGS["c5"]["cfg"]["abits"] == 6
    assert bench.CONFIGS["c4"]["arch"] == "sdxl" and bench.CONFIGS["c4"]["res"] == 128


def _abel_total(cd, cf, P, c0, c1, wvk):
    """What one workgroup of csrc/gemm_wxa8.hip adds up over the chunks [c0, c1) of its K range: WVK wave sequences (every
    WVK-th chunk of each 4-chunk K tile), each with a running total T that a flush multiplies by δ_c − δ_next and a
    cflush == 2 mark on the last chunk of a K tile clears behind that tile (coefficient of the wave's last chunk: δ_c)."""
    from fractions import Fraction
    tot = Fraction(0)
    mych = 4 // wvk
    nk = (c1 - c0) // 4
    for kq in range(wvk):
        seq = [c0 + (i // mych) * 4 + (i % mych) * wvk + kq for i in range(nk * mych)]
        T = 0
        for i, g in enumerate(seq):
            T += int(P[g])
            last = i == len(seq) - 1
            clr = (not last) and (i % mych == mych - 1) and int(cf[g // 4 * 4 + 3]) == 2
            coef = cd[g] - (0 if (last or clr) else cd[seq[i + 1]])
            tot += coef * T
            if clr:
                T = 0
    return tot


def test_summation_by_parts_matches_group_sums_on_planned_tables():
    """The GEMM keeps running int32 totals: a flush adds (δ_c − δ_next)·T_c (csrc/gemm_wxa8.hip).  On the 32-wide chunk tables
    plan_act produces — every chunk of a group carries the group's δ, padding chunks carry zero codes — that equals
    Σ_g δ_g·P_g exactly (integer partial sums, exact rational arithmetic): for any K-tile split where each split restarts
    its total at zero, for one or two wave sequences per workgroup (WVK), and with the clear marks of mark_clears."""
    from fractions import Fraction
    from dgq_amd.plan import mark_clears, seg_limit
    g = torch.Generator().manual_seed(5)
    for C, taps, G in ((320, 1, 16), (64, 9, 8), (1280, 1, 16)):
        K = C * taps
        d, z = synth._group_params(K, G, 8, "abel", 0)
        shape = (1, 1, -1) if taps == 1 else (1, -1, 1)
        lay = plan_act(d.view(*shape), z.view(*shape), "linear" if taps == 1 else "conv", C, taps, 8)
        assert KCHUNK == 32 and lay.Kp % KTILE == 0
        assert lay.Kp <= K + lay.n_groups * (KCHUNK - 1) + KTILE - 1       # at most 31 padding codes per group + the K tile
        nch = lay.Kp // KCHUNK
        valid = (lay.kperm.view(nch, KCHUNK) >= 0)
        P = torch.randint(-50000, 50000, (nch,), generator=g)
        P = torch.where(valid.any(dim=1), P, torch.zeros_like(P))          # all-padding chunks contribute nothing
        cd = [Fraction(float(x)) for x in lay.cdelta]
        ref = sum(cd[c] * int(P[c]) for c in range(nch))
        marks = [lay.cflush, mark_clears(lay.cflush, 8, 4), mark_clears(lay.cflush, 8, 8)]
        forced = lay.cflush.clone()
        forced[3::8] = 2                                                   # a clear behind every other K tile, inside groups too
        marks.append(forced)
        tiles = nch // 4
        for cf in marks:
            for wvk in (1, 2):
                for splits in (1, 2, 3):
                    per = -(-tiles // splits)
                    tot = Fraction(0)
                    for s0 in range(0, tiles, per):
                        tot += _abel_total(cd, cf, P, 4 * s0, min(nch, 4 * (s0 + per)), wvk)
                    assert tot == ref, (C, taps, G, splits, wvk)
    # clear marks: uniform segments of whole K tiles, W4A8 2176 codes, W8A8 1024, never longer than what keeps |T| < 2^22 (W4: the
    # kernels read float(T) off the bits of a biased total) or <= 2^24 (W8: by conversion)
    assert seg_limit(8, 4) == 2176 and seg_limit(8, 8) == 1024 and seg_limit(6, 4) == 8704
    for ab, wb in ((8, 4), (8, 8), (6, 4), (4, 4)):
        worst = seg_limit(ab, wb) * (1 << (ab - 1)) * (15 if wb == 4 else 128)
        assert worst < (1 << 22) if wb == 4 else worst <= (1 << 24)
    # the bit trick itself, on the host: bits(1.5·2^23) + T read as a float, minus 1.5·2^23, is float(T) for every |T| < 2^22
    import numpy as np
    T = np.concatenate([np.array([0, 1, -1, (1 << 22) - 1, -(1 << 22) + 1], dtype=np.int64),
                        np.random.default_rng(0).integers(-(1 << 22) + 1, 1 << 22, 100000)])
    biased = (np.int64(0x4B400000) + T).astype(np.int32).view(np.float32)
    assert np.array_equal(biased - np.float32(12582912.0), T.astype(np.float32))
    cf = mark_clears(torch.zeros(100, dtype=torch.uint8), 8, 8)
    assert cf.tolist().count(2) == 3 and int(cf[31]) == 2 and int(cf[63]) == 2 and int(cf[95]) == 2


def test_bench_two_rank_dry_run_under_gloo(tmp_path):
    """The N > 1 path of bench.py executed end to end WITHOUT GPUs: `python bench.py --gpus 2` under
    DGQ_BENCH_BACKEND=gloo starts torch.distributed.run as a child with two CPU ranks; rank 0 writes the synthetic ckpt, rank 1
    waits at the barrier and reads the weights back memory-mapped, each rank denoises its own rank-seeded prompt, the timed
    windows are bracketed by barriers and MAX-reduced, and rank 0 prints ONE JSON line with n_gpus = 2.  The model is the FP tiny
    UNet (the quantized kernels need a GPU): the line says so (``dry_run``) and is not a measurement."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DGQ_BENCH_BACKEND="gloo", DGQ_BENCH_CKPT_DIR=str(tmp_path), OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "2"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and "dry_run" in d
    assert d["value"] > 0 and len(d["windows"]["ms_per_step_all"]) == 2
    assert len(d["load"]["model_ready_s_per_rank"]) == 2 and all(v > 0 for v in d["load"]["model_ready_s_per_rank"])
    assert d["config"]["parallelism"].startswith("replicas x2")
    files = [f for f in os.listdir(tmp_path) if f.endswith(".pth")]
    assert len(files) == 1                                   # one ckpt, written by rank 0 only


def test_skip_concatenation_in_place_host_logic():
    """unet._CatInPlace + ops.OutputRedirect on CPU tensors (pure host logic): a producer whose final layer call takes the redirect leaves
    its rows in the concatenation buffer and `join` returns the buffer itself (no torch.cat); anything that does not take it — or a skip
    tensor modified afterwards — falls back to torch.cat with the same result.  The kernels' side of it (dgq_gemm_extra_t.y2, strided y):
    tests/test_gpu_kernels.py::test_output_redirect_into_a_concatenation_buffer."""
    import types
    from dgq_amd import ops
    from dgq_amd.diffusers_rewrite import unet as U
    B, H, W = 2, 4, 4
    # two up resnets: the first pops skip 1 (C2 = 6, with h of C1 = 10), the second pops skip 0 (C2 = 4, C1 = 8)
    mk = lambda c: types.SimpleNamespace(norm1=types.SimpleNamespace(num_channels=c))
    fake = types.SimpleNamespace(up_blocks=[types.SimpleNamespace(resnets=[mk(16), mk(12)])])
    g = torch.Generator().manual_seed(0)

    def layer(value, take):
        """a module's final layer call: stores `value` [B, C, H, W] where a pending redirect asks for it (when `take`)"""
        M, N = B * H * W, value.shape[1]
        rows = value.permute(0, 2, 3, 1).reshape(M, N)
        out, out2 = ops.take_redirect(M, N, value.dtype) if take else (None, None)
        if out is not None:
            out.copy_(rows)
            return out.view(B, H, W, N).permute(0, 3, 1, 2)
        if out2 is not None:
            out2.copy_(rows)
        return value.contiguous(memory_format=torch.channels_last)

    for take_skip, take_h, touch in ((True, True, False), (True, False, False), (False, True, False), (True, True, True)):
        cat = U._CatInPlace(fake, torch.float32, torch.device("cpu"))
        s0, s1 = torch.randn(B, 4, H, W, generator=g), torch.randn(B, 6, H, W, generator=g)
        h1, h0 = torch.randn(B, 10, H, W, generator=g), torch.randn(B, 8, H, W, generator=g)
        k0 = cat.produce_skip(lambda: layer(s0, take_skip), tuple(s0.shape))
        k1 = cat.produce_skip(lambda: layer(s1, take_skip), tuple(s1.shape))
        assert ops.pending_redirect() is None
        a = cat.produce_h(lambda: layer(h1, take_h))
        if touch:
            k1.add_(0.0)                                   # an in-place op on the skip tensor: the copy in the buffer is no longer trusted
        x1 = cat.join(a)
        assert torch.equal(x1, torch.cat([h1, k1], dim=1))
        in_place = take_skip and take_h and not touch
        assert (x1.data_ptr() == a.data_ptr()) == in_place
        b = cat.produce_h(lambda: layer(h0, take_h))
        x0 = cat.join(b)
        assert torch.equal(x0, torch.cat([h0, k0], dim=1)) and not cat.entries
    # a redirect whose views do not have the layer's output shape is left alone
    ops.set_redirect(ops.OutputRedirect(out=torch.empty(5, 3)))
    try:
        assert ops.take_redirect(5, 4, torch.float32) == (None, None) and not ops.pending_redirect().taken
        assert ops.take_redirect(5, 3, torch.float32)[0] is not None and ops.pending_redirect().taken
        assert ops.take_redirect(5, 3, torch.float32) == (None, None)          # taken once
    finally:
        ops.set_redirect(None)
