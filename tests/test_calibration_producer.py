"""SURVEY.md §8(f)-1 — the DGQ activation-calibration PRODUCER (quant/calibration_group_quantization.py:44-129,
quant/quant_layer.py:301-429) against tests/golden/f8_calibration_mini.pt: the REAL reference's cali_model_aq run on a
two-level UNet composed of the reference's own block classes (make_golden.py `calib`; ARCH['mini'] is the same model in
this package).  The host half (spread-based axis choice, K-Means(G, random_state=0), per-cluster ranges -> δ, z) is
checked exactly from the reference's recorded ranges on CPU; the whole producer (statistics kernel, calibration forwards
on the integer path, writer) on the GPU.  scikit-learn is unpinned: the golden file records its version."""
import os

import pytest
import torch

from dgq_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f8_calibration_mini.pt")


def calib_data(c):
    xs = synth.named_randn("calib_x", (c["n"], 4, c["res"], c["res"]), 3)
    ts = torch.tensor([c["ts"][i // c["interval"]] for i in range(c["n"])], dtype=torch.int64)
    ctx = synth.named_randn("calib_ctx", (c["n"], 77, 768), 4)
    return xs, ts, ctx


def test_grouping_from_reference_ranges_is_exact():
    """done_group_num's host arithmetic: same axis, same δ, same z as the reference for every one of the 2 x 140 grouped
    quantizers, given the (min, max) vectors the reference itself recorded."""
    from dgq_amd.quant.quant_layer import group_params_from_ranges
    g = torch.load(GOLD)
    c = g["meta"]
    n = 0
    for t, qs in g["ranges"].items():
        for name, r in qs.items():
            d, z, in_wise = group_params_from_ranges(r["in_min"], r["in_max"], r["out_min"], r["out_max"], c["G"], c["mode"],
                                                     2 ** c["abits"])
            assert d.shape == r["delta"].shape, (name, d.shape, r["delta"].shape)       # the axis choice
            assert torch.equal(d, r["delta"]), (t, name, (d - r["delta"]).abs().max())
            assert torch.equal(z, r["zero_point"]), (t, name)
            assert torch.unique(torch.stack([d.flatten(), z.flatten()], 1), dim=0).shape[0] <= c["G"]
            n += 1
    assert n == 2 * 140


def test_mini_arch_matches_the_reference_composition():
    """ARCH['mini'] has the state-dict keys the reference's block classes produce (the golden act tables are keyed by them)."""
    g = torch.load(GOLD)
    from dgq_amd.diffusers_rewrite import UNet2DConditionModel
    with torch.device("meta"):
        net = UNet2DConditionModel("mini")
    paths = {n for n, m in net.named_modules() if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d))}
    for k in g["act"]["act_0"]:
        p = k[len("model."):].rsplit(".", 2)[0]                       # '<path>.aqtizer' or '<blk>.attnN'
        assert p in paths or any(q.startswith(p + ".") for q in paths), k


@pytest.mark.gpu
def test_minmax_statistics_kernel_vs_torch():
    from dgq_amd import ops
    for shape, dt in (((2, 37, 320), torch.float32), ((3, 8, 15, 40), torch.float32), ((4, 1152, 64), torch.float16),
                      ((1, 4096, 77), torch.bfloat16)):
        x = torch.randn(shape, generator=torch.Generator().manual_seed(len(shape))).to("cuda", dt)
        C = shape[-1]
        rmin, rmax, cmin, cmax = ops.minmax_rows_cols(x.view(-1, C))
        x2 = x.float().view(-1, C)
        assert torch.equal(rmin, x2.min(dim=1)[0]) and torch.equal(rmax, x2.max(dim=1)[0])
        assert torch.equal(cmin, x2.min(dim=0)[0]) and torch.equal(cmax, x2.max(dim=0)[0])


@pytest.mark.gpu
def test_act_group_quant_producer_vs_reference_golden(tmp_path):
    """The whole producer on the GPU: reset -> scalar self-init forward -> set_group_num -> recording forwards ->
    done_group_num -> act_<t> dict, then the file is LOADED by load_cali_model and run.  Against the reference's result on
    the same model / data / numpy seed: identical key set and shapes; the calibration forwards themselves are quantized
    (scalar δ) and differ from the reference's CPU run at rounding level, so ranges agree closely, not bitwise."""
    import numpy as np
    from dgq_amd.diffusers_rewrite import UNet2DConditionModel
    from dgq_amd.quant import QuantModel, Scaler, load_cali_model, act_group_quant
    from dgq_amd.quant.quant_block import QuantBasicTransformerBlock
    g = torch.load(GOLD)
    c = g["meta"]
    unet = UNet2DConditionModel("mini")
    synth.load_synth_weights(unet, "mini", 0)
    wq = {"bits": c["wbits"], "channel_wise": True, "scaler": Scaler.MINMAX}
    aq = {"bits": c["abits"], "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": True}
    sm = {"softmax_a_bit": c["abits"], "t2i_log_quant": True, "t2i_real_time": True, "t2i_start_peak": True, "log_max_1": False}
    qnn = QuantModel(model=unet, wq_params=wq, aq_params=aq, softmax_aq_params=sm, aq_mode=[1, 0], tib_recon=False).cuda().eval()
    wpath = str(tmp_path / "w.pth")
    torch.save(synth.synth_weight_ckpt("mini", c["wbits"], 0), wpath)
    xs, ts, ctx = calib_data(c)
    load_cali_model(qnn, (xs[:1], ts[:1], ctx[:1]), use_aq=False, path=wpath)
    for m in qnn.modules():                                  # src/quantize_act.py flow: attentions quantise during calibration
        if isinstance(m, QuantBasicTransformerBlock):
            m.attn1.use_aq = m.attn2.use_aq = True
    np.random.seed(c["np_seed"])
    out = str(tmp_path / "act.pth")
    act = act_group_quant("sd", qnn, (xs, ts, ctx), path=out, group_num=c["G"], interval=c["interval"], group_mode=c["mode"])
    saved = torch.load(out)
    assert sorted(saved) == sorted(g["act"]) == ["act_0", "act_1"]
    # the statistics that entered the last interval's grouping, against the reference's (same data, same batches)
    rel_rng, in_order = [], []
    for name, m in qnn.model.named_modules():
        if hasattr(m, "last_ranges") and name in g["ranges"][1]:
            r = g["ranges"][1][name]
            worst_q = 0.0
            for mine, ref in zip(m.last_ranges, (r["in_min"], r["in_max"], r["out_min"], r["out_max"])):
                assert mine.shape == ref.shape, name
                e = ((mine - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()
                rel_rng.append(e)
                worst_q = max(worst_q, e)
            in_order.append((name, worst_q))
    print("first quantizers in module order:", ["%s %.2g" % (n.split(".", 2)[-1], e) for n, e in in_order[:8]])
    # the calibration forwards are themselves quantized (scalar 8-bit activations): like every fake-quant graph they
    # amplify rounding differences with depth, so the statistics agree tightly where the graph starts and
    # at the few-percent level further down
    # the two quantizers the graph EXECUTES first (named_modules lists attentions before resnets, like the reference's classes)
    first = [e for n, e in in_order if n.endswith("down_blocks.0.resnets.0.conv1.aqtizer") or n.endswith("down_blocks.0.resnets.0.conv2.aqtizer")]
    assert len(first) == 2 and max(first) < 1e-4, first
    rel_rng.sort()
    print("range vectors vs reference: n=%d median rel-max %.3g, 90%% %.3g, worst %.3g"
          % (len(rel_rng), rel_rng[len(rel_rng) // 2], rel_rng[int(0.9 * len(rel_rng))], rel_rng[-1]))
    same_axis = close = total = 0
    worst = 0.0
    elem_same = elem_total = 0
    for t in saved:
        assert sorted(saved[t]) == sorted(g["act"][t]), set(saved[t]) ^ set(g["act"][t])
        for k, v in saved[t].items():
            ref = g["act"][t][k]
            if not k.endswith(".delta"):
                continue
            total += 1
            if v.shape == ref.shape:
                same_axis += 1
                rel = ((v.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
                worst = max(worst, rel)
                close += rel < 5e-2
                elem_same += int(((v.float() - ref.float()).abs() <= 2e-2 * ref.float().abs()).sum())
                elem_total += v.numel()
            assert v.numel() == 1 or torch.unique(v).numel() <= c["G"]
    print("producer vs reference: %d quantizers, same axis %d, every δ within 5%% for %d (worst rel %.3g); "
          "channels whose δ agrees within 2%%: %d of %d" % (total, same_axis, close, worst, elem_same, elem_total))
    assert len(rel_rng) >= 130 and rel_rng[len(rel_rng) // 2] < 6e-2
    assert same_axis >= 0.95 * total and elem_same >= 0.6 * elem_total
    # the produced file is a valid cali_ckpt: merge with the weights (results/merge.py) and run it time-aware
    merged = dict(saved)
    merged["weight"] = torch.load(wpath)
    mpath = str(tmp_path / "merged.pth")
    torch.save(merged, mpath)
    load_cali_model(qnn, (xs[:1], ts[:1], ctx[:1]), use_aq=True, path=mpath, time_aware_aqtizer=True, num_inference_steps=2,
                    use_group=True)
    qnn.disable_out_quantization()
    with torch.no_grad():
        y = qnn(xs[:2].cuda(), torch.tensor(901), ctx[:2].cuda())[0]
    assert torch.isfinite(y).all()


@pytest.mark.gpu
def test_quantizer_statistics_vs_reference_golden():
    """VERDICT r2 item 6b — the statistics half of the producer pinned quantizer by quantizer, for every tensor rank the graph
    feeds a quantizer (3-D Linear, unfolded conv, 4-D attention q / k, 2-D): tests/golden/f8b_quantizer_statistics.pt = the
    REFERENCE's UniformAffineQuantizer driven as cali_model_aq drives it (scalar self-init, group_num = G, three batches,
    done_group_num) on name-keyed inputs.  On the GPU (dgq_minmax_rows_cols + the folds of observe / done_group_num):
    the per-batch (min, max) vectors, the folded ranges, δ and z are BIT-IDENTICAL (minima / maxima have no rounding); the
    scalar EMA branch (group_num = 1) agrees to an fp32 ulp.  Together with test_grouping_from_reference_ranges_is_exact and
    the statistics-kernel test this pins every link between a quantizer's input and its table."""
    from dgq_amd.quant import Scaler
    from dgq_amd.quant.quant_layer import UniformAffineQuantizer
    from tests.golden.recipes import qstat_batches
    g = torch.load(os.path.join(os.path.dirname(GOLD), "f8b_quantizer_statistics.pt"))
    G = g["meta"]["G"]
    for name, rec in g["cases"].items():
        xs = [x.cuda() for x in qstat_batches(name, tuple(rec["shape"]))]
        q = UniformAffineQuantizer(bits=8, channel_wise=False, scaler=Scaler.MINMAX, leaf_param=True)
        q.observe(xs[0])                                     # scalar self-initialisation
        assert torch.equal(q.delta.data.cpu().reshape(()), rec["init_delta"].reshape(())), name
        assert torch.equal(torch.as_tensor(q.zero_point).cpu().reshape(()), rec["init_zp"].reshape(())), name
        q.group_num = G
        for i, x in enumerate(xs):
            q.observe(x)
            if rec["per_batch"]:
                mine = q.min_max_per_in_channel[-1] + q.min_max_per_out_channel[-1]
                for a, b in zip(mine, rec["per_batch"][i]):
                    assert torch.equal(a.cpu(), b), (name, i)
        q.done_group_num(G, "minmax")
        if "ranges" in rec:
            for a, b in zip(q.last_ranges, rec["ranges"]):
                assert torch.equal(a, b), name
        assert q.delta.data.shape == rec["delta"].shape, (name, q.delta.data.shape, rec["delta"].shape)
        assert torch.equal(q.delta.data.cpu(), rec["delta"]) and torch.equal(torch.as_tensor(q.zero_point).cpu(), rec["zero_point"]), name
        q2 = UniformAffineQuantizer(bits=8, channel_wise=False, scaler=Scaler.MINMAX, leaf_param=True)
        q2.observe(xs[0])
        q2.group_num = 1
        for x in xs[1:]:
            q2.observe(x)
        xmin, xmax, d, z = rec["ema"]
        assert abs(float(q2.x_min) - float(xmin)) <= 1e-6 * abs(float(xmin)) and abs(float(q2.x_max) - float(xmax)) <= 1e-6 * abs(float(xmax)), name
        assert abs(float(q2.delta) - float(d)) <= 1e-6 * float(d) and float(torch.as_tensor(q2.zero_point)) == float(z), name
