"""SURVEY.md §8(f)-4 — weight PTQ (BRECQ / AdaRound reconstruction: quant/reconstruction.py:13-200,
quant/reconstruction_util.py:13-198, quant/calibration.py:100-206, quant/adaptive_rounding.py:39-70) against
tests/golden/f9_weight_ptq_mini.pt: the REAL reference's ``cali_model`` run on the mini UNet composed of the reference's own
block classes (make_golden.py `recon`), plus known-answer vectors of the soft quantiser and the rounding regulariser with
the gradients torch autograd derives for them in the reference.

CPU: the oracle's restatement of those formulas against the known answers, the temperature schedule, the reconstruction
schedule (which layers / blocks, in which order) and the C-ABI exports.  GPU: the HIP kernels against the oracle, then the
whole driver against the reference's trajectory."""
import os

import numpy as np
import pytest
import torch

from dgq_amd import synth
from oracle import dgq_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f9_weight_ptq_mini.pt")


def recon_data(c):
    xs = synth.named_randn("recon_x", (c["n"], 4, c["res"], c["res"]), 5)
    ts = torch.tensor([c["ts"][i * len(c["ts"]) // c["n"]] for i in range(c["n"])], dtype=torch.int64)
    ctx = synth.named_randn("recon_ctx", (c["n"], 77, 768), 6)
    return xs, ts, ctx


def build_mini(c, device="cpu"):
    from dgq_amd.diffusers_rewrite import UNet2DConditionModel
    from dgq_amd.quant import QuantModel, Scaler
    unet = UNet2DConditionModel("mini")
    synth.load_synth_weights(unet, "mini", 0)
    wq = {"bits": c["wbits"], "channel_wise": True, "scaler": Scaler.MINMAX, "leaf_param": False}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": False}
    sm = {"softmax_a_bit": 8, "t2i_log_quant": True, "t2i_real_time": True, "t2i_start_peak": True, "log_max_1": False}
    qnn = QuantModel(model=unet, wq_params=wq, aq_params=aq, softmax_aq_params=sm, aq_mode=[1, 0], tib_recon=False)
    return qnn.to(device).eval()


# ------------------------------------------------------------------------------------------------ CPU
def test_oracle_soft_quantiser_and_regulariser_match_reference_known_answers():
    g = torch.load(GOLD)
    for name, k in g["kat"].items():
        alpha = k["alpha"].clone().requires_grad_(True)
        out = orc.adaround_soft(k["w"], k["delta"], k["zero_point"], alpha, k["bits"])
        assert torch.equal(out, k["out"]), name
        assert torch.equal(orc.adaround_soft_target(alpha.detach()), k["soft_tgt"])
        (out * k["gout"]).sum().backward()
        assert torch.equal(alpha.grad, k["galpha"]), name
        for b, r in k["reg"].items():
            alpha.grad = None
            v = orc.adaround_round_loss(alpha, b)
            (0.01 * v).backward()
            assert torch.equal(v.detach(), r["value"]) and torch.equal(alpha.grad, r["galpha"]), (name, b)


def test_temperature_schedule_matches_reference():
    from dgq_amd.quant.reconstruction_util import LinearTempDecay
    g = torch.load(GOLD)
    for (iters, warm), pts in g["sched"].items():
        lf = LinearTempDecay(t_max=iters, rel_start_decay=warm, start_b=20, end_b=2)
        for t, b in pts:
            assert lf(t) == b, (iters, warm, t)


def test_reconstruction_schedule_matches_reference():
    """Which layers are reconstructed alone, which as blocks, in which order, and where the cache stays on the device —
    identical to the order the reference's recon_model visits (calibration.py:113-141)."""
    from dgq_amd.quant.calibration import recon_targets
    g = torch.load(GOLD)
    qnn = build_mini(g["meta"])
    mine = [(kind, name[len("unet."):], keep) for kind, name, _, keep in recon_targets(qnn, "unet")]
    assert mine == [tuple(o) for o in g["order"]]


def test_loss_func_reconstruction_terms_on_cpu():
    """lp_loss and the Fisher-weighted forms are plain tensor formulas (device-agnostic): against hand evaluation."""
    from dgq_amd.quant.reconstruction_util import LossFunc, RLOSS, lp_loss
    gen = torch.Generator().manual_seed(3)
    p, t, gr = (torch.randn(4, 6, 5, 5, generator=gen) for _ in range(3))
    assert torch.allclose(lp_loss(p, t, 2.0), ((p - t) ** 2).sum(1).mean())
    dummy = torch.nn.Identity()
    lf = LossFunc(dummy, round_loss=RLOSS.NONE, rec_loss=RLOSS.FISHER_DIAG, max_count=10)
    assert torch.allclose(lf(p, t, gr), ((p - t).pow(2) * gr.pow(2)).sum(1).mean())
    lf = LossFunc(dummy, round_loss=RLOSS.NONE, rec_loss=RLOSS.FISHER_FULL, max_count=10)
    a, ga = (p - t).abs(), gr.abs()
    assert torch.allclose(lf(p, t, gr), ((a * ga).sum((1, 2, 3)).view(-1, 1, 1, 1) * a * ga).mean() / 100)
    assert lf.count == 1


def test_weight_ptq_has_no_cpu_path():
    from dgq_amd.quant.adaptive_rounding import AdaRoundQuantizer
    from dgq_amd.quant.quant_layer import UniformAffineQuantizer, Scaler
    w = torch.randn(4, 8)
    uq = UniformAffineQuantizer(bits=4, channel_wise=True, scaler=Scaler.MINMAX)
    uq.init_from(w)
    q = AdaRoundQuantizer(uq, w)
    q.soft_tgt = True
    with pytest.raises(RuntimeError):
        q(w)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_adaround_kernels_vs_oracle_and_reference_known_answers():
    from dgq_amd import ops
    g = torch.load(GOLD)
    dev = "cuda"
    cases = [(n, k["w"], k["delta"], k["zero_point"], k["alpha"], k["bits"], k["gout"]) for n, k in g["kat"].items()]
    gen = torch.Generator().manual_seed(11)
    for N, C, kh, bits in ((64, 64, 3, 4), (320, 77, 1, 4), (13, 1280, 1, 8)):           # conv / linear shapes, W4 / W8
        w = torch.randn(N, C, kh, kh, generator=gen) * 0.2 if kh > 1 else torch.randn(N, C, generator=gen) * 0.2
        d, z = orc.minmax_channel(w, bits)
        alpha = torch.randn(w.shape, generator=gen) * 4
        cases.append(("rand%d" % N, w, d, z, alpha, bits, torch.randn(w.shape, generator=gen)))
    for name, w, d, z, alpha, bits, gout in cases:
        a_ref = alpha.clone().requires_grad_(True)
        out_ref = orc.adaround_soft(w, d, z, a_ref, bits)
        (out_ref * gout).sum().backward()
        a = alpha.to(dev).requires_grad_(True)
        out = ops.adaround_soft(w.to(dev), d.to(dev), z.to(dev), a, bits)
        (out * gout.to(dev)).sum().backward()
        # expf / the division inside sigmoid differ from the host libm by an ulp (h moves by ~1e-7), and floor + h + z is
        # rounded at the magnitude of the code (up to 2^bits): half an ulp of that, times δ
        assert torch.allclose(out.cpu(), out_ref.detach(), rtol=0, atol=1e-6 * float(d.max()) * 2 ** bits), name
        assert torch.allclose(a.grad.cpu(), a_ref.grad, rtol=2e-5, atol=1e-9), name
        assert (a.grad.cpu() == 0).eq(a_ref.grad == 0).float().mean() > 0.9999, name         # the clamp masks
        for b in (20.0, 7.3, 2.0):
            a_ref.grad = None
            v_ref = orc.adaround_round_loss(a_ref, b)
            (0.01 * v_ref).backward()
            a.grad = None
            v = ops.adaround_reg(a, b)
            (0.01 * v).backward()
            assert abs(float(v.detach()) - float(v_ref.detach())) <= 2e-5 * abs(float(v_ref.detach())) + 1e-4, (name, b)
            assert torch.allclose(a.grad.cpu(), a_ref.grad, rtol=1e-4, atol=1e-8), (name, b)


@pytest.mark.gpu
def test_cali_model_weight_ptq_vs_reference_golden(tmp_path):
    """The whole driver on the GPU — weight init, 36 targets (18 single layers, 18 blocks) reconstructed in the reference's
    order with its sample sequence (same torch seed), hard rounding afterwards, `_weight_only` ckpt — against the reference's
    CPU run: loss trajectories, rounding decisions of all 120 layers, α of five layers, ckpt schema; and the file loads."""
    from dgq_amd.quant import load_cali_model
    from dgq_amd.quant import reconstruction as rec
    from dgq_amd.quant.calibration import cali_model
    from dgq_amd.quant.reconstruction_util import RLOSS
    g = torch.load(GOLD)
    c = g["meta"]
    qnn = build_mini(c, "cuda")
    names = {id(m): n for n, m in qnn.named_modules()}
    losses = {}

    def on_loss(target, lf):
        lf.record = True
        losses[names[id(target)]] = lf
    rec.ON_LOSS_CREATED = on_loss
    try:
        torch.manual_seed(c["seed"])
        path = str(tmp_path / "cali_ckpt.pth")
        out = cali_model(qnn, w_cali_data=recon_data(c), a_cali_data=None, use_aq=False, path=path, running_stat=False,
                         interval=c["n"], tib_recon=False, iters=c["iters"], batch_size=c["batch_size"], w=c["w"], asym=True,
                         warmup=c["warmup"], opt_mode=RLOSS.MSE, multi_gpu=False, no_recon=False, resume_w=None)
    finally:
        rec.ON_LOSS_CREATED = None
    ck = torch.load(path + "_weight_only")["weight"]
    assert {k: (tuple(v.shape), str(v.dtype)) for k, v in ck.items()} == g["schema"]
    assert sorted(out["weight"]) == sorted(ck)
    # loss trajectories: same targets, same schedule, reconstruction and rounding terms at fp32-rounding distance
    assert list(losses) == [o[1] for o in g["order"]]
    worst_rec = worst_round = 0.0
    for name, lf in losses.items():
        ref = g["traj"][name]
        assert len(lf.history) == len(ref) == c["iters"]
        for (cnt, tot, rec_l, rnd, b), (rc, rtot, rrec, rrnd, rb) in zip(lf.history, ref):
            assert cnt == rc and b == (rb if rc >= c["iters"] * c["warmup"] else 0.0)
            worst_rec = max(worst_rec, abs(rec_l - rrec) / max(abs(rrec), 1e-12))
            worst_round = max(worst_round, abs(rnd - rrnd) / max(abs(rrnd), 1.0))
    print("loss trajectories vs reference: worst rel reconstruction %.3g, worst rel rounding term %.3g" % (worst_rec, worst_round))
    assert worst_rec < 2e-3 and worst_round < 1e-4
    # rounding decisions and α
    flips = total = 0
    for name, r in g["layers"].items():
        a = ck[name + ".wqtizer.alpha"].float()
        assert tuple(a.shape) == r["shape"]
        assert torch.equal(ck[name + ".wqtizer.delta"], r["delta"]) and torch.equal(ck[name + ".wqtizer.zero_point"], r["zero_point"])
        up = torch.from_numpy(np.unpackbits(r["up"].numpy())[:a.numel()].astype(bool)).view(a.shape)
        flips += int(((a >= 0) != up).sum())
        total += a.numel()
        assert abs(float(a.double().abs().sum()) - r["abs_sum"]) <= 1e-4 * r["abs_sum"], name
        if "alpha" in r:
            assert (a - r["alpha"]).abs().max() < 2.5e-3, name          # at most a couple of Adam steps of 1e-3 apart
            assert (a - r["alpha"]).abs().mean() < 1e-4, name
    print("rounding decisions vs reference: %d of %d differ" % (flips, total))
    assert flips <= 2e-4 * total
    # the produced file is a valid weight ckpt: AdaRound layers, hard rounding, runs
    qnn2 = build_mini(c, "cuda")
    xs, ts, ctx = recon_data(c)
    load_cali_model(qnn2, (xs[:1], ts[:1], ctx[:1]), use_aq=False, path=path + "_weight_only")
    qnn2.disable_out_quantization()                                   # as get_qmodel does after loading (load_qmodel_util.py:61)
    with torch.no_grad():
        y2 = qnn2(xs[:2].cuda(), torch.tensor(901), ctx[:2].cuda())[0]
        qnn.set_quant_state(True, False)
        qnn.disable_out_quantization()
        y1 = qnn(xs[:2].cuda(), torch.tensor(901), ctx[:2].cuda())[0]
    assert torch.isfinite(y2).all() and torch.allclose(y1, y2, atol=1e-5)
