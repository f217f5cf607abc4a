"""SURVEY.md §8(f)-3 — the caller side of the drop-in: the diffusers pipeline loop (restated in dgq_amd/pipeline.py; HF
diffusers itself is not installed), PNDM's N+1-call schedule and its slot aliasing, and the reference's OWN UNet classes
wrapped by this package's QuantModel / load_cali_model (duck-typing; dev container only — /root/reference does not
travel)."""
import os
import types

import pytest
import torch

from dgq_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fake_eps(x, t):
    import math
    return 0.3 * x * math.cos(t / 100.0) + 0.1 * torch.sin(3.0 * x + t)


@pytest.mark.parametrize("n", [25, 50, 8])
def test_pndm_restatement_matches_vendored_scheduler(n):
    """tests/golden/f7_pndm_schedule.pt = the reference's vendored diffusers PNDMScheduler (SD-v1-4 config) run by
    make_golden.py `pndm`: the timestep list (N + 1 calls, one repeated) and every intermediate latent, bit for bit."""
    from dgq_amd.scheduler import PNDMScheduler
    g = torch.load(os.path.join(GOLD, "f7_pndm_schedule.pt"))[n]
    sch = PNDMScheduler(n)
    assert sch.timesteps == g["timesteps"] and len(sch.timesteps) == n + 1
    assert sch.timesteps[1] == sch.timesteps[2]                       # the aliased pair
    x = g["samples"][0].clone()
    for i, t in enumerate(sch.timesteps):
        x = sch.step(_fake_eps(x, t), t, x)
        assert torch.equal(x, g["samples"][i + 1]), (n, i, (x - g["samples"][i + 1]).abs().max())


@pytest.mark.parametrize("n", [4, 1, 8])
def test_euler_ancestral_restatement_matches_vendored_scheduler(n):
    """tests/golden/f7_euler_ancestral_schedule.pt = the reference's vendored diffusers EulerAncestralDiscreteScheduler in
    SDXL-turbo's configuration (make_golden.py `euler`): timesteps, sigmas, init_noise_sigma, every scaled model input and
    every latent of a run with a seeded CPU generator — bit for bit."""
    from dgq_amd.scheduler import EulerAncestralDiscreteScheduler
    g = torch.load(os.path.join(GOLD, "f7_euler_ancestral_schedule.pt"))[n]
    sch = EulerAncestralDiscreteScheduler(n)
    assert sch.timesteps == g["timesteps"] and torch.equal(sch.sigmas, g["sigmas"])
    assert float(sch.init_noise_sigma) == g["init_noise_sigma"]
    if n == 4:
        assert sch.timesteps == [999.0, 749.0, 499.0, 249.0]
    gen = torch.Generator().manual_seed(g["noise_seed"])
    x = g["samples"][0].clone()
    for i, t in enumerate(sch.timesteps):
        xi = sch.scale_model_input(x, t)
        assert torch.equal(xi, g["inputs"][i]), (n, i)
        x = sch.step(_fake_eps(xi, t), t, x, generator=gen)
        assert torch.equal(x, g["samples"][i + 1]), (n, i, (x - g["samples"][i + 1]).abs().max())


def test_pipeline_loop_calls_the_unet_like_diffusers_does():
    """Keyword set, 0-d int64 timestep tensor, CFG batch, return_dict=False + [0] (pipeline_stable_diffusion.py:1027-1035)."""
    from dgq_amd.pipeline import stable_diffusion_denoise, sdxl_turbo_denoise
    calls = []

    class FakeUNet:
        config = types.SimpleNamespace(in_channels=4, sample_size=8, time_cond_proj_dim=None, addition_time_embed_dim=256)

        def __call__(self, sample, t, **kw):
            calls.append((tuple(sample.shape), t, kw))
            return (0.1 * sample,)
    lat = torch.randn(1, 4, 8, 8)
    out = stable_diffusion_denoise(FakeUNet(), lat, torch.randn(2, 77, 768), num_inference_steps=25)
    assert out.shape == lat.shape and len(calls) == 26
    shp, t, kw = calls[0]
    assert shp == (2, 4, 8, 8) and t.dim() == 0 and t.dtype == torch.int64 and int(t) == 961
    assert set(kw) == {"encoder_hidden_states", "timestep_cond", "cross_attention_kwargs", "added_cond_kwargs", "return_dict"}
    assert kw["return_dict"] is False and kw["added_cond_kwargs"] is None
    calls.clear()
    sdxl_turbo_denoise(FakeUNet(), lat, torch.randn(1, 77, 2048), torch.randn(1, 1280), torch.zeros(1, 6))
    assert [int(c[1]) for c in calls] == [999, 749, 499, 249]
    assert set(calls[0][2]["added_cond_kwargs"]) == {"text_embeds", "time_ids"} and calls[0][0] == (1, 4, 8, 8)


def test_pndm_calls_alias_onto_time_aware_slots():
    """26 calls -> 25 slots (calibration.py:301-304): calls 1 and 2 share slot 1; every slot 0..24 is visited."""
    from dgq_amd.scheduler import PNDMScheduler
    from dgq_amd.runtime import slot_for_timestep
    slots = [slot_for_timestep(t, 25) for t in PNDMScheduler(25).timesteps]
    assert slots[:4] == [0, 1, 1, 2] and sorted(set(slots)) == list(range(25)) and len(slots) == 26
    slots50 = [slot_for_timestep(t, 50) for t in PNDMScheduler(50).timesteps]
    assert len(slots50) == 51 and sorted(set(slots50)) == list(range(50))


@pytest.mark.gpu
def test_quantmodel_under_the_pndm_pipeline_loop(tmp_path):
    """QuantModel (tiny arch, W4A8 g16, time-aware tables for 25 steps) driven by the pipeline loop with PNDM: slot
    sequence, kwargs swallowing, hipGraph replay per slot (the aliased call replays slot 1's graph) == eager."""
    from dgq_amd.pipeline import stable_diffusion_denoise
    from dgq_amd.runtime import build_synthetic_qnn
    c = dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=25)
    qnn, _ = build_synthetic_qnn("tiny", c, 16, 2, 25, ckpt_dir=str(tmp_path))
    lat = synth.named_randn("latent", (1, 4, 16, 16), 5).cuda()
    ctx = synth.named_randn("ctx", (2, 77, 64), 6).cuda()
    seen = []
    orig = qnn.activate_slot
    qnn.activate_slot = lambda s: (seen.append(s), orig(s))[1]
    eager = stable_diffusion_denoise(qnn, lat, ctx, 25)
    assert seen[:4] == [0, 1, 1, 2] and len(seen) == 26 and seen[-1] == 24
    qnn.prepare_slots()
    qnn.enable_graphs(True)
    replay = stable_diffusion_denoise(qnn, lat, ctx, 25)
    assert len(qnn._graphs) == 25                                   # one graph per slot; the aliased call reuses slot 1's
    assert torch.isfinite(replay).all()
    e = ((replay - eager).norm() / eager.norm()).item()
    assert e < 1e-5, e


@pytest.mark.skipif(not os.path.isdir("/root/reference/quant"), reason="the reference is mounted in the build container only")
def test_reference_unet_classes_are_wrapped_by_duck_typing(tmp_path):
    """INTEGRATION.md's claim, executed: the REFERENCE's diffusers_rewrite.UNet2DConditionModel (its Attention /
    BasicTransformerBlock / ResnetBlock2D classes) goes through THIS package's QuantModel + load_cali_model:
    same module surgery counts as the reference's QuantModel (282 QuantLayer, 16 + 22 Quant blocks), every ckpt key
    loads, loader side effects as fixture F6, and — in the floating-point state, which runs on CPU — the wrapped model
    reproduces the bare reference UNet's output."""
    import json
    from oracle import ref_harness as rh
    ref = rh.import_reference("sd")
    from dgq_amd.quant import QuantModel, QuantLayer, Scaler, load_cali_model
    from dgq_amd.quant.quant_block import QuantBasicTransformerBlock, QuantResnetBlock2D
    torch.manual_seed(0)
    unet = ref.dr.UNet2DConditionModel()
    unet.load_state_dict(synth.synth_state_dict("sd", 0))
    inp = synth.synth_inputs("sd", 2, 1, 16)
    with torch.no_grad():
        y_ref = unet(inp["sample"], torch.tensor(981), inp["encoder_hidden_states"])[0].clone()
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MINMAX}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": True}
    sm = {"softmax_a_bit": 8, "t2i_log_quant": True, "t2i_real_time": True, "t2i_start_peak": True, "log_max_1": False}
    qnn = QuantModel(model=unet, wq_params=wq, aq_params=aq, softmax_aq_params=sm, aq_mode=[1, 0], tib_recon=False).eval()
    n_layers = sum(isinstance(m, QuantLayer) for m in qnn.modules())
    n_tb = sum(isinstance(m, QuantBasicTransformerBlock) for m in qnn.modules())
    n_rb = sum(isinstance(m, QuantResnetBlock2D) for m in qnn.modules())
    assert (n_layers, n_tb, n_rb) == (282, 16, 22)
    assert qnn.config.in_channels == 4 and qnn.config.sample_size == 64 and qnn.config.time_cond_proj_dim is None
    # floating-point state: the Quant blocks' glue around the reference's own submodules
    qnn.set_quant_state(False, False)
    with torch.no_grad():
        y = qnn(inp["sample"], torch.tensor(981), inp["encoder_hidden_states"], timestep_cond=None,
                cross_attention_kwargs=None, added_cond_kwargs=None, return_dict=False)[0]
    e = ((y - y_ref).norm() / y_ref.norm()).item()
    assert e < 1e-5, e
    # checkpoint load (reference format) + loader side effects, without the data-dependent init forward (GPU only)
    path = str(tmp_path / "ck.pth")
    synth.write_cali_ckpt(path, "sd", 4, 8, 16, num_slots=2, seed=0, batch=2, res=16, start_peak=True)
    load_cali_model(qnn, (), use_aq=True, path=path, time_aware_aqtizer=True, num_inference_steps=2, use_group=True,
                    init_forward=False)
    qnn.disable_out_quantization()                                    # as get_qmodel does after the load (load_qmodel_util.py:61)
    side = json.load(open(os.path.join(GOLD, "f1_f6_schema_sd.json")))["loader_side_effects"]
    mism = [n for n, m in qnn.named_modules() if isinstance(m, QuantLayer) and n in side
            and (bool(m.use_group_num), bool(m.use_wq), bool(m.use_aq)) !=
            (side[n]["use_group_num"], side[n]["use_wq"], side[n]["use_aq"])]
    assert not mism, mism[:5]
    assert qnn.time_aware is not None and sorted(qnn.time_aware["slots"]) == [0, 1]
