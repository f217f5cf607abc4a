"""GPU parity tests of the HIP kernels, called through the C ABI (ctypes): integer work bit-exact
against the oracle, floating outputs against the reference's golden vectors (tests/golden) within the
tolerance written in each test."""
import os

import pytest
import torch

from oracle import dgq_oracle as orc
from tests.golden import recipes

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return torch.load(os.path.join(GOLD, name), map_location="cpu")


@pytest.fixture(scope="module")
def dev():
    from dgq_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


# ------------------------------------------------------------------------------------------ weights
def test_int4_pack_unpack_bit_exact(dev):
    from dgq_amd import ops
    g = torch.Generator().manual_seed(0)
    N, K = 77, 200
    codes = torch.randint(0, 16, (N, K), generator=g, dtype=torch.uint8)
    # arbitrary permutation with holes (padding), Kp multiple of 128
    Kp = 384
    perm = torch.full((Kp,), -1, dtype=torch.int32)
    pos = torch.randperm(Kp, generator=g)[:K]
    perm[pos] = torch.randperm(K, generator=g).to(torch.int32)
    packed = ops.pack_weight(codes.to(dev), perm, Kp, 4)
    assert packed.shape == (N, Kp // 2)
    un = ops.unpack_w4(packed, Kp).cpu()
    expect = torch.zeros(N, Kp, dtype=torch.uint8)
    m = perm >= 0
    expect[:, m] = codes[:, perm[m].long()]
    assert torch.equal(un, expect)
    # identity order
    packed = ops.pack_weight(codes.to(dev), None, 256, 4)
    un = ops.unpack_w4(packed, 256).cpu()
    assert torch.equal(un[:, :K], codes) and int(un[:, K:].sum()) == 0
    # layout 1 (what the kernels read) = layout 0 with the 8-byte halves of every 16 bytes exchanged in rows with bit 4 set
    from dgq_amd import _lib
    p0 = torch.empty_like(packed)
    _lib.check(_lib.load().dgq_pack_w4(_lib.ptr(codes.to(dev)), N, K, None, 256, 0, _lib.ptr(p0), _lib.stream()), "dgq_pack_w4")
    a, b = p0.view(N, -1, 2, 8).cpu(), packed.view(N, -1, 2, 8).cpu()
    swap = ((torch.arange(N) >> 4) & 1).bool()
    assert torch.equal(b[~swap], a[~swap]) and torch.equal(b[swap], a[swap].flip(2))
    assert torch.equal(ops.unpack_w4(p0, 256, layout=0).cpu(), un)


def test_weight_codes_match_oracle(dev):
    from dgq_amd import ops
    f2 = gold("f2_quantizers.pt")
    for bits in (4, 8):
        v = f2["minmax_ch_b%d" % bits]
        codes = ops.quantize_weight(v["w"].to(dev), v["delta"].to(dev), v["zp"].to(dev), None, bits).cpu()
        expect = orc.uaq_codes(v["w"], v["delta"], v["zp"], bits).reshape(v["w"].shape[0], -1)
        assert torch.equal(codes.float(), expect)
    v = f2["adaround_b4"]
    codes = ops.quantize_weight(v["w"].to(dev), v["delta"].to(dev), v["zp"].to(dev), v["alpha"].to(dev), 4).cpu()
    expect = torch.clamp(torch.floor(v["w"] / v["delta"]) + (v["alpha"] >= 0).float() + v["zp"], 0, 15)
    assert torch.equal(codes.float(), expect)
    # dequantised codes reproduce the reference's fake-quant weight exactly
    assert torch.equal(v["delta"] * (codes.float() - v["zp"]), v["y"])


# ------------------------------------------------------------------------------------------ MFMA layout
def _gemm_exact_case(dev, M, N, Kp, wbits, seed=1, frag=False):
    """frag: also hand the library the fragment-major weight image (dgq_gemm_extra_t.wfrag) — what the short-K panel kernel reads"""
    from dgq_amd import _lib, ops
    import ctypes
    g = torch.Generator().manual_seed(seed)
    ws = ops.workspace(dev)
    s = torch.randint(-16, 16, (M, Kp), generator=g, dtype=torch.int32)
    s[::7, ::13] = -128            # int8 extremes, sparsely (keeps every fp32 step exact)
    s[3::11, 5::17] = 127
    hi = 16 if wbits == 4 else 256
    q = torch.randint(0, hi, (N, Kp), generator=g, dtype=torch.int32)
    nch = Kp // 32
    cd = torch.tensor([2.0 ** ((i % 5) - 2) for i in range(nch)])
    gend = torch.tensor([(i % 3 == 1 or i == nch - 1) for i in range(nch)])
    fl = gend.to(torch.uint8)
    fl[3::12] = 2                  # clears of the running totals behind every third K tile: at group ends and inside groups alike
    fl[6::16] = 2                  # (marks elsewhere are ignored)
    # groups = runs of chunks ending at a group end; scale of a group = scale of its last chunk
    gscale = cd.clone()
    for i in range(nch - 2, -1, -1):
        if not gend[i]:
            gscale[i] = gscale[i + 1]
    woff = 0 if wbits == 4 else 128
    qs = (q - woff)
    acc = torch.zeros(M, N, dtype=torch.float64)
    for c in range(nch):
        acc += gscale[c].double() * (s[:, 32 * c:32 * c + 32].double() @ qs[:, 32 * c:32 * c + 32].double().T)
    alpha = torch.tensor([2.0 ** ((n % 3) - 1) for n in range(N)])
    zw = torch.tensor([float((n * 7) % 16) for n in range(N)]) - woff
    gamma = torch.tensor([float(n % 11) - 5 for n in range(N)])
    rowsum = torch.tensor([float((m * 3) % 17) - 8 for m in range(M)])
    expect = alpha[None, :].double() * (acc - zw[None, :].double() * rowsum[:, None].double()) + gamma[None, :].double()

    codes = s.to(torch.int8).to(dev)
    if wbits == 4:
        wp = ops.pack_weight(q.to(torch.uint8).to(dev), None, Kp, 4)
    else:
        wp = qs.to(torch.int8).to(dev)
    y = torch.empty(M, N, dtype=torch.float32, device=dev)
    lib = _lib.load()
    extra = None
    if frag:
        wf = ops.pack_weight(q.to(torch.uint8).to(dev), None, Kp, 4, ops.W4_FRAG_LAYOUT)
        ex = _lib.GemmExtra()
        ex.res_div, ex.fq_T, ex.fq_D = 1, 1, 1
        ex.wfrag = wf.data_ptr()
        extra = ctypes.byref(ex)
    t = lambda x: x.to(dev).contiguous()
    cdg, flg, al, zwg, ga, rs = t(gscale), t(fl), t(alpha), t(zw), t(gamma), t(rowsum)
    rc = lib.dgq_gemm_wxa8(_lib.ptr(codes), _lib.ptr(rs), 1, M, Kp, _lib.ptr(wp), wbits, N, 0,
                           _lib.ptr(cdg), _lib.ptr(flg), None, None, 1, ctypes.c_float(128.0),
                           _lib.ptr(al), _lib.ptr(zwg), _lib.ptr(ga), None,
                           _lib.ptr(y), 0, N, _lib.ptr(ws), ws.numel(), extra, _lib.stream())
    _lib.check(rc, "dgq_gemm_wxa8")
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().double(), expect), (M, N, Kp, wbits, (y.cpu().double() - expect).abs().max())
    # per-M epilogue, L=5
    L = 5
    md = torch.tensor([2.0 ** (i - 2) for i in range(L)])
    mz = torch.tensor([float(100 + 9 * i) for i in range(L)])
    vn = torch.tensor([float((n * 5) % 23) - 11 for n in range(N)])
    acc1 = s.double() @ qs.double().T
    mi = torch.arange(M) % L
    expect = alpha[None, :].double() * md[mi][:, None].double() * (
        acc1 - zw[None, :].double() * rowsum[:, None].double()
        + (128.0 - mz[mi][:, None].double()) * vn[None, :].double()) + gamma[None, :].double()
    mdg, mzg, vng = t(md), t(mz), t(vn)
    rc = lib.dgq_gemm_wxa8(_lib.ptr(codes), _lib.ptr(rs), 1, M, Kp, _lib.ptr(wp), wbits, N, 1,
                           None, None, _lib.ptr(mdg), _lib.ptr(mzg), L, ctypes.c_float(128.0),
                           _lib.ptr(al), _lib.ptr(zwg), _lib.ptr(ga), _lib.ptr(vng),
                           _lib.ptr(y), 0, N, _lib.ptr(ws), ws.numel(), extra, _lib.stream())
    _lib.check(rc, "dgq_gemm_wxa8")
    torch.cuda.synchronize()
    got = y.cpu().double()
    assert rel_l2(got, expect) < 1e-6, (M, N, Kp, wbits, rel_l2(got, expect))


def test_gemm_exact_integer(dev):
    """int8 x int4 -> exact integers through V_MFMA_I32_32X32X32_I8 (asymmetric data; catches any row/col or k-order
    mistake), incl. clears of the running totals inside and between groups. All scales are powers of two, so the fp
    epilogue is exact too."""
    for (M, N, Kp, wbits) in ((200, 136, 384, 4), (64, 320, 128, 4), (130, 72, 256, 8), (100, 200, 2048, 4),
                              (300, 136, 1024, 8), (2, 320, 1280, 4)):      # the last three take the split-K path
        _gemm_exact_case(dev, M, N, Kp, wbits)


@pytest.mark.parametrize("tile", ["32,64,1", "32,128,1", "64,64,1", "64,128,1", "128,64,1", "128,128,1", "256,256,1", "32,64,3", "64,128,2"])
def test_gemm_exact_integer_every_tile(tile, dev, monkeypatch):
    """the same on every tile shape of the family (DGQ_GEMM_FORCE = BM,BN,splits), ragged M / N edges included"""
    monkeypatch.setenv("DGQ_GEMM_FORCE", tile)
    _gemm_exact_case(dev, 203, 332, 640, 4, seed=3)
    if tile.startswith("256,256"):
        _gemm_exact_case(dev, 300, 700, 512, 4, seed=5)      # the 256-row kernel: ragged column / row tiles, both staged halves of every wave
    if tile in ("32,64,1", "64,64,1", "128,128,1", "32,64,3"):
        _gemm_exact_case(dev, 170, 200, 384, 8, seed=4)


def test_gemm_exact_integer_big_kernel(dev, monkeypatch):
    """the 256-row ping-pong kernel (gemm_wxa8_big.hip; the plan names it by BM = 256): exact integers on ragged M / N edges, one
    and several workgroup tiles per dimension, K from a single tile (no steady state) to 17 tiles (every ring stage reused, a
    clear of the running totals inside) — per-K on its 256x128 tile and per-M on 256x256"""
    monkeypatch.setenv("DGQ_GEMM_FORCE", "256,256,1")
    for (M, N, Kp, seed) in ((203, 332, 640, 3), (300, 700, 128, 5), (515, 260, 256, 6), (130, 513, 2176, 7), (700, 300, 384, 8)):
        _gemm_exact_case(dev, M, N, Kp, 4, seed=seed)


@pytest.mark.parametrize("case", recipes.f3_wide_cases(), ids=lambda c: c["name"])
def test_f3_wide_layers_every_kernel_vs_reference(case, dev, monkeypatch):
    """F3c: a wide Linear layer (M = 300, N = 256, K = 9216) with a REAL plan_act table — per-K with 16 DGQ groups
    (flush coefficients that are not powers of two, a clear of the running totals inside the K range), per-M and scalar — against the
    reference's own output (tests/golden/f3c_layers_wide.pt), on every member of the GEMM family that takes the shape: the planner's
    choice, the 256-row kernel (DGQ_GEMM_FORCE=256,256,1) and a K-split tile launch.
    Each within 2e-5 of the reference; the tile family and the 256-row kernel share their epilogue AND their summation order inside a
    K tile sequence only up to the order of the group sums, so they are compared at 1e-6 (per-M / scalar: bit for bit)."""
    from dgq_amd import ops
    from dgq_amd.plan import plan_act
    g = gold("f3c_layers_wide.pt")[case["name"]]
    inp = recipes.f3_inputs(case)
    w = inp["w"].to(dev)
    C = w.shape[1]
    pw = ops.PackedWeight(w, g["wdelta"].to(dev), g["wzp"].to(dev), None, inp["b"].to(dev), 4, C, 1)
    lay = plan_act(inp["adelta"], inp["azp"], "linear", C, 1, 8)
    ab = ops.ActBinding(lay, pw, 8)
    if lay.mode == "perK":
        assert int((ab.cflush == 2).sum()) >= 1, "the plan of this case must carry a clear of the running totals"
    x = inp["x"].to(dev)
    monkeypatch.setattr(ops, "GEMM_FUSE", False)                 # (K = 9216 does not fit the fused form anyway)
    outs = {}
    for plan in (None, "256,256,1", "64,64,4", "128,128,1"):
        if plan is None:
            monkeypatch.delenv("DGQ_GEMM_FORCE", raising=False)
        else:
            monkeypatch.setenv("DGQ_GEMM_FORCE", plan)
        y = ops.quant_linear(x, ab)
        torch.cuda.synchronize()
        err = rel_l2(y.cpu(), g["y"])
        assert err < 2e-5, (plan, err)
        outs[plan] = y.clone()
    ref = outs["128,128,1"]
    for plan, y in outs.items():
        if lay.mode == "perK":
            assert rel_l2(y, ref) < 1e-6, plan
        elif plan is not None and plan != "64,64,4":
            assert torch.equal(y, ref), plan                     # unsplit integer sums: the same numbers through the same epilogue


FUSED_PLANS = ["F1,10,1,1", "F1,5,1,1", "F1,5,1,2", "F1,4,1,2"]


@pytest.mark.parametrize("plan", FUSED_PLANS)
@pytest.mark.parametrize("mode", ["perM", "perK"])
def test_gemm_quantise_on_load_exact_integer(plan, mode, dev, monkeypatch):
    """dgq_gemm_act_t (gemm_panel.hip, FUSE): the GEMM quantises its own rows.  Inputs are exact multiples of power-of-two scales, so
    the codes, the row sums and every fp32 step are exact: y must EQUAL the float64 evaluation of quant_layer.py:295-299 + :659 on
    integer data — per-M (natural K order, K padding written as zero codes) and per-K (codes scattered through kdst into their DGQ
    groups, group padding zero), ragged M / N, every fused configuration incl. K waves."""
    from dgq_amd import _lib, ops
    import ctypes
    monkeypatch.setenv("DGQ_GEMM_FORCE", plan)
    monkeypatch.setenv("DGQ_GEMM_FUSE_ALL", "1")           # wherever it fits, not only where the planner finds it faster
    g = torch.Generator().manual_seed(11)
    lib = _lib.load()
    ws = ops.workspace(dev)
    for (M, N, K, Kp) in ((203, 332, 320, 640 if mode == "perK" else 384), (64, 320, 100, 256 if mode == "perK" else 128), (97, 1290, 1280, 1664 if mode == "perK" else 1280)):
        nch = Kp // 32
        s = torch.randint(-16, 16, (M, K), generator=g, dtype=torch.int32)          # the centred codes the quantiser must find
        s[::7, ::13] = -128
        s[3::11, 5::17] = 127
        q = torch.randint(0, 16, (N, Kp), generator=g, dtype=torch.int32)
        alpha = torch.tensor([2.0 ** ((n % 3) - 1) for n in range(N)])
        zw = torch.tensor([float((n * 7) % 16) for n in range(N)])
        gamma = torch.tensor([float(n % 11) - 5 for n in range(N)])
        t = lambda v: v.to(dev).contiguous()
        wp = ops.pack_weight(q.to(torch.uint8).to(dev), None, Kp, 4)
        wf = ops.pack_weight(q.to(torch.uint8).to(dev), None, Kp, 4, ops.W4_FRAG_LAYOUT)
        act = _lib.GemmAct()
        ex = _lib.GemmExtra()
        ex.res_div, ex.fq_T, ex.fq_D = 1, 1, 1
        ex.wfrag = wf.data_ptr()
        y = torch.empty(M, N, dtype=torch.float32, device=dev)
        al, zwg, ga = t(alpha), t(zw), t(gamma)
        if mode == "perM":
            L = 5
            md = torch.tensor([2.0 ** (i - 2) for i in range(L)])
            mz = torch.tensor([float(100 + 9 * i) for i in range(L)])
            mi = torch.arange(M) % L
            x = (s.double() + 128.0 - mz[mi][:, None].double()) * md[mi][:, None].double()          # rne(x/δ) + z = s + 128 exactly
            vn = torch.tensor([float((n * 5) % 23) - 11 for n in range(N)])
            acc = s.double() @ q[:, :K].double().T
            rowsum = s.double().sum(1)
            expect = alpha[None].double() * md[mi][:, None].double() * (acc - zw[None].double() * rowsum[:, None]
                                                                       + (128.0 - mz[mi][:, None].double()) * vn[None].double()) + gamma[None].double()
            xg, mdg, mzg, vng = t(x.float()), t(md), t(mz), t(vn)
            act.x, act.x_dtype, act.ldx, act.K, act.bits, act.rows_per_image = xg.data_ptr(), 0, K, K, 8, 1
            ex.act = ctypes.cast(ctypes.pointer(act), ctypes.c_void_p)
            rc = lib.dgq_gemm_wxa8(_lib.ptr(wf), _lib.ptr(wf), 1, M, Kp, _lib.ptr(wp), 4, N, 1, None, None, _lib.ptr(mdg), _lib.ptr(mzg), L,
                                   ctypes.c_float(128.0), _lib.ptr(al), _lib.ptr(zwg), _lib.ptr(ga), _lib.ptr(vng), _lib.ptr(y), 0, N,
                                   _lib.ptr(ws), ws.numel(), ctypes.byref(ex), _lib.stream())
        else:
            # K source channels spread over the nch chunks (chunk c gets every channel k with k % nch == c: a permutation with holes)
            pos = []
            fill = [0] * nch
            for k in range(K):
                c = k % nch
                pos.append(32 * c + fill[c])
                fill[c] += 1
            assert max(fill) <= 32
            kdst = torch.tensor(pos, dtype=torch.int32)
            cd = torch.tensor([2.0 ** ((i % 5) - 2) for i in range(nch)])
            gend = torch.tensor([(i % 3 == 1 or i == nch - 1) for i in range(nch)])
            gscale = cd.clone()
            for i in range(nch - 2, -1, -1):
                if not gend[i]:
                    gscale[i] = gscale[i + 1]
            czp = torch.tensor([float(90 + 7 * (i % 9)) for i in range(nch)])
            for i in range(nch - 2, -1, -1):
                if not gend[i]:
                    czp[i] = czp[i + 1]
            ch = kdst.long() // 32
            x = (s.double() + 128.0 - czp[ch][None].double()) * gscale[ch][None].double()
            sp = torch.zeros(M, Kp, dtype=torch.float64)
            sp[:, kdst.long()] = s.double()
            acc = torch.zeros(M, N, dtype=torch.float64)
            for c in range(nch):
                acc += gscale[c].double() * (sp[:, 32 * c:32 * c + 32] @ q[:, 32 * c:32 * c + 32].double().T)
            rowsum = (sp * gscale.double().repeat_interleave(32)[None]).sum(1)
            expect = alpha[None].double() * (acc - zw[None].double() * rowsum[:, None]) + gamma[None].double()
            xg, cdg, flg, kdg, czg = t(x.float()), t(gscale), t(gend.to(torch.uint8)), t(kdst), t(czp)
            act.x, act.x_dtype, act.ldx, act.K, act.bits, act.rows_per_image = xg.data_ptr(), 0, K, K, 8, 1
            act.kdst, act.czp = kdg.data_ptr(), czg.data_ptr()
            ex.act = ctypes.cast(ctypes.pointer(act), ctypes.c_void_p)
            rc = lib.dgq_gemm_wxa8(_lib.ptr(wf), _lib.ptr(wf), 1, M, Kp, _lib.ptr(wp), 4, N, 0, _lib.ptr(cdg), _lib.ptr(flg), None, None, 1,
                                   ctypes.c_float(128.0), _lib.ptr(al), _lib.ptr(zwg), _lib.ptr(ga), None, _lib.ptr(y), 0, N,
                                   _lib.ptr(ws), ws.numel(), ctypes.byref(ex), _lib.stream())
        _lib.check(rc, "dgq_gemm_wxa8")
        torch.cuda.synchronize()
        got = y.cpu().double()
        assert torch.equal(got, expect), (plan, mode, M, N, K, (got - expect).abs().max())


@pytest.mark.parametrize("M,K,N,mode,fold", [(8192, 320, 320, "perK", "ln"), (8192, 320, 320, "perM", "ln"), (2048, 640, 640, "perK", None),
                                            (512, 1280, 1280, "perM", None), (512, 1280, 1280, "perK", "ln"), (154, 768, 320, "perK", None),
                                            (2048, 640, 5120, "perK", "geglu_out"), (8192, 320, 320, "scalar", "gn_silu")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_quant_linear_fused_equals_two_launches(M, K, N, mode, fold, dtype, dev, monkeypatch):
    """ops.quant_linear / quant_conv2d (1x1) with quantise-on-load inside the GEMM against the dgq_quant_act + dgq_gemm_wxa8 pair on the
    same layer: per-M without a folded norm the codes and the int32 sums are the same numbers -> equal outputs; per-K differs in
    the order of the fp32 group sums, a folded LayerNorm / GroupNorm in the rounding of the statistics (isolated code flips)."""
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    monkeypatch.setenv("DGQ_GEMM_FUSE_ALL", "1")
    g = torch.Generator().manual_seed(3)
    w = torch.randn(N, K, generator=g) * 0.05
    wd, wz = synth.channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.randn(N, generator=g).to(dev), 4, K, 1)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "fused|%d" % K, 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", K, 1, 8)
    elif mode == "perM":
        d, z = synth._group_params(64, 16, 8, "fused|%d" % K, 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
    else:
        lay = plan_act(torch.tensor(0.031), torch.tensor(117.0), "linear", K, 1, 8)
    ab = ops.ActBinding(lay, pw, 8)
    x = torch.randn(M, K, generator=g).to(dev).to(dtype)
    res = torch.randn(M, N if fold != "geglu_out" else N // 2, generator=g).to(dev).to(dtype)

    def run():
        if fold == "gn_silu":                                  # a 1x1 convolution behind GroupNorm + SiLU (Transformer2D proj_in, shortcuts)
            xi = x.view(2, 64, 64, K).permute(0, 3, 1, 2)
            gam, bet = torch.linspace(0.5, 1.5, K, device=dev), torch.linspace(-0.2, 0.2, K, device=dev)
            return ops.quant_conv2d(xi, ab, 1, 1, 1, 0, norm=(32, 1e-5, gam, bet, 1)).permute(0, 2, 3, 1).reshape(M, N)
        ln = (torch.linspace(0.5, 1.5, K, device=dev), torch.linspace(-0.1, 0.1, K, device=dev), 1e-5) if fold == "ln" else None
        if fold == "geglu_out":
            return ops.quant_linear(x, ab, geglu=True)
        return ops.quant_linear(x, ab, residual=res, ln=ln)
    assert ops.act_fuses(ab, M, K, dtype)
    y1 = run()
    monkeypatch.setattr(ops, "GEMM_FUSE", False)
    y0 = run()
    torch.cuda.synchronize()
    assert y1.shape == y0.shape and torch.isfinite(y1.float()).all()
    if mode != "perK" and fold is None and dtype == torch.float32:
        assert torch.equal(y1, y0)
    else:
        tol = 2e-3 if fold in ("ln", "gn_silu") else (3e-3 if dtype != torch.float32 else 2e-6)
        assert rel_l2(y1, y0) < tol, rel_l2(y1, y0)


@pytest.mark.parametrize("case", ["linear_perK_ln_res", "linear_perM_res", "linear_perK_small", "linear_perK_geglu", "conv1x1_gn_silu", "conv3x3_perK_gn_res",
                                  "conv3x3_perM_small", "linear_perK_splitk"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_quant_layers_half_io_equal_the_fp32_path_rounded_once(case, dtype, dev):
    """What "bf16 / fp16 between the layers" means here, as an operator-level identity (a stated bound instead of a
    measured one): a quantized layer on 16-bit tensors == the SAME layer on those values as fp32 tensors, its fp32 result rounded ONCE
    to the tensor type.  Loads convert exactly, statistics / quantiser / integer contraction / epilogue (residual, GEGLU, GroupNorm
    prologue) are the fp32 path's, only the store rounds — so the whole cost of the 16-bit modes is the tensors' own 2^-9 / 2^-12.
    Asserted bit for bit for bf16; fp16 within one ulp on <= 2e-3 of the elements (the compiler folds the epilogue's last fp32
    operation and the fp16 conversion into v_fma_mixlo_f16, which rounds once where fp32-then-fp16 rounds twice)."""
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    g = torch.Generator().manual_seed(len(case))

    def binding(N, C, taps, mode, L):
        w = torch.randn(N, C * taps, generator=g) * 0.05
        wd, wz = synth.channel_minmax(w, 4)
        wq = w.view(N, C, 3, 3) if taps == 9 else w
        pw = ops.PackedWeight(wq.to(dev), wd.to(dev), wz.to(dev), None, torch.randn(N, generator=g).to(dev), 4, C, taps)
        kind = "linear" if taps == 1 and not case.startswith("conv") else "conv"
        if mode == "perK":
            d, z = synth._group_params(C * taps, 16, 8, "half|%s" % case, 0)
            lay = plan_act(d.view(1, 1, -1) if kind == "linear" else d.view(1, -1, 1), z.view(1, 1, -1) if kind == "linear" else z.view(1, -1, 1), kind, C, taps, 8)
        else:
            d, z = synth._group_params(L, 16, 8, "half|%s" % case, 0)
            lay = plan_act(d.view(1, -1, 1) if kind == "linear" else d.view(1, 1, -1), z.view(1, -1, 1) if kind == "linear" else z.view(1, 1, -1), kind, C, taps, 8)
        return ops.ActBinding(lay, pw, 8)

    if case.startswith("linear"):
        M, K, N, mode = {"linear_perK_ln_res": (2048, 640, 640, "perK"), "linear_perM_res": (8192, 320, 320, "perM"), "linear_perK_small": (154, 768, 320, "perK"),
                         "linear_perK_geglu": (512, 320, 2560, "perK"), "linear_perK_splitk": (128, 5120, 1280, "perK")}[case]
        ab = binding(N, K, 1, mode, 64)
        x = torch.randn(M, K, generator=g).to(dev).to(dtype)
        res = torch.randn(M, N, generator=g).to(dev).to(dtype) if "res" in case else None
        ln = (torch.linspace(0.5, 1.5, K, device=dev), torch.linspace(-0.1, 0.1, K, device=dev), 1e-5) if "ln" in case else None
        run = lambda xx, rr: ops.quant_linear(xx, ab, geglu=True) if "geglu" in case else ops.quant_linear(xx, ab, residual=rr, ln=ln)
    else:
        B, C, H, N, k, mode = {"conv1x1_gn_silu": (2, 320, 32, 320, 1, "perK"), "conv3x3_perK_gn_res": (2, 320, 16, 320, 3, "perK"),
                               "conv3x3_perM_small": (1, 64, 8, 160, 3, "perM")}[case]
        ab = binding(N, C, k * k, mode, H * H)
        x = (torch.randn(B, C, H, H, generator=g) * 1.3 + 0.2).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
        res = torch.randn(B, N, H, H, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last) if "res" in case else None
        norm = (32, 1e-5, torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev), 1) if "gn" in case else None
        run = lambda xx, rr: ops.quant_conv2d(xx, ab, k, k, 1, k // 2, norm=norm, residual=rr)
    y16 = run(x, res)
    y32 = run(x.float(), None if res is None else res.float())
    torch.cuda.synchronize()
    assert y16.dtype == dtype and y32.dtype == torch.float32 and y16.shape == y32.shape
    want = y32.to(dtype)
    if dtype == torch.bfloat16:
        assert torch.equal(y16, want), (case, (y16.float() - want.float()).abs().max().item(), float((y16 != want).float().mean()))
    else:
        ne = y16 != want
        assert float(ne.float().mean()) <= 2e-3, float(ne.float().mean())
        ulp = torch.maximum(want.float().abs(), torch.tensor(6.2e-5, device=dev)) * 2.0 ** -10
        assert bool(((y16.float() - want.float()).abs() <= ulp).all())


def test_int4_fragment_major_layout(dev):
    """dgq_pack_w4 layout 2 (the panel kernel's weight image): unpacks to the same codes; block (j, p), lane (h << 5) | (n & 31) holds
    the two words of K half h of chunk 2p, then of chunk 2p + 1, of column 32j + (n & 31); columns past N are zero"""
    from dgq_amd import ops
    g = torch.Generator().manual_seed(2)
    N, K, Kp = 77, 200, 256
    codes = torch.randint(0, 16, (N, K), generator=g, dtype=torch.uint8)
    p1 = ops.pack_weight(codes.to(dev), None, Kp, 4)
    p2 = ops.pack_weight(codes.to(dev), None, Kp, 4, ops.W4_FRAG_LAYOUT)
    assert p2.shape == (96, Kp // 2)
    assert torch.equal(ops.unpack_w4(p2, Kp, layout=2).cpu()[:N], ops.unpack_w4(p1, Kp).cpu())
    # explicit image check against layout 0
    from dgq_amd import _lib
    p0 = torch.empty_like(p1)
    _lib.check(_lib.load().dgq_pack_w4(_lib.ptr(codes.to(dev)), N, K, None, Kp, 0, _lib.ptr(p0), _lib.stream()), "dgq_pack_w4")
    w0 = torch.zeros(96, Kp // 8, dtype=torch.int32)
    w0[:N] = p0.cpu().view(torch.int32).view(N, Kp // 8)
    img = p2.cpu().view(torch.int32).view(3, Kp // 64, 64, 4)            # [column tile][chunk pair][lane][word]
    for j in range(3):
        for pr in range(Kp // 64):
            for lane in (0, 5, 31, 32, 47, 63):
                n, h = 32 * j + (lane & 31), lane >> 5
                expect = [w0[n, (2 * pr) * 4 + 2 * h], w0[n, (2 * pr) * 4 + 2 * h + 1], w0[n, (2 * pr + 1) * 4 + 2 * h], w0[n, (2 * pr + 1) * 4 + 2 * h + 1]]
                assert img[j, pr, lane].tolist() == [int(e) for e in expect]


# ------------------------------------------------------------------------------------------ activation codes
@pytest.mark.parametrize("case", [c for c in recipes.f3_cases() if c["state"] == "wa"], ids=lambda c: c["name"])
def test_f3_layers_vs_reference(case, dev):
    """QuantLayer.forward parity (quant_layer.py:626-661): HIP path vs the reference's golden output.
    Tolerance: 2e-5 relative L2 (integer accumulation here vs fp32 GEMM in the reference; the survey's
    probe measured 3e-7 for the identity itself) and activation codes bit-exact vs the oracle."""
    from dgq_amd import ops
    from dgq_amd.plan import plan_act
    g = gold("f3_layers.pt")[case["name"]]
    inp = recipes.f3_inputs(case)
    w = inp["w"].to(dev)
    taps = 1 if case["kind"] == "linear" else case["k"] ** 2
    C = w.shape[1]
    pw = ops.PackedWeight(w, g["wdelta"].to(dev), g["wzp"].to(dev), None, inp["b"].to(dev), case["wbits"], C, taps)
    lay = plan_act(inp["adelta"], inp["azp"], case["kind"], C, taps, case["abits"])
    ab = ops.ActBinding(lay, pw, case["abits"])
    x = inp["x"].to(dev)
    if case["kind"] == "linear":
        y = ops.quant_linear(x, ab)
        xu = inp["x"]
        k_of = None
    else:
        y = ops.quant_conv2d(x, ab, case["k"], case["k"], case["stride"], case["padding"])
    torch.cuda.synchronize()
    err = rel_l2(y.cpu(), g["y"])
    assert err < 2e-5, err

    # activation codes bit-exact vs the oracle's integer codes on the unfolded operand
    if case["kind"] == "linear":
        x2 = inp["x"].reshape(-1, inp["x"].shape[-1])
        if inp["x"].dim() == 3:
            q = orc.uaq_codes(inp["x"], inp["adelta"], inp["azp"], case["abits"]).reshape(x2.shape)
        else:
            q = orc.uaq_codes(x2, inp["adelta"], inp["azp"], case["abits"])
        xs = x.reshape(-1, x.shape[-1]).contiguous()
        codes, rowsum, M = ops.quant_act(xs, xs.shape[0], 1, 1, C, 1, 1, 1, 0, ab)
    else:
        import torch.nn.functional as F
        cols = F.unfold(inp["x"], kernel_size=case["k"], padding=case["padding"], stride=case["stride"])
        q = orc.uaq_codes(cols, inp["adelta"], inp["azp"], case["abits"])       # [B, K_ref, L]
        q = q.permute(0, 2, 1).reshape(-1, cols.shape[1])                        # [M, K_ref]
        xc = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        codes, rowsum, M = ops.quant_act(xc, x.shape[0], x.shape[2], x.shape[3], C, case["k"], case["k"],
                                         case["stride"], case["padding"], ab)
    torch.cuda.synchronize()
    off = 2 ** (case["abits"] - 1)
    if lay.mode == "perK":
        kperm = lay.kperm
    else:
        from dgq_amd.plan import natural_kperm
        kperm = natural_kperm(C, taps)
    m = kperm >= 0
    got = codes.cpu().int()
    assert torch.equal(got[:, m], (q[:, kperm[m].long()] - off).int())
    assert int(got[:, ~m].abs().sum()) == 0


# ------------------------------------------------------------------------------------------ attention side
def test_fakequant_rows_and_logquant(dev):
    from dgq_amd import ops
    f2 = gold("f2_quantizers.pt")
    # per-last-dim and per-token broadcast layouts, [2,12,20] viewed as rows x C
    v = f2["uaq_bcast_lastdim"]
    x = v["x"].reshape(-1, 20).to(dev).contiguous()
    y = ops.fakequant_rows(x.clone(), 12, 20, 2, v["delta"].reshape(-1).to(dev), v["zp"].reshape(-1).to(dev), 0, 8)
    assert torch.equal(y.cpu().reshape(v["y"].shape), v["y"])
    v = f2["uaq_bcast_dim1"]
    x = v["x"].reshape(-1, 20).to(dev).contiguous()
    y = ops.fakequant_rows(x.clone(), 12, 20, 1, v["delta"].reshape(-1).to(dev), v["zp"].reshape(-1).to(dev), 0, 8)
    assert torch.equal(y.cpu().reshape(v["y"].shape), v["y"])
    for bits in (4, 6, 8):
        for tag in ("pos", "negz", "bigz"):
            v = f2["uaq_b%d_%s" % (bits, tag)]
            x = v["x"].reshape(1, -1).to(dev).contiguous()
            y = ops.fakequant_rows(x.clone(), 1, x.shape[1], 0, v["delta"].reshape(1).to(dev),
                                   v["zp"].reshape(1).to(dev), 0, bits)
            assert torch.equal(y.cpu().reshape(-1), v["y"])
    # log quantizer, real-time δ = global max; a handful of exact-tie elements may differ by one code
    for bits in (6, 8):
        v = f2["logq_rt_b%d" % bits]
        p = v["x"].to(dev).contiguous()
        d = ops.max_f32(p)
        assert float(d.cpu()) == float(v["x"].max())
        y = ops.logquant_f32(p.clone(), d, bits).cpu()
        mism = (y != v["y"]).float().mean().item()
        assert mism < 1e-3, mism
        assert rel_l2(y, v["y"]) < 1e-2
        v = f2["logq_fixed_b%d" % bits]
        y = ops.logquant_f32(v["x"].to(dev).contiguous(), v["delta"].reshape(1).to(dev), bits).cpu()
        assert (y != v["y"]).float().mean().item() < 1e-3


def test_static_log_quantizer_init_and_forward_vs_reference_golden(dev):
    """Non-real-time T2ILogQuantizer (quant_layer_text.py:49-76): first forward runs the 0.999 / 0.9999 / 0.99999
    quantile search for δ on the device, later forwards reuse it.  Golden `logq_static_b*` = the REAL reference's δ and
    output on the same probabilities; also through the fused attention kernel's static-δ mode (mode 2) with that δ."""
    from dgq_amd import ops
    from dgq_amd.quant.quant_layer_text import T2ILogQuantizer
    f2 = gold("f2_quantizers.pt")
    for bits in (6, 8):
        v = f2["logq_static_b%d" % bits]
        q = T2ILogQuantizer(bits=bits, always_zero=True, real_time=False)
        x = v["x"].to(dev)
        y = q(x.clone())
        assert q.init and not q.real_time
        d = float(torch.as_tensor(q.delta).detach().cpu())
        # torch.quantile on the device interpolates like the CPU one; allow 1 ulp-level difference of the picked quantile
        assert abs(d - float(v["delta"])) <= 2e-7 * abs(float(v["delta"])), (d, float(v["delta"]))
        mism = (y.cpu() != v["y"]).float().mean().item()
        assert mism < 1e-3, mism
        assert rel_l2(y.cpu(), v["y"]) < 1e-2
        y2 = q(x.clone())                                     # δ is frozen after the first call
        assert torch.equal(y2.cpu(), y.cpu())
        # a second tensor with a larger maximum must NOT move δ (that is what separates it from real-time mode)
        x3 = (x * 0.5).contiguous()
        q(x3.clone())
        assert float(torch.as_tensor(q.delta).detach().cpu()) == d


# ------------------------------------------------------------------------------------------ fused attention
@pytest.mark.parametrize("D,T,S,H", [(40, 200, 200, 2), (8, 70, 77, 8), (16, 130, 40, 3), (64, 96, 77, 2),
                                     (80, 257, 257, 2), (160, 64, 77, 2), (160, 256, 256, 8)])
@pytest.mark.parametrize("mode,skip", [(0, 0), (1, 0), (1, 1), (2, 0), (3, 1)])
def test_fused_attention_vs_reference_formulas(D, T, S, H, mode, skip, dev):
    """dgq_attention_f32 against the materialised reference sequence of sd.py:183-201 evaluated with torch on the
    CPU (scores·scale -> softmax -> aqtizer_w with column bypass -> @ v).  FP mode within 1e-5; quantised modes
    allow isolated one-code flips of a probability that sits on a rounding tie (rel-L2 < 2e-3)."""
    from dgq_amd import ops
    g = torch.Generator().manual_seed(D * 1000 + T + mode)
    B, bits = 2, 8
    q = torch.randn(B, T, H * D, generator=g)
    k = torch.randn(B, S, H * D, generator=g)
    v = torch.randn(B, S, H * D, generator=g)
    scale = D ** -0.5
    qh, kh, vh = (x.view(B, -1, H, D).transpose(1, 2) for x in (q, k, v))
    p = torch.softmax(torch.matmul(qh, kh.transpose(-2, -1)) * scale, dim=-1)
    delta = None
    if mode == 1:
        pq = orc.log_quant(p[..., skip:], p[..., skip:].max(), bits)
    elif mode == 2:
        delta = torch.tensor([0.37 * float(p.max())])
        pq = orc.log_quant(p[..., skip:], delta[0], bits)
    elif mode == 3:
        delta = torch.tensor([float(p.max()) / 255.0])
        pq = orc.uaq(p[..., skip:], delta[0], torch.tensor(0.0), bits)
    else:
        pq = p[..., skip:]
    pf = torch.cat([p[..., :skip], pq], dim=-1) if skip else pq
    ref = torch.matmul(pf, vh).transpose(1, 2).reshape(B, T, H * D)
    o = ops.attention_f32(q.to(dev), k.to(dev), v.to(dev), H, D, scale, mode, skip,
                          delta.to(dev) if delta is not None else None, bits)
    torch.cuda.synchronize()
    err = rel_l2(o.cpu(), ref)
    # a probability whose −log2(p/δ) (or p/δ) lies within fp32 rounding of a tie gets the neighbouring code (measured:
    # ~3e-6 of the entries, the rate fp32 evaluation-order differences imply); every other row agrees to fp32 accuracy
    assert err < (1e-5 if mode == 0 else 4e-3), err
    row_err = (o.cpu() - ref).view(B * T, -1).norm(dim=1) / ref.view(B * T, -1).norm(dim=1)
    assert row_err.median().item() < 1e-5
    assert (row_err > 1e-4).float().mean().item() < 0.02


@pytest.mark.parametrize("D,T,S,H", [(40, 200, 200, 2), (80, 96, 77, 3), (160, 64, 77, 2)])
@pytest.mark.parametrize("mode", [1, 3])
def test_attention_fused_qkv_quantizers_bit_identical(D, T, S, H, mode, dev):
    """aqtizer_q/k/v applied inside dgq_attention_f32's operand loads == dgq_fakequant_rows followed by the same
    attention, bit for bit (scalar, per-token with the start-peak key bypass, per-head-dim tables)."""
    from dgq_amd import ops
    assert ops.attention_fuses_fakequant(D, mode)
    g = torch.Generator().manual_seed(D + T + mode)
    B, bits, skip = 2, 8, 1
    q, k, v = (torch.randn(B, n, H * D, generator=g).to(dev) for n in (T, S, S))

    def table(n, lo):
        d = (torch.rand(n, generator=g) * 0.02 + lo).to(dev)
        z = torch.randint(100, 156, (n,), generator=g).float().to(dev)
        return d, z
    fq_q = (1,) + table(T, 0.02) + (0, bits)                 # per query token
    fq_k = (1,) + table(S - skip, 0.02) + (skip, bits)       # per key token, key 0 bypassed
    fq_v = (2,) + table(D, 0.02) + (0, bits)                 # per head-dim element
    delta = torch.tensor([0.004], device=dev) if mode == 3 else None
    scale = D ** -0.5
    for fqs in ((fq_q, fq_k, fq_v), ((0,) + table(1, 0.03) + (0, bits), None, fq_v)):
        qq, kk, vv = q.clone(), k.clone(), v.clone()
        for ten, f, n in ((qq, fqs[0], T), (kk, fqs[1], S), (vv, fqs[2], S)):
            if f is not None:
                ops.fakequant_rows(ten.view(B * n, H * D), n, D, f[0], f[1], f[2], f[3], f[4])
        ref = ops.attention_f32(qq, kk, vv, H, D, scale, mode, skip, delta, bits)
        os.environ["DGQ_ATTN_I8"] = "0"                      # the bf16x3 products: same arithmetic as the unfused run
        try:
            out = ops.attention_f32(q, k, v, H, D, scale, mode, skip, delta, bits, fq=fqs)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("DGQ_ATTN_I8", None)
        assert torch.equal(out, ref)


@pytest.mark.parametrize("D,T,S,H", [(40, 200, 200, 2), (8, 70, 77, 8), (64, 300, 77, 2), (80, 257, 257, 2), (160, 64, 77, 2),
                                     (40, 4096, 4096, 1)])
@pytest.mark.parametrize("mode,skip,qmode,kmode,vmode,bits", [(1, 1, 1, 1, 2, 8), (1, 0, 0, 1, 1, 8), (3, 0, 0, 0, 0, 6),
                                                               (2, 1, 1, 0, 2, 6),
                                                               # a per-head-dim table on q or k: one exact Q plane x three K planes
                                                               (1, 1, 2, 1, 2, 8), (1, 0, 1, 2, 0, 8), (3, 0, 2, 2, 1, 6), (2, 1, 2, 0, 2, 6)])
def test_attention_int8_scores_vs_exact_formula(D, T, S, H, mode, skip, qmode, kmode, vmode, bits, dev):
    """with scalar / per-token aqtizer_q and aqtizer_k, Q·K^T runs as ONE int8 contraction
    (V_MFMA_I32_32X32X32_I8) with the zero-point / scale algebra in the epilogue.  Checked against (a) the attention
    formulas of sd.py:165-201 evaluated in float64 on the dequantised operands (the int path is EXACT up to the fp32
    rounding of the final scale, so scores agree to ~1e-7) and (b) the bf16x3 path of the same call."""
    from dgq_amd import ops
    g = torch.Generator().manual_seed(D * 7 + T + mode)
    B = 2
    q, k, v = (torch.randn(B, n, H * D, generator=g).to(dev) for n in (T, S, S))
    kskip = skip                                                # start-peak bypasses key 0 in aqtizer_k too (sd.py:176-180)
    qmax = float(2 ** bits - 1)

    def table(fmode, ntok, lo=0.02):
        n = 1 if fmode == 0 else (ntok if fmode == 1 else D)
        d = (torch.rand(n, generator=g) * 0.02 + lo).to(dev)
        z = torch.randint(int(qmax * 0.4), int(qmax * 0.6) + 1, (n,), generator=g).float().to(dev)
        return d, z
    fq_q = (qmode,) + table(qmode, T) + (0, bits)
    fq_k = (kmode,) + table(kmode, S - kskip) + (kskip, bits)
    fq_v = (vmode,) + table(vmode, S) + (0, bits)
    delta = torch.tensor([0.004], device=dev) if mode >= 2 else None
    scale = D ** -0.5
    out = ops.attention_f32(q, k, v, H, D, scale, mode, skip, delta, bits, fq=(fq_q, fq_k, fq_v))
    os.environ["DGQ_ATTN_I8"] = "0"
    try:
        ref3 = ops.attention_f32(q, k, v, H, D, scale, mode, skip, delta, bits, fq=(fq_q, fq_k, fq_v))
        torch.cuda.synchronize()
    finally:
        os.environ.pop("DGQ_ATTN_I8", None)

    # float64 formula on the dequantised operands
    def dq(x, f, ntok):
        fm, d, z, sk, _ = f
        xh = x.view(B, ntok, H, D).double()
        if fm == 0:
            dd, zz = d.double().view(1, 1, 1, 1), z.double().view(1, 1, 1, 1)
        elif fm == 1:
            dd, zz = d.double().view(1, -1, 1, 1), z.double().view(1, -1, 1, 1)
        else:
            dd, zz = d.double().view(1, 1, 1, -1), z.double().view(1, 1, 1, -1)
        body = xh[:, sk:]
        codes = torch.clamp(torch.round(body.float() / dd.float()).double() + zz, 0, qmax)
        return torch.cat([xh[:, :sk], dd * (codes - zz)], dim=1).transpose(1, 2)          # [B,H,ntok,D]
    qh, kh, vh = dq(q, fq_q, T), dq(k, fq_k, S), dq(v, fq_v, S)
    p = torch.softmax(torch.matmul(qh, kh.transpose(-2, -1)) * scale, dim=-1).float()
    pq = p[..., skip:]
    if mode in (1, 2):
        dl = pq.max() if mode == 1 else delta.cpu().to(p.device)[0]
        code = torch.clamp(torch.round(-torch.log2(pq / dl)), 0, qmax)
        pq = dl * 2.0 ** (-code)
    else:
        pq = delta[0] * torch.clamp(torch.round(pq / delta[0]), 0, qmax)
    pf = torch.cat([p[..., :skip], pq], dim=-1).double()
    ref = torch.matmul(pf, vh).transpose(1, 2).reshape(B, T, H * D).float()
    for name, other in (("float64 formula", ref), ("bf16x3 path", ref3)):
        row_err = (out - other).view(B * T, -1).norm(dim=1) / other.view(B * T, -1).norm(dim=1).clamp_min(1e-20)
        # isolated log2-code flips of probabilities that sit on a rounding tie (as in the other attention tests)
        # the int8 contraction is exact; the bf16x3 products (like the reference's fp32 matmul) carry ~1e-7·Σ|q||k| of
        # absolute score error, which the softmax turns into ~1e-5 relative error of every probability at these magnitudes
        assert row_err.median().item() < (1e-5 if name.startswith("float64") else 1e-4), (name, row_err.median().item())
        # a row holds S probabilities, each flips with the same small rate (measured ~1e-5 per entry): scale the bound
        assert (row_err > 1e-4).float().mean().item() < max(0.02, 2.5e-5 * S), (name, (row_err > 1e-4).float().mean().item())
        assert rel_l2(out.cpu(), other.cpu()) < 4e-3, (name, rel_l2(out.cpu(), other.cpu()))


# ------------------------------------------------------------------------------------------ fused GroupNorm + SiLU
@pytest.mark.parametrize("C,H,k", [(64, 12, 3), (320, 16, 3), (96, 9, 1)])
def test_fused_groupnorm_silu_quant_codes(C, H, k, dev):
    """norm -> SiLU -> conv with the GroupNorm folded into dgq_quant_act (dgq_groupnorm_scale_shift) against
    F.group_norm + F.silu + unfused quantisation: codes may differ only where the (differently rounded) normalised
    value sits on a rounding boundary (< 1e-3 of the elements, never by more than one step)."""
    import torch.nn.functional as F
    from dgq_amd import ops
    from dgq_amd.plan import plan_act
    g = torch.Generator().manual_seed(C + H)
    B, N, taps = 2, 32, k * k
    x = (torch.randn(B, C, H, H, generator=g) * 1.3 + 0.2)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w = torch.randn(N, C, k, k, generator=g) * 0.05
    from dgq_amd.synth import channel_minmax, _group_params
    wd, wz = channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, None, 4, C, taps)
    d, z = _group_params(C * taps, 8, 8, "gnf", 0)
    ab = ops.ActBinding(plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, taps, 8), pw, 8)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    y_ref_in = F.silu(F.group_norm(xg, 32, gamma.to(dev), beta.to(dev), 1e-5)).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    c_ref, rs_ref, M = ops.quant_act(y_ref_in.permute(0, 2, 3, 1), B, H, H, C, k, k, 1, pad, ab)
    sc, sh = ops.groupnorm_scale_shift(xg.permute(0, 2, 3, 1), B, H * H, C, 32, 1e-5, gamma.to(dev), beta.to(dev))
    c_fus, rs_fus, _ = ops.quant_act(xg.permute(0, 2, 3, 1), B, H, H, C, k, k, 1, pad, ab, (sc, sh, 1))
    torch.cuda.synchronize()
    diff = (c_ref.int() - c_fus.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3
    # and the statistics themselves
    mean = x.view(B, 32, -1).mean(-1)
    var = x.view(B, 32, -1).var(-1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    exp_scale = (rstd.repeat_interleave(C // 32, 1) * gamma[None])
    assert rel_l2(sc.cpu(), exp_scale) < 1e-6


@pytest.mark.parametrize("C,T,layout", [(320, 70, "perK"), (64, 33, "perM"), (1280, 9, "perK"), (96, 40, "scalar")])
def test_fused_layernorm_quant_codes(C, T, layout, dev):
    """norm -> Linear with the LayerNorm folded into dgq_quant_act (per-row statistics inside the kernel) against
    F.layer_norm + unfused quantisation: codes may differ only where the (differently rounded) normalised value sits on
    a rounding boundary (< 1e-3 of the elements, never by more than one step)."""
    import torch.nn.functional as F
    from dgq_amd import ops
    from dgq_amd.plan import plan_act
    from dgq_amd.synth import channel_minmax, _group_params
    g = torch.Generator().manual_seed(C + T)
    B, N = 2, 48
    x = (torch.randn(B, T, C, generator=g) * 1.7 - 0.4)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w = torch.randn(N, C, generator=g) * 0.05
    wd, wz = channel_minmax(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.zeros(N, device=dev), 4, C, 1)
    if layout == "perK":
        d, z = _group_params(C, 8, 8, "lnf", 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", C, 1, 8)
    elif layout == "perM":
        d, z = _group_params(T, 4, 8, "lnf", 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", C, 1, 8)
    else:
        lay = plan_act(torch.tensor(0.031), torch.tensor(131.0), "linear", C, 1, 8)
    ab = ops.ActBinding(lay, pw, 8)
    xg = x.to(dev)
    ref_in = F.layer_norm(xg, (C,), gamma.to(dev), beta.to(dev), 1e-5)
    c_ref, rs_ref, M = ops.quant_act(ref_in.view(B * T, C), B * T, 1, 1, C, 1, 1, 1, 0, ab)
    c_fus, rs_fus, _ = ops.quant_act(xg.view(B * T, C), B * T, 1, 1, C, 1, 1, 1, 0, ab, None, (gamma.to(dev), beta.to(dev), 1e-5))
    torch.cuda.synchronize()
    diff = (c_ref.int() - c_fus.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3
    y_ref = ops.quant_linear(ref_in, ab)
    y_fus = ops.quant_linear(xg, ab, ln=(gamma.to(dev), beta.to(dev), 1e-5))
    torch.cuda.synchronize()
    assert rel_l2(y_fus.cpu(), y_ref.cpu()) < 2e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("D,T,S,H,mode,skip", [(40, 200, 200, 2, 1, 0), (80, 96, 77, 3, 1, 1), (160, 64, 77, 2, 3, 1), (80, 256, 1024, 4, 1, 1),
                                               (40, 384, 600, 2, 3, 0)])
def test_attention_half_io_equals_fp32_path(D, T, S, H, mode, skip, dtype, dev):
    """dgq_attention on fp16 / bf16 tensors (the reference's --fp16 mode) == the fp32 entry point on the same values,
    rounded once to the output dtype: only loads and the final store follow the tensor dtype.  The last two cases are key-split
    launches (both halves' parts fp32, their sum rounded once by attn3_add16_kernel)."""
    from dgq_amd import ops
    g = torch.Generator().manual_seed(D + T)
    B, bits = 2, 8
    q, k, v = (torch.randn(B, n, H * D, generator=g).to(dev).to(dtype) for n in (T, S, S))
    delta = torch.tensor([0.004], device=dev) if mode == 3 else None
    fq = ((1, torch.full((T,), 0.03, device=dev), torch.full((T,), 128.0, device=dev), 0, bits), None, None)
    o_h = ops.attention(q, k, v, H, D, D ** -0.5, mode, skip, delta, bits, fq=fq)
    o_f = ops.attention(q.float(), k.float(), v.float(), H, D, D ** -0.5, mode, skip, delta, bits, fq=fq)
    torch.cuda.synchronize()
    assert o_h.dtype == dtype
    assert torch.equal(o_h, o_f.to(dtype))


@pytest.mark.parametrize("D,T,S,mode,skip,fused", [(80, 256, 1024, 1, 1, True), (64, 320, 300, 3, 0, True), (160, 64, 256, 2, 0, False),
                                                  (40, 512, 2048 - 5, 1, 0, True)])
def test_attention_key_split_equals_unsplit(D, T, S, mode, skip, fused, dev, monkeypatch):
    """Under-filled grids run every (query block, batch·head) as two workgroups over the two halves of the key tiles
    (attn_bf16x3.hip: launch_attn3): statistics halves merged by a small kernel, the second half's part of o added by another.
    Against the same call unsplit (DGQ_ATTN_SPLIT=0): the row statistics merge in another order, so l differs in its last bits
    (and with it the real-time δ) and a log2 / uniform code of a probability on a rounding tie may move — one such flip changes
    that p̂ by a factor 2.  Asserted: relative L2 within the 4e-3 every attention test grants the float64 formula (measured
    6e-4 with the real-time δ, <= 1e-6 with a static one), >= 99 % (static δ: 99.9 %) of the elements within 1e-5 of the output's
    scale (the flips are isolated: an ulp of a0 ~ 2e-6 in log2 units is also their rate), the split call deterministic; unfused cases also against the float64 formula."""
    from dgq_amd import ops
    B, H, bits = 2, 4, 8
    g = torch.Generator().manual_seed(D + T + S)
    q, k, v = (torch.randn(B, n, H * D, generator=g).to(dev) for n in (T, S, S))
    scale = D ** -0.5
    delta = None if mode == 1 else torch.tensor([1.0 if mode == 2 else 1.0 / 255.0], device=dev)
    fq = None
    if fused:
        tab = lambda n: (torch.rand(n, generator=g).to(dev) * 0.02 + 0.02, torch.randint(100, 156, (n,), generator=g).float().to(dev))
        fq = ((1,) + tab(T) + (0, 8), (1,) + tab(S - skip) + (skip, 8), (2,) + tab(D) + (0, 8))
    monkeypatch.setenv("DGQ_ATTN_SPLIT", "0")
    o1 = ops.attention(q, k, v, H, D, scale, mode, skip, delta, bits, fq=fq).clone()
    monkeypatch.setenv("DGQ_ATTN_SPLIT", "100000")
    o2 = ops.attention(q, k, v, H, D, scale, mode, skip, delta, bits, fq=fq).clone()
    o3 = ops.attention(q, k, v, H, D, scale, mode, skip, delta, bits, fq=fq).clone()
    torch.cuda.synchronize()
    assert torch.equal(o2, o3), "the split call is not deterministic"
    err = (o1 - o2).abs()
    assert rel_l2(o2.cpu(), o1.cpu()) < (4e-3 if mode == 1 else 2e-4), rel_l2(o2.cpu(), o1.cpu())
    assert float((err <= 1e-5 * float(o1.abs().max())).float().mean()) >= (0.99 if mode == 1 else 0.999)
    if not fused:
        qh, kh, vh = (x.double().view(B, -1, H, D).transpose(1, 2) for x in (q, k, v))
        p = torch.softmax(torch.matmul(qh, kh.transpose(-2, -1)) * scale, dim=-1).float()
        pq = orc.log_quant(p.cpu(), delta[0].cpu(), bits).to(dev) if mode == 2 else orc.uaq(p.cpu(), delta[0].cpu(), torch.tensor(0.0), bits).to(dev)
        ref = torch.matmul(pq.double(), vh).transpose(1, 2).reshape(B, T, H * D).float()
        assert rel_l2(o2.cpu(), ref.cpu()) < 4e-3


@pytest.mark.parametrize("D,mode,skip", [(40, 1, 1), (64, 1, 0), (40, 3, 0)])
def test_attention_wide_blocks(D, mode, skip, dev):
    """T = 2048 with 32 (batch, head) pairs selects the 8-wave (256-row) blocks of the bf16x3 kernels; checked against the
    materialised reference sequence evaluated in float64 on the GPU, and a ragged T (2048 - 37) exercises the tail block."""
    from dgq_amd import ops
    B, H, S, bits = 2, 16, 333, 8
    for T in (2048, 2048 - 37):
        g = torch.Generator().manual_seed(D + T + mode)
        q, k, v = (torch.randn(B, n, H * D, generator=g).to(dev) for n in (T, S, S))
        scale = D ** -0.5
        qh, kh, vh = (x.double().view(B, -1, H, D).transpose(1, 2) for x in (q, k, v))
        p = torch.softmax(torch.matmul(qh, kh.transpose(-2, -1)) * scale, dim=-1).float()
        delta = None
        if mode == 1:
            pq = orc.log_quant(p[..., skip:].cpu(), p[..., skip:].max().cpu(), bits).to(dev)
        else:
            delta = torch.tensor([float(p.max()) / 255.0], device=dev)
            pq = orc.uaq(p[..., skip:].cpu(), delta[0].cpu(), torch.tensor(0.0), bits).to(dev)
        pf = torch.cat([p[..., :skip], pq], dim=-1) if skip else pq
        ref = torch.matmul(pf.double(), vh).transpose(1, 2).reshape(B, T, H * D).float()
        o = ops.attention(q, k, v, H, D, scale, mode, skip, delta, bits)
        torch.cuda.synchronize()
        assert rel_l2(o.cpu(), ref.cpu()) < 4e-3
        row_err = (o - ref).view(B * T, -1).norm(dim=1) / ref.view(B * T, -1).norm(dim=1)
        assert row_err.median().item() < 1e-5
        assert (row_err > 1e-4).float().mean().item() < 0.05      # a row holds 16 heads x 333 probabilities: more tie flips per row


# ------------------------------------------------------------------------------------------ batched small-M linear
@pytest.mark.parametrize("M,K,wbits,abits,dtype", [(2, 1280, 4, 8, torch.float32), (8, 1280, 4, 6, torch.float32),
                                                     (16, 128, 8, 8, torch.float32), (2, 320, 4, 8, torch.float16)])
def test_linear_smallm_batch_equals_the_gemm_path(M, K, wbits, abits, dtype, dev):
    """dgq_linear_smallm_batch (every time_emb_proj of a forward in one launch) against the regular two-kernel path
    (dgq_quant_act with the SiLU prologue + dgq_gemm_wxa8) layer by layer: the same integer sums and the same epilogue
    expression, so the outputs are bit-identical."""
    from dgq_amd import ops
    from dgq_amd.plan import plan_act
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * 2).to(dev, dtype)
    binds = []
    for N in (320, 640, 1280, 77, 1280):
        w = torch.randn(N, K, generator=g) * 0.05
        wd, wz = orc.minmax_channel(w, wbits)
        pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.randn(N, generator=g).to(dev), wbits, K, 1)
        d = torch.tensor(0.01 + 0.03 * float(torch.rand(1, generator=g)))
        z = torch.tensor(float(torch.randint(int(2 ** abits * 0.3), int(2 ** abits * 0.7), (1,), generator=g)))
        binds.append(ops.ActBinding(plan_act(d, z, "linear", K, 1, abits), pw, abits))
    outs = ops.linear_smallm_batch(x, binds, pre_act=1)
    for ab, y in zip(binds, outs):
        ref = ops.quant_linear(x, ab, pre_act=1)
        assert y.shape == ref.shape and y.dtype == dtype
        assert torch.equal(y, ref), (ab.pw.N, (y.float() - ref.float()).abs().max().item())


@pytest.mark.parametrize("M,K,with_ln", [(154, 768, False), (512, 320, True), (64, 1280, True)])
def test_quant_linear_multi_equals_single_calls(M, K, with_ln, dev):
    """dgq_quant_act_batch + dgq_gemm_wxa8_batch (layers sharing one input: q/k/v, the cross-attention k/v of the text
    context) against one dgq_quant_act + dgq_gemm_wxa8 per layer: same kernels, same arithmetic -> bit-identical,
    for mixed per-K (grouped) and per-token / scalar tables and mixed widths in one call."""
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(2, M // 2, K, generator=g).to(dev)
    ln = None
    if with_ln:
        ln = ((1 + 0.1 * torch.randn(K, generator=g)).to(dev), (0.05 * torch.randn(K, generator=g)).to(dev), 1e-5)
    binds = []
    for i, (N, mode) in enumerate(((320, "perK"), (640, "perM"), (320, "perK"), (1280, "scalar"), (640, "perK"), (320, "perM"),
                                   (320, "perK"), (320, "perK"), (640, "perK"), (320, "perK"), (1280, "perK"))):
        w = torch.randn(N, K, generator=g) * 0.05
        wd, wz = orc.minmax_channel(w, 4)
        pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.randn(N, generator=g).to(dev), 4, K, 1)
        if mode == "perK":
            d, z = synth._group_params(K, 16, 8, "multi|%d" % i, 0)
            lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", K, 1, 8)
        elif mode == "perM":
            d, z = synth._group_params(M // 2, 16, 8, "multi|%d" % i, 0)
            lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
        else:
            lay = plan_act(torch.tensor(0.03), torch.tensor(120.0), "linear", K, 1, 8)
        binds.append(ops.ActBinding(lay, pw, 8))
    outs = ops.quant_linear_multi(x, binds, ln=ln)
    for ab, y in zip(binds, outs):
        ref = ops.quant_linear(x, ab, ln=ln)
        assert y.shape == ref.shape
        assert torch.equal(y, ref), (ab.pw.N, ab.mode, (y - ref).abs().max().item())


@pytest.mark.parametrize("M,K,N,mode,dtype", [(512, 320, 2560, "perK", torch.float32), (300, 640, 5120, "perM", torch.float32),
                                              (1024, 1280, 10240, "perK", torch.bfloat16), (96, 64, 200, "perK", torch.float16)])
def test_gemm_geglu_epilogue_equals_projection_then_geglu(M, K, N, mode, dtype, dev):
    """ff.net.0 with the GEGLU in the GEMM epilogue (weight rows (value, gate) interleaved at pack time, dgq_gemm_extra_t.geglu)
    against the same layer's plain output followed by value·gelu(gate) (diffusers_rewrite/sd.py:210-222): the projection
    values are the same bit for bit (same tiles, same accumulation order per column), so only the final product differs
    by the rounding of one multiply-chain — and by the output dtype's rounding for half types."""
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 1.2).to(dev, dtype)
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g) * 0.1
    wd, wz = orc.minmax_channel(w, 4)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "geglu|%d" % N, 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "linear", K, 1, 8)
    else:
        d, z = synth._group_params(M, 16, 8, "geglu|%d" % N, 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "linear", K, 1, 8)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, b.to(dev), 4, K, 1)
    y = ops.quant_linear(x, ops.ActBinding(lay, pw, 8)).float()
    a, gt = y.chunk(2, dim=-1)
    ref = a * torch.nn.functional.gelu(gt)
    rp = torch.stack([torch.arange(N // 2), torch.arange(N // 2) + N // 2], 1).flatten()
    pwi = ops.PackedWeight(w[rp].to(dev), wd[rp].to(dev), wz[rp].to(dev), None, b[rp].to(dev), 4, K, 1)
    out = ops.quant_linear(x, ops.ActBinding(lay, pwi, 8), geglu=True)
    torch.cuda.synchronize()
    assert out.shape == (M, N // 2) and out.dtype == dtype
    tol = 1e-6 if dtype == torch.float32 else (2e-3 if dtype == torch.float16 else 8e-3)
    assert rel_l2(out.float().cpu(), ref.cpu()) < tol, rel_l2(out.float().cpu(), ref.cpu())


@pytest.mark.parametrize("B,C,H,N,k,mode,dtype", [(2, 64, 16, 64, 3, "perK", torch.float32), (2, 320, 32, 320, 3, "perM", torch.float32),
                                                   (1, 128, 8, 96, 1, "perK", torch.float32), (2, 96, 16, 640, 3, "perK", torch.bfloat16),
                                                   (2, 64, 64, 160, 3, "perK", torch.float32)])
def test_groupnorm_from_gemm_epilogue_partials(B, C, H, N, k, mode, dtype, dev):
    """GroupNorm statistics out of the producing GEMM's epilogue (dgq_gemm_extra_t.gn_partial + dgq_groupnorm_from_partials)
    against the standalone statistics kernels on the stored tensor: the per-(batch, channel) scale / shift agree to
    rounding (1e-5 relative), for one source and for a channel concat of two; with a residual in the epilogue; the conv
    output itself is bit-identical with and without the partials."""
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    g = torch.Generator().manual_seed(B + C + N)
    taps = k * k
    x = torch.randn(B, C, H, H, generator=g).to(dev, dtype)
    w = torch.randn(N, C, k, k, generator=g) * 0.05
    wd, wz = orc.minmax_channel(w, 4)
    pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.randn(N, generator=g).to(dev), 4, C, taps)
    if mode == "perK":
        d, z = synth._group_params(C * taps, 16, 8, "gnp|%d" % N, 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, taps, 8, kw=k)
    else:
        d, z = synth._group_params(H * H, 16, 8, "gnp|%d" % N, 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "conv", C, taps, 8, kw=k)
    ab = ops.ActBinding(lay, pw, 8)
    res = torch.randn(B, N, H, H, generator=g).to(dev, dtype)
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).to(dev)
    beta = (0.1 * torch.randn(N, generator=g)).to(dev)
    for residual in (None, res):
        y = ops.quant_conv2d(x, ab, k, k, 1, k // 2, residual=residual, gn_out=True)
        y0 = ops.quant_conv2d(x, ab, k, k, 1, k // 2, residual=residual, gn_out=False)
        assert torch.equal(y, y0) and getattr(y, "_dgq_gn", None) is not None and getattr(y0, "_dgq_gn", None) is None
        sc, sh = ops.groupnorm_from_partials(y._dgq_gn, 32, 1e-5, gamma, beta)
        ys = y.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        sc0, sh0 = ops.groupnorm_scale_shift(ys, B, H * H, N, 32, 1e-5, gamma, beta)
        torch.cuda.synchronize()
        assert rel_l2(sc.cpu(), sc0.cpu()) < 1e-5 and rel_l2(sh.cpu(), sh0.cpu()) < 1e-5, (rel_l2(sc.cpu(), sc0.cpu()), rel_l2(sh.cpu(), sh0.cpu()))
    # channel concat of two producers: statistics of torch.cat([y, y2], 1) from the two partial buffers
    y2 = ops.quant_conv2d(x.flip(0), ab, k, k, 1, k // 2)
    cat = ops.cat_channels(y, y2)
    g2 = torch.cat([gamma, gamma.flip(0)]); b2 = torch.cat([beta, beta.flip(0)])
    sc, sh = ops.groupnorm_from_partials(cat._dgq_gn, 32, 1e-6, g2, b2)
    cs = cat.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
    sc0, sh0 = ops.groupnorm_scale_shift(cs, B, H * H, 2 * N, 32, 1e-6, g2, b2)
    torch.cuda.synchronize()
    assert rel_l2(sc.cpu(), sc0.cpu()) < 1e-5 and rel_l2(sh.cpu(), sh0.cpu()) < 1e-5


@pytest.mark.parametrize("force", ["32,64,3", "64,64,2"])
def test_groupnorm_partials_from_the_splitk_combine(force, dev, monkeypatch):
    """The same statistics when the producing GEMM is K-split: the combine kernel (splitk_epilogue_kernel) writes the partials
    — four lanes per (16-row block, 4 columns) unit, merged by two equal-count Chan steps.  Output bit-identical to the combine
    without partials; scale / shift equal to the statistics pass over the stored tensor to 1e-5."""
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    monkeypatch.setenv("DGQ_GEMM_FORCE", force)
    for (B, C, H, N, k, dtype) in ((2, 64, 16, 72, 3, torch.float32), (1, 96, 8, 320, 3, torch.float32), (2, 64, 8, 64, 3, torch.bfloat16)):
        g = torch.Generator().manual_seed(B + C + N)
        taps = k * k
        x = torch.randn(B, C, H, H, generator=g).to(dev, dtype)
        w = torch.randn(N, C, k, k, generator=g) * 0.05
        wd, wz = orc.minmax_channel(w, 4)
        pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.randn(N, generator=g).to(dev), 4, C, taps)
        d, z = synth._group_params(C * taps, 16, 8, "gnsk|%d" % N, 0)
        ab = ops.ActBinding(plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, taps, 8, kw=k), pw, 8)
        assert ops._lib.load().dgq_gemm_plan_splits(B * H * H, N, ab.Kp, 4, 0, ops.WORKSPACE_BYTES) > 1
        gamma = (1 + 0.1 * torch.randn(N, generator=g)).to(dev)
        beta = (0.1 * torch.randn(N, generator=g)).to(dev)
        res = torch.randn(B, N, H, H, generator=g).to(dev, dtype)
        for residual in (None, res):
            y = ops.quant_conv2d(x, ab, k, k, 1, 1, residual=residual, gn_out=True)
            y0 = ops.quant_conv2d(x, ab, k, k, 1, 1, residual=residual, gn_out=False)
            assert torch.equal(y, y0) and getattr(y, "_dgq_gn", None) is not None
            sc, sh = ops.groupnorm_from_partials(y._dgq_gn, 8, 1e-5, gamma, beta)
            ys = y.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
            sc0, sh0 = ops.groupnorm_scale_shift(ys, B, H * H, N, 8, 1e-5, gamma, beta)
            torch.cuda.synchronize()
            assert rel_l2(sc.cpu(), sc0.cpu()) < 1e-5 and rel_l2(sh.cpu(), sh0.cpu()) < 1e-5, (force, N, rel_l2(sc.cpu(), sc0.cpu()))


# ------------------------------------------------------------------------------------------ weight-only state
@pytest.mark.parametrize("shape", [(2, 8, 9, 11, 20, 3, 1, 1), (1, 4, 16, 16, 64, 3, 2, 1), (3, 12, 7, 5, 40, 1, 1, 0), (2, 320, 8, 8, 70, 3, 1, 1),
                                   (2, 320, 12, 12, 4, 3, 1, 1), (1, 64, 9, 9, 7, 3, 2, 1)])      # the last two: the N <= 8 kernel (conv_out)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_weight_only_conv_kernel_vs_float64(shape, dtype, dev):
    """dgq_conv2d_f32w (exact-fp32 MFMA, im2col folded into the load) against F.conv2d evaluated in float64: ragged M / N / K
    edges, stride 2, 1x1, padding; fp32 within 2e-6 relative (the fp32 accumulation itself), fp16 / bf16 I/O within their rounding."""
    from dgq_amd import ops
    B, C, H, W, N, k, stride, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, C, H, W, generator=g).to(dev, dtype)
    w = torch.randn(N, C, k, k, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    wn = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(dev)
    y = ops.conv2d_f32w(x, wn, b.to(dev), k, k, stride, pad)
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double(), b.double(), stride=stride, padding=pad)
    assert y.shape == ref.shape and y.dtype == dtype
    tol = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 6e-3}[dtype]
    assert rel_l2(y.double().cpu(), ref) < tol
    # Linear form
    xl = torch.randn(5, 7, C * 3, generator=g).to(dev, dtype)
    wl = torch.randn(N, C * 3, generator=g) * 0.1
    yl = ops.conv2d_f32w(xl, wl.to(dev), None, 1, 1, 1, 0)
    refl = xl.double().cpu() @ wl.double().t()
    assert yl.shape == refl.shape and rel_l2(yl.double().cpu(), refl) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [4, 72])
def test_weight_only_conv_kernel_with_folded_groupnorm_silu(N, dtype, dev):
    """conv(SiLU(GroupNorm(x))) in one launch of dgq_conv2d_f32w (conv_out of the UNets: FP conv behind conv_norm_out + SiLU)
    against the float64 composition, for the N <= 8 form and the tiled form."""
    from dgq_amd import ops
    g = torch.Generator().manual_seed(N)
    B, C, H = 2, 64, 12
    x = (torch.randn(B, C, H, H, generator=g) * 2 + 0.5).to(dev, dtype)
    w = torch.randn(N, C, 3, 3, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    wn = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(dev)
    y = ops.conv2d_f32w(x, wn, b.to(dev), 3, 3, 1, 1, norm=(8, 1e-5, gamma.to(dev), beta.to(dev), 1))
    xn = torch.nn.functional.silu(torch.nn.functional.group_norm(x.double().cpu(), 8, gamma.double(), beta.double(), 1e-5))
    ref = torch.nn.functional.conv2d(xn, w.double(), b.double(), padding=1)
    # (bf16: the tensor's own 2^-9 rounding of the output; the statistics and the contraction stay fp32)
    assert rel_l2(y.double().cpu(), ref) < (5e-6 if dtype == torch.float32 else 6e-3), rel_l2(y.double().cpu(), ref)


# ------------------------------------------------------------------------------------------ 3x3 convolution, quantiser inside the GEMM launch
@pytest.mark.parametrize("geom", [(2, 320, 64, 64, 320), (1, 320, 32, 32, 320), (2, 64, 16, 24, 160), (1, 128, 8, 8, 320), (2, 32, 4, 8, 160), (1, 320, 64, 64, 160)],
                         ids=lambda g: "x".join(str(v) for v in g))
@pytest.mark.parametrize("mode", ["perK", "perM"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3x3_quantise_inside_gemm_equals_two_launches(geom, mode, dtype, dev, monkeypatch):
    """csrc/gemm_convq.hip (dgq_gemm_act_t with kh > 1): a workgroup stages the input patch of its 4 x 8 output positions (GroupNorm + SiLU
    folded, zeros outside the image as F.unfold pads them), quantises the unfolded operand slab by slab into LDS and contracts it —
    quant_layer.py:626-661 as ONE kernel, no int8 code matrix.  Codes, row sums and the integer contraction are those of
    dgq_quant_act + dgq_gemm_wxa8: per-M outputs equal bit for bit; per-K outputs up to the order of the fp32 group sums (the tile
    family splits a K tile's chunks over two waves, this kernel does not: 1e-6 where both forms quantise on the block-staged lane order).  With GroupNorm prologue, residual and the
    GroupNorm partials of the output (compared through the scale / shift they finalise to)."""
    if dtype == torch.bfloat16 and geom[1] * geom[2] > 320 * 32:
        pytest.skip("half-type coverage on the small geometries")
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    B, C, H, W, N = geom
    gen = torch.Generator().manual_seed(sum(geom) + (1 if mode == "perK" else 0))
    K = C * 9
    w = (torch.randn(N, C, 3, 3, generator=gen) * 0.05).to(dev)
    x = (torch.randn(B, C, H, W, generator=gen) * 1.3 + 0.2).to(dev).to(dtype)
    wd, wz = synth.channel_minmax(w.cpu(), 4)
    pw = ops.PackedWeight(w, wd.to(dev), wz.to(dev), None, torch.randn(N, generator=gen).to(dev), 4, C, 9)
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "convq|%d" % K, 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, 9, 8)
    else:
        d, z = synth._group_params(H * W, 16, 8, "convq|%d" % K, 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "conv", C, 9, 8)
    ab = ops.ActBinding(lay, pw, 8)
    res = torch.randn(B, N, H, W, generator=gen).to(dev).to(dtype)
    norm = (32, 1e-5, torch.randn(C, generator=gen).to(dev), torch.randn(C, generator=gen).to(dev), 1) if C % 32 == 0 else None
    assert ops.conv_act_fuses(ab, B, H, W, C, 3, 3, 1, 1, dtype)
    outs = []
    for fuse in (False, True):
        monkeypatch.setattr(ops, "CONV_FUSE", fuse)
        y = ops.quant_conv2d(x, ab, 3, 3, 1, 1, norm=norm, residual=res)
        gn = ops._gn_of(y)
        assert gn is not None
        sc, sh = ops.groupnorm_from_partials(gn, 32, 1e-5, torch.ones(N, device=dev), torch.zeros(N, device=dev))
        outs.append((y.float().clone(), sc.clone(), sh.clone()))
    torch.cuda.synchronize()
    (y0, sc0, sh0), (y1, sc1, sh1) = outs
    assert torch.isfinite(y1).all()
    if mode == "perM":
        assert torch.equal(y0, y1), (y0 - y1).abs().max().item()
    else:
        # (below 2048 rows the two-launch form quantises on the row-wise scatter kernel, whose fp32 row sums Σ δ_k·s add up in another
        # order than the block-staged kernel's, which this one reproduces: y = α·(acc − z_w·rowsum) cancels, 1e-7 of a row sum shows as 1e-5)
        tol = (1e-6 if B * H * W >= 2048 else 2e-5) if dtype == torch.float32 else 3e-3
        assert rel_l2(y1, y0) < tol, rel_l2(y1, y0)
    assert rel_l2(sc1, sc0) < 1e-5 and rel_l2(sh1, sh0) < 1e-4, (rel_l2(sc1, sc0), rel_l2(sh1, sh0))


# ------------------------------------------------------------------------------------------ implicit-im2col convolution
@pytest.mark.parametrize("case", [c for c in recipes.f3_cases() if c["kind"] == "conv" and c["state"] == "wa" and c["layout"] == "scalar"
                                  and c["wbits"] == 4], ids=lambda c: c["name"])
def test_implicit_conv_bit_identical_to_materialised(case, dev, monkeypatch):
    """Scalar-δ convolutions (the reference's native path F.conv2d(aqtizer(x), ŵ), quant_layer.py:659) through the implicit-im2col
    GEMM (dgq_gemm_conv_t: the input quantised once per pixel, taps gathered by the LDS-DMA, out-of-image taps = the code of 0.0)
    against the materialising pass: the SAME output bit for bit (integer contraction, exact integer row sums), hence the same
    2e-5 distance to the reference's golden; 3x3 stride 1, 3x3 stride 2 pad 1, 1x1 stays on the ordinary path."""
    from dgq_amd import ops
    from dgq_amd.plan import plan_act
    g = gold("f3_layers.pt")[case["name"]]
    inp = recipes.f3_inputs(case)
    w = inp["w"].to(dev)
    C, taps = w.shape[1], case["k"] ** 2
    pw = ops.PackedWeight(w, g["wdelta"].to(dev), g["wzp"].to(dev), None, inp["b"].to(dev), 4, C, taps)
    ab = ops.ActBinding(plan_act(inp["adelta"], inp["azp"], "conv", C, taps, case["abits"]), pw, case["abits"])
    assert ab.mode == "scalar"
    x = inp["x"].to(dev)
    monkeypatch.setattr(ops, "CONV_IMPLICIT", False)
    y0 = ops.quant_conv2d(x, ab, case["k"], case["k"], case["stride"], case["padding"])
    monkeypatch.setattr(ops, "CONV_IMPLICIT", True)
    y1 = ops.quant_conv2d(x, ab, case["k"], case["k"], case["stride"], case["padding"])
    torch.cuda.synchronize()
    assert torch.equal(y0, y1), (case["name"], (y0 - y1).abs().max().item())
    assert rel_l2(y1.cpu(), g["y"]) < 2e-5


@pytest.mark.parametrize("geom", [(2, 64, 24, 24, 3, 1, 1, 96), (1, 320, 16, 16, 3, 1, 1, 320), (2, 48, 17, 13, 3, 2, 1, 80), (1, 640, 8, 8, 3, 1, 1, 1280),
                                  (3, 32, 9, 11, 5, 1, 2, 40)], ids=lambda g: "x".join(str(v) for v in g))
def test_implicit_conv_geometries(geom, dev, monkeypatch):
    """the implicit path on more geometries (C not a multiple of the K tile, ragged images, stride 2, 5x5, K-split shapes, a
    zero point that makes the padded value's code non-trivial), with residual and GroupNorm prologue: bit-identical to the
    materialising pass"""
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    B, C, H, W, k, stride, pad, N = geom
    gen = torch.Generator().manual_seed(sum(geom))
    w = (torch.randn(N, C, k, k, generator=gen) * 0.05).to(dev)
    x = (torch.randn(B, C, H, W, generator=gen) * 1.5 + 0.3).to(dev)
    wd, wz = synth.channel_minmax(w.cpu(), 4)
    pw = ops.PackedWeight(w, wd.to(dev), wz.to(dev), None, torch.randn(N, generator=gen).to(dev), 4, C, k * k)
    ab = ops.ActBinding(plan_act(torch.tensor(0.037), torch.tensor(97.0), "conv", C, k * k, 8), pw, 8)
    assert ab.mode == "scalar"
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, N, Ho, Wo, generator=gen).to(dev)
    norm = None
    if C % 32 == 0:
        norm = (32, 1e-5, torch.randn(C, generator=gen).to(dev), torch.randn(C, generator=gen).to(dev), 1)
    outs = []
    for flag in (False, True):
        monkeypatch.setattr(ops, "CONV_IMPLICIT", flag)
        outs.append(ops.quant_conv2d(x, ab, k, k, stride, pad, norm=norm, residual=res))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), (geom, (outs[0] - outs[1]).abs().max().item())


@pytest.mark.parametrize("case", ["conv_128x128_b8_c320", "conv_64x64_b8_c640", "linear_m32768_geglu_in", "linear_m8192_k5120"])
def test_c5_full_size_launch_plans_equal_the_small_tile_forms(case, dev, monkeypatch):
    """the launch plans config C5 takes at its FULL size (8 prompts, 128 x 128 latents: M = 131 072 / 32 768 / 8 192 rows —
    the implicit-im2col operand, the 256-row kernel, the wide tiles) checked at those sizes, not only on small geometries: the planner's
    launch against the same layer forced onto the materialising pass / the 64 x 64 tile.  Scalar-δ layers (C5: one (δ, z) per layer):
    integer sums and one epilogue — bit for bit."""
    from dgq_amd import ops, synth, _lib
    from dgq_amd.plan import plan_act
    gen = torch.Generator().manual_seed(len(case))
    if case.startswith("conv"):
        B, C, H, N = (8, 320, 128, 320) if "c320" in case else (8, 640, 64, 640)
        w = (torch.randn(N, C, 3, 3, generator=gen) * 0.05).to(dev)
        x = (torch.randn(B, C, H, H, generator=gen) * 1.5 + 0.3).to(dev).contiguous(memory_format=torch.channels_last)
        wd, wz = synth.channel_minmax(w.cpu(), 4)
        pw = ops.PackedWeight(w, wd.to(dev), wz.to(dev), None, torch.randn(N, generator=gen).to(dev), 4, C, 9)
        ab = ops.ActBinding(plan_act(torch.tensor(0.045), torch.tensor(29.0), "conv", C, 9, 6), pw, 6)
        assert ab.mode == "scalar"
        res = torch.randn(B, N, H, H, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
        norm = (32, 1e-5, torch.randn(C, generator=gen).to(dev), torch.randn(C, generator=gen).to(dev), 1)
        y1 = ops.quant_conv2d(x, ab, 3, 3, 1, 1, norm=norm, residual=res)                       # the planner's form (implicit operand)
        monkeypatch.setattr(ops, "CONV_IMPLICIT", False)
        monkeypatch.setenv("DGQ_GEMM_FORCE", "64,64,1")
        y0 = ops.quant_conv2d(x, ab, 3, 3, 1, 1, norm=norm, residual=res)                       # int8 im2col matrix + 64 x 64 tiles
    else:
        M, K, N = (32768, 640, 5120) if "geglu" in case else (8192, 5120, 1280)
        w = torch.randn(N, K, generator=gen) * 0.05
        wd, wz = synth.channel_minmax(w, 4)
        pw = ops.PackedWeight(w.to(dev), wd.to(dev), wz.to(dev), None, torch.randn(N, generator=gen).to(dev), 4, K, 1)
        ab = ops.ActBinding(plan_act(torch.tensor(0.045), torch.tensor(29.0), "linear", K, 1, 6), pw, 6)
        x = torch.randn(M, K, generator=gen).to(dev)
        res = torch.randn(M, N, generator=gen).to(dev)
        y1 = ops.quant_linear(x, ab, residual=res)
        monkeypatch.setattr(ops, "GEMM_FUSE", False)
        monkeypatch.setenv("DGQ_GEMM_FORCE", "64,64,1")
        y0 = ops.quant_linear(x, ab, residual=res)
    torch.cuda.synchronize()
    assert torch.isfinite(y1).all() and torch.equal(y0, y1), (case, (y0 - y1).abs().max().item())


@pytest.mark.parametrize("case", ["conv3x3_perK_tile", "conv3x3_perM_convq", "conv1x1_fused", "conv3x3_splitk", "fp_conv_in"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_output_redirect_into_a_concatenation_buffer(case, dtype, dev):
    """dgq_gemm_extra_t.y2 / a strided y (ops.OutputRedirect): a layer's output stored INTO rows [:, :C1] of a [M][C1 + C2] buffer, or
    stored there AS WELL as into its own tensor ([:, C1:]) — the two halves of torch.cat([h, skip], 1) of channels-last tensors
    (sd.py:558-613) — equals the plain call bit for bit, the rest of the buffer untouched; through the tile family, the conv kernel with
    its quantiser inside, the fused 1x1 form, a K-split launch with its combine, and the FP conv_in kernel."""
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    gen = torch.Generator().manual_seed(len(case))
    B, C, H, N, k, mode = {"conv3x3_perK_tile": (2, 64, 16, 96, 3, "perK"), "conv3x3_perM_convq": (1, 320, 32, 320, 3, "perM"),
                           "conv1x1_fused": (2, 320, 32, 320, 1, "perK"), "conv3x3_splitk": (2, 1280, 8, 1280, 3, "perK"),
                           "fp_conv_in": (2, 4, 16, 320, 3, None)}[case]
    x = (torch.randn(B, C, H, H, generator=gen) * 1.3 + 0.2).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    res = torch.randn(B, N, H, H, generator=gen).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    M, C1 = B * H * H, 64
    if mode is None:
        w = torch.randn(N, C * 9, generator=gen).to(dev) * 0.1
        b = torch.randn(N, generator=gen).to(dev)
        run = lambda **kw: ops.conv2d_f32w(x, w, b, 3, 3, 1, 1, out2=kw.get("out2"), gn_out=True)
    else:
        w = (torch.randn(N, C, k, k, generator=gen) * 0.05).to(dev)
        wd, wz = synth.channel_minmax(w.cpu(), 4)
        pw = ops.PackedWeight(w, wd.to(dev), wz.to(dev), None, torch.randn(N, generator=gen).to(dev), 4, C, k * k)
        if mode == "perK":
            d, z = synth._group_params(C * k * k, 16, 8, "rd|%s" % case, 0)
            lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, k * k, 8)
        else:
            d, z = synth._group_params(H * H, 16, 8, "rd|%s" % case, 0)
            lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "conv", C, k * k, 8)
        ab = ops.ActBinding(lay, pw, 8)
        run = lambda **kw: ops.quant_conv2d(x, ab, k, k, 1, k // 2, residual=res, **kw)
    y0 = run()
    buf = torch.full((M, C1 + N + 32), 7.0, dtype=dtype, device=dev)
    y1 = run(out2=buf[:, C1:C1 + N])                                   # second copy
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(buf[:, C1:C1 + N].view(B, H, H, N).permute(0, 3, 1, 2), y0)
    assert bool((buf[:, :C1] == 7.0).all()) and bool((buf[:, C1 + N:] == 7.0).all())
    if mode is None:
        # ... and the FP kernel's GroupNorm partials of its output (conv_in feeds norm1 of the first resnet) against the statistics pass
        gam, bet = torch.linspace(0.5, 1.5, N, device=dev), torch.linspace(-0.2, 0.2, N, device=dev)
        sc, sh = ops.groupnorm_from_partials(ops._gn_of(y1), 32, 1e-5, gam, bet)
        sc0, sh0 = ops.groupnorm_scale_shift(y1.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1), B, H * H, N, 32, 1e-5, gam, bet)
        torch.cuda.synchronize()
        assert rel_l2(sc.cpu(), sc0.cpu()) < 1e-5 and rel_l2(sh.cpu(), sh0.cpu()) < 1e-4
    if mode is not None:
        buf.fill_(7.0)
        y2 = run(out=buf[:, C1:C1 + N])                                # the output itself lives in the buffer
        torch.cuda.synchronize()
        assert y2.data_ptr() == buf[:, C1:].data_ptr() and torch.equal(y2, y0)
        assert bool((buf[:, :C1] == 7.0).all()) and bool((buf[:, C1 + N:] == 7.0).all())
        g0, g2 = ops._gn_of(y0), ops._gn_of(y2)
        assert (g0 is None) == (g2 is None)
        if g0 is not None:
            assert torch.equal(g0["parts"][0][0], g2["parts"][0][0])


# ------------------------------------------------------------------------------------------ step glue (glue.hip)
@pytest.mark.parametrize("dim,tdtype", [(320, torch.int64), (256, torch.float32), (320, torch.float32)])
def test_timestep_embedding_is_the_torch_chain(dim, tdtype, dev):
    """dgq_timestep_embedding against Timesteps.forward's own torch chain (diffusers_rewrite/sd.py:19-39) on the device: the same
    operations in the same order — equal bit for bit where the device libm functions coincide, within 2 ulp of an fp32 value in
    [-1, 1] (2.4e-7) otherwise; and against the float64 formula on the host within fp32 evaluation error of the argument."""
    import math
    from dgq_amd import ops
    for tv in ([981], [1], [999.0, 749.0, 0.0, 512.0, 3.0, 64.0]):
        t = torch.tensor(tv, dtype=tdtype, device=dev)
        if len(tv) == 1:
            t = t.expand(2)                                                   # the expanded single timestep (stride 0)
        half = dim // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=dev) / (half - 0.0))
        ang = t[:, None].float() * freqs[None, :]
        want = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
        got = ops.timestep_embedding(t, dim)
        torch.cuda.synchronize()
        assert got.shape == want.shape
        assert (got - want).abs().max().item() <= 2.4e-7, (tv, (got - want).abs().max().item())
        f64 = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float64) / half)
        a64 = t.cpu().double()[:, None] * f64[None, :]
        w64 = torch.cat([torch.cos(a64), torch.sin(a64)], dim=-1)
        assert (got.cpu().double() - w64).abs().max().item() <= 1e-3           # |arg| <= 999 carries ~1e-4 of fp32 error


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_cfg_ddim_step_is_the_eager_chain_bit_for_bit(dtype, dev):
    """guidance + DDIM update as one launch == the ten eager torch kernels of the pipeline loop (pipeline_stable_diffusion.py:1037-1044
    + scheduling_ddim.py step): same fp32 operations, same order, torch's `/ host scalar` as `* (1 / scalar)`; 16-bit tensors: every
    statement's result rounded to the type, as each eager kernel stores it."""
    from dgq_amd.scheduler import DDIMScheduler
    sch = DDIMScheduler(50)
    g = torch.Generator().manual_seed(3)
    from dgq_amd import ops
    calls = []
    real = ops.cfg_ddim_step
    ops.cfg_ddim_step = lambda *a: calls.append(real(*a)) or calls[-1]
    try:
        for t in (sch.timesteps[0], sch.timesteps[17], sch.timesteps[-1]):
            for P, cl in ((1, True), (1, False), (3, True)):          # channels-last: the layout the UNet's output has
                eps = torch.randn(2 * P, 4, 64, 64, generator=g).to(dev, dtype)
                if cl:
                    eps = eps.contiguous(memory_format=torch.channels_last)
                x = torch.randn(P, 4, 64, 64, generator=g).to(dev, dtype)
                e_u, e_c = eps.chunk(2)
                want = sch.step(e_u + 7.5 * (e_c - e_u), t, x)
                n0 = len(calls)
                got = sch.step_guided(eps, t, x, 7.5)
                torch.cuda.synchronize()
                assert len(calls) == n0 + 1 and calls[-1] is not None, "the single-launch form was not taken"
                assert torch.equal(got, want), (t, P, cl, (got - want).abs().max().item())
    finally:
        ops.cfg_ddim_step = real
    # host tensors take the torch statements (pipeline glue, no device op involved)
    eps, x = torch.randn(2, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    e_u, e_c = eps.chunk(2)
    assert torch.equal(sch.step_guided(eps, 1, x, 7.5), sch.step(e_u + 7.5 * (e_c - e_u), 1, x))


@pytest.mark.parametrize("D,T,S,mode,qmode", [(40, 4096, 1024, 2, 2), (40, 4096, 512, 2, 1), (64, 4096, 320 - 64, 3, 1), (64, 4096, 1024, 2, 2)])
def test_attention_two_tiles_per_stage_is_bit_identical(D, T, S, mode, qmode, dev, monkeypatch):
    """The wide (8-wave) self-attention launches walk their key tiles two per ring stage (attn_bf16x3.hip, TPS = 2: one barrier per
    64 keys) where the tile count is even and >= 8; tile order and per-row arithmetic are those of the 4-wave, one-tile-per-stage
    kernels, so with a static softmax δ (nothing exchanged between batch items) the batch-2 call on the wide form equals the two
    batch-1 calls, which the planner puts on the 4-wave form (too few workgroups for the wide one; key split off), bit for bit."""
    from dgq_amd import ops
    B, H, bits = 2, 8, 8
    g = torch.Generator().manual_seed(D + T + S + mode)
    q, k, v = (torch.randn(B, n, H * D, generator=g).to(dev) for n in (T, S, S))
    delta = torch.tensor([1.0 / 255.0 if mode == 3 else 0.7], device=dev)
    tab = lambda n: (torch.rand(n, generator=g).to(dev) * 0.02 + 0.02, torch.randint(100, 156, (n,), generator=g).float().to(dev))
    fq = ((qmode,) + tab(T if qmode == 1 else D) + (0, 8), (1,) + tab(S) + (0, 8), (2,) + tab(D) + (0, 8))
    monkeypatch.setenv("DGQ_ATTN_SPLIT", "0")
    o2 = ops.attention(q, k, v, H, D, D ** -0.5, mode, 0, delta, bits, fq=fq).clone()
    o1 = torch.cat([ops.attention(q[b:b + 1].contiguous(), k[b:b + 1].contiguous(), v[b:b + 1].contiguous(), H, D, D ** -0.5, mode, 0, delta, bits, fq=fq)
                    for b in range(B)])
    torch.cuda.synchronize()
    assert torch.isfinite(o2).all()
    assert torch.equal(o1, o2), (o1 - o2).abs().max().item()


ONE_LAUNCH_CASES = [
    # D, T, S, mode, skip, qmode, kmode, vmode, dtype      (qmode / kmode / vmode: 0 scalar, 1 per token, 2 per head-dim)
    (40, 4096, 77, 1, 1, 1, 1, 2, torch.float32),    # SD 64x64 cross-attention: 512 workgroups, int8 scores, start-peak, real-time δ
    (40, 4096, 77, 1, 1, 2, 1, 1, torch.float32),    # ... one Q plane x three K planes, three V planes
    (80, 1024, 77, 1, 1, 1, 1, 2, torch.float32),
    (80, 1024, 77, 1, 1, 2, 2, 1, torch.float32),
    (160, 256, 77, 1, 1, 1, 0, 0, torch.float32),    # scalar aqtizer_k (δk folded into the per-query constants)
    (160, 256, 256, 1, 0, 1, 1, 2, torch.float32),   # SD 16x16 self-attention: eight key tiles kept in registers
    (160, 256, 256, 1, 0, 2, 1, 1, torch.float32),
    (160, 64, 64, 1, 0, 1, 1, 2, torch.float32),
    (160, 64, 77, 1, 1, 2, 1, 2, torch.float32),
    (160, 200, 250, 1, 0, 1, 1, 2, torch.float32),   # ragged rows and keys
    (64, 1024, 77, 3, 0, 0, 0, 0, torch.float32),    # SDXL C5: uniform softmax quantiser, scalar tables
    (64, 4096, 77, 2, 0, 1, 1, 2, torch.float32),    # static log2 δ (no exchange)
    (64, 1024, 256, 3, 0, 1, 1, 2, torch.float16),
    (40, 4096, 77, 1, 1, 1, 1, 2, torch.bfloat16),
    (80, 1024, 200, 1, 0, 1, 1, 1, torch.float16),
]


@pytest.mark.parametrize("D,T,S,mode,skip,qmode,kmode,vmode,dtype", ONE_LAUNCH_CASES)
def test_attention_one_launch_is_bit_identical(D, T, S, mode, skip, qmode, kmode, vmode, dtype, dev, monkeypatch):
    """Key ranges of at most 8 tiles run statistics, the real-time δ maximum (exchanged between the resident workgroups inside the
    launch) and P·V as ONE kernel behind the pre-pass (csrc/attn_one.hip).  Scores, row statistics, the δ maximum and the P·V
    arithmetic are those of the three launches without a key split, so the output is equal bit for bit (DGQ_ATTN_ONE=0
    DGQ_ATTN_SPLIT=0 select that form); no workgroup gives up the exchange."""
    from dgq_amd import ops
    B, H, bits = 2, 8, 8
    g = torch.Generator().manual_seed(D + T + S + mode + 7 * qmode + 3 * vmode)
    q, k, v = ((torch.randn(B, n, H * D, generator=g) * 1.1).to(dev).to(dtype) for n in (T, S, S))
    delta = None if mode == 1 else torch.tensor([1.0 / 255.0 if mode == 3 else 0.8], device=dev)
    tab = lambda n: (torch.rand(n, generator=g).to(dev) * 0.02 + 0.02, torch.randint(100, 156, (n,), generator=g).float().to(dev))
    ntab = lambda m, ntok: 1 if m == 0 else (ntok if m == 1 else D)
    fq = ((qmode,) + tab(ntab(qmode, T)) + (0, 8), (kmode,) + tab(ntab(kmode, S - skip)) + (skip, 8), (vmode,) + tab(ntab(vmode, S)) + (0, 8))
    before = ops.attention_sync_timeouts()
    monkeypatch.setenv("DGQ_ATTN_ONE", "0")
    monkeypatch.setenv("DGQ_ATTN_SPLIT", "0")
    o3 = ops.attention(q, k, v, H, D, D ** -0.5, mode, skip, delta, bits, fq=fq).clone()
    monkeypatch.delenv("DGQ_ATTN_ONE")
    monkeypatch.delenv("DGQ_ATTN_SPLIT")
    o1 = ops.attention(q, k, v, H, D, D ** -0.5, mode, skip, delta, bits, fq=fq).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(o1.float()).all()
    assert torch.equal(o1, o3), (o1.float() - o3.float()).abs().max().item()
    assert ops.attention_sync_timeouts() == before


def test_attention_one_launch_delta_exchange_under_load(dev, monkeypatch):
    """The δ exchange of the single-launch form repeated back to back on changing inputs, between other launches that keep the caches
    warm with the previous call's words: every call still equals the three-launch form bit for bit (a stale δ slot or counter would
    show as a different quantisation grid), and no workgroup gives up."""
    from dgq_amd import ops
    B, H, D, T, S, bits = 2, 8, 40, 4096, 77, 8
    g = torch.Generator().manual_seed(11)
    tab = lambda n: (torch.rand(n, generator=g).to(dev) * 0.02 + 0.02, torch.randint(100, 156, (n,), generator=g).float().to(dev))
    fq = ((1,) + tab(T) + (0, 8), (1,) + tab(S - 1) + (1, 8), (2,) + tab(D) + (0, 8))
    before = ops.attention_sync_timeouts()
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for it in range(24):
        q, k, v = ((torch.randn(B, n, H * D, generator=g) * (0.5 + 0.1 * it)).to(dev) for n in (T, S, S))
        o1 = ops.attention(q, k, v, H, D, D ** -0.5, 1, 1, None, bits, fq=fq).clone()
        if it % 3 == 0:
            scratch.fill_(it)                               # other traffic between the calls
        monkeypatch.setenv("DGQ_ATTN_ONE", "0")
        monkeypatch.setenv("DGQ_ATTN_SPLIT", "0")
        o3 = ops.attention(q, k, v, H, D, D ** -0.5, 1, 1, None, bits, fq=fq).clone()
        monkeypatch.delenv("DGQ_ATTN_ONE")
        monkeypatch.delenv("DGQ_ATTN_SPLIT")
        assert torch.equal(o1, o3), (it, (o1 - o3).abs().max().item())
    assert ops.attention_sync_timeouts() == before


@pytest.mark.parametrize("B,C,Hs,N,mode", [(2, 64, 8, 64, "perK"), (2, 320, 32, 320, "perK"), (2, 128, 16, 96, "perK"), (2, 320, 32, 64, "perM"),
                                           (1, 64, 8, 32, "perM"), (2, 64, 16, 64, "scalar")])
def test_upsample_folded_into_the_conv_quantiser(B, C, Hs, N, mode, dev):
    """Upsample2D.forward = conv(F.interpolate(x, 2x, nearest)) (diffusers_rewrite/sd.py): ops.quant_conv2d(upsample=True) reads the
    (H/2) x (W/2) source through the (h/2, w/2) mapping inside the quantise-on-load pass (scatter and block-staged conv kernels,
    dgq_quant_act_args_t.ups) and materialises the interpolate for layers on other variants — bit-identical either way."""
    from dgq_amd import ops, synth
    from dgq_amd.plan import plan_act
    g = torch.Generator().manual_seed(B * C + Hs + N)
    K = C * 9
    w = (torch.randn(N, C, 3, 3, generator=g) * 0.05).to(dev)
    x = (torch.randn(B, C, Hs, Hs, generator=g) * 1.3 + 0.2).to(dev)
    wd, wz = synth.channel_minmax(w.cpu(), 4)
    pw = ops.PackedWeight(w, wd.to(dev), wz.to(dev), None, torch.randn(N, generator=g).to(dev), 4, C, 9)
    H = 2 * Hs
    if mode == "perK":
        d, z = synth._group_params(K, 16, 8, "ups|%d" % K, 0)
        lay = plan_act(d.view(1, -1, 1), z.view(1, -1, 1), "conv", C, 9, 8)
    elif mode == "perM":
        d, z = synth._group_params(H * H, 16, 8, "ups|%d" % K, 0)
        lay = plan_act(d.view(1, 1, -1), z.view(1, 1, -1), "conv", C, 9, 8)
    else:
        lay = plan_act(torch.tensor(0.031), torch.tensor(121.0), "conv", C, 9, 8)
    ab = ops.ActBinding(lay, pw, 8)
    res = torch.randn(B, N, H, H, generator=g).to(dev)
    want = ops.quant_conv2d(torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest"), ab, 3, 3, 1, 1, residual=res)
    got = ops.quant_conv2d(x, ab, 3, 3, 1, 1, residual=res, upsample=True)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (B, N, H, H)
    assert torch.equal(got, want), (got - want).abs().max().item()
    gw, gg = getattr(want, "_dgq_gn", None), getattr(got, "_dgq_gn", None)
    assert (gw is None) == (gg is None)
    if gw is not None:
        # (the materialised call may run with its quantiser inside the GEMM launch, csrc/gemm_convq.hip, whose 16-row partial blocks are
        # halves of 4 x 8 position tiles, the folded call's are 16 consecutive rows: the same statistics through another partition)
        fin = lambda gn: ops.groupnorm_from_partials(gn, 32, 1e-5, torch.ones(N, device=dev), torch.zeros(N, device=dev)) if N % 32 == 0 else (gn["parts"][0][0],) * 2
        (s0, h0), (s1, h1) = fin(gw), fin(gg)
        assert rel_l2(s1, s0) < 1e-5 and rel_l2(h1, h0) < 1e-4
