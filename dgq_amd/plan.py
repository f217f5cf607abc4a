"""Host-side planning for the W4A8 GEMM: how DGQ's per-timestep activation-quantizer tables
(``delta`` / ``zero_point`` of ``UniformAffineQuantizer``, shapes per SURVEY.md §5.4) are turned into
the K permutation, 64-wide chunk tables and epilogue vectors the HIP kernels consume.

Pure CPU/torch index arithmetic (no kernels) so it is unit-tested without a GPU.

K orders:
  * reference order  k_ref = c·(kh·kw) + tap     (``w.view(N,-1)`` / ``F.unfold`` row order, quant_layer.py:557,634)
  * natural physical kp = tap·C + c               (channels-last friendly; used for per-M / scalar scales)
  * grouped physical: DGQ groups (= distinct (δ,z) pairs, arbitrary channel sets, quant_layer.py:405-418)
    made contiguous, each padded with zero codes to a multiple of 32 (one MFMA_I32_32x32x32_I8 slice is
    then scaled by a single δ; rounds 1-2 padded to 64 for MFMA_I32_16x16x64_I8: K = 320 in 16 groups
    gave Kp = 1024, now 512), whole K padded to a multiple of 128 (the GEMM's K tile).

The GEMM sums by parts over running int32 totals T_c (Σ_g δ_g·P_g = Σ_c (δ_c − δ_{c+1})·T_c); ``cflush`` marks where
a group ends (1); ``mark_clears`` adds, per weight width, the chunks after which the running total is cleared (2) so
that |T| never exceeds 2^24 and its fp32 conversion in the flush stays exact (a clear may fall inside a group: the
coefficient of the chunk in front of it is then the full δ_c).
"""
from dataclasses import dataclass
from typing import Optional

import torch

KCHUNK = 32
KTILE = 128


def round_up(x, m):
    return (x + m - 1) // m * m


def natural_kperm(C: int, taps: int):
    """kperm[kp] -> k_ref for the natural physical order, padded with -1 to a multiple of KTILE."""
    K = C * taps
    kp = torch.arange(K)
    tap, c = kp // C, kp % C
    perm = torch.full((round_up(K, KTILE),), -1, dtype=torch.int32)
    perm[:K] = (c * taps + tap).to(torch.int32)
    return perm


@dataclass
class ActLayout:
    """What one activation quantizer looks like to the kernels."""
    mode: str                       # 'scalar' | 'perM' | 'perK'
    # perM / scalar
    mdelta: Optional[torch.Tensor] = None     # [L] f32
    mzp: Optional[torch.Tensor] = None
    L: int = 1
    # perK
    kperm: Optional[torch.Tensor] = None      # [Kp] int32: k_ref or -1   (weight packing)
    ksrc: Optional[torch.Tensor] = None       # [Kp] int32: (dh<<24)|(dw<<16)|c or -1 (activation gather)
    cdelta: Optional[torch.Tensor] = None     # [Kp/32] f32
    czp: Optional[torch.Tensor] = None
    cflush: Optional[torch.Tensor] = None     # [Kp/32] u8: 0 inside a group, 1 group end (2 after mark_clears: clear the running total)
    kcoef: Optional[torch.Tensor] = None      # [K] f64 in k_ref order: δ_k·(offset − z_k), for U[n]
    Kp: int = 0
    n_groups: int = 0


def classify_act_params(delta: torch.Tensor, kind: str):
    """Which axis a ckpt (δ,z) pair addresses.
    kind 'linear': input [B,T,K]; (1,1,K) -> perK, (1,T,1) -> perM, () -> scalar.
    kind 'conv'  : quantizer sees unfolded [B, C·kh·kw, L]; (1,K,1) -> perK, (1,1,L) -> perM."""
    if delta.dim() == 0 or delta.numel() == 1:
        return "scalar"
    if delta.dim() != 3 or delta.shape[0] != 1:
        raise ValueError("unsupported activation-quantizer shape %s" % (tuple(delta.shape),))
    if kind == "linear":
        if delta.shape[1] == 1:
            return "perK"
        if delta.shape[2] == 1:
            return "perM"
    else:
        if delta.shape[2] == 1:
            return "perK"
        if delta.shape[1] == 1:
            return "perM"
    raise ValueError("unsupported activation-quantizer shape %s for %s" % (tuple(delta.shape), kind))


def seg_limit(abits: int, wbits: int) -> int:
    """Codes (a multiple of KTILE: clears sit behind whole K tiles) a running int32 total may span before it is cleared:
    |s| <= 2^(abits−1) and |qw'| <= 15 (W4, unsigned nibbles) or 128 (W8, centred), so |T| <= 2^24 and float(T) is exact."""
    wmax = 15 if wbits == 4 else 128
    return max(KTILE, (1 << 24) // ((1 << (abits - 1)) * wmax) // KTILE * KTILE)


def mark_clears(cflush: torch.Tensor, abits: int, wbits: int) -> torch.Tensor:
    """cflush with value 2 on the last chunk of every seg_limit-long segment (uniform segments from chunk 0; always the
    last chunk of a K tile, the only place where dgq_gemm_wxa8 honours the mark)."""
    out = cflush.clone()
    step = seg_limit(abits, wbits) // KCHUNK
    out[step - 1::step] = 2
    return out


def plan_act(delta: torch.Tensor, zp: torch.Tensor, kind: str, C: int, taps: int, abits: int, kw: int = 0) -> ActLayout:
    """delta/zp as stored in the cali_ckpt (CPU tensors). ``kw``: kernel width (default: square kernel)."""
    if kw <= 0:
        kw = int(round(taps ** 0.5))
    delta = delta.detach().float().cpu()
    zp = torch.as_tensor(zp).detach().float().cpu()
    mode = classify_act_params(delta, kind)
    if mode == "scalar":
        return ActLayout("scalar", mdelta=delta.reshape(1).clone(), mzp=zp.reshape(1).clone(), L=1)
    if mode == "perM":
        d = delta.reshape(-1).clone()
        z = zp.reshape(-1).expand_as(d).clone() if zp.numel() == 1 else zp.reshape(-1).clone()
        return ActLayout("perM", mdelta=d, mzp=z, L=d.numel())
    # ---- perK: group = distinct (δ,z) pair
    K = C * taps
    d = delta.reshape(-1)
    z = zp.reshape(-1)
    if d.numel() != K or z.numel() != K:
        raise ValueError("per-K activation table has %d entries, layer has K=%d" % (d.numel(), K))
    pairs = torch.stack([d, z], dim=1)
    uniq, inv = torch.unique(pairs, dim=0, return_inverse=True)
    G = uniq.shape[0]
    k_ref = torch.arange(K)
    c_of, tap_of = k_ref // taps, k_ref % taps
    # sort by (group, tap, c): members of a group stay close in memory (channels-last gather)
    key = (inv.long() * taps + tap_of) * C + c_of
    order = torch.argsort(key)
    counts = torch.bincount(inv, minlength=G)
    padded = ((counts + KCHUNK - 1) // KCHUNK) * KCHUNK
    Kp = round_up(int(padded.sum()), KTILE)
    kperm = torch.full((Kp,), -1, dtype=torch.int32)
    ksrc = torch.full((Kp,), -1, dtype=torch.int32)
    nch = Kp // KCHUNK
    cdelta = torch.ones(nch)
    czp = torch.zeros(nch)
    cflush = torch.zeros(nch, dtype=torch.uint8)
    pos, src = 0, 0
    for g in range(G):
        n = int(counts[g])
        ks = order[src:src + n]
        kperm[pos:pos + n] = ks.to(torch.int32)
        ksrc[pos:pos + n] = (((tap_of[ks] // kw) << 24) | ((tap_of[ks] % kw) << 16) | c_of[ks]).to(torch.int32)
        c0, c1 = pos // KCHUNK, (pos + int(padded[g])) // KCHUNK
        cdelta[c0:c1] = uniq[g, 0]
        czp[c0:c1] = uniq[g, 1]
        cflush[c1 - 1] = 1
        pos += int(padded[g])
        src += n
    if pos < Kp:                       # tail chunk added to reach the K tile: all-zero codes
        cdelta[pos // KCHUNK:] = uniq[G - 1, 0]
        czp[pos // KCHUNK:] = uniq[G - 1, 1]
        cflush[-1] = 1
    offset = act_offset(abits)
    kcoef = d.double() * (offset - z.double())
    return ActLayout("perK", kperm=kperm, ksrc=ksrc, cdelta=cdelta, czp=czp, cflush=cflush, kcoef=kcoef, Kp=Kp,
                     n_groups=G)


def act_offset(abits: int) -> float:
    """Code offset o = 2^(b−1): q ∈ [0, 2^b−1] -> s = q − o ∈ [−2^(b−1), 2^(b−1)−1] fits int8 for every b ≤ 8, and
    centred codes keep the fp32 epilogue free of the cancellation an all-positive operand would cause."""
    return float(2 ** (abits - 1))
