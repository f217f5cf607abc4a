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
    |s| <= 2^(abits−1) and |qw'| <= 15 (W4, unsigned nibbles) or 128 (W8, centred).  W8: |T| <= 2^24, float(T) by conversion is
    exact.  W4: |T| < 2^22 — the per-K W4 kernels keep their totals biased by bits(1.5·2^23) and read float(T) off the bits
    (csrc/gemm_device.h:dgq_total_to_float), exact on that range."""
    wmax, bound = (15, 1 << 22) if wbits == 4 else (128, 1 << 24)
    return max(KTILE, bound // ((1 << (abits - 1)) * wmax) // KTILE * KTILE)


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
    # ---- perK: group = distinct (δ,z) pair.  Pure index arithmetic on K <= 23040 entries, vectorised in numpy (one call per
    # layer and timestep slot: 7000 calls for a 50-slot SD model; the torch.unique(dim=0) + per-group Python loop it replaces
    # took 30 of the 34 s of QuantModel.prepare_slots on a 64-thread host).
    import numpy as np
    K = C * taps
    d = delta.reshape(-1)
    z = zp.reshape(-1)
    if d.numel() != K or z.numel() != K:
        raise ValueError("per-K activation table has %d entries, layer has K=%d" % (d.numel(), K))

    def ordered_bits(x):                       # float32 -> uint32 whose unsigned order is the float order
        b = x.numpy().view(np.uint32)
        return np.where(b >> 31, ~b, b | np.uint32(0x80000000)).astype(np.uint64)
    key = (ordered_bits(d.contiguous()) << np.uint64(32)) | ordered_bits(z.contiguous())
    ukey, first, inv = np.unique(key, return_index=True, return_inverse=True)     # groups in lexicographic (δ, z) order
    inv = inv.reshape(-1).astype(np.int64)
    G = int(ukey.shape[0])
    k_ref = np.arange(K, dtype=np.int64)
    c_of, tap_of = k_ref // taps, k_ref % taps
    # sort by (group, tap, c): members of a group stay close in memory (channels-last gather)
    order = np.argsort((inv * taps + tap_of) * C + c_of, kind="stable")
    counts = np.bincount(inv, minlength=G)
    padded = (counts + KCHUNK - 1) // KCHUNK * KCHUNK
    starts = np.cumsum(padded) - padded                      # first packed position of each group
    first_sorted = np.cumsum(counts) - counts                # first sorted index of each group
    used = int(padded.sum())
    Kp = round_up(used, KTILE)
    g_sorted = inv[order]
    dest = starts[g_sorted] + (np.arange(K, dtype=np.int64) - first_sorted[g_sorted])
    kperm = np.full((Kp,), -1, dtype=np.int32)
    ksrc = np.full((Kp,), -1, dtype=np.int32)
    kperm[dest] = order.astype(np.int32)
    to = tap_of[order]
    ksrc[dest] = (((to // kw) << 24) | ((to % kw) << 16) | c_of[order]).astype(np.int32)
    nch = Kp // KCHUNK
    gd, gz = d.numpy()[first], z.numpy()[first]               # (δ, z) of each group
    reps = padded // KCHUNK
    tail = nch - int(reps.sum())                             # chunks added to reach the K tile: all-zero codes, last group's (δ, z)
    cdelta = np.concatenate([np.repeat(gd, reps), np.full((tail,), gd[G - 1], dtype=np.float32)]).astype(np.float32)
    czp = np.concatenate([np.repeat(gz, reps), np.full((tail,), gz[G - 1], dtype=np.float32)]).astype(np.float32)
    cflush = np.zeros((nch,), dtype=np.uint8)
    cflush[(starts + padded) // KCHUNK - 1] = 1
    if tail:
        cflush[-1] = 1
    offset = act_offset(abits)
    kcoef = d.double() * (offset - z.double())
    return ActLayout("perK", kperm=torch.from_numpy(kperm), ksrc=torch.from_numpy(ksrc), cdelta=torch.from_numpy(cdelta),
                     czp=torch.from_numpy(czp), cflush=torch.from_numpy(cflush), kcoef=kcoef, Kp=Kp, n_groups=G)


def flush_coefficients(cdelta: torch.Tensor, cflush: torch.Tensor) -> torch.Tensor:
    """The per-chunk coefficients of the GEMM's summation by parts for an UNSPLIT launch, as the kernels form them in their
    prologue (gemm_wxa8.hip / gemm_wxa8_big.hip): coef_c = δ_c − δ_{c+1} (non-zero at group ends only); the last chunk and the last
    chunk of a K tile that carries a clear mark (cflush == 2 on that tile's last chunk) take the full δ_c.  Followed by one clear
    flag per K tile (1.0 where the totals are cleared behind the tile; never behind the last).  fp32 [Kp/32 + Kp/128] — the
    256-row kernel reads it with scalar loads instead of testing a staged LDS table between its MFMAs."""
    d = cdelta.detach().float().cpu()
    f = cflush.detach().cpu()
    nch = d.numel()
    per = KTILE // KCHUNK
    nk = nch // per
    dn = torch.cat([d[1:], d[-1:]])
    last_of_tile = torch.arange(nch) % per == per - 1
    clr_tile = (f.view(nk, per)[:, per - 1] == 2)
    full = (torch.arange(nch) == nch - 1) | (last_of_tile & clr_tile.repeat_interleave(per))
    coef = torch.where(full, d, d - dn)
    flags = clr_tile.float()
    flags[-1] = 0.0
    return torch.cat([coef, flags]).contiguous()


def act_offset(abits: int) -> float:
    """Code offset o = 2^(b−1): q ∈ [0, 2^b−1] -> s = q − o ∈ [−2^(b−1), 2^(b−1)−1] fits int8 for every b ≤ 8, and
    centred codes keep the fp32 epilogue free of the cancellation an all-positive operand would cause."""
    return float(2 ** (abits - 1))
