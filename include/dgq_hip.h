/* dgq_hip.h — C ABI of libdgq_hip.so: the MI355X (gfx950) kernels behind DGQ's quantized-UNet path.
 *
 * The reference (ugonfor/DGQ) has NO FFI: its per-layer operator boundary is the Python method
 * QuantLayer.forward (quant/quant_layer.py:626-661) and the attention-side quantizer calls in
 * Attention.Attention_forward (diffusers_rewrite/sd.py:151-207).  Each entry point below names the
 * reference code it replaces.  Conventions:
 *   - plain C types only: raw DEVICE pointers, ints, floats and a hipStream_t passed as void*;
 *   - the caller owns every buffer; the library never allocates, frees or keeps device pointers;
 *   - every launch is asynchronous on `stream`; no host synchronisation, safe under hipGraph capture;
 *   - return 0 on success, a negative DGQ_E* otherwise (message via dgq_last_error()); no exceptions,
 *     no abort().
 * dtype codes for floating tensors: 0 = f32, 1 = f16, 2 = bf16.
 */
#ifndef DGQ_HIP_H
#define DGQ_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGQ_OK 0
#define DGQ_EINVAL (-1)   /* bad shape / alignment / argument */
#define DGQ_EUNSUPPORTED (-2)
#define DGQ_ELAUNCH (-3)  /* HIP reported a launch error */

#define DGQ_F32 0
#define DGQ_F16 1
#define DGQ_BF16 2

#define DGQ_KCHUNK 32     /* K granularity of one MFMA_I32_32x32x32_I8 slice = DGQ group padding */
#define DGQ_KTILE 128     /* Kp (padded, permuted K) must be a multiple of this */

/* ABI revision, major·100 + minor.  A host binding must refuse a library whose revision differs from the header it was written
 * against (dgq_amd/_lib.py does): the argument structs are passed by pointer and have grown between revisions.
 *   100  rounds 1-4
 *   110  round 5: dgq_gemm_extra_t += wfrag, act; dgq_quant_act_args_t += ups; W4 per-K clear marks (cflush == 2) at least every
 *        2176 codes (the kernels read float(T) off a biased int32 total: |T| < 2^22), plan.seg_limit
 *   120  round 6: dgq_attention_sync_timeouts; the attention workspace's δ area is 512 bytes; dgq_gemm_act_t += kpat, B, H, W, kh, kw,
 *        stride, pad; dgq_gemm_conv_act_fuses
 *   121  round 6: dgq_cfg_ddim_step takes the tensors' dtype (void pointers + `dtype`); dgq_attention_workspace_bytes grew by one fp32
 *        part area (key-split launches on 16-bit tensors)
 *   122  round 6: dgq_gemm_extra_t += y2, ldy2; dgq_conv2d_f32w takes y2 / ldy2 / gn_partial */
#define DGQ_ABI_VERSION 122
int dgq_version(void);
const char* dgq_last_error(void);

/* ---- weights (load time) -------------------------------------------------------------------------
 * dgq_quantize_weight: integer codes of wqtizer(self.w) — UniformAffineQuantizer.forward
 * (quant_layer.py:295-299: clamp(rne(w/δ)+z, 0, 2^b−1)) or, when alpha != NULL, AdaRoundQuantizer hard
 * mode (adaptive_rounding.py:51,58-70: clamp(floor(w/δ)+(α≥0)+z, 0, 2^b−1)).  The reference recomputes
 * these on EVERY forward (quant_layer.py:642-643); here they are computed once.
 * w [N][K] f32, delta/zp [N] f32, alpha [N][K] f32 or NULL, codes [N][K] u8. */
int dgq_quantize_weight(const float* w, const float* delta, const float* zp, const float* alpha,
                        int N, int K, int bits, uint8_t* codes, void* stream);

/* dgq_pack_w4: codes u8 [N][K] (values 0..15) -> packed [N][Kp/2] bytes.  kperm [Kp] gives for each
 * packed position the source k (or -1 = zero padding); NULL = identity (Kp == K).  Layout per 8
 * consecutive kp: one 32-bit word, byte j holds kp+j in its low nibble and kp+4+j in its high nibble,
 * so that (word & 0x0F0F0F0F) and ((word >> 4) & 0x0F0F0F0F) are 4 consecutive int8 each.
 * layout 2 (dgq_gemm_extra_t.wfrag): fragment-major for V_MFMA_I32_32X32X32_I8 B operands — N padded to 32-column tiles (zero
 * rows), [ceil(N/32)][Kp/64] blocks of 1 KiB; in block (j, p) lane l = (k half h << 5) | (n & 31) owns the 16 bytes at l·16: the two
 * words of K half h of chunk 2p, then those of chunk 2p + 1, column n = 32j + (l & 31).  `packed` holds ceil(N/32)·32·Kp/2 bytes.
 * layout 0: rows as described.  layout 1 (what dgq_gemm_wxa8 and dgq_linear_smallm_batch read): in rows n with (n & 16) != 0 the two
 * 8-byte halves of every 16 packed bytes (= one 32-wide chunk) are exchanged — the GEMM stages 16-byte pieces into LDS by
 * DMA and a lane reads the 8 bytes of its K half; with the exchange the 32 lanes of a read cover all 64 LDS banks. */
int dgq_pack_w4(const uint8_t* codes, int N, int K, const int32_t* kperm, int Kp, int layout, uint8_t* packed, void* stream);
/* exact inverse (test hook for the "bit-exact int4 unpack" requirement): out [N][Kp] u8 */
int dgq_unpack_w4(const uint8_t* packed, int N, int Kp, int layout, uint8_t* out, void* stream);
/* W8: out[n][kp] = (int8)(codes[n][kperm[kp]] - 128), 0 for padding */
int dgq_pack_w8(const uint8_t* codes, int N, int K, const int32_t* kperm, int Kp, int8_t* packed, void* stream);

/* ---- activations: quantise-on-load pre-pass ------------------------------------------------------
 * Replaces `x = self.aqtizer(x)` (quant_layer.py:640-641 -> :295-299) and, for grouped convs, the
 * F.unfold that precedes it (quant_layer.py:630-638): writes int8 codes s = q − offset of the (implicit)
 * unfolded operand, row-major [M][Kp], in the permuted/padded K order of the packed weight, plus one
 * float per row.
 *
 * x is a channels-last tensor [B][H][W][C] (a Linear input [B,T,K] is B=B·T... H=W=1, C=K); geometry
 * (kh,kw,stride,pad) as the conv; M = B·Ho·Wo.  Source of packed position kp:
 *   ksrc == NULL : natural order kp = tap·C + c   (tap = dh·kw + dw)
 *   ksrc != NULL : ksrc[kp] = (dh << 24) | (dw << 16) | c, or -1 for padding (zero code); koff (optional, NULL = derive
 *                  from ksrc) = the same table resolved for this geometry, (dh·W + dw)·ldc + c or -1, read by rows whose
 *                  taps all lie inside the image (no bounds checks); klds (optional) = (dh·kw + dw)·C + c or -1: the
 *                  index into the [tap][C] strip a wave stages in LDS (kh·kw > 1 and kh·kw·C·4 <= 64 KB).
 * Quantiser parameters:
 *   per_m == 0 : cdelta/czp [Kp/32] — one (δ,z) per 32-wide chunk (DGQ groups are chunk aligned);
 *                rowsum[m] = Σ_kp δ(kp)·s[m,kp]
 *   per_m == 1 : mdelta/mzp [L] indexed by (m % L) (L=1: scalar quantiser);
 *                rowsum[m] = Σ_kp s[m,kp]  (exact integer in f32)
 * bits: activation bits b; code offset = 2^(b-1) (s = q − 2^(b−1) is a centred int8).  Out-of-image taps read 0.0 and are
 * quantised like any value (F.unfold pads before the quantizer).
 * pre_scale/pre_shift [B][C] (or NULL): the element is first mapped to x·scale + shift (a GroupNorm folded into
 * the load, see dgq_groupnorm_scale_shift); pre_act == 1 then applies SiLU (resnet norm→SiLU→conv, and SiLU(temb) →
 * time_emb_proj); pre_act == 2 reads rows of 2C elements and forms x[c]·gelu(x[C+c]) (GEGLU in front of ff.net.2,
 * sd.py:210-236); out-of-image taps stay 0.
 * ln_gamma/ln_beta [C] (or NULL) + ln_eps: nn.LayerNorm over the C elements of each row folded into the load
 * (norm1/2/3 of BasicTransformerBlock in front of to_q/to_k/to_v/ff.net.0.proj, sd.py:245-268): the wave that
 * quantises a row computes its mean / biased variance and maps x to (x − μ)·rstd·γ + β.  1x1 only, no other prologue.
 * ksplits >= 1 splits every row's K range over that many waves (low-M layers); rowsum then has
 * dgq_quant_act_parts(Kp, ksplits) x M entries ([part][m]) which dgq_gemm_wxa8 adds in a fixed order. */
int dgq_quant_act(const void* x, int x_dtype, int B, int H, int W, int C,
                  int kh, int kw, int stride, int pad,
                  const int32_t* ksrc, const int32_t* koff, const int32_t* klds, int Kp,
                  int per_m, const float* delta, const float* zp, int L,
                  int bits, int8_t* codes, float* rowsum, int ksplits,
                  const float* pre_scale, const float* pre_shift, int pre_act,
                  const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream);
int dgq_quant_act_parts(int Kp, int ksplits);
/* The same with the arguments in a struct, for 1..8 problems in ONE launch (dgq_quant_act_batch): the q / k / v
 * projections of an attention quantise one input with three tables, the to_k / to_v of every cross-attention quantise
 * the one text context.  All problems of a batch must have the same row count M, dtype, scale mode (per_m) and kernel
 * variant (dgq_quant_act_variant: 0 LDS-staged, 1 table gather, 2 natural order, 3 / 4 LDS scatter, 5 block-staged conv) —
 * DGQ_EINVAL otherwise. */
typedef struct dgq_quant_act_args {
    const void* x; int x_dtype; int B, H, W, C, kh, kw, stride, pad;
    const int32_t* ksrc; const int32_t* koff; const int32_t* klds;
    const int32_t* kdst;      /* optional [kh·kw·C]: packed position kp of element (tap, c) — the inverse of ksrc; enables the
                                 coalesced-read / LDS-scatter path for per-K conv layers (ksplits must be 1) */
    int Kp;
    int per_m; const float* delta; const float* zp; int L; int bits;
    int8_t* codes; float* rowsum; int ksplits;
    const float* pre_scale; const float* pre_shift; int pre_act;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const int32_t* kpat;      /* optional [Kp] (convolutions): (dh·PW + dw)·C + c of packed position kp inside the input patch of
                                 a tile, PW from dgq_quant_act_conv_tile; -1 = padding.  Natural order (ksrc == NULL) too.
                                 Enables the block-staged conv path (variant 5): a workgroup stages the input patch of a tile
                                 of output positions once in LDS with the GroupNorm / SiLU prologue applied once per element
                                 (the per-row paths apply it once per tap), then gathers every row's codes from it. */
    int ups;                  /* 0, or 1: x holds (H/2) x (W/2) pixels per image and input pixel (hi, wi) reads (hi/2, wi/2) — the
                                 F.interpolate(x, scale_factor=2, mode="nearest") in front of Upsample2D's conv (diffusers_rewrite/sd.py
                                 Upsample2D.forward) folded into the load.  H, W are the UPSAMPLED dims; convolutions on the scatter /
                                 block-staged paths only (dgq_quant_act_variant 3 / 4 / 5, asked with ups set), DGQ_EINVAL otherwise. */
} dgq_quant_act_args_t;
int dgq_quant_act_batch(int n, const dgq_quant_act_args_t* args, void* stream);
int dgq_quant_act_variant(const dgq_quant_act_args_t* args);
/* tile id (1: 4x8, 2: 4x4, 3: 2x4 output positions; 0: the patch does not fit, no block-staged path) and patch width PW
 * = (TW − 1)·stride + kw of the block-staged conv path for a geometry: what the caller needs to build `kpat` */
int dgq_quant_act_conv_tile(int C, int kh, int kw, int stride, int Kp, int* patch_w);

/* GroupNorm of a channels-last tensor x [B][HW][C] as per-(b,c) scale/shift (biased variance, eps as F.group_norm):
 * GN(x) = x·scale + shift.  Replaces norm1/norm2 of QuantResnetBlock2D.forward (quant_block.py:98-119) together with
 * the SiLU that follows, which dgq_quant_act applies while loading.  partial_ws: B·G·slices·3 floats;
 * slices > 1: a second small kernel merges the slices in slice order. */
int dgq_groupnorm_scale_shift(const void* x, int x_dtype, int B, int HW, int C, int G, float eps,
                              const float* gamma, const float* beta, float* scale, float* shift,
                              float* partial_ws, int slices, void* stream);
/* The same scale / shift from the partial statistics a producing dgq_gemm_wxa8 left behind (dgq_gemm_extra_t.gn_partial):
 * partial [B·HW/16][C][2] = (mean, Σ(x − mean)²) per 16-row block and channel; HW % 16 == 0.  One small launch, no pass
 * over the tensor; the blocks and channels of a (batch, group) are merged in a fixed order (Chan), biased variance.
 * partial2 / C2 (optional, NULL / 0): a second source whose channels FOLLOW the first's — the statistics of
 * torch.cat([a, b], dim=1) from those of a and b (the skip concatenations in front of the up blocks' norm1). */
int dgq_groupnorm_from_partials(const float* partial, int C1, const float* partial2, int C2, int B, int HW, int G, float eps,
                                const float* gamma, const float* beta, float* scale, float* shift, void* stream);

/* ---- elementwise glue of a denoise step as single launches ------------------------------------------------------------------
 * Timesteps.forward (diffusers_rewrite/sd.py:19-39: flip_sin_to_cos, no frequency shift): out [rows][dim] =
 * cat(cos(t·f), sin(t·f)), f_j = exp(−ln(10000)·j / (dim/2)).  t: rows values of int64 (t_is_float = 0) or fp32 (1), element
 * stride t_stride (0 = the one timestep expanded over the batch, sd.py:550).  Same operation order as the torch chain. */
int dgq_timestep_embedding(const void* t, int t_is_float, int64_t t_stride, int rows, int dim, void* out, int out_dtype, void* stream);
/* Classifier-free guidance + DDIM update (eta = 0) of the pipeline loop (pipeline_stable_diffusion.py:1037-1044,
 * schedulers/scheduling_ddim.py step): eps = e_u + guidance·(e_c − e_u) (eps_cond == NULL: eps = e_u),
 * out = s3·((sample − s1·eps)·inv_s2) + s4·eps with s1 = sqrt(1−a_t), inv_s2 = 1/sqrt(a_t), s3 = sqrt(a_prev), s4 = sqrt(1−a_prev); the four
 * tensors of one dtype (DGQ_F32 / F16 / BF16: a 16-bit chain rounds after every statement, as the eager one does),
 * n elements per tensor = whole [C][HW] images; sample / out contiguous [n/(C·HW)][C][HW]; the eps halves in the same layout
 * (eps_channels_last = 0) or as [.][HW][C] (1: the UNet's channels-last output); evaluated in the order of the eager torch chain. */
int dgq_cfg_ddim_step(const void* eps_uncond, const void* eps_cond, const void* sample, void* out, int dtype, int64_t n, int C, int HW,
                      int eps_channels_last, float guidance, float s1, float inv_s2, float s3, float s4, void* stream);

/* ---- weight-only state (use_wq without use_aq: quant_layer.py:642-659 with unquantised activations) -----------------
 * y[m][n] = Σ_k x_unfolded[m][k]·w[n][k] + bias[n] in exact fp32 (V_MFMA_F32_32X32X2_F32), the im2col of a convolution folded
 * into the operand load: x channels-last [B][H][W][C] (x_dtype), w [N][kh·kw·C] fp32 = the dequantised weight δw·(qw − zw) with
 * K in (tap, c) order, y [B·Ho·Wo][ldy] (y_dtype).  A Linear layer is B = rows, H = W = kh = kw = stride = 1, pad = 0.
 * Optional prologue as in dgq_quant_act: pre_scale / pre_shift [B][C] (a GroupNorm folded into the load: x·scale + shift) and
 * pre_act = 1 (SiLU) — conv_out(SiLU(conv_norm_out(x))) of the UNets in one launch.
 * Replaces F.linear / F.conv2d on the dequantised (or, for the FP conv_in / conv_out, the original) weight. 
 * y2 (or NULL), ldy2: a second copy of the output rows, as dgq_gemm_extra_t.y2; gn_partial (or NULL; M % 16 == 0, N > 8): the
 * output's GroupNorm partials [M/16][N][2], as dgq_gemm_extra_t.gn_partial. */
int dgq_conv2d_f32w(const void* x, int x_dtype, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                    const float* w, const float* bias, int N, void* y, int y_dtype, int ldy,
                    const float* pre_scale, const float* pre_shift, int pre_act, void* y2, int ldy2, float* gn_partial, void* stream);

/* ---- the hot kernel: W4A8 / W8A8 MFMA GEMM with fused dequantisation ------------------------------
 * Replaces F.linear / `w.view(N,-1) @ unfolded` / F.conv2d on fake-quantised operands
 * (quant_layer.py:652-659, :562).  Integer identity, s = qx − offset, zw' = zw − woff:
 *   per_m == 0: y[m,n] = alpha[n]·( Σ_chunks cdelta[c]·Σ_{k∈c} s[m,k]·qw'[n,k]  −  zw[n]·rowsum[m] ) + gamma[n]
 *   per_m == 1: y[m,n] = alpha[n]·mdelta[m%L]·( Σ_k s·qw' − zw[n]·rowsum[m] + (offset − mzp[m%L])·vn[n] ) + gamma[n]
 * codes [M][Kp] int8; wpacked: w_bits==4 -> [N][Kp/2] (dgq_pack_w4), w_bits==8 -> [N][Kp] int8;
 * alpha = δw, zw = zero point in the stored code domain (zw − 8·0 for W4: unsigned nibbles; zw − 128 for W8);
 * gamma = bias (+ alpha·U for per_m==0, U[n] = Σ_k δx_k(offset − zx_k)(qw'[n,k] − zw[n]), precomputed per slot);
 * vn[n] = Σ_k qw'[n,k] − K·zw[n].  y [M][ldy] of y_dtype.  cdelta / cflush have Kp/32 entries (one per 32-wide chunk).
 * Every chunk of a DGQ group carries the group's δ in cdelta; the kernel sums by parts, Σ_c (cdelta[c] − cdelta[c+1])·T_c
 * with T_c the running int32 total after chunk c, so a group boundary is wherever cdelta changes.  cflush[c] == 2 on the
 * LAST chunk of a K tile (c % 4 == 3) asks for the running totals to be cleared behind that tile (the coefficient in front
 * of the clear is then the full cdelta): the caller places such marks so that float(T) is exact — W8: |T| <= 2^24 (converted);
 * W4: |T| < 2^22, REQUIRED: the W4 per-K kernels keep T biased by bits(1.5·2^23) and read the float off the bits
 * (dgq_amd/plan.py:seg_limit: every 2176 codes for W4A8, 1024 for W8A8); marks on other chunks are ignored,
 * other values (0 inside a group, 1 at a group end) are informative. */
/* Optional epilogue extras (host struct, passed by pointer; NULL = none), applied in this order to the fp32 result:
 *   fq_mode != 0 : the attention-side quantizer of the projection output, aqtizer_{q,k,v} (sd.py:174-182,199):
 *                  y = δ·(clamp(rne(y/δ)+z, 0, fq_qmax) − z) with (δ,z) = table[0] (mode 1), table[(m % fq_T) − fq_skip]
 *                  (mode 2, per token; tokens < fq_skip bypass: start_peak) or table[n % fq_D] (mode 3, per head-dim);
 *   residual     : y += residual[(m / res_div)·ldr + n]  (x + attn(x), x + ff(x), shortcut + conv2(...) of the Quant
 *                  blocks, quant_block.py:98-119,165-186; res_div = Ho·Wo broadcasts one row per image: conv1(...) +
 *                  time_emb_proj(...)[:, :, None, None]) — dtype res_dtype (DGQ_F32/F16/BF16; ldr in elements), may alias
 *                  nothing written by this call; res_div >= 1;
 *   geglu != 0   : FeedForward's GEGLU (diffusers_rewrite/sd.py:210-222) for a ff.net.0 whose weight rows were
 *                  interleaved at pack time (row 2i = value column i, row 2i+1 = gate column i; alpha / zw / gamma / vn
 *                  likewise): y has N/2 columns, y[m, i] = h[m, 2i]·gelu(h[m, 2i+1]) (erf form).  No other extra, no K
 *                  split, N % 4 == 0;
 *   gn_partial   : (or NULL) [M/16][N][2] floats: per 16-row block and column the mean and the sum of squared deviations of
 *                  the values this call stores (after residual; as rounded to y_dtype) — GroupNorm statistics of the output
 *                  for dgq_groupnorm_from_partials, instead of a pass over the tensor.  M % 16 == 0, N % 4 == 0, 16-byte
 *                  aligned.  Written by the GEMM's own epilogue, or — a K-split launch — by its combine kernel
 *                  (dgq_gemm_plan_splits tells which the shape gets; only the GEGLU epilogue forces an unsplit launch). */
/* Implicit im2col A operand (dgq_gemm_extra_t.conv): a convolution whose activation quantizer is ONE (δ, z) pair — the
 * reference's native path F.conv2d(aqtizer(x), ŵ), quant_layer.py:659 — quantises every input pixel once:
 *   codes_in [B·H·W][ldc] int8 = dgq_quant_act of the NHWC input as a 1x1 layer in natural order (ldc = its Kp >= C),
 *   pixsum   [pixsum_parts][B·H·W] = that call's rowsum (Σ_c s of a pixel, in the K-split parts the pass wrote).
 * dgq_gemm_wxa8 then takes the rows of the unfolded operand straight from codes_in: K order kp = tap·C + c (the natural
 * conv order of dgq_pack_w4's kperm == NULL image of the [N][kh][kw][C] weight), row m = (b·Ho + ho)·Wo + wo, tap (dh, dw)
 * reads pixel (ho·stride − pad + dh, wo·stride − pad + dw); a tap outside the image reads `fill` (16 bytes of the code of
 * the value 0.0, i.e. z − offset: the native F.conv2d of quant_layer.py:659 pads with 0.0 BEHIND the quantizer, which is the code
 * z — the caller offers this operand only while 0 <= z <= 2^b − 1 — followed by 16 zero bytes for the K padding).
 * `codes` / `rowsum` of the call: codes is ignored (pass codes_in), rowsum [M] is an OUTPUT scratch the call fills first
 * (Σ over the taps of pixsum, C·zero_code for a tap outside).  per_m == 1 with L == 1, C % 16 == 0, w_bits == 4. */
typedef struct dgq_gemm_conv {
    const int8_t* codes_in;
    const float* pixsum;
    const int8_t* fill;
    int B, H, W, C, ldc, kh, kw, stride, pad, Ho, Wo;
    float zero_code;
    int pixsum_parts;
} dgq_gemm_conv_t;

/* Quantise-on-load INSIDE the GEMM (dgq_gemm_extra_t.act; Linear / 1x1 layers): instead of reading the int8 codes a dgq_quant_act
 * launch wrote, the short-K kernel quantises the rows of its own workgroup tile straight from the floating-point input into its LDS
 * operand panel — `x = self.aqtizer(x)` (quant_layer.py:640-641) and F.linear (:659) in ONE launch, no code matrix in HBM.  Same
 * quantiser arithmetic as dgq_quant_act (exact-division codes, the same folded prologues); `codes` / `rowsum` of the call are
 * ignored (pass any non-NULL pointers).  The library takes it only where dgq_gemm_act_fuses() says so; otherwise DGQ_EINVAL.
 *   x [M][ldx] of x_dtype (the layer's input rows; K = C source channels, ldx >= K, 16-byte aligned rows);
 *   per_m == 0: kdst [K] = packed position kp of source channel c (the inverse of dgq_quant_act's ksrc), czp [Kp/32] the chunks'
 *               zero points (the scales are the call's cdelta); per_m == 1: natural order, tables mdelta / mzp / L of the call;
 *   bits: activation bits; pre_scale / pre_shift [M / rows_per_image][K] (or NULL) + pre_act (0 / 1 = SiLU): a GroupNorm folded
 *   into the load as in dgq_quant_act; ln_gamma / ln_beta [K] (or NULL) + ln_eps: a LayerNorm over each row folded into the load. */
typedef struct dgq_gemm_act {
    const void* x; int x_dtype; int ldx; int K;
    const int32_t* kdst; const float* czp; int bits;
    const float* pre_scale; const float* pre_shift; int rows_per_image; int pre_act;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    /* round 6 (revision 120): a k x k convolution in this form (csrc/gemm_convq.hip) — kh > 1: x is the NHWC input [B][H][W][ldx] of K = C
     * channels, the call's M = B·Ho·Wo and Kp covers C·kh·kw; kpat [Kp] = for every packed position the index (dh·PW + dw)·C + c into a
     * workgroup's staged (4 + kh − 1) x PW x C input patch, PW = 8 + kw − 1, −1 for padding (ops.ActBinding.kpat_for); per_m == 0: czp as
     * above (kdst unused); pre_scale / pre_shift per image ([B][C]); no LayerNorm.  Zero-initialised (kh == 0): a Linear / 1x1 layer. */
    const int32_t* kpat; int B, H, W, kh, kw, stride, pad;
} dgq_gemm_act_t;

typedef struct dgq_gemm_extra {
    const void* residual;
    int ldr;
    int res_div;
    int res_dtype;
    int fq_mode;
    const float* fq_delta;
    const float* fq_zp;
    int fq_T, fq_D, fq_skip;
    float fq_qmax;
    int geglu;
    float* gn_partial;
    const dgq_gemm_conv_t* conv;     /* (or NULL) implicit im2col A operand, see dgq_gemm_conv_t */
    const float* flush_coef;         /* (or NULL; per_m == 0) [Kp/32 + Kp/128] floats, 16-byte aligned: the summation-by-parts
                                      * coefficients of an unsplit launch, coef_c = cdelta[c] − cdelta[c+1] (the last chunk, and a
                                      * K tile's last chunk under a clear mark cflush == 2: cdelta[c]), then one clear flag per K
                                      * tile.  Derived from cdelta / cflush (the library forms the same table itself when this
                                      * is NULL); given here, the 256-row kernel reads it with scalar loads. */
    const void* wfrag;               /* (or NULL; w_bits == 4) the SAME weights as `wpacked` in dgq_pack_w4 layout 2 (fragment-major,
                                      * [ceil(N/32)·32][Kp/2] bytes, 16-byte aligned): lets the library take its short-K kernel
                                      * (gemm_panel.hip: the activations' whole K slice in LDS, weights streamed straight into MFMA
                                      * B fragments) for the launches it is faster on; results are those of the other kernels. */
    const dgq_gemm_act_t* act;       /* (or NULL) quantise-on-load inside the GEMM, see dgq_gemm_act_t; needs wfrag */
    void* y2;                        /* (or NULL) a SECOND copy of the output rows, y_dtype, row pitch ldy2 >= N elements: the layer's output
                                      * stored as well into its slot of a channel-concatenation buffer (the skip connections of the
                                      * up path, diffusers_rewrite/sd.py:558-613: torch.cat([h, skip], 1) of channels-last tensors is
                                      * [M][C1 + C2] rows) — with `y` / ldy pointing into such a buffer for the other half, no
                                      * concatenation launch is needed.  Not with the GEGLU epilogue. */
    int ldy2;
} dgq_gemm_extra_t;

int dgq_gemm_wxa8(const int8_t* codes, const float* rowsum, int rowsum_parts, int M, int Kp,
                  const void* wpacked, int w_bits, int N,
                  int per_m, const float* cdelta, const uint8_t* cflush,
                  const float* mdelta, const float* mzp, int L, float offset,
                  const float* alpha, const float* zw, const float* gamma, const float* vn,
                  void* y, int y_dtype, int ldy, void* workspace, size_t workspace_bytes,
                  const dgq_gemm_extra_t* extra, void* stream);
/* 1 where dgq_gemm_wxa8 / _batch accept dgq_gemm_extra_t.act for a layer of this shape (n_problems of one launch, all alike): the
 * whole padded K must fit the short-K kernel's LDS panel and the launch must be one it would give that kernel anyway. */
int dgq_gemm_act_fuses(int M, int N, int K, int Kp, int w_bits, int per_m, int n_problems, int x_dtype, int y_dtype);
/* ... and whether a dgq_gemm_wxa8 call with extra.act describing a kh x kw convolution (act.kh > 1) is accepted: 3x3, stride 1, pad 1,
 * W4, N = 160 or 320, H % 4 == 0, W % 8 == 0, the 6 x 10 x C fp32 input patch + a 10-tile operand slab within the LDS (C <= 340). */
int dgq_gemm_conv_act_fuses(int B, int H, int W, int C, int kh, int kw, int stride, int pad, int N, int Kp, int w_bits, int per_m,
                            int x_dtype, int y_dtype);
/* The same with the arguments in a struct, for 1..8 problems in ONE launch (dgq_gemm_wxa8_batch): problems that share
 * weight bits, scale mode (per_m) and output dtype, e.g. the q / k / v projections of one attention or the to_k / to_v of
 * every cross-attention (same text context).  The launch plan (tile shape) of problem 0 serves all; no K split. */
typedef struct dgq_gemm_args {
    const int8_t* codes; const float* rowsum; int rowsum_parts; int M, Kp;
    const void* wpacked; int w_bits; int N;
    int per_m; const float* cdelta; const uint8_t* cflush; const float* mdelta; const float* mzp; int L; float offset;
    const float* alpha; const float* zw; const float* gamma; const float* vn;
    void* y; int y_dtype; int ldy;
    const dgq_gemm_extra_t* extra;
} dgq_gemm_args_t;
int dgq_gemm_wxa8_batch(int n, const dgq_gemm_args_t* args, void* stream);
/* Small tile grids are split along K (deterministic: fp32 partial slabs [S][M][N] in the caller's `workspace`,
 * summed in a fixed order by a second kernel). dgq_gemm_workspace_bytes returns what the preferred split of a
 * shape needs; with workspace == NULL (or too small) fewer / no splits are used — results do not depend on it
 * beyond fp32 summation order. */
size_t dgq_gemm_workspace_bytes(int M, int N, int Kp);
int dgq_gemm_plan_splits(int M, int N, int Kp, int w_bits, int per_m, size_t workspace_bytes);

/* ---- attention-side quantizers --------------------------------------------------------------------
 * dgq_fakequant_rows: aqtizer_q/k/v (sd.py:174-182,199 -> quant_layer.py:295-299) applied on the
 * projection output viewed as [rows][C] (row = b·T + t, col = h·D + d): y = δ·(clamp(rne(x/δ)+z,0,2^b−1) − z).
 *   mode 0: scalar (delta[0]);  mode 1: per token, index (row % T) − skip;  mode 2: per head-dim, index col % D.
 * Tokens t < skip are copied unquantised (start_peak bypass of token 0, sd.py:176-180). In-place allowed. */
int dgq_fakequant_rows(const void* x, void* y, int dtype, int rows, int C, int T, int D,
                       int mode, const float* delta, const float* zp, int skip, int bits, void* stream);

/* dgq_max_f32: out[0] = max over p[rows][S] excluding columns < skip_cols (real-time δ = x.max(),
 * quant_layer_text.py:96-97). `out` must be pre-set to 0 by the caller; probabilities are >= 0. */
int dgq_max_f32(const float* p, int64_t rows, int S, int skip_cols, float* out, void* stream);
/* dgq_logquant_f32: T2ILogQuantizer.forward (quant_layer_text.py:101-105) on softmax probabilities,
 * y = δ·2^(−clamp(rne(−log2(p/δ)),0,2^b−1)), δ read from device memory; columns < skip_cols are copied
 * (start_peak, sd.py:191-195). In-place allowed. */
int dgq_logquant_f32(const float* p, float* y, int64_t rows, int S, int skip_cols, const float* delta,
                     int bits, void* stream);

/* dgq_attention_f32: the attention core of Attention.Attention_forward (sd.py:183-201) without materialising the
 * probabilities:  o = aqtizer_w(softmax((q·kᵀ)·scale)) · v, all fp32 (exact fp32 MFMA).
 * q [B][T][H·D], k/v [B][S][H·D], o [B][T][H·D] (the projection layout: no head transposes).
 * mode 0: no quantiser; 1: T2ILogQuantizer real-time (δ = max probability over the whole [B,H,T,S≥skip] tensor,
 *         quant_layer_text.py:96-105, found by a first statistics pass); 2: T2ILogQuantizer with δ = delta_in[0];
 *         3: UniformAffineQuantizer always_zero with δ = delta_in[0] (quant_block.py:145-156).
 * skip = 1 bypasses key column 0 (start_peak, sd.py:191-195).  workspace: caller-owned, 256-byte aligned,
 * >= dgq_attention_workspace_bytes(...) (δ scalar, per-row softmax statistics, bf16 split planes of K and V).
 * head_dim D ∈ {8,16,40,64,80,160}.  Quantised modes run on the bf16 MFMA with exact three-way bf16 operand splits
 * (fp32-equivalent accuracy); mode 0 on the exact fp32 MFMA.
 * fq (optional, NULL = none): the aqtizer_q / aqtizer_k / aqtizer_v fake-quantizers (sd.py:165-181) applied to the
 * operands as they are loaded, fq[0..2] = q, k, v, each addressed exactly like dgq_fakequant_rows (mode 0 scalar,
 * 1 per token with entry t − skip, 2 per head-dim element; mode < 0 = none for that operand; tokens < skip pass
 * through).  Only where dgq_attention_fuses_fakequant(D, mode) returns 1; DGQ_EUNSUPPORTED otherwise. */
typedef struct dgq_attn_fq {
    int mode;                  /* -1 none, 0 scalar, 1 per token, 2 per head-dim element */
    int skip;                  /* leading tokens left unquantised (start_peak key 0) */
    int bits;
    const float* delta;        /* device */
    const float* zero_point;   /* device */
} dgq_attn_fq_t;
int dgq_attention_f32(const float* q, const float* k, const float* v, float* o, int B, int H, int T, int S, int D,
                      float scale, int mode, int skip, const float* delta_in, int bits, const dgq_attn_fq_t* fq,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The same for q/k/v/o of dtype DGQ_F32 / DGQ_F16 / DGQ_BF16 (the reference's --fp16 mode, quant_model.py:183-201):
 * the arithmetic is unchanged (operands widened to fp32 and split exactly, fp32 accumulation), only the loads and the
 * final store follow the tensors' dtype.  Half types are served where dgq_attention_fuses_fakequant(D, mode) is 1. */
int dgq_attention(const void* q, const void* k, const void* v, void* o, int dtype, int B, int H, int T, int S, int D,
                  float scale, int mode, int skip, const float* delta_in, int bits, const dgq_attn_fq_t* fq,
                  void* workspace, size_t workspace_bytes, void* stream);
int dgq_attention_fuses_fakequant(int D, int mode);
size_t dgq_attention_workspace_bytes(int B, int H, int T, int S, int D);
/* Calls whose key range is at most 8 tiles (S <= 256: every cross-attention over the text tokens, the 16x16 / 8x8 self-attentions)
 * run statistics, the real-time δ maximum and P·V in ONE launch (csrc/attn_one.hip; results equal to the three-launch form bit for
 * bit).  Under mode 1 the workgroups of that launch exchange their maxima through the workspace and wait for each other with a
 * bounded poll; the form is taken only for grids that are resident as a whole.  dgq_attention_sync_timeouts: how many workgroups ever
 * gave up that poll in this process (device counter, read with a synchronous copy; 0 unless a grid was not resident — then the
 * affected call's output is invalid); < 0 on a runtime error.  DGQ_ATTN_ONE=0 in the environment keeps every call on three launches. */
int dgq_attention_sync_timeouts(void);

/* ---- batched small-M Linear ------------------------------------------------------------------------
 * dgq_linear_smallm_batch: y_l = W_l·aqtizer_l(act(x)) + b_l for up to 24 quantized Linear layers that share one input
 * x [M][K] of M <= 16 rows — the time_emb_proj(SiLU(temb)) projections of every resnet block of a forward
 * (quant_block.py:98-119), which do not depend on the latents — in ONE launch: quantise-on-load (scalar activation
 * quantizer of each layer, SiLU prologue when pre_act == 1) + integer dot products + the per_m epilogue of dgq_gemm_wxa8
 * (same arithmetic, term for term).  Weights in the natural K order (dgq_pack_w4 layout 1 / dgq_pack_w8 with kperm == NULL... Kp
 * padded to DGQ_KTILE); K <= 2048.  probs: host array, copied into the kernel arguments. */
typedef struct dgq_smallm_problem {
    const void* wpacked;
    const float* alpha;
    const float* zw;
    const float* gamma;
    const float* vn;
    const float* mdelta;
    const float* mzp;
    void* y;
    int ldy, N, Kp, w_bits, a_bits;
} dgq_smallm_problem_t;
int dgq_linear_smallm_batch(const void* x, int x_dtype, int M, int K, int64_t ldx, int pre_act, int n_problems,
                            const dgq_smallm_problem_t* probs, int y_dtype, void* stream);

/* ---- calibration producer --------------------------------------------------------
 * dgq_minmax_rows_cols: the statistics UniformAffineQuantizer.record_min_max_ema collects for DGQ's grouping
 * (quant/quant_layer.py:301-313): for x viewed as [rows][C] (row stride ldx elements, any fp dtype) the row-wise
 * (rowmin/rowmax [rows]) and column-wise (colmin/colmax [C]) minima / maxima, fp32.  Either pair may be NULL.
 * partial_ws: 2·slices·C floats of caller-owned scratch for the column pass (slices = row slices reduced in parallel). */
int dgq_minmax_rows_cols(const void* x, int dtype, int rows, int C, int64_t ldx,
                         float* rowmin, float* rowmax, float* colmin, float* colmax,
                         float* partial_ws, int slices, void* stream);

/* ---- weight PTQ: AdaRound soft quantiser + rounding regulariser -------------------------
 * The elementwise device ops of the reconstruction loop (reference: quant/adaptive_rounding.py:39-70 with soft_tgt,
 * quant/reconstruction_util.py:68-70), forward and backward; w, alpha, out, gout, galpha are [N][K] fp32 (K = the
 * flattened C·kh·kw), delta / zp [N] per output channel.
 *   soft_fwd : out = δ·(clamp(floor(w/δ) + h(α) + z, 0, 2^bits − 1) − z),  h(α) = clamp(sigmoid(α)·1.2 − 0.1, 0, 1)
 *   soft_bwd : galpha = gout · ∂out/∂α  (the clamps pass the gradient on their closed intervals, as torch.clamp)
 *   reg_fwd  : partial[blk] = Σ_blk (1 − |2h(α) − 1|^b); blk < dgq_adaround_reg_blocks(numel); the caller adds them up
 *   reg_bwd  : galpha = g[0] · ∂/∂α Σ (1 − |2h(α) − 1|^b); g is a 1-element device tensor (no host sync) */
int dgq_adaround_soft_fwd(const float* w, const float* delta, const float* zp, const float* alpha, int N, int K, int bits,
                          float* out, void* stream);
int dgq_adaround_soft_bwd(const float* gout, const float* w, const float* delta, const float* zp, const float* alpha, int N,
                          int K, int bits, float* galpha, void* stream);
int dgq_adaround_reg_blocks(int64_t numel);
int dgq_adaround_reg_fwd(const float* alpha, int64_t numel, float b, float* partial, void* stream);
int dgq_adaround_reg_bwd(const float* alpha, int64_t numel, float b, const float* g, float* galpha, void* stream);

#ifdef __cplusplus
}
#endif
#endif
