"""ctypes binding of libdgq_hip.so (include/dgq_hip.h).  There is NO fallback: if the shared
library is missing or a call fails, the product raises."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DGQ_HIP_LIB") or os.path.join(_HERE, "csrc", "libdgq_hip.so")   # override: A/B builds of the kernels

_vp, _i, _f, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
ABI_VERSION = 122          # DGQ_ABI_VERSION of include/dgq_hip.h: the struct layouts below are that revision's

# name -> argtypes; every function returns int except dgq_last_error
SIGNATURES = {
    "dgq_version": [],
    "dgq_quantize_weight": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "dgq_pack_w4": [_vp, _i, _i, _vp, _i, _i, _vp, _vp],
    "dgq_unpack_w4": [_vp, _i, _i, _i, _vp, _vp],
    "dgq_pack_w8": [_vp, _i, _i, _vp, _i, _vp, _vp],
    "dgq_quant_act": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i,
                      _vp, _vp, _i, _vp, _vp, _f, _vp],
    "dgq_groupnorm_scale_shift": [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "dgq_quant_act_parts": [_i, _i],
    "dgq_gemm_wxa8": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _i,
                      _vp, ctypes.c_size_t, _vp, _vp],
    "dgq_gemm_workspace_bytes": [_i, _i, _i],
    "dgq_gemm_plan_splits": [_i, _i, _i, _i, _i, ctypes.c_size_t],
    "dgq_gemm_act_fuses": [_i, _i, _i, _i, _i, _i, _i, _i, _i],
    "dgq_gemm_conv_act_fuses": [_i] * 14,
    "dgq_groupnorm_from_partials": [_vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "dgq_fakequant_rows": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp],
    "dgq_max_f32": [_vp, _i64, _i, _i, _vp, _vp],
    "dgq_logquant_f32": [_vp, _vp, _i64, _i, _i, _vp, _i, _vp],
    "dgq_attention_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _i, _vp, _vp, ctypes.c_size_t, _vp],
    "dgq_attention": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _i, _vp, _vp, ctypes.c_size_t, _vp],
    "dgq_conv2d_f32w": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp],
    "dgq_attention_fuses_fakequant": [_i, _i],
    "dgq_attention_workspace_bytes": [_i, _i, _i, _i, _i],
    "dgq_attention_sync_timeouts": [],
    "dgq_minmax_rows_cols": [_vp, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "dgq_linear_smallm_batch": [_vp, _i, _i, _i, _i64, _i, _i, _vp, _i, _vp],
    "dgq_quant_act_batch": [_i, _vp, _vp],
    "dgq_quant_act_variant": [_vp],
    "dgq_quant_act_conv_tile": [_i, _i, _i, _i, _i, _vp],
    "dgq_gemm_wxa8_batch": [_i, _vp, _vp],
    "dgq_adaround_soft_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "dgq_adaround_soft_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "dgq_adaround_reg_blocks": [_i64],
    "dgq_adaround_reg_fwd": [_vp, _i64, _f, _vp, _vp],
    "dgq_adaround_reg_bwd": [_vp, _i64, _f, _vp, _vp, _vp],
    "dgq_timestep_embedding": [_vp, _i, _i64, _i, _i, _vp, _i, _vp],
    "dgq_cfg_ddim_step": [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _f, _f, _f, _f, _f, _vp],
}



class GemmExtra(ctypes.Structure):
    """dgq_gemm_extra_t of include/dgq_hip.h"""
    _fields_ = [("residual", _vp), ("ldr", _i), ("res_div", _i), ("res_dtype", _i), ("fq_mode", _i), ("fq_delta", _vp), ("fq_zp", _vp),
                ("fq_T", _i), ("fq_D", _i), ("fq_skip", _i), ("fq_qmax", _f), ("geglu", _i), ("gn_partial", _vp), ("conv", _vp), ("flush_coef", _vp), ("wfrag", _vp), ("act", _vp), ("y2", _vp), ("ldy2", _i)]


class GemmAct(ctypes.Structure):
    """dgq_gemm_act_t of include/dgq_hip.h"""
    _fields_ = [("x", _vp), ("x_dtype", _i), ("ldx", _i), ("K", _i), ("kdst", _vp), ("czp", _vp), ("bits", _i),
                ("pre_scale", _vp), ("pre_shift", _vp), ("rows_per_image", _i), ("pre_act", _i),
                ("ln_gamma", _vp), ("ln_beta", _vp), ("ln_eps", _f),
                ("kpat", _vp), ("B", _i), ("H", _i), ("W", _i), ("kh", _i), ("kw", _i), ("stride", _i), ("pad", _i)]


class GemmConv(ctypes.Structure):
    """dgq_gemm_conv_t of include/dgq_hip.h"""
    _fields_ = [("codes_in", _vp), ("pixsum", _vp), ("fill", _vp), ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("ldc", _i), ("kh", _i),
                ("kw", _i), ("stride", _i), ("pad", _i), ("Ho", _i), ("Wo", _i), ("zero_code", _f), ("pixsum_parts", _i)]


class SmallMProblem(ctypes.Structure):
    """dgq_smallm_problem_t of include/dgq_hip.h"""
    _fields_ = [("wpacked", _vp), ("alpha", _vp), ("zw", _vp), ("gamma", _vp), ("vn", _vp), ("mdelta", _vp), ("mzp", _vp),
                ("y", _vp), ("ldy", _i), ("N", _i), ("Kp", _i), ("w_bits", _i), ("a_bits", _i)]


class QuantActArgs(ctypes.Structure):
    """dgq_quant_act_args_t of include/dgq_hip.h"""
    _fields_ = [("x", _vp), ("x_dtype", _i), ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("kh", _i), ("kw", _i), ("stride", _i),
                ("pad", _i), ("ksrc", _vp), ("koff", _vp), ("klds", _vp), ("kdst", _vp), ("Kp", _i), ("per_m", _i), ("delta", _vp), ("zp", _vp),
                ("L", _i), ("bits", _i), ("codes", _vp), ("rowsum", _vp), ("ksplits", _i), ("pre_scale", _vp), ("pre_shift", _vp),
                ("pre_act", _i), ("ln_gamma", _vp), ("ln_beta", _vp), ("ln_eps", _f), ("kpat", _vp), ("ups", _i)]


class GemmArgs(ctypes.Structure):
    """dgq_gemm_args_t of include/dgq_hip.h"""
    _fields_ = [("codes", _vp), ("rowsum", _vp), ("rowsum_parts", _i), ("M", _i), ("Kp", _i), ("wpacked", _vp), ("w_bits", _i),
                ("N", _i), ("per_m", _i), ("cdelta", _vp), ("cflush", _vp), ("mdelta", _vp), ("mzp", _vp), ("L", _i),
                ("offset", _f), ("alpha", _vp), ("zw", _vp), ("gamma", _vp), ("vn", _vp), ("y", _vp), ("y_dtype", _i), ("ldy", _i),
                ("extra", _vp)]


class AttnFq(ctypes.Structure):
    """dgq_attn_fq_t of include/dgq_hip.h"""
    _fields_ = [("mode", _i), ("skip", _i), ("bits", _i), ("delta", _vp), ("zero_point", _vp)]


_lib = None


def load():
    """Loads the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "dgq_amd: %s not found — build it with `make -C dgq_amd/csrc` (or __graft_entry__.build()); "
                "there is no CPU fallback" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = ctypes.c_size_t if name.endswith("_workspace_bytes") else ctypes.c_int
        if lib.dgq_version() != ABI_VERSION:
            raise RuntimeError("dgq_amd: %s is ABI revision %d, this binding was written against %d (include/dgq_hip.h: "
                               "DGQ_ABI_VERSION) — rebuild with `make -C dgq_amd/csrc`" % (LIB_PATH, lib.dgq_version(), ABI_VERSION))
        lib.dgq_last_error.argtypes = []
        lib.dgq_last_error.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def last_error():
    return load().dgq_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, last_error()))


DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "dgq_amd kernels need device tensors (no CPU path)"
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("dgq_amd: no GPU visible — the quantized path runs only through the HIP kernels "
                           "(no CPU fallback)")
    load()
