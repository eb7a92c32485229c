"""ctypes binding of libradmmm_hip.so (the C-ABI in include/radmmm_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
compute call below goes through the C-ABI with raw device pointers.  There is NO CPU
or eager fallback: if the shared library is missing the import of this module raises,
and every op raises on a non-GPU tensor.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RADMMM_LIB_PATH") or os.path.join(_HERE, "libradmmm_hip.so")   # override: A/B builds of the same ABI



def debug_env(name: str, default: Optional[str] = None) -> Optional[str]:
    """Value of an experiment / test switch (forced kernel choices, A/B paths): honoured only under RADMMM_DEBUG=1, like
    the library's own (csrc/error.cpp debug_env).  Supported switches are read with os.environ directly: RADMMM_PRECISION,
    RADMMM_GEMM_CUS, RADMMM_CHECK_SATURATION, RADMMM_LIB_PATH, RADMMM_LSTM."""
    if os.environ.get("RADMMM_DEBUG", "0") in ("", "0"):
        return default
    return os.environ.get(name, default)


ACT_NONE, ACT_SOFTPLUS, ACT_RELU, ACT_LEAKY = 0, 1, 2, 3
SCALE = {"tanh": 0, "exp": 1, "sigmoid": 2, "translate": 3}
ACT = {"none": 0, "softplus": 1, "relu": 2, "leaky_relu": 3}


class RadmmmError(RuntimeError):
    pass


class RowGemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int),
        ("a_item_stride", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int), ("b_tap_stride", C.c_int64), ("b_layout", C.c_int),
        ("C", C.c_void_p), ("ldc", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("taps", C.c_int), ("dil", C.c_int), ("sign", C.c_int),
        ("T", C.c_int),
        ("lens", C.c_void_p),
        ("a_mask_mode", C.c_int),
        ("bias", C.c_void_p),
        ("pconv", C.c_int), ("premask", C.c_int), ("postmask", C.c_int),
        ("ratio_taps", C.c_int), ("ratio_dil", C.c_int),
        ("add", C.c_void_p), ("ldadd", C.c_int),
        ("dact_src", C.c_void_p), ("lddact", C.c_int), ("dact", C.c_int),
        ("rowscale", C.c_int),
        ("act", C.c_int),
        ("C2", C.c_void_p), ("ldc2", C.c_int), ("c2_accum", C.c_int),
        ("Ch", C.c_void_p), ("Cl", C.c_void_p), ("ldch", C.c_int), ("ch_scale", C.c_float),
        ("C2h", C.c_void_p), ("C2l", C.c_void_p), ("ldc2h", C.c_int), ("c2h_scale", C.c_float),
        ("split_fmt", C.c_int), ("ch_x8_exp", C.c_int), ("c2h_x8_exp", C.c_int),
        ("sat_flag", C.c_void_p),
        ("Clo", C.c_void_p),
        ("colsum_out", C.c_void_p), ("colsum_scratch", C.c_void_p),
        ("dact_h", C.c_void_p), ("dact_x", C.c_void_p), ("lddact_h", C.c_int), ("dact_x8_exp", C.c_int),
        ("c2_src", C.c_void_p * 3), ("n_c2_src", C.c_int),
    ]


class CsItem(C.Structure):
    """radmmm_cs_item: one deferred column-sum final (partials [nparts][cols] -> out [cols])"""
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("nparts", C.c_int), ("cols", C.c_int)]


class DactItem(C.Structure):
    """radmmm_dact_item: one saved tensor of radmmm_dact_mul_rows_multi with its split pair and partial-sum rows"""
    _fields_ = [("saved", C.c_void_p), ("yh", C.c_void_p), ("yl", C.c_void_p), ("ylo16", C.c_void_p), ("part", C.c_void_p)]


class SplitOpts(C.Structure):
    """radmmm_split_opts: format of a split producer's second array (SPLIT_F16 / SPLIT_X8A / SPLIT_X8B), exponent of its
    8-bit parts, optional device saturation flag."""
    _fields_ = [("fmt", C.c_int), ("x8_exp", C.c_int), ("sat_flag", C.c_void_p), ("lo16", C.c_void_p)]


SPLIT_F16, SPLIT_X8A, SPLIT_X8B = 0, 1, 2


def split_opts(fmt: int = 0, x8_exp: int = 0, sat_flag: Optional[torch.Tensor] = None, lo16: Optional[torch.Tensor] = None):
    """byref(SplitOpts) for a C-ABI call (the struct is read during the call only)."""
    return C.byref(SplitOpts(fmt, x8_exp, sat_flag.data_ptr() if sat_flag is not None else None,
                             lo16.data_ptr() if lo16 is not None else None))


class RowGemmH3Desc(C.Structure):
    _fields_ = [
        ("base", RowGemmDesc),
        ("Ah", C.c_void_p), ("Al", C.c_void_p), ("lda_h", C.c_int),
        ("Bh", C.c_void_p), ("Bl", C.c_void_p), ("ldb_h", C.c_int), ("b_tap_stride_h", C.c_int64),
        ("acc_scale", C.c_float),
        ("nprod", C.c_int),
        ("a8_exp", C.c_int), ("b8_exp", C.c_int),
        ("extra_tap", C.c_int), ("extra_a_rows", C.c_int),
    ]


class WnItem(C.Structure):
    """radmmm_wn_item: one tensor of radmmm_weightnorm_fwd_h3_multi"""
    _fields_ = [("v", C.c_void_p), ("g", C.c_void_p), ("Wh", C.c_void_p), ("Wl", C.c_void_p), ("inv_norm", C.c_void_p),
                ("Cout", C.c_int), ("Cin", C.c_int), ("taps", C.c_int), ("ldk", C.c_int), ("perm_split", C.c_int),
                ("off_lo", C.c_int), ("off_hi", C.c_int)]


class TpItem(C.Structure):
    """radmmm_tp_item: one pair of radmmm_transpose_f16_pair_multi"""
    _fields_ = [("src_h", C.c_void_p), ("src_l", C.c_void_p), ("dst_h", C.c_void_p), ("dst_l", C.c_void_p),
                ("src_batch", C.c_int64), ("dst_batch", C.c_int64), ("ld_src", C.c_int), ("ld_dst", C.c_int),
                ("batches", C.c_int), ("rows", C.c_int), ("cols", C.c_int)]


class WgradDesc(C.Structure):
    _fields_ = [
        ("GY", C.c_void_p), ("ldgy", C.c_int),
        ("X", C.c_void_p), ("ldx", C.c_int),
        ("P", C.c_void_p), ("ldp", C.c_int), ("split_stride", C.c_int64),
        ("R", C.c_int),
        ("Mc", C.c_int), ("Nc", C.c_int),
        ("taps", C.c_int), ("dil", C.c_int),
        ("T", C.c_int), ("lens", C.c_void_p), ("x_mask_mode", C.c_int),
        ("splits", C.c_int),
    ]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or rad_mmm_amd/csrc/build.sh (hipcc --offload-arch=gfx950). rad_mmm_amd has no fallback path.")
    lib = C.CDLL(LIB_PATH)
    lib.radmmm_last_error.restype = C.c_char_p
    lib.radmmm_abi_version.restype = C.c_int
    lib.radmmm_gemm_cu_slots.restype = C.c_int
    lib.radmmm_gemm_cu_slots.argtypes = []
    if lib.radmmm_abi_version() != 4:
        raise ImportError("libradmmm_hip.so ABI version mismatch")
    i, i64, p = C.c_int, C.c_int64, C.c_void_p
    f = C.c_float
    so = C.POINTER(SplitOpts)
    sig = {
        "radmmm_rowgemm_f32": [C.POINTER(RowGemmDesc), p],
        "radmmm_wgrad_f32": [C.POINTER(WgradDesc), p],
        "radmmm_rowgemm_h3": [C.POINTER(RowGemmH3Desc), p],
        "radmmm_weightnorm_fwd": [p, p, p, p, i, i, i, i, i, i, i, p],
        "radmmm_weightnorm_bwd": [p, p, p, p, i, i64, p, p, i, i, i, i, i, i, i, p, p],
        "radmmm_wn_input_fwd": [p, i, p, i, p, i, i, i, i, p, p, so, p],
        "radmmm_wn_input_bwd": [p, i, p, i, i, p, i, i, i, i, p],
        "radmmm_squeeze_rows": [p, p, i, i, i, i, i, i, p],
        "radmmm_unsqueeze_rows": [p, p, i, i, i, i, i, i, p],
        "radmmm_affine_coupling_fwd": [p, i, p, i, p, p, i, i, i, p],
        "radmmm_affine_coupling_bwd": [p, i, p, i, p, p, p, p, i, i, i, p],
        "radmmm_dact_mul": [p, i, p, i, p, i, i, i, i, i, i, p, i, i, p, p, i, f, so, p],
        "radmmm_colsum": [p, i, p, p, i, i, i, i, p, i, i, i, p],
        "radmmm_masked_reduce": [p, i, i, i, i64, i64, i64, p, i, p, p, p],
        "radmmm_masked_reduce_bwd": [p, i, i, i, i64, i64, i64, p, i, p, p, p],
        "radmmm_fused_add_tanh_sigmoid_multiply": [p, p, i, p, i, i, i, p],
        "radmmm_film_fwd": [p, i, p, i, p, i, p, p, p, p, p, i, i, i, i, p],
        "radmmm_film_bwd": [p, i, p, i, p, i, p, p, p, p, f, i, p, p, i, p, i, p, i, p, p, p, i, i, i, p],
        "radmmm_film_bwd_sums": [p, i, p, i, p, i, p, p, p, p, p, p, p, i, i, p],
        "radmmm_film_bwd_apply": [p, i, p, i, p, i, p, p, p, p, f, i, p, p, i, p, i, p, i, p, i, i, p],
        "radmmm_split_f16": [p, i, p, p, i, i, i, f, so, p],
        "radmmm_transpose_split_act": [p, i, i, i, i, i, i, p, i, f, p, p, p, p, i, p],
        "radmmm_wgrad_h3_tiles": [i, i, i],
        "radmmm_wgrad_rm_tiles": [i, i, i],
        "radmmm_wgrad_rm": [p, p, i, p, p, i, i, i, p, i, p, i, i64, i, i, i, i, i, f, p],
        "radmmm_wgrad_rm8": [p, p, i, i, p, p, i, i, i, i, p, i, p, i, i64, i, i, i, i, i, f, p],
        "radmmm_betabinom_prior": [i, i, C.c_double, p, p],
        "radmmm_prior_zoom_batch": [p, i, p, i, i, p],
        "radmmm_energy_average": [p, p, i, i, i, i, p],
        "radmmm_lu_weight_fwd": [p, p, p, p, p, i, p, i, i, p, p],
        "radmmm_lu_weight_bwd": [p, p, p, p, p, i, p, i, i, p, p, p, p, p],
        "radmmm_instnorm_fwd": [p, i, p, p, p, i, p, p, p, i, i, i, f, i, p],
        "radmmm_instnorm_bwd": [p, i, p, i, p, i, p, p, p, p, i, p, p, p, i, i, i, i, p],
        "radmmm_pq_spline_inv": [p, i, p, i, p, i, i, i, i, p],
        "radmmm_sumsq": [p, i64, p, p],
        "radmmm_radam_step": [p, p, p, p, i64, p, f, f, f, f, f, i, p],
        "radmmm_transpose_split_act_colsum": [p, i, i, i, i, i, i, p, i, f, p, p, p, p, i, p, i, i, i, p],
        "radmmm_colsum_final": [p, p, i, i, p],
        "radmmm_colsum_final_multi": [C.POINTER(CsItem), i, p], "radmmm_rowgemm_h3_colsum_rows": [C.POINTER(RowGemmH3Desc)],
        "radmmm_dact_mul_transposed": [p, i, p, i, i, i, i, i, i, i, f, p, p, i, so, p, p, i, p, p],
        "radmmm_dact_mul_rows": [p, i, p, i, i, i, i, i, i, p, i, i, f, p, p, i, so, p, p],
        "radmmm_dact_mul_rows_multi": [p, i, C.POINTER(DactItem), i, i, i, i, i, i, f, i, so, p],
        "radmmm_lstm_fwd": [p, p, p, p, p, p, p, p, i, i, i, p],
        "radmmm_stream_create_masked": [i, C.POINTER(C.c_void_p)],
        "radmmm_stream_destroy": [p],
        "radmmm_ctc_monotonic": [p, p, p, p, p, p, i, i, i, p],
        "radmmm_lstm_bwd": [p, p, p, p, p, p, p, p, i, i, i, p, p],
        "radmmm_wgrad_h3": [p, p, p, p, p, p, i, i, i, p, i, i64, i, i, i, i, i, f, i, p],
        "radmmm_weightnorm_fwd_h3": [p, p, p, p, p, i, i, i, i, i, i, i, f, so, p],
        "radmmm_transpose_f16_pair": [p, p, i, i64, p, p, i, i64, i, i, i, i, i, p],
        "radmmm_weightnorm_fwd_h3_multi": [p, i, f, so, p], "radmmm_transpose_f16_pair_multi": [p, i, i, i, p],
        "radmmm_pq_spline_fwd": [p, i, p, i, p, i, p, i, i, i, p],
        "radmmm_pq_spline_bins": [p, i, p, i, p, p, p, i, i, i, p],
        "radmmm_pq_spline_bwd": [p, i, p, i, p, i, p, p, i, p, i, i, i, i, p],
        "radmmm_attn_fwd": [p, p, p, p, p, p, i, i, i, i, f, p],
        "radmmm_attn_bwd": [p, p, p, p, p, p, p, p, p, p, p, i, i, i, i, f, p],
        "radmmm_mas_width1": [p, p, p, p, p, i, i, i, p],
        "radmmm_mas_width1_prob": [p, p, p, p, p, i, i, i, p],
        "radmmm_stft_mel": [p, p, p, p, p, i, i, i, i, i, f, p],
    }
    missing = [n for n in sig if not hasattr(lib, n)]
    if missing:
        raise ImportError(f"libradmmm_hip.so lacks symbols {missing}: rebuild it")
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, args in {"radmmm_colsum_scratch_floats": [i, i], "radmmm_rowgemm_h3_colsum_scratch_floats": [i, i],
                       "radmmm_masked_reduce_scratch_floats": [i, i, i],
                       "radmmm_mas_scratch_bytes": [i, i, i], "radmmm_ctc_monotonic_scratch_floats": [i, i, i],
                       "radmmm_film_bwd_scratch_floats": [i, i],
                       "radmmm_stft_mel_scratch_floats": [i, i, i, i, i],
                       "radmmm_lstm_scratch_bytes": [i, i, i], "radmmm_lstm_hseq_bytes": [i, i, i],
                       "radmmm_sumsq_scratch_floats": []}.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int64
    return lib


lib = _load()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RadmmmError(f"{what} failed ({rc}): {lib.radmmm_last_error().decode()}")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RadmmmError("rad_mmm_amd ops need GPU tensors (there is no CPU path)")
    if t.dtype is torch.bfloat16:
        raise RadmmmError("a bfloat16 tensor reached the fp32 C ABI (an autocast region upstream?): cast it to float")
    return t.data_ptr()


def f32c(t: torch.Tensor) -> torch.Tensor:
    """contiguous fp32 view/copy"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def rowgemm(**kw) -> None:
    d = RowGemmDesc()
    d.sign = 1
    d.taps = 1
    d.dil = 1
    d.ratio_taps = 1
    d.ratio_dil = 1
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = ptr(v)
        setattr(d, k, v)
    check(lib.radmmm_rowgemm_f32(C.byref(d), stream()), "radmmm_rowgemm_f32")


_H3_KEYS = {"Ah", "Al", "lda_h", "Bh", "Bl", "ldb_h", "b_tap_stride_h", "acc_scale", "nprod", "a8_exp", "b8_exp", "extra_tap",
            "extra_a_rows"}


# Measurement hook (bench.py's roofline leg): LAUNCH_TIMER(kw) -> True brackets that rowgemm_h3 launch with a pair of HIP
# events on the stream it is launched on; the pairs collect in LAUNCH_EVENTS.  None (the default) costs one comparison.
LAUNCH_TIMER = None
LAUNCH_EVENTS: list = []


def rowgemm_h3(**kw) -> None:
    """split-f16 row GEMM: keys of RowGemmDesc (epilogue, shapes) + the split operands."""
    if LAUNCH_TIMER is not None and LAUNCH_TIMER(kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _rowgemm_h3(kw)
        e1.record()
        LAUNCH_EVENTS.append((e0, e1))
        return
    _rowgemm_h3(kw)


def rowgemm_h3_colsum_rows(**kw) -> int:
    """rows of partial column sums this launch would leave in colsum_scratch with colsum_out = None (0: it cannot defer)"""
    return int(lib.radmmm_rowgemm_h3_colsum_rows(C.byref(_h3_desc(kw))))


def _rowgemm_h3(kw) -> None:
    check(lib.radmmm_rowgemm_h3(C.byref(_h3_desc(kw)), stream()), "radmmm_rowgemm_h3")


def _h3_desc(kw):
    d = RowGemmH3Desc()
    d.base.sign = 1
    d.base.taps = 1
    d.base.dil = 1
    d.base.ratio_taps = 1
    d.base.ratio_dil = 1
    d.acc_scale = 1.0
    for k, v in kw.items():
        if k == "c2_src":                               # list of 1 .. 3 fp32 tensors (radmmm_rowgemm_desc.c2_src / n_c2_src)
            d.base.c2_src = (C.c_void_p * 3)(*[ptr(t) for t in v], *([None] * (3 - len(v))))
            d.base.n_c2_src = len(v)
            continue
        if isinstance(v, torch.Tensor):
            v = ptr(v)
        setattr(d if k in _H3_KEYS else d.base, k, v)
    return d


def wgrad(**kw) -> None:
    d = WgradDesc()
    d.taps = 1
    d.dil = 1
    d.splits = 1
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = ptr(v)
        setattr(d, k, v)
    check(lib.radmmm_wgrad_f32(C.byref(d), stream()), "radmmm_wgrad_f32")


# Every custom autograd Function of this package computes in fp32 through raw pointers: under torch.autocast
# (Lightning `precision: bf16-mixed`) a stock op inside a forward would hand a bf16 tensor to an fp32 kernel.
# amp_fwd switches autocast off inside forward (casting floating tensor arguments to fp32), amp_bwd runs
# backward in the same state.
amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
amp_bwd = torch.amp.custom_bwd(device_type="cuda")


def fp32_region(fn):
    """Module-level forwards of this package also hold a few stock matmuls (W @ mean of the whitening conv, the LSTM's
    input projection, hipBLASLt GEMMs of the predictors): run them with autocast off, like the kernels they feed."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        with torch.autocast(device_type="cuda", enabled=False):
            return fn(*args, **kwargs)
    return wrapped
