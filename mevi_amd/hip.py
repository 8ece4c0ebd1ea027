"""ctypes binding of libmevi_hip.so (C ABI: include/mevi_hip.h).

The library is loaded lazily and loudly: a missing .so or a missing GPU raises --
there is no eager/CPU fallback anywhere in mevi_amd.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (loads the HIP runtime the extension binds to)

from .build import LIB

_lib = None


class MeviHipError(RuntimeError):
    pass


class IpTopkStats(ctypes.Structure):
    _fields_ = [("n_chunks", c_int64), ("n_failed_queries", c_int64), ("n_fallback_chunks", c_int64),
                ("filter_ms", c_double), ("compact_ms", c_double), ("filter_flops", c_double),
                ("max_err_ratio", c_double), ("err_bound", c_double), ("n_second_pass_queries", c_int64),
                ("n_filter_candidates", c_int64), ("max_launch_candidates", c_int64), ("n_list_overflows", c_int64),
                ("n_i8_queries", c_int64), ("n_i8_unproven", c_int64)]


_SIGNATURES = {
    "mevi_abi_version": (c_int, []),
    "mevi_last_error": (ctypes.c_char_p, []),
    "mevi_ip_topk_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "mevi_ip_topk_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mevi_ip_index_bytes": (c_size_t, [c_int64, c_int64]),
    "mevi_ip_index_build_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "mevi_ip_topk_indexed_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "mevi_ip_topk_indexed_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mevi_ip_index8_bytes": (c_size_t, [c_int64, c_int64]),
    "mevi_ip_index8_build_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "mevi_ip_topk_indexed8_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "mevi_ip_topk_indexed8_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                          c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mevi_topk_merge_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "mevi_topk_merge_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mevi_pack_lists_i64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mevi_topk_merge_packed_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "mevi_rq_encode_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "mevi_rq_encode_fast_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "mevi_rq_encode_fast_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t,
                                        c_void_p]),
    "mevi_rq_encode_fast_stats": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "mevi_gemm_nt_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                 c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mevi_split_kp": (c_int64, [c_int64]),
    "mevi_split_rows_f16": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mevi_rmsnorm_split_f16": (c_int, [c_void_p, c_int64, c_void_p, c_float, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "mevi_gemm_nt_split_to_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int64,
                                            c_int64, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mevi_gemm_nt_split_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                       c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mevi_gemm_rmsnorm_supported": (c_int, [c_int64, c_int64, c_int64]),
    "mevi_gemm_norm_fold_supported": (c_int, [c_int64]),
    "mevi_split_rows_ssq_f16": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mevi_row_rscale_f32": (c_int, [c_void_p, c_int64, c_int64, c_float, c_float, c_void_p, c_void_p]),
    "mevi_gemm_nt_split_normed_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int,
                                              c_void_p]),
    "mevi_gemm_nt_split_normed_to_split": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p, c_float,
                                                   c_void_p, c_void_p, c_float, c_int64, c_int64, c_int64, c_void_p, c_float, c_int,
                                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "mevi_gemm_nt_split_residual_stream": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_float, c_float, c_void_p,
                                                   c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                                   c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mevi_gemm_nt_rmsnorm_split_f32": (c_int, [c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                               c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mevi_gemm_nt_rmsnorm_split_to_split": (c_int, [c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_float, c_int64, c_int64,
                                                    c_int64, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mevi_rmsnorm_f32": (c_int, [c_void_p, c_int64, c_void_p, c_float, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "mevi_add_layernorm_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_float,
                                       c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "mevi_gather_rows_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "mevi_scatter_rows_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "mevi_scale_f32": (c_int, [c_void_p, c_float, c_int64, c_void_p, c_void_p]),
    "mevi_attention_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                   c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                   c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "mevi_attention_cached_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                          c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                          c_int64, c_int, c_float, c_void_p]),
    "mevi_attention_varlen_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                          c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                          c_int, c_float, c_void_p]),
    "mevi_attention_split_f16": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                         c_void_p, c_int64, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                         c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_float, c_void_p,
                                         c_void_p]),
    "mevi_attention_cached_split_f16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                                c_void_p, c_int64, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                                c_void_p, c_int64, c_int64, c_int64, c_int, c_float, c_void_p]),
    "mevi_attention_varlen_split_f16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                                c_int, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                                c_int64, c_int64, c_int, c_float, c_void_p]),
    "mevi_adaptive_logits_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                         c_int64, c_void_p, c_void_p]),
    "mevi_adaptive_logits_rows_f32": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                              c_int64, c_void_p, c_void_p]),
    "mevi_gemm_nt_split_head_supported": (c_int, [c_int64, c_int64, c_int64, c_int64]),
    "mevi_gemm_nt_split_head_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                            c_int64, c_float, c_int64, c_void_p, c_void_p]),
    "mevi_logits_finish_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "mevi_beam_step_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "mevi_beam_step_tree_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                        c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mevi_row_softmax_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "mevi_pair_dot_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                  c_void_p, c_void_p]),
    "mevi_segment_sort_desc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "mevi_segment_aggregate_sort_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                                c_void_p, c_void_p]),
    "mevi_rq_neg_dist_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "mevi_gather_sub_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "mevi_cluster_means_workspace_bytes": (ctypes.c_size_t, [c_int64, c_int64, c_int64]),
    "mevi_cluster_means_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "mevi_format_f32_list": (c_int64, [c_void_p, c_int64, c_void_p, c_int64]),
    "mevi_format_i64_list": (c_int64, [c_void_p, c_int64, c_void_p, c_int64]),
    "mevi_format_ranked_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, ctypes.c_int32]),
    "mevi_parse_i64_list": (c_int64, [ctypes.c_char_p, c_int64, c_void_p, c_int64]),
    "mevi_parse_f64_list": (c_int64, [ctypes.c_char_p, c_int64, c_void_p, c_int64]),
    "mevi_parse_tsv_columns": (c_int64, [c_void_p, c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_void_p,
                                         c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64]),
    "mevi_cluster_ranks_i32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                       ctypes.c_int32, c_void_p, c_void_p, c_void_p]),
    "mevi_ensemble_rank_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_int64, ctypes.c_int32, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "mevi_first_hits_i64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mevi_ip_topk_set_growth": (None, [c_double]),
    "mevi_ip_topk_set_profiling": (None, [c_int]),
    "mevi_ip_topk_get_stats": (None, [ctypes.POINTER(IpTopkStats)]),
}


def exported_symbols():
    """Names every entry point include/mevi_hip.h declares."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise MeviHipError(
                f"{LIB} not found: build it with `python -m mevi_amd.build` "
                "(or __graft_entry__.build()); mevi_amd has no CPU fallback")
        L = ctypes.CDLL(LIB)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().mevi_last_error().decode("utf-8", "replace")
        raise MeviHipError(f"{what} failed with status {status}: {msg}")


def require_gpu():
    if not torch.cuda.is_available():
        raise MeviHipError("no MI355X visible (torch.cuda.is_available() is False); "
                           "mevi_amd has no CPU fallback")


# Host cost of a launch: the small-batch passes (the reference's own batch sizes) are ~250-550 launches during which the GPU
# mostly waits for Python.  cProfile of a 128-query tower pass (tools/prof_host.py, 4.5 ms for 247 launches): the public
# `torch.cuda.current_stream()` took 7.7 us per call and `with torch.cuda.device(...)` 4-8 us -- more than the ctypes call
# itself.  The raw hooks below are what torch's own Python wrappers end in.
# They are private: a torch build without them (CPU-only wheel, a rename) falls back to the public calls instead of failing
# the import -- `require_gpu()` / MeviHipError stay the way a missing GPU is reported.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)
_cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def stream_ptr():
    """The current device's current stream as an integer handle (ctypes passes it as void *); the capturing stream inside
    `torch.cuda.graph`."""
    return _raw_stream(_cur_device())


class _NoGuard:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(dev):
    """`with device_guard(t.device):` = `with torch.cuda.device(t.device):`, for free when that device is already current."""
    idx = dev.index
    if idx is None or idx == _cur_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def on_device(fn):
    """Run `fn` with its first tensor argument's GPU as the current device, so that `stream_ptr()` and the launch refer
    to the device the pointers live on (the public classes take `device=`; callers need not torch.cuda.set_device)."""
    import functools

    @functools.wraps(fn)
    def run(*args, **kw):
        t = next((a for a in args if torch.is_tensor(a)), None)
        if t is None or not t.is_cuda or t.device.index == _cur_device():
            return fn(*args, **kw)
        with torch.cuda.device(t.device):
            return fn(*args, **kw)

    return run


def ptr(t):
    """Device address of a tensor's first element, as the integer ctypes passes for a void * argument."""
    return t.data_ptr()
