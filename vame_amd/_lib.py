"""ctypes binding of libvame_hip.so (the gfx950 kernels) -- fails loudly when the library is missing.

There is no CPU fallback: every op in :mod:`vame_amd.ops` goes through this C ABI
(``include/vame_hip.h``), needs the library built for gfx950 and tensors in HIP device memory.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvame_hip.so")
_lib = None


class VameHipError(RuntimeError):
    pass


_SIGS = {
    "vame_source_id": (c_char_p, []),
    "vame_version": (c_int, []),
    "vame_last_error": (c_char_p, []),
    "vame_clock_stamp": (c_int, [c_void_p, c_int, c_void_p]),
    "vame_window_gather_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "vame_gemm_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_int64, c_int,
                              c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "vame_gemm_group_f32": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_int64, c_int,
                                    c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "vame_gemm_bf16x6_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vame_gemm_group_bf16x6_f32": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                           c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vame_gru_pack_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_gru_pack_batch_f32": (c_int, [c_void_p, c_int, c_void_p]),
    "vame_gru_pack_x_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vame_gru_stash_floats": (c_int64, [c_int, c_int, c_int]),
    "vame_gru_seq_fwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "vame_gru_seq_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "vame_gru_seq_bwd_has_kernel": (c_int, [c_int, c_int]),
    "vame_gru_seq_fwd_has_kernel": (c_int, [c_int, c_int]),
    "vame_gru_cell_fwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                      c_int, c_int, c_void_p]),
    "vame_gru_cell_bwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int,
                                      c_void_p]),
    "vame_latent_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "vame_loss_finish_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "vame_latent_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                    c_void_p, c_void_p]),
    "vame_head_stream_ws_floats": (c_int64, [c_int, c_int, c_int]),
    "vame_head_stream_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                             c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_mse_fwd_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "vame_nuclear_state_doubles": (c_int64, [c_int]),
    "vame_nuclear_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_kmeans_assign_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vame_timesum_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "vame_colsum_ws_floats": (c_int64, [c_int64, c_int]),
    "vame_colsum_f32": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "vame_colsum_batch_f32": (c_int, [c_void_p, c_int, c_void_p]),
    "vame_adam_amsgrad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                                      c_float, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_mask_scale_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_float, c_void_p, c_int64, c_int, c_void_p]),
    "vame_linear_group_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_axpy_f32": (c_int, [c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "vame_index_copy_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vame_gru_coop_flag_ints": (c_int64, [c_int, c_int, c_int]),
    "vame_gru_coop_supported": (c_int, [c_int, c_int, c_int]),
    "vame_gru_coop_set_poll_limit": (c_int, [c_int]),
    "vame_gru_coop_fwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "vame_gru_coop_xbuf_floats": (c_int64, [c_int, c_int, c_int]),
    "vame_gru_coop_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "vame_gru_wide_supported": (c_int, [c_int]),
    "vame_gru_wide_fwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "vame_gru_wide_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "vame_gru_cell_bwd_frag_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p]),
    "vame_hmm_emission_f64": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_hmm_ws_doubles": (c_int64, [c_int64, c_int, c_int]),
    "vame_hmm_forward_f64": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_hmm_backward_f64": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_hmm_stats_doubles": (c_int64, [c_int, c_int]),
    "vame_hmm_stats_ws_doubles": (c_int64, [c_int, c_int]),
    "vame_hmm_stats_f64": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_hmm_viterbi_ws_bytes": (c_int64, [c_int64, c_int, c_int]),
    "vame_hmm_viterbi_f64": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_prep_zscore_mask_f64": (c_int, [c_void_p, c_int, c_int64, c_int64, c_double, c_double, c_double, c_int, c_void_p, c_int64, c_void_p]),
    "vame_prep_ws_bytes": (c_int64, [c_int]),
    "vame_prep_fill_last_valid_f64": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "vame_prep_fill_across_features_f64": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "vame_prep_rowstats_f64": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vame_prep_savgol_f64": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
}
EXPORTS = tuple(_SIGS)


def _bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here = the .so does not export the declared ABI
        fn.restype = res
        fn.argtypes = args
    return lib


def lib():
    """The loaded library; raises if the HIP extension has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VameHipError(
                f"{LIB_PATH} not found: the gfx950 HIP extension is not built. Run `python __graft_entry__.py` "
                "(or `make`) first; vame_amd has no CPU fallback.")
        _lib = _bind(LIB_PATH)
    return _lib


def device(index=None):
    """The HIP device the path runs on (current device, or `index`); raises when no MI355X is visible."""
    import torch
    if not torch.cuda.is_available():
        raise VameHipError("vame_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    if index is not None:
        torch.cuda.set_device(index)
    return torch.device("cuda", torch.cuda.current_device())


def require_device_tensor(t):
    if not t.is_cuda:
        raise VameHipError("vame_amd ops need CUDA(HIP) tensors; there is no CPU path")


_raw_stream = None


def stream_handle():
    """hipStream_t of torch's current stream: every kernel of the path is launched on it.  (The raw-handle query: building a
    torch.cuda.Stream object per launch costs ~2 us of a host-bound small-batch step, ~110 launches.)"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def check(rc, what=""):
    if rc != 0:
        msg = lib().vame_last_error()
        raise VameHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
