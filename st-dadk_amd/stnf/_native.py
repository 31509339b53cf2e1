"""ctypes binding of libstdadk.so (C ABI: include/stdadk.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing, or a
tensor is not a contiguous fp32 tensor on a HIP device, the call raises — loudly.
torch is used here only for device memory and the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STDADK_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libstdadk.so"))

MAX_HIDDEN = 8
BASIS_KIND = {"wendland": 0, "gaussian": 1, "triangular": 2}


class NativeLibraryError(RuntimeError):
    pass


class MlpDesc(C.Structure):
    _fields_ = [("n_hidden", C.c_int32), ("in_dim", C.c_int32), ("hidden", C.c_int32 * MAX_HIDDEN),
                ("out_dim", C.c_int32), ("layernorm", C.c_int32), ("ln_eps", C.c_float),
                ("dropout_p", C.c_float)]


class MlpTensors(C.Structure):
    _fields_ = [("W", C.c_void_p * (MAX_HIDDEN + 1)), ("b", C.c_void_p * (MAX_HIDDEN + 1)),
                ("ln_g", C.c_void_p * MAX_HIDDEN), ("ln_b", C.c_void_p * MAX_HIDDEN),
                ("W_bf16", C.c_void_p * (MAX_HIDDEN + 1)), ("WT_bf16", C.c_void_p * (MAX_HIDDEN + 1))]


class BF16Region(C.Structure):
    _fields_ = [("off", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32), ("dst", C.c_void_p),
                ("dst_t", C.c_void_p)]


class BF16Shadow(C.Structure):
    _fields_ = [("n", C.c_int32), ("r", BF16Region * MAX_HIDDEN)]


class BasisDesc(C.Structure):
    _fields_ = [("p", C.c_int32), ("basis", C.c_int32), ("n_levels", C.c_int32),
                ("side", C.c_int32 * 8), ("Ks", C.c_int64), ("Kt", C.c_int64),
                ("s_centers", C.c_void_p), ("s_bw", C.c_void_p), ("t_centers", C.c_void_p),
                ("t_bw", C.c_void_p)]


ABI_VERSION = 8            # STDADK_ABI_VERSION of include/stdadk.h this binding was written against
MAX_Q = 8
LOSS_MSE, LOSS_PINBALL = 0, 1


class LossDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("y_cols", C.c_int32), ("tau", C.c_float * MAX_Q),
                ("nc_weight", C.c_float), ("nc_power", C.c_int32)]


GRADSQ_PARTS = 512


class OptimDesc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("lr_dev", C.c_void_p), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("step_dev", C.c_void_p),
                ("max_norm", C.c_float), ("sumsq_parts", C.c_void_p), ("ema_decay", C.c_float),
                ("shadow", C.POINTER(BF16Shadow)), ("nonfinite_step", C.c_void_p)]


class AdamGroup(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("lr_dev", C.c_void_p), ("max_norm", C.c_float),
                ("sumsq_parts", C.c_void_p), ("n_parts", C.c_int32), ("shadow", C.POINTER(BF16Shadow))]


SPARSITY_KINDS = {"none": 0, "element": 1, "group": 2, "sparse_group": 3}


class SparsityDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lambda_l1", C.c_float), ("lambda_group", C.c_float),
                ("apply_spatial", C.c_int32), ("apply_temporal", C.c_int32)]


class KnotTrain(C.Structure):
    _fields_ = [("centers_init", C.c_void_p), ("gradient_damping", C.c_int32),
                ("damping_threshold", C.c_float), ("damping_strength", C.c_float),
                ("domain_weight", C.c_float), ("movement_weight", C.c_float),
                ("penalty_grad_scale", C.c_float), ("penalty_loss_scale", C.c_float)]


FLAG_DENSE = 1
FLAG_W0_T = 2
FLAG_LOG_BW = 4
FLAG_WINDOW = 8
FLAG_PREBINNED = 16
FLAG_BF16 = 32
FLAG_SCATTERED = 64
KNOT_CELLS = 32          # cells per axis of the knot cell lists (csrc/window.h)

_lib = None

_SIGNATURES = {
    "stdadk_abi_version": (C.c_int, []),
    "stdadk_last_error": (C.c_char_p, []),
    "stdadk_rbf_build_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_void_p]),
    "stdadk_mlp_workspace_bytes": (C.c_size_t, [C.POINTER(MlpDesc), C.c_int64]),
    "stdadk_mlp_forward_f32": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpTensors), C.c_void_p,
                                         C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_int32, C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p]),
    "stdadk_mlp_backward_f32": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpTensors),
                                          C.POINTER(MlpTensors), C.c_void_p, C.c_int64, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64,
                                          C.POINTER(C.c_void_p), C.c_void_p]),
    "stdadk_mse_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "stdadk_loss_f32": (C.c_int, [C.POINTER(LossDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "stdadk_delta_head_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "stdadk_delta_head_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                 C.c_int32, C.c_int32, C.c_float, C.c_float,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "stdadk_sparsity_f32": (C.c_int, [C.POINTER(SparsityDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "stdadk_spatial_partial_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.POINTER(MlpTensors),
                                             C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32,
                                             C.c_void_p]),
    "stdadk_temporal_partial_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.POINTER(MlpTensors),
                                              C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "stdadk_forward_parts_f32": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(MlpTensors), C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "stdadk_train_step_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.POINTER(MlpTensors),
                                        C.POINTER(MlpTensors), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_float, C.POINTER(LossDesc),
                                        C.POINTER(SparsityDesc), C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64,
                                        C.c_int32, C.POINTER(OptimDesc), C.c_void_p]),
    "stdadk_train_step_next_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.POINTER(MlpTensors),
                                             C.POINTER(MlpTensors), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int64, C.c_float, C.POINTER(LossDesc),
                                             C.POINTER(SparsityDesc), C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64,
                                             C.c_int32, C.POINTER(OptimDesc), C.c_void_p, C.c_int64, C.c_int32,
                                             C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.c_void_p]),
    "stdadk_sumsq2_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "stdadk_adamw_ema2_f32": (C.c_int, [C.POINTER(AdamGroup), C.POINTER(AdamGroup), C.c_float, C.c_float,
                                        C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_float, C.c_float,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "stdadk_step_uses_window": (C.c_int32, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.c_int32]),
    "stdadk_step_workspace_bytes": (C.c_size_t, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.c_int64,
                                                 C.c_int32]),
    "stdadk_forward_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.POINTER(MlpTensors),
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_int32, C.c_uint64, C.c_void_p,
                                     C.c_int32, C.c_void_p]),
    "stdadk_backward_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.POINTER(MlpTensors),
                                      C.POINTER(MlpTensors), C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_uint64, C.c_void_p, C.c_int32, C.c_void_p]),
    "stdadk_knot_backward_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.POINTER(MlpTensors),
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_int32,
                                           C.POINTER(KnotTrain), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "stdadk_knot_grad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "stdadk_knot_grad_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p]),
    "stdadk_train_fwd_bwd_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc),
                                           C.POINTER(MlpTensors), C.POINTER(MlpTensors), C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                           C.POINTER(LossDesc),
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64,
                                           C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "stdadk_train_fwd_bwd_indexed_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc),
                                                   C.POINTER(MlpTensors), C.POINTER(MlpTensors), C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                   C.c_float, C.POINTER(LossDesc), C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_int32,
                                                   C.c_void_p, C.c_void_p]),
    "stdadk_bin_batch_f32": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(MlpDesc), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                       C.c_size_t, C.c_int32, C.c_void_p]),
    "stdadk_gather_batch_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "stdadk_bin_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "stdadk_bin_obs_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "stdadk_knot_windows_i32": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "stdadk_profile_enable": (C.c_int, [C.c_int32]),
    "stdadk_profile_collect": (C.c_int64, [C.c_char_p, C.c_int64]),
    "stdadk_gemm_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "stdadk_gemm_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "stdadk_sumsq_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "stdadk_step_advance": (C.c_int, [C.c_void_p, C.c_void_p]),
    "stdadk_adamw_ema_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_float, C.c_void_p, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_float,
                                       C.c_void_p, C.c_int32, C.c_float, C.c_float, C.POINTER(BF16Shadow),
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "stdadk_bf16_shadow_refresh": (C.c_int, [C.c_void_p, C.POINTER(BF16Shadow), C.c_void_p]),
}


def lib():
    """Load libstdadk.so once; raise NativeLibraryError (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"libstdadk.so not found at {LIB_PATH}: build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (st-dadk_amd/csrc/build.sh). "
                f"This package has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError => the .so is stale
            fn.restype, fn.argtypes = res, args
        if handle.stdadk_abi_version() != ABI_VERSION:
            raise NativeLibraryError("libstdadk.so ABI version mismatch")
        _lib = handle
    return _lib


def exported_symbols():
    return list(_SIGNATURES)


def _check(rc, what):
    if rc != 0:
        msg = lib().stdadk_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def dry_run():
    """STDADK_DRY_RUN=1 (read by the library when it is loaded): every entry point validates and plans but launches
    NOTHING -- the hook of the host-side sanitizer pass (tools/build_asan.sh, tests/test_host_sanitizer.py), which runs
    without a GPU.  Host tensors are then accepted as stand-ins for device buffers (the host never dereferences
    them); whatever comes out is meaningless.  Not a CPU path: without the variable host tensors raise."""
    return _DRY_RUN


# read once, as the library does (api.cpp reads it when it is loaded): a lookup in os.environ per tensor argument was
# a fifth of the host time of a learnable-knot step (38 tensors per step)
_DRY_RUN = os.environ.get("STDADK_DRY_RUN", "") == "1"


def _on_device(tensor):
    return tensor.is_cuda or _DRY_RUN


def _dev(tensor, name):
    """Validate a device tensor and return its address."""
    if tensor is None:
        return None
    if _DRY_RUN and not tensor.is_cuda:
        if not tensor.is_contiguous():
            raise RuntimeError(f"{name}: tensor must be contiguous")
        return tensor.data_ptr()
    if not tensor.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on a HIP device (cuda:N), got {tensor.device}; "
                           f"the MI355X build of stnf has no CPU path")
    if tensor.dtype not in (torch.float32, torch.uint8, torch.int32, torch.bfloat16):
        raise RuntimeError(f"{name}: unsupported dtype {tensor.dtype}")
    if not tensor.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous")
    if tensor.device.index != _current_device():
        # the launch stream is the CURRENT device's current stream (_stream below)
        raise RuntimeError(f"{name}: tensor lives on {tensor.device} but the current device is cuda:"
                           f"{torch.cuda.current_device()}; call under torch.cuda.device({tensor.device.index})")
    return tensor.data_ptr()


# the raw accessors of torch (a tensor on a HIP device exists, so the runtime is initialised): current_device() and
# current_stream() go through lazy initialisation checks and build a Stream object per call -- per tensor argument
# and per native call that is tens of microseconds of a 0.11 ms step
_current_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream of the current device (every tensor of a call is checked to live there)."""
    if _DRY_RUN and not torch.cuda.is_available():
        return None
    if _raw_stream is not None:
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


# --------------------------------------------------------------------------------------------
def rbf_build(coords, t, X, s_centers, s_bw, basis, t_centers, t_bw, out):
    """out[b,:] = [X | phi | psi]; `out` is (B, ld) fp32 with ld >= p+Ks+Kt.  Any of the three
    column groups may be absent (X None/p==0, s_centers None, t_centers None)."""
    B = out.shape[0]
    ld = out.stride(0) if out.dim() == 2 else 0
    if out.dim() != 2 or out.stride(1) != 1:
        raise RuntimeError("rbf_build: out must be 2-D with unit column stride")
    p = 0 if X is None or X.numel() == 0 else X.shape[1]
    Ks = 0 if s_centers is None else s_centers.shape[0]
    Kt = 0 if t_centers is None else t_centers.shape[0]
    if Ks and (coords.shape[0] != B or coords.shape[1] != 2):
        raise RuntimeError(f"rbf_build: coords must be ({B}, 2), got {tuple(coords.shape)}")
    if Kt and t.numel() != B:
        raise RuntimeError(f"rbf_build: t must have {B} elements, got {t.numel()}")
    if p and X.shape[0] != B:
        raise RuntimeError("rbf_build: X row count mismatch")
    if not _on_device(out):
        raise RuntimeError("rbf_build: expected tensors on a HIP device; this build has no CPU path")
    rc = lib().stdadk_rbf_build_f32(
        _dev(coords, "coords") if Ks else None, _dev(t, "t") if Kt else None,
        _dev(X, "X") if p else None, B, p,
        _dev(s_centers, "s_centers") if Ks else None, _dev(s_bw, "s_bw") if Ks else None, Ks,
        BASIS_KIND[basis], _dev(t_centers, "t_centers") if Kt else None,
        _dev(t_bw, "t_bw") if Kt else None, Kt, out.data_ptr(), ld, _stream())
    _check(rc, "stdadk_rbf_build_f32")
    return out


def make_desc(in_dim, hidden, out_dim, layernorm, dropout_p, ln_eps=1e-5):
    if len(hidden) > MAX_HIDDEN:
        raise RuntimeError(f"at most {MAX_HIDDEN} hidden layers are supported")
    d = MlpDesc()
    d.n_hidden, d.in_dim, d.out_dim = len(hidden), in_dim, out_dim
    for i, h in enumerate(hidden):
        d.hidden[i] = h
    d.layernorm, d.ln_eps, d.dropout_p = int(bool(layernorm)), ln_eps, float(dropout_p)
    return d


def make_tensors(Ws, bs, gs, betas, bf16=None):
    """Pack per-layer tensors (lists; gs/betas may be None) into the ABI struct.  `bf16`: per Linear index
    None or the pair (W_bf16 (out,in), WT_bf16 (in,out)) of torch.bfloat16 operand copies (FLAG_BF16)."""
    s = MlpTensors()
    for i, (w, b) in enumerate(zip(Ws, bs)):
        s.W[i], s.b[i] = _dev(w, f"W[{i}]"), _dev(b, f"b[{i}]")
    if gs is not None:
        for i, (g, be) in enumerate(zip(gs, betas)):
            s.ln_g[i], s.ln_b[i] = _dev(g, f"ln_g[{i}]"), _dev(be, f"ln_b[{i}]")
    if bf16 is not None:
        for i, pair in enumerate(bf16):
            if pair is not None:
                if pair[0].dtype != torch.bfloat16 or pair[1].dtype != torch.bfloat16:
                    raise RuntimeError("make_tensors: the operand copies must be torch.bfloat16")
                s.W_bf16[i], s.WT_bf16[i] = _dev(pair[0], f"W_bf16[{i}]"), _dev(pair[1], f"WT_bf16[{i}]")
    return s


def make_bf16_shadow(regions):
    """stdadk_bf16_shadow from [(element offset into the flat fp32 buffer, rows, cols, dst, dst_t)]."""
    if len(regions) > MAX_HIDDEN:
        raise RuntimeError(f"at most {MAX_HIDDEN} bf16 shadow regions")
    sh = BF16Shadow()
    sh.n = len(regions)
    for i, (off, rows, cols, dst, dst_t) in enumerate(regions):
        sh.r[i].off, sh.r[i].rows, sh.r[i].cols = int(off), int(rows), int(cols)
        sh.r[i].dst, sh.r[i].dst_t = _dev(dst, "shadow dst"), _dev(dst_t, "shadow dst_t")
    return sh


def bf16_shadow_refresh(p, shadow):
    """Rewrite every region's bf16 copies from the fp32 buffer `p` (stdadk_bf16_shadow_refresh)."""
    rc = lib().stdadk_bf16_shadow_refresh(_dev(p, "p"), C.byref(shadow), _stream())
    _check(rc, "stdadk_bf16_shadow_refresh")


def mlp_workspace_bytes(desc, B):
    n = lib().stdadk_mlp_workspace_bytes(C.byref(desc), B)
    if n == 0:
        raise RuntimeError("stdadk_mlp_workspace_bytes: invalid descriptor")
    return n


def _mask_array(masks):
    if masks is None:
        return None
    arr = (C.c_void_p * MAX_HIDDEN)()
    for i, m in enumerate(masks):
        arr[i] = _dev(m, f"drop_mask[{i}]") if m is not None else None
    return arr


def mlp_forward(desc, params, features, B, y_pred, workspace, training, seed=0, masks=None):
    rc = lib().stdadk_mlp_forward_f32(C.byref(desc), C.byref(params), _dev(features, "features"),
                                      features.stride(0), B, _dev(y_pred, "y_pred"),
                                      workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                      int(training), seed, _mask_array(masks), _stream())
    _check(rc, "stdadk_mlp_forward_f32")


def mlp_backward(desc, params, grads, features, B, dY, workspace, seed=0, masks=None):
    rc = lib().stdadk_mlp_backward_f32(C.byref(desc), C.byref(params), C.byref(grads),
                                       _dev(features, "features"), features.stride(0), B,
                                       _dev(dY, "dY"), workspace.data_ptr(),
                                       workspace.numel() * workspace.element_size(), seed,
                                       _mask_array(masks), _stream())
    _check(rc, "stdadk_mlp_backward_f32")


def gemm(A, a_km, Bm, b_km, M, N, K, bias=None, out=None, workspace=None):
    """C[M,N] = Aop Bop (+bias) with the MLP's fp32 MFMA GEMM (layouts: include/stdadk.h)."""
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    need = lib().stdadk_gemm_workspace_bytes(M, N, K)
    if workspace is None and need:
        workspace = torch.empty(need // 4, device=A.device, dtype=torch.float32)
    rc = lib().stdadk_gemm_f32(_dev(A, "A"), A.stride(0), int(a_km), _dev(Bm, "B"), Bm.stride(0),
                               int(b_km), M, N, K, _dev(bias, "bias"), _dev(out, "C"), out.stride(0),
                               workspace.data_ptr() if workspace is not None else None,
                               workspace.numel() * 4 if workspace is not None else 0, _stream())
    _check(rc, "stdadk_gemm_f32")
    return out


def make_basis(p, basis, sides, s_centers, s_bw, t_centers, t_bw):
    """ABI descriptor of the knot tables; `sides` = level side lengths of a uniform grid, or (with FLAG_SCATTERED)
    the knot COUNT of every level of a scattered table, or None."""
    b = BasisDesc()
    b.p, b.basis = int(p), BASIS_KIND[basis]
    b.n_levels = len(sides) if sides else 0
    if b.n_levels > 8:
        b.n_levels = 0          # more levels than the window path tracks: dense path only
    for i in range(b.n_levels):
        b.side[i] = int(sides[i])
    b.Ks, b.Kt = s_centers.shape[0], t_centers.shape[0]
    b.s_centers, b.s_bw = _dev(s_centers, "s_centers"), _dev(s_bw, "s_bw")
    b.t_centers, b.t_bw = _dev(t_centers, "t_centers"), _dev(t_bw, "t_bw")
    return b


def step_uses_window(basis, desc, flags):
    return bool(lib().stdadk_step_uses_window(C.byref(basis), C.byref(desc), flags))


def step_workspace_bytes(basis, desc, B, flags):
    n = lib().stdadk_step_workspace_bytes(C.byref(basis), C.byref(desc), B, flags)
    if n == 0:
        raise RuntimeError("stdadk_step_workspace_bytes: invalid descriptors")
    return n


def forward(basis, desc, params, coords, t, X, B, y_pred, workspace, flags, training=False, seed=0,
            step_dev=None):
    rc = lib().stdadk_forward_f32(C.byref(basis), C.byref(desc), C.byref(params), _dev(coords, "coords"),
                                  _dev(t, "t"), _dev(X, "X"), B, _dev(y_pred, "y_pred"),
                                  workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                  int(training), seed, _dev(step_dev, "step_dev"), flags, _stream())
    _check(rc, "stdadk_forward_f32")


def backward(basis, desc, params, grads, B, dY, workspace, flags, seed=0, step_dev=None):
    rc = lib().stdadk_backward_f32(C.byref(basis), C.byref(desc), C.byref(params), C.byref(grads), B,
                                   _dev(dY, "dY"), workspace.data_ptr(),
                                   workspace.numel() * workspace.element_size(), seed,
                                   _dev(step_dev, "step_dev"), flags, _stream())
    _check(rc, "stdadk_backward_f32")


def make_loss(kind="mse", Q=1, y_cols=None, taus=None, nc_weight=0.0, nc_power=1):
    """stdadk_loss_desc for an objective on (B,Q) predictions: kind "mse" or "pinball" (check loss
    with levels `taus`, optional prediction-level non-crossing penalty)."""
    if kind not in ("mse", "pinball"):
        raise ValueError(f"unknown loss kind '{kind}'; use 'mse' or 'pinball'")
    if Q < 1 or Q > MAX_Q:
        raise ValueError(f"Q={Q} outside 1..{MAX_Q}")
    ld = LossDesc()
    ld.kind = LOSS_MSE if kind == "mse" else LOSS_PINBALL
    ld.y_cols = Q if y_cols is None else int(y_cols)
    if kind == "pinball":
        taus = list(taus if taus is not None else [])
        if len(taus) != Q:
            raise ValueError(f"pinball loss needs {Q} quantile levels, got {len(taus)}")
        for i, q in enumerate(taus):
            ld.tau[i] = float(q)
        if nc_weight and int(nc_power) not in (1, 2):
            raise ValueError(f"Unsupported power={nc_power}; use 1 or 2.")
        ld.nc_weight = float(nc_weight)
        ld.nc_power = int(nc_power)
    return ld


def loss(desc, y_pred, y, grad_scale, dY=None, loss_sum=None):
    """stdadk_loss_f32 on y_pred (B,Q), y (B,y_cols); desc None = MSE."""
    B, Q = y_pred.shape
    rc = lib().stdadk_loss_f32(C.byref(desc) if desc is not None else None, _dev(y_pred, "y_pred"),
                               _dev(y, "y"), B, Q, grad_scale, _dev(dY, "dY"), _dev(loss_sum, "loss_sum"),
                               _stream())
    _check(rc, "stdadk_loss_f32")


def _rows(tensor, name, cols):
    """Address + row stride of a (Q, cols) fp32 device matrix whose rows are contiguous."""
    if not _on_device(tensor) or tensor.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected a float32 tensor on a HIP device, got {tensor.dtype} on "
                           f"{tensor.device}; the MI355X build of stnf has no CPU path")
    if tensor.dim() != 2 or tensor.shape[1] != cols or (cols > 1 and tensor.stride(1) != 1):
        raise RuntimeError(f"{name}: expected shape (Q, {cols}) with contiguous rows, got {tuple(tensor.shape)}")
    return tensor.data_ptr(), tensor.stride(0)


def delta_head(delta, Wo, bo):
    """delta (Q, d+1) rows (any row stride) -> Wo (Q,d), bo (Q): cumulative sums over quantiles."""
    Q, d = Wo.shape
    ptr, ldd = _rows(delta, "delta", d + 1)
    rc = lib().stdadk_delta_head_f32(ptr, ldd, Q, d, _dev(Wo, "Wo"), _dev(bo, "bo"), _stream())
    _check(rc, "stdadk_delta_head_f32")


def delta_head_backward(delta, dWo, dbo, lambda_grad, lambda_loss, d_delta, loss_sum=None):
    Q, d = dWo.shape
    ptr, ldd = _rows(delta, "delta", d + 1)
    gptr, ldg = _rows(d_delta, "d_delta", d + 1)
    if ldg != ldd:
        raise RuntimeError("delta_head_backward: delta and d_delta must share the row stride")
    rc = lib().stdadk_delta_head_backward_f32(ptr, _dev(dWo, "dWo"), _dev(dbo, "dbo"), ldd, Q, d,
                                              lambda_grad, lambda_loss, gptr, _dev(loss_sum, "loss_sum"),
                                              _stream())
    _check(rc, "stdadk_delta_head_backward_f32")


def spatial_partial(basis, desc, params, coords, out, workspace, flags):
    """stdadk_spatial_partial_f32: out (S, h0) = the per-site half of layer 0's pre-activation."""
    S = coords.shape[0]
    if tuple(out.shape) != (S, desc.hidden[0]):
        raise RuntimeError(f"spatial_partial: out has shape {tuple(out.shape)}")
    rc = lib().stdadk_spatial_partial_f32(C.byref(basis), C.byref(desc), C.byref(params), _dev(coords, "coords"), S,
                                          _dev(out, "out"), workspace.data_ptr(),
                                          workspace.numel() * workspace.element_size(), flags, _stream())
    _check(rc, "stdadk_spatial_partial_f32")


def temporal_partial(basis, desc, params, t_values, out, flags):
    """stdadk_temporal_partial_f32: out (T, h0) = the per-time half of layer 0's pre-activation."""
    T = t_values.numel()
    if tuple(out.shape) != (T, desc.hidden[0]):
        raise RuntimeError(f"temporal_partial: out has shape {tuple(out.shape)}")
    rc = lib().stdadk_temporal_partial_f32(C.byref(basis), C.byref(desc), C.byref(params), _dev(t_values, "t_values"),
                                           T, _dev(out, "out"), flags, _stream())
    _check(rc, "stdadk_temporal_partial_f32")


def forward_parts(desc, params, sp, tp, y_pred):
    """stdadk_forward_parts_f32: y_pred (T*S, Q), rows time-major, from the two halves of layer 0."""
    S, T = sp.shape[0], tp.shape[0]
    if y_pred.shape[0] != S * T:
        raise RuntimeError("forward_parts: y_pred must have T*S rows")
    rc = lib().stdadk_forward_parts_f32(C.byref(desc), C.byref(params), _dev(sp, "sp"), S, _dev(tp, "tp"), T,
                                        _dev(y_pred, "y_pred"), _stream())
    _check(rc, "stdadk_forward_parts_f32")


def make_sparsity(kind, lambda_l1=0.01, lambda_group=0.01, apply_spatial=True, apply_temporal=True):
    if kind not in SPARSITY_KINDS:
        raise ValueError(f"Unknown penalty_type: {kind}")
    s = SparsityDesc()
    s.kind, s.lambda_l1, s.lambda_group = SPARSITY_KINDS[kind], float(lambda_l1), float(lambda_group)
    s.apply_spatial, s.apply_temporal = int(bool(apply_spatial)), int(bool(apply_temporal))
    return s


def sparsity(sp, W0, dW0, w0_t, p, Ks, Kt, grad_scale=1.0, loss_scale=0.0, loss_sum=None, penalties=None):
    """stdadk_sparsity_f32: W0 / dW0 are (D, H0) with w0_t, else (H0, D); penalties (2,) accumulates
    (spatial, temporal)."""
    D = p + Ks + Kt
    H0 = W0.shape[1] if w0_t else W0.shape[0]
    if tuple(W0.shape) != ((D, H0) if w0_t else (H0, D)):
        raise RuntimeError(f"sparsity: W0 has shape {tuple(W0.shape)}, D = {D}")
    if dW0 is not None and (dW0.shape != W0.shape or dW0.stride() != W0.stride()):
        raise RuntimeError("sparsity: dW0 must have the shape and strides of W0")
    if penalties is not None and penalties.numel() != 2:
        raise RuntimeError("sparsity: penalties must hold 2 floats")
    rc = lib().stdadk_sparsity_f32(C.byref(sp), _dev(W0, "W0"), _dev(dW0, "dW0"), W0.stride(0), int(bool(w0_t)),
                                   H0, p, Ks, Kt, grad_scale, loss_scale, _dev(loss_sum, "loss_sum"),
                                   _dev(penalties, "penalties"), _stream())
    _check(rc, "stdadk_sparsity_f32")


def train_fwd_bwd(basis, desc, params, grads, coords, t, X, y, B, grad_scale, loss_sum, y_pred,
                  workspace, flags, seed=0, step_dev=None, aux_stream=None, loss_desc=None):
    rc = lib().stdadk_train_fwd_bwd_f32(C.byref(basis), C.byref(desc), C.byref(params), C.byref(grads),
                                        _dev(coords, "coords"), _dev(t, "t"), _dev(X, "X"), _dev(y, "y"),
                                        B, grad_scale,
                                        C.byref(loss_desc) if loss_desc is not None else None,
                                        _dev(loss_sum, "loss_sum"), _dev(y_pred, "y_pred"),
                                        workspace.data_ptr(),
                                        workspace.numel() * workspace.element_size(), seed,
                                        _dev(step_dev, "step_dev"), flags, _stream(),
                                        aux_stream.cuda_stream if aux_stream is not None else None)
    _check(rc, "stdadk_train_fwd_bwd_f32")


def train_fwd_bwd_indexed(basis, desc, params, grads, coords_all, t_all, X_all, y_all, idx, grad_scale,
                          loss_sum, y_pred, workspace, flags, seed=0, step_dev=None, aux_stream=None,
                          loss_desc=None):
    """The fused step on rows `idx` (contiguous int64 device tensor) of the resident arrays (window path)."""
    if idx.dtype != torch.int64 or not _on_device(idx) or not idx.is_contiguous():
        raise RuntimeError("train_fwd_bwd_indexed: idx must be a contiguous int64 tensor on the device")
    rc = lib().stdadk_train_fwd_bwd_indexed_f32(
        C.byref(basis), C.byref(desc), C.byref(params), C.byref(grads), _dev(coords_all, "coords"),
        _dev(t_all, "t"), _dev(X_all, "X"), _dev(y_all, "y"), idx.data_ptr(), idx.numel(), grad_scale,
        C.byref(loss_desc) if loss_desc is not None else None, _dev(loss_sum, "loss_sum"),
        _dev(y_pred, "y_pred"), workspace.data_ptr(), workspace.numel() * workspace.element_size(), seed,
        _dev(step_dev, "step_dev"), flags, _stream(),
        aux_stream.cuda_stream if aux_stream is not None else None)
    _check(rc, "stdadk_train_fwd_bwd_indexed_f32")


def bin_batch(basis, desc, coords_all, t_all, X_all, y_all, idx, workspace, flags):
    """Batch preparation of the window path on its own (rows `idx` of the resident arrays binned into
    `workspace`) on the current stream; the step then runs with FLAG_PREBINNED."""
    if idx.dtype != torch.int64 or not _on_device(idx) or not idx.is_contiguous():
        raise RuntimeError("bin_batch: idx must be a contiguous int64 tensor on the device")
    rc = lib().stdadk_bin_batch_f32(C.byref(basis), C.byref(desc), _dev(coords_all, "coords"), _dev(t_all, "t"),
                                    _dev(X_all, "X"), _dev(y_all, "y"), idx.data_ptr(), idx.numel(),
                                    y_all.shape[1] if y_all is not None else 0, workspace.data_ptr(),
                                    workspace.numel() * workspace.element_size(), flags, _stream())
    _check(rc, "stdadk_bin_batch_f32")


def make_knot_train(centers_init, gradient_damping=False, damping_threshold=0.3, damping_strength=1.0,
                    domain_weight=0.0, movement_weight=0.0, penalty_grad_scale=1.0, penalty_loss_scale=0.0):
    kt = KnotTrain()
    kt.centers_init = _dev(centers_init, "centers_init")
    kt.gradient_damping = 1 if gradient_damping else 0
    kt.damping_threshold, kt.damping_strength = float(damping_threshold), float(damping_strength)
    kt.domain_weight, kt.movement_weight = float(domain_weight), float(movement_weight)
    kt.penalty_grad_scale, kt.penalty_loss_scale = float(penalty_grad_scale), float(penalty_loss_scale)
    return kt


def knot_backward(basis, desc, params, coords, B, workspace, flags, knot_train, d_centers, d_log_bw,
                  loss_sum=None):
    """Gradients w.r.t. the learnable knots for the batch whose backward just ran on `workspace`."""
    rc = lib().stdadk_knot_backward_f32(C.byref(basis), C.byref(desc), C.byref(params), _dev(coords, "coords"),
                                        B, workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                        flags, C.byref(knot_train) if knot_train is not None else None,
                                        _dev(d_centers, "d_centers"), _dev(d_log_bw, "d_log_bw"),
                                        _dev(loss_sum, "loss_sum"), _stream())
    _check(rc, "stdadk_knot_backward_f32")


def knot_grad(coords, d_phi, centers, log_bw, basis, d_centers, d_log_bw):
    """stdadk_knot_grad_f32: dL/dphi (B, Ks) -> gradients w.r.t. the centres (Ks,2) and log-bandwidths (Ks,)."""
    B, Ks = d_phi.shape
    if d_phi.stride(1) != 1:
        raise RuntimeError("knot_grad: d_phi must have unit column stride")
    need = lib().stdadk_knot_grad_workspace_bytes(B, Ks)
    ws = torch.empty(max(need // 4, 1), device=d_phi.device, dtype=torch.float32)
    rc = lib().stdadk_knot_grad_f32(_dev(coords, "coords"), B, d_phi.data_ptr(), d_phi.stride(0),
                                    _dev(centers, "centers"), _dev(log_bw, "log_bw"), Ks, BASIS_KIND[basis],
                                    _dev(d_centers, "d_centers"), _dev(d_log_bw, "d_log_bw"), ws.data_ptr(),
                                    ws.numel() * 4, _stream())
    _check(rc, "stdadk_knot_grad_f32")


def gather_batch(coords, t, y, X, idx, coords_out, t_out, y_out, X_out):
    """Rows idx (int64 device tensor) of the resident observation arrays -> contiguous batch buffers."""
    if idx.dtype != torch.int64 or not _on_device(idx) or not idx.is_contiguous():
        raise RuntimeError("gather_batch: idx must be a contiguous int64 tensor on the device")
    B = idx.numel()
    Q = y.shape[1] if y is not None else 0
    p = X.shape[1] if X is not None else 0
    rc = lib().stdadk_gather_batch_f32(_dev(coords, "coords"), _dev(t, "t"), _dev(y, "y"), _dev(X, "X"),
                                       idx.data_ptr(), B, Q, p, _dev(coords_out, "coords_out"),
                                       _dev(t_out, "t_out"), _dev(y_out, "y_out"), _dev(X_out, "X_out"),
                                       _stream())
    _check(rc, "stdadk_gather_batch_f32")


def bin_obs(coords, G):
    """(keys[B], cell_start[G*G+1], perm[B]) int32 device tensors of the window path's binning."""
    B = coords.shape[0]
    dev = coords.device
    keys = torch.empty(B, dtype=torch.int32, device=dev)
    cell_start = torch.empty(G * G + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty(max(lib().stdadk_bin_workspace_bytes(B, G) // 4, 1), dtype=torch.float32, device=dev)
    rc = lib().stdadk_bin_obs_f32(_dev(coords, "coords"), B, G, keys.data_ptr(), cell_start.data_ptr(),
                                  perm.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream())
    _check(rc, "stdadk_bin_obs_f32")
    return keys, cell_start, perm


def knot_windows(coords, sides, p):
    B, L = coords.shape[0], len(sides)
    dev = coords.device
    out = [torch.empty(B, L, dtype=torch.int32, device=dev) for _ in range(3)]
    arr = (C.c_int32 * L)(*sides)
    rc = lib().stdadk_knot_windows_i32(_dev(coords, "coords"), B, arr, L, p, out[0].data_ptr(),
                                       out[1].data_ptr(), out[2].data_ptr(), _stream())
    _check(rc, "stdadk_knot_windows_i32")
    return out


def mse(y_pred, y, grad_scale, dY=None, loss_sum=None):
    rc = lib().stdadk_mse_f32(_dev(y_pred, "y_pred"), _dev(y, "y"), y_pred.numel(), grad_scale,
                              _dev(dY, "dY"), _dev(loss_sum, "loss_sum"), _stream())
    _check(rc, "stdadk_mse_f32")


SUMSQ_PARTS = 256


def sumsq(g, parts, step_inc=None):
    """parts[0:256] = partial sums of g^2 (`parts` is a 256-float slice, fully overwritten);
    step_inc (device int32) is advanced by one when given."""
    if parts.numel() != SUMSQ_PARTS:
        raise RuntimeError(f"sumsq: parts must hold {SUMSQ_PARTS} floats")
    rc = lib().stdadk_sumsq_f32(_dev(g, "g"), g.numel(), _dev(parts, "parts"), _dev(step_inc, "step_inc"),
                                _stream())
    _check(rc, "stdadk_sumsq_f32")


def step_advance(step_dev):
    _check(lib().stdadk_step_advance(_dev(step_dev, "step_dev"), _stream()), "stdadk_step_advance")


def adamw_ema(p, g, m, v, ema, lr, betas, eps, weight_decay, step, max_norm=0.0, sumsq_parts=None,
              grad_mul=1.0, ema_decay=0.0, lr_dev=None, step_dev=None, shadow=None, loss_watch=None,
              nonfinite_step=None):
    """`loss_watch` / `nonfinite_step`: the non-finite guard of include/stdadk.h (both or neither)."""
    rc = lib().stdadk_adamw_ema_f32(_dev(p, "p"), _dev(g, "g"), _dev(m, "m"), _dev(v, "v"),
                                    _dev(ema, "ema"), p.numel(), lr, _dev(lr_dev, "lr_dev"),
                                    betas[0], betas[1], eps, weight_decay, int(step),
                                    _dev(step_dev, "step_dev"), max_norm, _dev(sumsq_parts, "sumsq"),
                                    0 if sumsq_parts is None else sumsq_parts.numel(),
                                    grad_mul, ema_decay, C.byref(shadow) if shadow is not None else None,
                                    _dev(loss_watch, "loss_watch"), _dev(nonfinite_step, "nonfinite_step"), _stream())
    _check(rc, "stdadk_adamw_ema_f32")


def make_optim(p, g, m, v, ema, lr, lr_dev, betas, eps, weight_decay, step_dev, max_norm, sumsq_parts, ema_decay,
               shadow=None, nonfinite_step=None):
    o = OptimDesc()
    o.nonfinite_step = _dev(nonfinite_step, "nonfinite_step")
    o._keep = shadow                 # the descriptor points at it
    o.shadow = C.pointer(shadow) if shadow is not None else None
    o.p, o.g, o.m, o.v, o.ema = _dev(p, "p"), _dev(g, "g"), _dev(m, "m"), _dev(v, "v"), _dev(ema, "ema")
    o.n = p.numel()
    o.lr, o.lr_dev = float(lr), _dev(lr_dev, "lr_dev")
    o.beta1, o.beta2, o.eps, o.weight_decay = float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
    o.step_dev = _dev(step_dev, "step_dev")
    o.max_norm = float(max_norm)
    if sumsq_parts is not None and sumsq_parts.numel() != GRADSQ_PARTS:
        raise RuntimeError(f"make_optim: sumsq_parts must hold {GRADSQ_PARTS} floats")
    o.sumsq_parts = _dev(sumsq_parts, "sumsq_parts")
    o.ema_decay = float(ema_decay)
    return o


def train_step(basis, desc, params, grads, coords, t, X, y, idx, B, grad_scale, loss_sum, workspace, flags, optim,
               seed=0, loss_desc=None, sparsity_desc=None):
    """The whole single-GPU step in one call: forward, objective (+ first-layer sparsity penalties), backward,
    clip + AdamW + EMA."""
    if idx is not None and (idx.dtype != torch.int64 or not _on_device(idx) or not idx.is_contiguous()):
        raise RuntimeError("train_step: idx must be a contiguous int64 tensor on the device")
    rc = lib().stdadk_train_step_f32(
        C.byref(basis), C.byref(desc), C.byref(params), C.byref(grads), _dev(coords, "coords"), _dev(t, "t"),
        _dev(X, "X"), _dev(y, "y"), idx.data_ptr() if idx is not None else None, B, grad_scale,
        C.byref(loss_desc) if loss_desc is not None else None,
        C.byref(sparsity_desc) if sparsity_desc is not None else None, _dev(loss_sum, "loss_sum"), workspace.data_ptr(),
        workspace.numel() * workspace.element_size(), seed, flags, C.byref(optim), _stream())
    _check(rc, "stdadk_train_step_f32")


def train_step_next(basis, desc, params, grads, coords_all, t_all, X_all, y_all, idx, grad_scale, loss_sum, workspace,
                    flags, optim, next_idx, next_workspace, seed=0, loss_desc=None, sparsity_desc=None):
    """train_step on rows `idx` of the resident arrays whose optimiser launch also bins rows `next_idx` into
    `next_workspace` (stdadk_train_step_next_f32).  Returns True when the next batch was binned (step on
    next_workspace with FLAG_PREBINNED then), False when nothing was prepared."""
    for nm, ix in (("idx", idx), ("next_idx", next_idx)):
        if ix.dtype != torch.int64 or not _on_device(ix) or not ix.is_contiguous():
            raise RuntimeError(f"train_step_next: {nm} must be a contiguous int64 tensor on the device")
    if next_workspace.data_ptr() == workspace.data_ptr():
        raise RuntimeError("train_step_next: the next batch needs a workspace of its own")
    done = C.c_int32(0)
    rc = lib().stdadk_train_step_next_f32(
        C.byref(basis), C.byref(desc), C.byref(params), C.byref(grads), _dev(coords_all, "coords"), _dev(t_all, "t"),
        _dev(X_all, "X"), _dev(y_all, "y"), idx.data_ptr(), idx.numel(), grad_scale,
        C.byref(loss_desc) if loss_desc is not None else None,
        C.byref(sparsity_desc) if sparsity_desc is not None else None, _dev(loss_sum, "loss_sum"), workspace.data_ptr(),
        workspace.numel() * workspace.element_size(), seed, flags, C.byref(optim), next_idx.data_ptr(), next_idx.numel(),
        y_all.shape[1] if y_all is not None else 0, next_workspace.data_ptr(),
        next_workspace.numel() * next_workspace.element_size(), C.byref(done), _stream())
    _check(rc, "stdadk_train_step_next_f32")
    return bool(done.value)


def sumsq2(g0, parts0, g1, parts1, step_inc=None):
    """stdadk_sumsq_f32 on two gradient buffers in one launch (step_inc advanced once)."""
    if parts0.numel() != SUMSQ_PARTS or parts1.numel() != SUMSQ_PARTS:
        raise RuntimeError(f"sumsq2: parts must hold {SUMSQ_PARTS} floats each")
    rc = lib().stdadk_sumsq2_f32(_dev(g0, "g0"), g0.numel(), _dev(parts0, "parts0"), _dev(g1, "g1"), g1.numel(),
                                 _dev(parts1, "parts1"), _dev(step_inc, "step_inc"), _stream())
    _check(rc, "stdadk_sumsq2_f32")


def make_adam_group(p, g, m, v, ema, lr, lr_dev=None, max_norm=0.0, sumsq_parts=None, shadow=None):
    gr = AdamGroup()
    gr._keep = shadow
    gr.shadow = C.pointer(shadow) if shadow is not None else None
    gr.p, gr.g, gr.m, gr.v, gr.ema = _dev(p, "p"), _dev(g, "g"), _dev(m, "m"), _dev(v, "v"), _dev(ema, "ema")
    gr.n = p.numel()
    gr.lr, gr.lr_dev = float(lr), _dev(lr_dev, "lr_dev")
    gr.max_norm = float(max_norm)
    gr.sumsq_parts = _dev(sumsq_parts, "sumsq")
    gr.n_parts = 0 if sumsq_parts is None else sumsq_parts.numel()
    return gr


def adamw_ema2(group0, group1, betas, eps, weight_decay, step, grad_mul=1.0, ema_decay=0.0, step_dev=None,
               loss_watch=None, nonfinite_step=None):
    """stdadk_adamw_ema_f32 on two parameter groups (make_adam_group) in one launch."""
    rc = lib().stdadk_adamw_ema2_f32(C.byref(group0), C.byref(group1), betas[0], betas[1], eps, weight_decay,
                                     int(step), _dev(step_dev, "step_dev"), grad_mul, ema_decay,
                                     _dev(loss_watch, "loss_watch"), _dev(nonfinite_step, "nonfinite_step"), _stream())
    _check(rc, "stdadk_adamw_ema2_f32")


def profile_enable(on=True):
    """Bracket every kernel launch of the library with HIP events (measurement aid)."""
    lib().stdadk_profile_enable(int(on))


def profile_collect():
    """[(kernel name, milliseconds)] per launch since profile_enable(True), in launch order."""
    need = lib().stdadk_profile_collect(None, 0)
    if need < 0:
        raise RuntimeError("stdadk_profile_collect failed")
    buf = C.create_string_buffer(int(need) + 16)
    lib().stdadk_profile_collect(buf, len(buf))
    out = []
    for line in buf.value.decode().splitlines():
        name, ms = line.rsplit("\t", 1)
        out.append((name, float(ms)))
    return out
