"""ctypes binding of the C ABI declared in include/bayesrul_amd.h.

The product path has NO CPU fallback: if the HIP library is missing or does not load, or
no gfx950 device is present, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbayesrul_amd.so")

NET_INCEPTION, NET_LINEAR = 0, 1
MODE_NORMAL, MODE_LRT, MODE_FLIPOUT, MODE_RADIAL = 0, 1, 2, 3
PREC_F32, PREC_BF16X3 = 0, 1
T_ACT1, T_MID, T_ACT2, T_H, T_Z, T_H2, T_H3, T_H4 = range(8)

EXPORTS = [
    "bnn_version", "bnn_last_error", "bnn_abi_sizeof", "bnn_plan_create", "bnn_plan_destroy",
    "bnn_plan_num_params", "bnn_plan_num_sites", "bnn_plan_num_layers", "bnn_plan_workspace_bytes",
    "bnn_plan_bind", "bnn_plan_site", "bnn_plan_layer", "bnn_plan_tensor", "bnn_sample_weights",
    "bnn_forward", "bnn_head_nll", "bnn_backward", "bnn_grad_finalize", "bnn_clipped_adam",
    "bnn_elbo_step", "bnn_elbo_evaluate", "bnn_predict", "bnn_export_noise", "bnn_profile_enable",
    "bnn_profile_select", "bnn_profile_name", "bnn_profile_read", "bnn_gather_windows", "bnn_det_step", "bnn_plan_validate",
    "bnn_det_forward",
]


class PlanDesc(C.Structure):
    _fields_ = [("net", C.c_int32), ("mode", C.c_int32), ("prec", C.c_int32), ("max_particles", C.c_int32),
                ("max_batch", C.c_int32), ("win_length", C.c_int32), ("n_features", C.c_int32),
                ("max_windows", C.c_int32)]


class Buffers(C.Structure):
    _fields_ = [("mu", C.c_void_p), ("rho", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
                ("grad", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class Noise(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("step", C.c_uint64), ("eps_w", C.c_void_p), ("radial_r", C.c_void_p),
                ("lrt_eps", C.POINTER(C.c_void_p)), ("sign_in", C.POINTER(C.c_void_p)),
                ("sign_out", C.POINTER(C.c_void_p))]


class ElboArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("batch", C.c_int32), ("particles", C.c_int32),
                ("global_batch_offset", C.c_int32), ("global_batch", C.c_int32), ("dataset_size", C.c_double),
                ("prior_loc", C.c_double), ("prior_scale", C.c_double), ("mode_override", C.c_int32),
                ("with_obs", C.c_int32), ("scaled", C.c_int32), ("reserved", C.c_int32)]


class AdamArgs(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("clip_norm", C.c_double), ("weight_decay", C.c_double), ("step", C.c_int64),
                ("grad_scale", C.c_double), ("freeze_loc", C.c_int32), ("freeze_scale", C.c_int32),
                ("torch_eps", C.c_int32), ("reserved", C.c_int32)]


class ElboOut(C.Structure):
    _fields_ = [("loss", C.c_void_p), ("kl", C.c_void_p), ("loglik", C.c_void_p), ("preds", C.c_void_p)]


class Dropout(C.Structure):
    _fields_ = [("p", C.c_double), ("seed", C.c_uint64), ("step", C.c_uint64), ("keep_act1", C.c_void_p),
                ("keep_act2", C.c_void_p), ("keep_h", C.c_void_p)]


class DetArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("batch", C.c_int32), ("objective", C.c_int32),
                ("dropout", C.POINTER(Dropout))]


_ABI_STRUCTS = [PlanDesc, Buffers, Noise, ElboArgs, AdamArgs, ElboOut, DetArgs, Dropout]
ABI_VERSION = 3
_lib: Optional[C.CDLL] = None


class NativeError(RuntimeError):
    """Raised for every failure of the native library (keeps the reference's RuntimeError
    contract: Optuna catches RuntimeError, tasks/hpsearch.py:93)."""


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} not found: build it with `python -m bayesrul_amd.csrc.build` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own HIP runtime, the library links the system one: when the system runtime is
    # mapped first and torch's initialises first, the library's later hipGetDevice finds no device.  Bring torch's
    # runtime up before the library is mapped (a no-op on a box without a GPU).
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:  # pragma: no cover - torch is the memory / stream provider of this package
        pass
    lib = C.CDLL(LIB_PATH)
    lib.bnn_last_error.restype = C.c_char_p
    lib.bnn_abi_sizeof.restype = C.c_size_t
    lib.bnn_abi_sizeof.argtypes = [C.c_int]
    lib.bnn_plan_destroy.restype = None
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise NativeError(f"{LIB_PATH} does not export {name}")
    if lib.bnn_version() != ABI_VERSION:
        raise NativeError(f"{LIB_PATH} has ABI version {lib.bnn_version()}, the binding expects {ABI_VERSION}: rebuild it")
    for i, st in enumerate(_ABI_STRUCTS):
        n = lib.bnn_abi_sizeof(i)
        if n != C.sizeof(st):
            raise NativeError(f"ABI mismatch for {st.__name__}: library {n} bytes, binding {C.sizeof(st)}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise NativeError(f"bayesrul_amd native error {rc}: {load().bnn_last_error().decode()}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor, None -> 0."""
    return 0 if t is None else int(t.data_ptr())
