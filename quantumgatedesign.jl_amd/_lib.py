"""ctypes loader for csrc/libqgd_hip.so (the C ABI of include/qgd.h).

This is the only gateway to the compute path.  If the library is missing it
raises -- there is no Python/CPU fallback for the stepper.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# QGD_LIB_PATH: an alternative build of the same library (e.g. one compiled with the in-kernel cycle stamps on)
LIB_PATH = os.environ.get("QGD_LIB_PATH") or os.path.join(CSRC, "libqgd_hip.so")

(QGD_OK, QGD_ERR_ARGUMENT, QGD_ERR_NO_DEVICE, QGD_ERR_STATE, QGD_ERR_UNSUPPORTED, QGD_ERR_NUMERIC, QGD_ERR_MEMORY,
 QGD_ERR_COMM) = range(8)
QGD_SHARD_TIME, QGD_SHARD_COLUMNS, QGD_UNIQUE_ID_BYTES = 0, 1, 128

EXPORTS = [
    "qgd_abi_version", "qgd_create", "qgd_destroy", "qgd_last_error", "qgd_set_nsteps", "qgd_set_target",
    "qgd_set_control_basis", "qgd_set_control_tables", "qgd_eval_forward", "qgd_discrete_adjoint",
    "qgd_apply_hamiltonian", "qgd_get_intermediate", "qgd_get_timings",
    "qgd_set_partition", "qgd_get_partition", "qgd_set_stream", "qgd_exchange_buffer",
    "qgd_dist_forward_begin", "qgd_dist_forward_end", "qgd_dist_adjoint_begin", "qgd_dist_adjoint_end",
    "qgd_dist_finish", "qgd_set_timing", "qgd_eval_adjoint", "qgd_set_operator_path", "qgd_get_operator_path", "qgd_eval_grad_forced", "qgd_eval_forward_forced",
    "qgd_register_host_buffer", "qgd_unregister_host_buffer", "qgd_create_csc", "qgd_cols_forward", "qgd_cols_adjoint",
    "qgd_set_lambda_derivatives", "qgd_set_cost_type",
    "qgd_comm_unique_id", "qgd_comm_init_rccl", "qgd_comm_destroy", "qgd_comm_info", "qgd_set_save_every",
    "qgd_set_memory_budget", "qgd_get_memory_plan", "qgd_set_comm_timeout", "qgd_set_small_path",
]
QGD_CREATE_DEFER_GRID = 1


class ProblemDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("n_cols", C.c_int32), ("n_ops", C.c_int32), ("n_ess", C.c_int32),
        ("order", C.c_int32), ("nsteps", C.c_int32), ("tf", C.c_double),
        ("system_sym", C.c_void_p), ("system_asym", C.c_void_p), ("sym_ops", C.c_void_p),
        ("asym_ops", C.c_void_p), ("u0", C.c_void_p), ("v0", C.c_void_p), ("guard", C.c_void_p),
        ("device", C.c_int32), ("reserved", C.c_int32),
    ]


class CSC(C.Structure):
    """qgd_csc of include/qgd.h."""
    _fields_ = [("colptr", C.c_void_p), ("rowval", C.c_void_p), ("nzval", C.c_void_p),
                ("index_base", C.c_int32), ("reserved", C.c_int32)]


class QGDError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"qgd error {code}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile libqgd_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    # make tracks the dependencies (four kernel translation units + the ABI host side)
    subprocess.check_call(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1)), "libqgd_hip.so"],
                          stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950). "
            "The stepper has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.qgd_abi_version.restype = C.c_int
    L.qgd_last_error.restype = C.c_char_p
    L.qgd_last_error.argtypes = [C.c_void_p]
    L.qgd_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(C.c_void_p)]
    L.qgd_create_csc.argtypes = [C.POINTER(ProblemDesc), C.POINTER(CSC), C.POINTER(CSC), C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_void_p)]
    L.qgd_destroy.argtypes = [C.c_void_p]
    L.qgd_destroy.restype = None
    L.qgd_set_nsteps.argtypes = [C.c_void_p, C.c_int32, C.c_double]
    L.qgd_set_target.argtypes = [C.c_void_p, C.c_void_p]
    L.qgd_set_control_basis.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.qgd_set_control_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.qgd_eval_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.qgd_discrete_adjoint.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    L.qgd_apply_hamiltonian.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.qgd_get_intermediate.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.qgd_get_timings.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.qgd_set_partition.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.qgd_get_partition.argtypes = [C.c_void_p, C.c_void_p]
    L.qgd_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.qgd_exchange_buffer.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.qgd_dist_forward_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    for name in ("qgd_dist_forward_end", "qgd_dist_adjoint_begin", "qgd_dist_adjoint_end"):
        getattr(L, name).argtypes = [C.c_void_p]
    L.qgd_dist_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.qgd_set_timing.argtypes = [C.c_void_p, C.c_int32, C.c_char_p]
    L.qgd_eval_grad_forced.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.qgd_eval_forward_forced.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.qgd_set_operator_path.argtypes = [C.c_void_p, C.c_int32]
    L.qgd_get_operator_path.argtypes = [C.c_void_p, C.c_void_p]
    L.qgd_eval_adjoint.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.qgd_cols_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.qgd_cols_adjoint.argtypes = [C.c_void_p, C.c_int32]
    L.qgd_register_host_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.qgd_unregister_host_buffer.argtypes = [C.c_void_p, C.c_void_p]
    L.qgd_set_lambda_derivatives.argtypes = [C.c_void_p, C.c_int32]
    L.qgd_set_cost_type.argtypes = [C.c_void_p, C.c_int32]
    L.qgd_set_save_every.argtypes = [C.c_void_p, C.c_int32]
    L.qgd_set_memory_budget.argtypes = [C.c_void_p, C.c_size_t]
    L.qgd_get_memory_plan.argtypes = [C.c_void_p, C.c_void_p]
    L.qgd_comm_unique_id.argtypes = [C.c_void_p]
    L.qgd_comm_init_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.qgd_comm_destroy.argtypes = [C.c_void_p]
    L.qgd_comm_info.argtypes = [C.c_void_p, C.c_void_p]
    L.qgd_set_comm_timeout.argtypes = [C.c_void_p, C.c_double]
    L.qgd_set_small_path.argtypes = [C.c_void_p, C.c_int32]
    _lib = L
    return L


def check(handle, rc):
    if rc != QGD_OK:
        msg = lib().qgd_last_error(handle)
        raise QGDError(rc, msg.decode() if msg else "unknown error")

