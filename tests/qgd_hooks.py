"""TEST INFRASTRUCTURE: the fault injector of tests/hooks/qgd_test_hooks.cpp (kept out of libqgd_hip.so and include/qgd.h)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "hooks", "libqgd_testhooks.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        # always through make: the hook writes a field of the library's handle, so it must be compiled against the very headers
        # the library was built from (the Makefile knows the dependency; a stale copy pokes the wrong field)
        csrc = os.path.join(os.path.dirname(_HERE), "quantumgatedesign.jl_amd", "csrc")
        subprocess.check_call(["make", "-C", csrc, "libqgd_hip.so", "testhooks"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.qgd_comm_debug_fail_at.argtypes = [C.c_void_p, C.c_int32]
        _lib.qgd_comm_debug_fail_at.restype = C.c_int
    return _lib


def comm_debug_fail_at(dp, collective):
    """The next collective call of DeviceProblem `dp` fails locally in front of its exchange `collective` - 1 (0: off)."""
    rc = lib().qgd_comm_debug_fail_at(dp.h, int(collective))
    if rc:
        raise RuntimeError(f"qgd_comm_debug_fail_at: {rc}")
