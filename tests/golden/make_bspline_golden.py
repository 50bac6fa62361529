"""Generates tests/golden/bspline_bsplvd.npz from the REFERENCE's own Fortran routines
(src/Fortran/bsplvb.f, bsplvd.f compiled into oracle/_ref/bspline_lib.so by oracle/Makefile).
Run in the build container only (the reference does not travel to the GPU box):

    make -C oracle ref && python tests/golden/make_bspline_golden.py

The fixture holds inputs (degree, n_basis, x) and the dbiatx output of bsplvd_, called
exactly as FortranBSplineControl does (src/Controls/FortranBSpline.jl:257-277).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "..", "oracle", "_ref", "bspline_lib.so")


def bsplvd(lib, knots, k, x, left, nderiv):
    a = np.zeros((k, k), order="F")
    out = np.zeros((k, nderiv), order="F")
    i64 = lambda v: C.byref(C.c_int64(v))
    lib.bsplvd_(knots.ctypes.data_as(C.c_void_p), i64(k), C.byref(C.c_double(x)), i64(left),
                a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), i64(nderiv))
    return out


def main():
    lib = C.CDLL(LIB)
    rng = np.random.default_rng(42)
    recs = []
    for degree, n_basis in [(2, 10), (2, 11), (3, 8), (5, 12), (8, 12), (16, 20), (16, 17)]:
        k = degree + 1
        n_knots = n_basis + k
        nd = n_knots - 2 * (k - 1)
        knots = np.concatenate([np.zeros(k - 1), np.linspace(0, 1, nd), np.ones(k - 1)])
        xs = np.concatenate([[0.0, 1.0, 0.5, 1.0 / (nd - 1)], rng.random(12)])
        for x in xs:
            left = min(int(np.floor(x * (nd - 1) + k)), n_knots - k)
            nderiv = min(k, 7)
            out = bsplvd(lib, knots, k, float(x), left, nderiv)
            recs.append((degree, n_basis, float(x), left, nderiv, out))
    np.savez(os.path.join(HERE, "bspline_bsplvd.npz"),
             degree=np.array([r[0] for r in recs]), n_basis=np.array([r[1] for r in recs]),
             x=np.array([r[2] for r in recs]), left=np.array([r[3] for r in recs]),
             nderiv=np.array([r[4] for r in recs]),
             values=np.array([np.pad(r[5], ((0, 17 - r[5].shape[0]), (0, 7 - r[5].shape[1]))) for r in recs]))
    print("wrote", len(recs), "records")


if __name__ == "__main__":
    main()
