"""A problem without control operators (pure drift): the forward sweep through every kernel family against expm."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from scipy.linalg import expm
from __graft_entry__ import import_package
qgd = import_package()
worst = 0.0
for N, c in ((4, 2), (20, 3), (64, 8), (80, 8), (144, 32), (300, 8)):
    prob = qgd.construct_rand_prob(N, 0, tf=0.1, nsteps=10, scale=1.0 / N)
    prob.u0 = np.asfortranarray(prob.u0[:, :c]); prob.v0 = np.asfortranarray(prob.v0[:, :c]); prob.N_initial_conditions = c
    order = 8
    hist = np.zeros((2 * N, order // 2 + 1, 11, c), order="F")
    qgd.eval_forward_(hist, prob, [], np.zeros(0), order=order)
    A = prob.system_asym - 1j * prob.system_sym
    psiT = expm(prob.tf * A) @ (prob.u0 + 1j * prob.v0)
    got = hist[:N, 0, -1, :] + 1j * hist[N:, 0, -1, :]
    e = np.abs(got - psiT).max() / np.abs(psiT).max()
    worst = max(worst, e)
    print(f"N={N} c={c} n_ops=0: final state vs expm {e:.1e}")
    qgd.clear_cache()
assert worst < 1e-11
