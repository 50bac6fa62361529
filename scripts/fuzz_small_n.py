"""Randomised shape sweep of the N <= 64 paths against the numpy statement of the algorithm (tests/proto_propagator.py):
dispersive problems (sparse ELL kernels, guard levels, 1-3 subsystems of 2-4 levels, carrier controls) and random dense
problems (dense MFMA kernels), N = 2..64 (padding to 16/32/48/64), 1..20 columns, orders 2..16, odd step counts from
1 to 200.  Usage: python scripts/fuzz_small_n.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import proto_propagator as pp
qgd = import_package()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for it in range(ncases):
    order = int(rng.choice([2, 4, 6, 8, 10, 12, 16]))
    nsteps = int(rng.choice([1, 2, 3, 5, 9, 17, 24, 37, 64, 101, 200]))
    if rng.random() < 0.6:      # dispersive: sparse operators, guard projector
        nsub = int(rng.integers(1, 4))
        sizes = tuple(int(rng.integers(2, 5)) for _ in range(nsub))
        ess = tuple(int(rng.integers(1, s + 1)) for s in sizes)
        if int(np.prod(ess)) == int(np.prod(sizes)) and rng.random() < 0.5 and max(sizes) > 2:
            ess = tuple(max(1, s - 1) for s in sizes)
        freqs = 2 * np.pi * (4.0 + rng.random(nsub))
        kerr = 2 * np.pi * 0.2 * (rng.random((nsub, nsub)) + 0.1); kerr = 0.5 * (kerr + kerr.T)
        tf = 0.3 * nsteps
        prob = qgd.DispersiveProblem(sizes, ess, freqs, freqs, kerr, tf, nsteps)
        nb = int(rng.integers(4, 9))
        ctrl = [qgd.CarrierControl(qgd.FortranBSplineControl(2, nb, prob.tf), [0.0, -float(kerr[k, k])][: int(rng.integers(1, 3))])
                for k in range(prob.N_operators)]
        kind = f"dispersive {sizes}/{ess}"
    else:
        N = int(rng.integers(2, 65)); n_ops = int(rng.integers(0, 6))
        prob = qgd.construct_rand_prob(N, n_ops, tf=0.02 * nsteps, nsteps=nsteps, scale=1.0 / max(N, 4))
        c = int(rng.integers(1, min(N, 20) + 1))
        prob.u0 = np.asfortranarray(prob.u0[:, :c]); prob.v0 = np.asfortranarray(prob.v0[:, :c]); prob.N_initial_conditions = c
        ctrl = [qgd.FortranBSplineControl(int(rng.choice([2, 16])), 20, prob.tf) for _ in range(n_ops)]
        kind = f"random dense N={N} ops={n_ops}"
    npar = qgd.get_number_of_control_parameters(ctrl) if ctrl else 0
    pcof = 0.2 * (rng.random(npar) - 0.5)
    c = prob.N_initial_conditions
    target = rng.random((prob.N_tot_levels, c)) + 1j * rng.random((prob.N_tot_levels, c))
    m = order // 2
    if npar:
        Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, m)
    else:
        Gp, Gq, off = [], [], []
    try:
        ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    except Exception as exc:
        print(f"[{it}] {kind}: statement raised {exc!r}; skipped", flush=True); continue
    dp = qgd.DeviceProblem(prob, order)
    if npar: dp.set_controls(ctrl)
    dp.set_target(target)
    hist = np.zeros((2 * prob.N_tot_levels, m + 1, nsteps + 1, c), order="F")
    href = pp.history_real(ref["ws"])
    if npar:
        grad, out3 = dp.discrete_adjoint(pcof, False, hist)
        eg = np.abs(grad - ref["grad"]).max() / max(np.abs(ref["grad"]).max(), 1e-300)
    else:
        out3 = dp.eval_forward(None, hist); eg = 0.0
    eh = np.abs(hist - href).max() / max(1.0, np.abs(href).max())
    worst = max(worst, eh, eg)
    flag = "" if max(eh, eg) < 1e-10 else "   <-- FAIL"
    print(f"[{it}] {kind} c={c} order={order} nsteps={nsteps} path={dp.operator_path()[0]}: history {eh:.1e} gradient {eg:.1e}{flag}", flush=True)
    dp.close(); qgd.clear_cache()
print("worst", worst)
assert worst < 1e-10
