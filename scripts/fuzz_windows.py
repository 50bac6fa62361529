"""Randomised check of the paths added in round 3 against the plain resident evaluation of the same handle type:
  * the bounded-memory time grid (random budgets -> 2..12 windows): gradient, scalars, the three reference-layout arrays;
  * the in-library RCCL route with one rank, both shard kinds;
  * the stepping adjoint history pass (QGD_PATHS=no_suffix) against the suffix-product pass;
on dispersive (sparse kernels, guard levels) and random dense problems, N = 2..300, orders 2..12, 20..900 steps.
Usage: python scripts/fuzz_windows.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
qgd = import_package()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for it in range(ncases):
    order = int(rng.choice([2, 4, 6, 8, 12]))
    nsteps = int(rng.choice([20, 37, 64, 101, 200, 333, 550, 900]))
    if rng.random() < 0.6:
        nsub = int(rng.integers(1, 4))
        sizes = tuple(int(rng.integers(2, 5)) for _ in range(nsub))
        ess = tuple(max(1, s - int(rng.integers(0, 2))) for s in sizes)
        freqs = 2 * np.pi * (4.0 + rng.random(nsub))
        kerr = 2 * np.pi * 0.2 * (rng.random((nsub, nsub)) + 0.1); kerr = 0.5 * (kerr + kerr.T)
        prob = qgd.DispersiveProblem(sizes, ess, freqs, freqs, kerr, 0.3 * nsteps, nsteps)
        ctrl = [qgd.CarrierControl(qgd.FortranBSplineControl(2, int(rng.integers(4, 9)), prob.tf), [0.0, -float(kerr[k, k])])
                for k in range(prob.N_operators)]
        kind = f"dispersive {sizes}/{ess}"
    else:
        N = int(rng.choice([3, 9, 20, 33, 64, 72, 100, 130, 160, 260, 300])); n_ops = int(rng.integers(1, 4))
        nsteps = min(nsteps, 200 if N > 64 else nsteps)
        prob = qgd.construct_rand_prob(N, n_ops, tf=0.02 * nsteps, nsteps=nsteps, scale=1.0 / max(N, 4))
        c = int(rng.integers(1, min(N, 20) + 1)) if N <= 100 else int(rng.choice([8, 24, 40, 72]))
        prob.u0 = np.asfortranarray(prob.u0[:, :c]); prob.v0 = np.asfortranarray(prob.v0[:, :c]); prob.N_initial_conditions = c
        ctrl = [qgd.FortranBSplineControl(int(rng.choice([2, 16])), 20, prob.tf) for _ in range(n_ops)]
        kind = f"random dense N={N} ops={n_ops}"
    npar = qgd.get_number_of_control_parameters(ctrl)
    pcof = 0.2 * (rng.random(npar) - 0.5)
    c, N = prob.N_initial_conditions, prob.N_tot_levels
    target = rng.random((N, c)) + 1j * rng.random((N, c))
    shape = (2 * N, order // 2 + 1, nsteps + 1, c)
    mk = lambda: [np.full(shape, np.nan, order="F"), np.full(shape, np.nan, order="F"), np.full((shape[0], shape[2], shape[3]), np.nan, order="F")]
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    ref = mk()
    g_ref, o_ref = dp.discrete_adjoint(pcof, False, *ref)
    full = dp.memory_plan()["window_bytes"]
    dp.close()
    sc = max(np.abs(g_ref).max(), 1e-300)
    errs = {}
    # windows
    frac = float(rng.uniform(0.12, 0.7))
    dp = qgd.DeviceProblem(prob, order)
    try:
        dp.set_memory_budget(int(full * frac))
        dp.set_controls(ctrl); dp.set_target(target)
        got = mk()
        g, o = dp.discrete_adjoint(pcof, False, *got)
        errs[f"windows({dp.memory_plan()['windows']})"] = max(np.abs(g - g_ref).max() / sc, np.abs(np.asarray(o) - np.asarray(o_ref)).max() / max(1.0, np.abs(o_ref).max()),
                                                              *[np.nanmax(np.abs(a - b)) / max(1.0, np.abs(b).max()) if np.isfinite(a).all() else 1.0 for a, b in zip(got, ref)])
    except qgd._lib.QGDError as exc:
        if exc.code != qgd._lib.QGD_ERR_MEMORY: raise
        errs["windows(budget below one step)"] = 0.0
    dp.close()
    # in-library RCCL, one rank
    for shard in ("time", "columns"):
        ev = qgd.RcclEvaluation(prob, order, ctrl, target, 0, 1, qgd.comm_unique_id(), shard=shard)
        g, o = ev.discrete_adjoint(pcof)
        errs["rccl-" + shard] = max(np.abs(g - g_ref).max() / sc, np.abs(np.asarray(o) - np.asarray(o_ref)).max() / max(1.0, np.abs(o_ref).max()))
        ev.close()
    # stepping adjoint history pass
    os.environ["QGD_PATHS"] = "no_suffix"
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    g, o = dp.discrete_adjoint(pcof)
    errs["no-suffix"] = np.abs(g - g_ref).max() / sc
    dp.close(); del os.environ["QGD_PATHS"]
    w = max(errs.values()); worst = max(worst, w)
    print(f"[{it}] {kind} c={c} order={order} nsteps={nsteps}: " + "  ".join(f"{k} {v:.1e}" for k, v in errs.items()) + ("" if w < 1e-10 else "   <-- FAIL"), flush=True)
    qgd.clear_cache()
print("worst", worst)
assert worst < 1e-10
