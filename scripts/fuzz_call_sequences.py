"""Stateful fuzz of ONE handle: a random sequence of calls (gradient with and without output arrays, history_precomputed,
forward-only with and without history, forced sweep, eval_adjoint, forced gradient, cost-type / timing / small-path /
save-every switches, two coefficient vectors) -- every result is compared with what a FRESH handle returns for the same
call.  What this looks for is state that leaks from one call into the next (the handle keeps a dozen validity flags).
    python3 scripts/fuzz_call_sequences.py [n_ops] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
nops = int(sys.argv[1]) if len(sys.argv) > 1 else 80
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = []
COSTS = ("Infidelity", "Tracking", "Norm")


def expected(prob, ctrl, target, order, pcofs, forcing, term):
    """results of every call kind from fresh handles (general kernels): exp[(kind, ip, cost)]"""
    exp = {}
    m = order // 2
    shape = (prob.real_system_size, m + 1, prob.nsteps + 1, prob.N_initial_conditions)
    for cost in COSTS:
        for ip, p in enumerate(pcofs):
            dp = qgd.DeviceProblem(prob, order); dp.set_small_path(False); dp.set_controls(ctrl); dp.set_target(target); dp.set_cost_type(cost)
            arrs = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((shape[0], shape[2], shape[3]), order="F")]
            g, o = dp.discrete_adjoint(p, False, *arrs)
            exp["grad", ip, cost] = (g, np.asarray(o)); exp["arrays", ip, cost] = arrs
            exp["fwd", ip, cost] = np.asarray(dp.eval_forward(p))
            hF = np.zeros(shape, order="F")
            exp["forced", ip, cost] = (np.asarray(dp.eval_forward_forced(p, forcing, hF)), hF)
            exp["adj", ip, cost] = dp.eval_adjoint(p, term, arrs[2])
            if len(p) <= 60 and prob.N_tot_levels <= 64:
                exp["gforced", ip, cost] = dp.eval_grad_forced(p)
            for save in (2, 3):
                dp.set_save_every(save)
                hs = np.zeros((shape[0], shape[1], 1 + prob.nsteps // save, shape[3]), order="F")
                dp.eval_forward(p, hs); exp["save", ip, cost, save] = hs
            dp.close()
    return exp


def run(name, prob, ctrl, pcof, target, order, windows, rng, general=False):
    m = order // 2
    shape = (prob.real_system_size, m + 1, prob.nsteps + 1, prob.N_initial_conditions)
    pcofs = [pcof, 0.6 * pcof[::-1].copy()]
    forcing = np.asfortranarray(0.2 * rng.standard_normal((shape[0], m, shape[2], shape[3])))
    term = rng.standard_normal((shape[0], shape[3]))
    os.environ["QGD_PATHS"] = "no_front"      # the fresh handles: the two-point propagator path throughout (the persistent handle takes
    try:                                       # the fused front wherever it is supported)
        exp = expected(prob, ctrl, target, order, pcofs, forcing, term)
    finally:
        os.environ["QGD_PATHS"] = "front"      # (the fused front wherever it is supported, not only on the grids it wins on)
    dp = qgd.DeviceProblem(prob, order)
    if windows:
        dp.set_memory_budget(int(dp.memory_plan()["window_bytes"] / windows * 1.15))
    # general=True: the persistent handle sees the same (linear) controls through their pointwise protocol only -- tables and
    # Jacobian uploaded per evaluation, NULL pcof (qgd_set_control_tables + qgd_set_control_basis) -- and must reproduce the basis path
    dp.set_controls([cases.PointwiseOnly(c) for c in ctrl] if general else ctrl); dp.set_target(target)
    cost, save, log = "Infidelity", 1, []

    def check(tag, a, b, tol):
        a, b = np.asarray(a, float), np.asarray(b, float)
        err = np.abs(a - b).max() / max(1.0, np.abs(b).max())
        if not (err <= tol) or not np.isfinite(a).all():
            bad.append((name, windows, tag, err, list(log[-6:]))); print(f"   MISMATCH {name} windows={windows} {tag}: {err:.2e} after {log[-6:]}", flush=True)

    kinds = ["grad", "grad_hp", "arrays", "fwd", "fwd_hist", "forced", "adj", "gforced", "cost", "timing", "small", "save"]
    for step in range(nops):
        kind = str(rng.choice(kinds)); ip = int(rng.integers(0, 2)); p = pcofs[ip]
        log.append((kind, ip, cost))
        gs = max(np.abs(exp["grad", ip, cost][0]).max(), 1e-2)
        if kind == "grad":
            g, o = dp.discrete_adjoint(p); check("grad", g / gs, exp["grad", ip, cost][0] / gs, 1e-10); check("grad scalars", o, exp["grad", ip, cost][1], 1e-11)
        elif kind == "grad_hp":
            try:
                g, o = dp.discrete_adjoint(p, True)
            except qgd._lib.QGDError as e:      # (no previous forward evaluation: a state error is the documented answer)
                assert e.code == qgd._lib.QGD_ERR_STATE, e
                continue
            check("grad_hp", g / gs, exp["grad", ip, cost][0] / gs, 1e-10); check("grad_hp scalars", o, exp["grad", ip, cost][1], 1e-11)
        elif kind == "arrays":
            arrs = [np.full(shape, np.nan, order="F"), np.full(shape, np.nan, order="F"), np.full((shape[0], shape[2], shape[3]), np.nan, order="F")]
            g, o = dp.discrete_adjoint(p, False, *arrs)
            check("arrays grad", g / gs, exp["grad", ip, cost][0] / gs, 1e-10)
            for nm, x, y in zip(("uv", "lam", "forc"), arrs, exp["arrays", ip, cost]):
                if nm == "lam": x = x.copy(); x[:, :, 0] = 0; x[:, 1:] = 0; y = y.copy(); y[:, 1:] = 0      # (j = 0 columns from n = 1)
                check("arrays " + nm, x, y, 1e-10)
        elif kind == "fwd":
            check("fwd", dp.eval_forward(p), exp["fwd", ip, cost], 1e-11)
        elif kind == "fwd_hist":
            if save == 1:
                h = np.full(shape, np.nan, order="F"); s = dp.eval_forward(p, h); check("fwd_hist", h, exp["arrays", ip, cost][0], 1e-10)
            else:
                h = np.full((shape[0], shape[1], 1 + prob.nsteps // save, shape[3]), np.nan, order="F"); s = dp.eval_forward(p, h)
                check("fwd_hist save", h, exp["save", ip, cost, save], 1e-10)
            check("fwd_hist scalars", s, exp["fwd", ip, cost], 1e-11)
        elif kind == "forced":
            if save != 1: dp.set_save_every(1); save = 1
            h = np.full(shape, np.nan, order="F"); s = dp.eval_forward_forced(p, forcing, h)
            check("forced scalars", s, exp["forced", ip, cost][0], 1e-11); check("forced history", h, exp["forced", ip, cost][1], 1e-10)
        elif kind == "adj":
            lam = dp.eval_adjoint(p, term, exp["arrays", ip, cost][2]); check("eval_adjoint", lam[:, 0], exp["adj", ip, cost][:, 0], 1e-10)
        elif kind == "gforced":
            if ("gforced", ip, cost) not in exp: continue
            check("forced gradient", dp.eval_grad_forced(p) / gs, exp["gforced", ip, cost] / gs, 1e-8)
        elif kind == "cost":
            cost = str(rng.choice(COSTS)); dp.set_cost_type(cost)
        elif kind == "timing":
            dp.set_timing(int(rng.integers(0, 2)))
        elif kind == "small":
            dp.set_small_path(bool(rng.integers(0, 2)))
        elif kind == "save":
            save = int(rng.choice([1, 2, 3])); dp.set_save_every(save)
    dp.close()


rng = np.random.default_rng(seed)
for name, case, order in (("cnot2", cases.cnot2_case(qgd, nsteps=36, tf=36.0), 8), ("guarded", cases.guarded_case(qgd, nsteps=40, tf=20.0), 6),
                          ("cnot3", cases.cnot3_case(qgd, nsteps=45, tf=45.0), 8), ("dense_guard", cases.dense_guard_case(qgd), 6)):
    for windows in (0, 3):
        print(name, "windows", windows, flush=True)
        run(name, *case, order, windows, rng)
    if name in ("cnot2", "guarded"):
        for windows in (0, 2):
            print(name, "general controls, windows", windows, flush=True)
            run(name + " (general controls)", *case, order, windows, rng, general=True)
print("mismatches:", len(bad))
for b in bad: print("  ", b)
sys.exit(1 if bad else 0)
