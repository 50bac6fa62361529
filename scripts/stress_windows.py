"""Allocation-churn stress of the windowed entry points (debugging aid): handles of varying sizes and budgets are created,
used (eval_adjoint with and without derivative columns, the reference-shaped discrete_adjoint, eval_forward with a stride) and
destroyed in a loop; every result is compared with a resident handle's.  Progress goes to stdout line by line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import faulthandler; faulthandler.enable()
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
rng = np.random.default_rng(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for it in range(iters):
    which = ("cnot3", "synthetic", "cnot2", "guarded")[it % 4]
    order = (8, 12, 8, 6)[it % 4]
    nsteps = int(rng.integers(40, 300))
    if which == "synthetic":
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=80, c=16, nsteps=nsteps, tf=0.002 * nsteps)
    else:
        prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=float(nsteps) / (1 if which.startswith("cnot") else 2))
    ref = qgd.DeviceProblem(prob, order); ref.set_controls(ctrl); ref.set_target(target)
    full = ref.memory_plan()["window_bytes"]
    windows = int(rng.integers(2, 6))
    dp = qgd.DeviceProblem(prob, order); dp.set_memory_budget(int(full / windows * 1.15) + (110 << 20 if which == 'synthetic' else 0)); dp.set_controls(ctrl); dp.set_target(target)
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    term = rng.standard_normal((shape[0], shape[3])); forc = 0.1 * rng.standard_normal((shape[0], shape[2], shape[3]))
    print(f"it {it}: {which} order {order} nsteps {nsteps} plan {dp.memory_plan()}", flush=True)
    for derivs in (False, True):
        ref.set_lambda_derivatives(derivs); dp.set_lambda_derivatives(derivs)
        a = ref.eval_adjoint(pcof, term, forc); b = dp.eval_adjoint(pcof, term, forc)
        for j in range(shape[1]):
            assert np.abs(a[:, j] - b[:, j]).max() <= 1e-10 * max(1.0, np.abs(a[:, j]).max()), (it, derivs, j)
        print(f"   eval_adjoint derivs={derivs} ok", flush=True)
        out_r = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((shape[0], shape[2], shape[3]), order="F")]
        out_w = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((shape[0], shape[2], shape[3]), order="F")]
        g0, _ = ref.discrete_adjoint(pcof, False, *out_r); g1, _ = dp.discrete_adjoint(pcof, False, *out_w)
        assert np.abs(g0 - g1).max() <= 1e-10 * np.abs(g0).max()
        for x, y in zip(out_r, out_w):
            assert np.abs(x - y).max() <= 1e-10 * max(1.0, np.abs(x).max())
        print(f"   discrete_adjoint with outputs ok", flush=True)
    ref.set_lambda_derivatives(False); dp.set_lambda_derivatives(False)
    save = int(rng.integers(2, 9))
    ref.set_save_every(save); dp.set_save_every(save)
    hs = (shape[0], shape[1], 1 + prob.nsteps // save, shape[3])
    h0 = np.zeros(hs, order="F"); h1 = np.zeros(hs, order="F")
    ref.eval_forward(pcof, h0); dp.eval_forward(pcof, h1)
    assert np.abs(h0 - h1).max() <= 1e-10 * max(1.0, np.abs(h0).max())
    print(f"   eval_forward save={save} ok", flush=True)
    if it % 3 == 0:
        dp.close(); ref.close()          # (the others are left to the garbage collector)
print("stress done", flush=True)
