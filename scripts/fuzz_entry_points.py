"""Cross-checks between the ENTRY POINTS of the C ABI on random problems: every quantity that two different calls (or two
layouts of the same handle) return must agree -- scalars included, which the kernel-level parity tests look at least.
    python3 scripts/fuzz_entry_points.py [n_cases] [seed] [large]
Per case (dispersive problems with guard levels on the sparse kernels, random dense problems with a diagonal or a dense guard
projector, N = 2..40, a few N > 64): plain gradient call vs general kernels (small path off) vs a windowed handle vs the
reference-shaped call; eval_forward scalars and history vs discrete_adjoint's; zero-forced sweep vs plain sweep; random-forced
sweep on windows vs resident; eval_adjoint fed with discrete_adjoint's terminal lambda and forcing vs its lambda history;
forced gradient vs adjoint gradient; history_precomputed; guard penalty and infidelity recomputed on the host from the
history (the oracle's closed forms)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package, import_oracle
import cases
qgd = import_package(); orc = import_oracle(); orc.lib()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = []


def check(tag, a, b, tol, scale=None):
    a, b = np.asarray(a, float), np.asarray(b, float)
    sc = scale if scale is not None else max(1.0, np.abs(b).max())
    err = np.abs(a - b).max() / sc if a.size else 0.0
    if not (err <= tol) or not np.isfinite(a).all():
        bad.append((case_id, tag, err)); print(f"      MISMATCH {tag}: {err:.2e} (tol {tol:.0e})", flush=True)


for it in range(ncases):
    order = int(rng.choice([2, 4, 6, 8, 12]))
    nsteps = int(rng.choice([3, 7, 16, 33, 60, 97]))
    r = rng.random()
    if len(sys.argv) > 3 and sys.argv[3] == "large":      # N > 64 only: the GEMM-style kernels
        r = 0.95
        nsteps = int(rng.choice([3, 7, 16, 33]))
    if r < 0.5:
        nsub = int(rng.integers(1, 3)); sizes = tuple(int(rng.integers(2, 5)) for _ in range(nsub))
        ess = tuple(max(1, s - int(rng.integers(0, 2))) for s in sizes)
        freqs = 2 * np.pi * (4.0 + rng.random(nsub)); kerr = 2 * np.pi * 0.2 * (rng.random((nsub, nsub)) + 0.1); kerr = 0.5 * (kerr + kerr.T)
        prob = qgd.DispersiveProblem(sizes, ess, freqs, freqs, kerr, 0.3 * nsteps, nsteps, gmres_abstol=1e-15, gmres_reltol=1e-15)
        ctrl = [qgd.CarrierControl(qgd.FortranBSplineControl(2, int(rng.integers(4, 8)), prob.tf), [0.0, -float(kerr[k, k])][: int(rng.integers(1, 3))])
                for k in range(prob.N_operators)]
        kind = f"dispersive {sizes}/{ess}"
    else:
        N = int(rng.integers(2, 41)) if r < 0.93 else int(rng.choice([72, 100, 130, 180])); n_ops = int(rng.integers(1, 4))
        prob = qgd.construct_rand_prob(N, n_ops, tf=0.02 * nsteps, nsteps=nsteps, gmres_abstol=1e-15, gmres_reltol=1e-15, scale=1.0 / max(N, 4))
        c = int(rng.integers(1, min(N, 9) + 1)) if N <= 64 else int(rng.choice([3, 8, 16, 40, N]))
        prob.u0 = np.asfortranarray(prob.u0[:, :c]); prob.v0 = np.asfortranarray(prob.v0[:, :c]); prob.N_initial_conditions = c
        g = rng.random()
        W = np.zeros((2 * N, 2 * N))
        if g < 0.45:
            wd = rng.random(N) * (rng.random(N) > 0.5); W[np.arange(N), np.arange(N)] = wd; W[N + np.arange(N), N + np.arange(N)] = wd
        elif g < 0.7:
            A = rng.standard_normal((N, N)) * 0.3; S = A @ A.T; W[:N, :N] = S; W[N:, N:] = S
        prob.guard_subspace_projector = W
        ctrl = [qgd.FortranBSplineControl(3, 6, prob.tf) for _ in range(n_ops)]
        kind = f"dense N={N} ops={n_ops} guard={'diag' if g < 0.45 else 'full' if g < 0.7 else 'none'}"
    npar = qgd.get_number_of_control_parameters(ctrl)
    pcof = 0.3 * (rng.random(npar) - 0.5)
    target = cases.rand_target(prob, seed=it)
    N, c, m = prob.N_tot_levels, prob.N_initial_conditions, order // 2
    case_id = f"[{it}] {kind} c={c} order={order} nsteps={nsteps}"
    print(case_id, flush=True)
    shape = (2 * N, m + 1, nsteps + 1, c)

    def handle(small=True, windows=0):
        dp = qgd.DeviceProblem(prob, order)
        dp.set_small_path(small)
        if windows:
            probe = dp.memory_plan()["window_bytes"]
            try:
                dp.set_memory_budget(int(probe / windows * 1.15) + (110 << 20 if N > 64 else 0))
            except qgd._lib.QGDError:
                dp.close(); return None
        dp.set_controls(ctrl); dp.set_target(target)
        return dp

    A = handle(True); B = handle(False); Wn = handle(False, 3) if nsteps >= 16 else None
    gA, oA = A.discrete_adjoint(pcof); gB, oB = B.discrete_adjoint(pcof)
    gs = np.abs(gB).max()
    check("small path vs general: gradient", gA, gB, 1e-11, gs); check("small path vs general: scalars", oA, oB, 1e-12)
    check("eval_forward scalars", B.eval_forward(pcof), oB, 1e-13); check("eval_forward scalars (small path)", A.eval_forward(pcof), oB, 1e-12)
    arrs = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((2 * N, nsteps + 1, c), order="F")]
    gC, oC = B.discrete_adjoint(pcof, False, *arrs)
    check("reference-shaped call: gradient", gC, gB, 1e-13, gs); check("reference-shaped call: scalars", oC, oB, 1e-13)
    gD, oD = B.discrete_adjoint(pcof, True); check("history_precomputed: gradient", gD, gB, 1e-13, gs)
    hist = np.zeros(shape, order="F"); sE = B.eval_forward(pcof, hist)
    check("eval_forward history vs discrete_adjoint history", hist, arrs[0], 1e-13); check("eval_forward(hist) scalars", sE, oB, 1e-13)
    # scalars recomputed on the host from the history (infidelity.jl:7-18, :732-752 through the oracle's closed forms)
    check("guard penalty vs host", oB[2], orc.guard_penalty_real(prob, hist), 1e-11)
    inf_dev = 1 - (oB[0] ** 2 + oB[1] ** 2) / prob.N_ess_levels ** 2
    check("infidelity vs host", inf_dev, orc.infidelity_real(hist[:, 0, -1, :], orc.target_real(target), prob.N_ess_levels), 1e-12)
    z = np.zeros((2 * N, m, nsteps + 1, c), order="F")
    check("zero-forced sweep scalars", B.eval_forward_forced(pcof, z), oB, 1e-12)
    ff = np.asfortranarray(0.2 * rng.standard_normal(z.shape)); hF = np.zeros(shape, order="F")
    sF = B.eval_forward_forced(pcof, ff, hF)
    check("forced sweep: guard penalty vs host", sF[2], orc.guard_penalty_real(prob, hF), 1e-11)
    # eval_adjoint fed with the adjoint sweep's own terminal lambda and forcing reproduces its lambda history
    lam = B.eval_adjoint(pcof, arrs[1][:, 0, -1, :], arrs[2])
    check("eval_adjoint vs discrete_adjoint lambda history", lam[:, 0], arrs[1][:, 0], 1e-11)
    if npar <= 40 and N <= 64:
        check("forced gradient vs adjoint gradient", B.eval_grad_forced(pcof), gB, 1e-9, gs)
    if Wn is not None:
        gW, oW = Wn.discrete_adjoint(pcof)
        check("windows: gradient", gW, gB, 1e-11, gs); check("windows: scalars", oW, oB, 1e-12)
        check("windows: eval_forward scalars", Wn.eval_forward(pcof), oB, 1e-12)
        arw = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((2 * N, nsteps + 1, c), order="F")]
        gW2, _ = Wn.discrete_adjoint(pcof, False, *arw)
        check("windows: reference-shaped gradient", gW2, gB, 1e-11, gs)
        for nm, x, y in zip(("uv_history", "lambda_history", "adjoint_forcing"), arw, arrs):
            check("windows: " + nm, x, y, 1e-10)
        hW = np.zeros(shape, order="F"); sW = Wn.eval_forward_forced(pcof, ff, hW)
        check("windows: forced sweep scalars", sW, sF, 1e-11); check("windows: forced sweep history", hW, hF, 1e-10)
        check("windows: eval_adjoint", Wn.eval_adjoint(pcof, arrs[1][:, 0, -1, :], arrs[2])[:, 0], lam[:, 0], 1e-10)
        if npar <= 40 and N <= 64:
            check("windows: forced gradient", Wn.eval_grad_forced(pcof), gB, 1e-9, gs)
        Wn.close()
    for cost in ("Tracking", "Norm"):
        A.set_cost_type(cost); B.set_cost_type(cost)
        g1, o1 = A.discrete_adjoint(pcof); g2, o2 = B.discrete_adjoint(pcof)
        check(f"{cost}: small path vs general gradient", g1, g2, 1e-11, max(np.abs(g2).max(), 1e-2))      # (:Norm without guard levels is conserved: its gradient is rounding, 1e-15); check(f"{cost}: scalars", o1, o2, 1e-12)
        if npar <= 40 and N <= 64:
            check(f"{cost}: forced gradient", B.eval_grad_forced(pcof), g2, 1e-9, max(np.abs(g2).max(), 1e-2))
    A.close(); B.close()
print("mismatches:", len(bad))
for b in bad: print("  ", b)
sys.exit(1 if bad else 0)
