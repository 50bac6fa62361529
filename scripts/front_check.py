"""Fused front (csrc/qgd_front.h) against the general path and the numpy statements, on the device: cnot3 at a few grid sizes.
   gpurun -- python scripts/front_check.py [nsteps ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
qgd = ge.import_package()
import cases, proto_propagator as pp

def run(nsteps, order=8):
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps))
    m = order // 2
    out = {}
    for tag, paths in (("front", "front"), ("general", "no_front")):
        os.environ["QGD_PATHS"] = paths
        dp = qgd.DeviceProblem(prob, order); dp.set_target(target); dp.set_controls(ctrl)
        hist = np.zeros(dp._hist_shape(), order="F"); lam = np.zeros(dp._hist_shape(), order="F")
        forc = np.zeros((2 * dp.N, nsteps + 1, dp.c), order="F")
        g, o3 = dp.discrete_adjoint(pcof, uv_history=hist, lambda_history=lam, adjoint_forcing=forc)
        g2, o32 = dp.discrete_adjoint(pcof)
        taken = dp.front_path_taken()
        t0 = time.perf_counter()
        for _ in range(50): dp.discrete_adjoint(pcof)
        dt = (time.perf_counter() - t0) / 50
        out[tag] = dict(g=g, g2=g2, o3=o3, hist=hist.copy(), lam=lam.copy(), forc=forc.copy(), taken=taken, ms=dt * 1e3, rep=dp.intermediate("repivoted"))
        dp.close()
    f, gnl = out["front"], out["general"]
    gs = np.abs(gnl["g"]).max()
    print(f"nsteps {nsteps}: front taken {f['taken']} / {gnl['taken']}; ms/eval front {f['ms']:.4f} general {gnl['ms']:.4f}; repivoted {f['rep']} {gnl['rep']}")
    print(f"   grad rel diff {np.abs(f['g'] - gnl['g']).max() / gs:.2e} (plain call {np.abs(f['g2'] - gnl['g']).max() / gs:.2e}); scalars {np.abs(f['o3'] - gnl['o3']).max():.2e}; "
          f"hist {np.abs(f['hist'] - gnl['hist']).max():.2e}; lam {np.abs(f['lam'] - gnl['lam']).max() / max(1e-300, np.abs(gnl['lam']).max()):.2e}; "
          f"forcing {np.abs(f['forc'] - gnl['forc']).max():.2e}")
    if nsteps <= 60:
        Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, m)
        r = pp.evaluate_local(prob, Gp, Gq, off, pcof, target, order)
        print(f"   vs numpy statement (local form): grad {np.abs(f['g'] - r['grad']).max() / gs:.2e}, hist {np.abs(pp.history_real(r['ws']) - f['hist']).max():.2e}")

for ns in ([int(a) for a in sys.argv[1:]] or [6, 20, 40, 300, 550]):
    run(ns)
