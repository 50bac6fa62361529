"""Random partitions of ONE problem over 2..8 ranks inside one process (qgd.LocalGroup: every rank a handle on the same GPU, the
collectives as copies): time windows and column blocks, random step counts (blocks that do not divide, ranks with a single
block), sparse and dense problems, N <= 64 and N > 64 -- gradient and scalars of every rank against the unpartitioned call.
    python3 scripts/fuzz_partitions.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import import_package
import cases
qgd = import_package()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
stream = torch.cuda.current_stream().cuda_stream
bad = []
for it in range(ncases):
    order = int(rng.choice([2, 4, 8, 12]))
    nsteps = int(rng.choice([9, 16, 23, 40, 64, 101, 150]))
    r = rng.random()
    if r < 0.35:
        prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps)); kind = "cnot3"
    elif r < 0.6:
        prob, ctrl, pcof, target = cases.guarded_case(qgd, nsteps=nsteps, tf=nsteps / 2.0); kind = "guarded"
    elif r < 0.8:
        prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=nsteps, tf=float(nsteps)); kind = "cnot2"
    else:
        N = int(rng.choice([24, 72, 100])); c = int(rng.choice([4, 8, 16]))
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=N, c=c, nsteps=nsteps, tf=0.01 * nsteps, seed=it); kind = f"dense N={N} c={c}"
    dp = qgd.DeviceProblem(prob, order); dp.set_small_path(False); dp.set_controls(ctrl); dp.set_target(target)
    g_ref, o_ref = dp.discrete_adjoint(pcof); dp.close()
    o_ref = np.asarray(o_ref); sc = max(1.0, np.abs(o_ref).max())
    for split in ("time", "columns"):
        world = int(rng.integers(2, 9))
        if split == "columns":
            world = min(world, prob.N_initial_conditions)
            if world < 2: continue
        try:
            if split == "time":
                backs = [qgd.DeviceBackend(prob, order, ctrl, target, rk, world, device=0, stream=stream) for rk in range(world)]
            else:
                backs = [qgd.ColumnBackend(prob, order, ctrl, target, rk, world, device=0, stream=stream) for rk in range(world)]
        except qgd._lib.QGDError as e:
            print(f"[{it}] {kind} nsteps={nsteps} order={order} {split} x{world}: refused ({str(e)[:60]})", flush=True); continue
        res = qgd.LocalGroup(backs).discrete_adjoint(pcof)
        eg = max(np.abs(g - g_ref).max() for g, _ in res) / np.abs(g_ref).max()
        eo = max(np.abs(np.asarray(o) - o_ref).max() for _, o in res) / sc
        flag = "" if (eg <= 1e-11 and eo <= 1e-12) else "   <-- MISMATCH"
        if flag: bad.append((it, kind, nsteps, order, split, world, eg, eo))
        print(f"[{it}] {kind} nsteps={nsteps} order={order} {split} x{world}: gradient {eg:.1e} scalars {eo:.1e}{flag}", flush=True)
        for b in backs: b.close()
print("mismatches:", len(bad))
for b in bad: print("  ", b)
sys.exit(1 if bad else 0)
