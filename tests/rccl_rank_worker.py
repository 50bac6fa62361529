"""One rank of a multi-process evaluation through the library's own communicator path (tests/test_gpu_rccl.py):
    python tests/rccl_rank_worker.py <case> <shard> <rank | threads> <world> <dir> [fail_at]
The transport is whatever QGD_RCCL_LIB names (tests/fake_rccl: shared memory between processes on one GPU).  Rank 0 draws the
unique id and leaves it in <dir>/uid; every rank compares its results with <dir>/ref.npz (the single-GPU evaluation) and
writes <dir>/rank<r>.npz.  With fail_at, rank `world-1` injects a local failure in front of that exchange
(tests/qgd_hooks.py: qgd_comm_debug_fail_at, a test-only hook outside the library) and every rank must come back with QGD_ERR_COMM instead of hanging: exit code 7 then.
`threads` instead of a rank: all `world` ranks as THREADS of this one process, each with its own handle and communicator (a GPU
box admits six processes on its card, so eight ranks cannot be eight processes there; ctypes releases the interpreter lock
inside the library, so the ranks do meet in the collectives).  The exit code is the ranks' common code, 9 when they differ."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime in the process)
from __graft_entry__ import import_package
import cases
import qgd_hooks


def problem(qgd, case):
    if case == "cnot3":
        return cases.cnot3_case(qgd, nsteps=120, tf=120.0) + (8,)
    if case == "cnot3_headline":         # the benchmark grid: 550 steps, 69-step windows on 8 ranks
        return cases.cnot3_case(qgd, nsteps=550, tf=550.0) + (8,)
    if case == "guarded":
        return cases.guarded_case(qgd, nsteps=60, tf=30.0) + (6,)
    if case == "dense":
        return cases.synthetic_case(qgd, N=100, c=32, nsteps=60, tf=0.6) + (12,)
    raise ValueError(case)


class RankExit(Exception):
    def __init__(self, code):
        self.code = code


def run_rank(qgd, case, shard, rank, world, d, fail_at):
    prob, ctrl, pcof, target, order = problem(qgd, case)
    uid_path = os.path.join(d, "uid")
    if rank == 0:
        uid = qgd.comm_unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 60:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.01)
        uid = open(uid_path, "rb").read()
    ev = qgd.RcclEvaluation(prob, order, ctrl, target, rank, world, uid, shard=shard)
    assert ev.dp.comm_info() == dict(rank=rank, world=world, shard=shard)
    if fail_at:
        ev.dp.set_comm_timeout(15000.0)
        if rank == world - 1:
            qgd_hooks.comm_debug_fail_at(ev.dp, fail_at)
        try:
            ev.discrete_adjoint(pcof)
        except qgd._lib.QGDError as e:
            print(f"rank {rank}: {e}", flush=True)
            raise RankExit(7 if e.code == qgd._lib.QGD_ERR_COMM else 3)
        raise RankExit(4)      # (no error at all: the failure was lost)
    ref = np.load(os.path.join(d, "ref.npz"))
    g_ref, o_ref, f_ref = ref["g"], ref["o"], ref["f"]
    scale = max(1.0, np.abs(o_ref).max())
    out = {}
    for rep in range(2):
        g, o = ev.discrete_adjoint(pcof)
        assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max(), (rank, rep, np.abs(g - g_ref).max() / np.abs(g_ref).max())
        assert np.abs(np.asarray(o) - o_ref).max() <= 1e-12 * scale, (rank, rep, o, o_ref)
        out[f"g{rep}"], out[f"o{rep}"] = g, np.asarray(o)
    f = ev.eval_forward(pcof)
    assert np.abs(np.asarray(f) - f_ref).max() <= 1e-12 * scale, (rank, f, f_ref)
    g, o = ev.discrete_adjoint(pcof, history_precomputed=True)
    assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max() and np.abs(np.asarray(o) - o_ref).max() <= 1e-12 * scale
    g, o = ev.discrete_adjoint(0.5 * pcof)                    # another point: nothing stale in the exchange buffers
    out["g_half"], out["o_half"] = g, np.asarray(o)
    # the reference-shaped call: the rank's share of the three arrays
    m1 = 1 + order // 2
    if shard == "time":
        lo, hi = ev.dp.window
        nt, cc, csl = hi - lo + 1, prob.N_initial_conditions, slice(None)
    else:
        lo, hi = 0, prob.nsteps
        nt, cc, csl = prob.nsteps + 1, ev.columns[1] - ev.columns[0], slice(*ev.columns)
    n2 = prob.real_system_size
    arrays = [np.zeros((n2, m1, nt, cc), order="F"), np.zeros((n2, m1, nt, cc), order="F"), np.zeros((n2, nt, cc), order="F")]
    g, o = ev.discrete_adjoint(pcof, False, *arrays)
    assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
    for name, a, key in zip(("uv_history", "lambda_history", "adjoint_forcing"), arrays, ("hist", "lam", "forc")):
        full = ref[key]
        share = full[:, :, lo:hi + 1, csl] if full.ndim == 4 else full[:, lo:hi + 1, csl]
        if name == "lambda_history" and shard == "time" and lo > 0:
            a, share = a[:, :, 1:], share[:, :, 1:]          # (a window's first point belongs to the rank before it)
        assert np.abs(a - share).max() <= 1e-11 * max(1.0, np.abs(share).max()), (rank, name, np.abs(a - share).max())
    np.savez(os.path.join(d, f"rank{rank}.npz"), **out)
    ev.close()
    print(f"rank {rank} of {world} ({case}, {shard}): ok", flush=True)


def main():
    case, shard, who, world, d = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    fail_at = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    qgd = import_package()
    if who != "threads":
        try:
            run_rank(qgd, case, shard, int(who), world, d, fail_at)
        except RankExit as e:
            sys.exit(e.code)
        return
    import threading, traceback
    codes = [None] * world

    def body(r):
        try:
            run_rank(qgd, case, shard, r, world, d, fail_at)
            codes[r] = 0
        except RankExit as e:
            codes[r] = e.code
        except BaseException:
            traceback.print_exc()
            codes[r] = 5
    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print("rank codes:", codes, flush=True)
    sys.exit(codes[0] if all(c == codes[0] for c in codes) else 9)


if __name__ == "__main__":
    main()
