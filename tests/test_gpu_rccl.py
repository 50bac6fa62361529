"""GPU tests of the multi-GPU path BEHIND the C ABI: after qgd_comm_init_rccl the library itself issues the RCCL
collectives of a partitioned evaluation (include/qgd.h, "several GPUs behind ONE call").  The test box has one GPU, so
the communicator has one rank -- every collective is still an RCCL call on the handle's stream -- and the result must
equal the unpartitioned evaluation.  Worlds of 2 and 3 ranks run the SAME library path as one process per rank with the
transport replaced by tests/fake_rccl (test_library_protocol_between_processes, test_failure_on_one_rank_ends_every_rank);
worlds of 2-8 ranks are also covered through the phase hooks the protocol is made of
(test_time_partitioned_matches_single_gpu, test_bench_two_processes_share_the_gpu, tests/test_distributed_cpu.py).
"""
import numpy as np
import pytest

import cases
import qgd_hooks

pytestmark = pytest.mark.gpu


def _reference(qgd, prob, ctrl, pcof, target, order):
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    g, o = dp.discrete_adjoint(pcof)
    f = dp.eval_forward(pcof)
    dp.close()
    return g, np.asarray(o), np.asarray(f)


@pytest.mark.parametrize("which,shard", [("cnot3", "time"), ("cnot3", "columns"), ("c5", "time"), ("c5", "columns"), ("cnot2", "time")])
def test_rccl_world1_reproduces_unpartitioned(qgd, which, shard):
    """cnot3 (550 steps, order 8) and config 5 (N=256, 256 columns, order 12, 200 steps) through the in-library
    protocol over an RCCL communicator of one rank: gradient and scalars to 1e-12 of the plain evaluation, for both
    splits, including history_precomputed reuse after a collective eval_forward."""
    if which == "c5":
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=256, c=256, nsteps=200, tf=2.0)
        order = 12
    elif which == "cnot3":
        prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=550, tf=550.0)
        order = 8
    else:
        prob, ctrl, pcof, target = cases.cnot2_case(qgd)
        order = 8
    g_ref, o_ref, f_ref = _reference(qgd, prob, ctrl, pcof, target, order)
    uid = qgd.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ev = qgd.RcclEvaluation(prob, order, ctrl, target, 0, 1, uid, shard=shard)
    assert ev.dp.comm_info() == dict(rank=0, world=1, shard=shard)
    scale = max(1.0, np.abs(o_ref).max())
    for rep in range(2):
        g, o = ev.discrete_adjoint(pcof)
        assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max(), (which, shard, rep)
        assert np.abs(np.asarray(o) - o_ref).max() <= 1e-12 * scale
    f = ev.eval_forward(pcof)                          # collective forward: the scalars are reduced
    assert np.abs(np.asarray(f) - f_ref).max() <= 1e-12 * scale
    g, o = ev.discrete_adjoint(pcof, history_precomputed=True)      # ... and must not be counted twice afterwards
    assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
    assert np.abs(np.asarray(o) - o_ref).max() <= 1e-12 * scale
    ev.close()


def test_rccl_optional_outputs_and_phase_names(qgd):
    """The reference-shaped call (uv_history, lambda_history, adjoint_forcing) under a communicator returns the
    rank's share -- with one rank: everything -- and the collectives show up as phases of the evaluation."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=48, tf=48.0)
    order = 8
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    ref = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((shape[0], shape[2], shape[3]), order="F")]
    g_ref, _ = dp.discrete_adjoint(pcof, False, *ref)
    dp.close()
    for shard in ("time", "columns"):
        ev = qgd.RcclEvaluation(prob, order, ctrl, target, 0, 1, qgd.comm_unique_id(), shard=shard)
        got = [np.zeros_like(a, order="F") for a in ref]
        ev.set_timing(1)
        g, _ = ev.discrete_adjoint(pcof, False, *got)
        names = set(ev.timings())
        assert "comm_reduce" in names and (("comm_gather_fwd" in names and "comm_gather_adj" in names) if shard == "time"
                                           else "comm_reduce_scal" in names), names
        assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
        for a, b in zip(got, ref):
            assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(b).max())
        ev.close()


def test_rccl_errors(qgd):
    prob, ctrl, pcof, target = cases.cnot2_case(qgd)
    dp = qgd.DeviceProblem(prob, 4)
    with pytest.raises(ValueError):
        dp.comm_init(b"short", 0, 1)
    with pytest.raises(qgd._lib.QGDError) as e:
        dp.comm_init(qgd.comm_unique_id(), 3, 2)
    assert e.value.code == qgd._lib.QGD_ERR_ARGUMENT
    assert dp.comm_info()["rank"] == -1
    dp.close()


@pytest.mark.parametrize("shard,where", [("time", 1), ("time", 2), ("time", 3), ("columns", 4), ("columns", 3), ("time", "timeout")])
def test_rccl_failure_mode(qgd, shard, where):
    """A collective call that cannot complete must END, with QGD_ERR_COMM and the communicator aborted (ncclCommAbort) --
    not leave this rank's stream, or the other ranks, waiting inside a collective: (a) a local failure injected in front
    of each exchange of both protocols (tests/qgd_hooks.py: qgd_comm_debug_fail_at, a test-only hook outside the library), (b) a time limit that expires (qgd_set_comm_timeout).
    Afterwards the handle has no communicator and accepts a fresh one, with which the evaluation is right again.
    (The reference's thread loop has no such state: an exception leaves Threads.@threads, src/forward_evolution.jl:48.)"""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=60, tf=60.0)
    order = 8
    g_ref, o_ref, _ = _reference(qgd, prob, ctrl, pcof, target, order)
    ev = qgd.RcclEvaluation(prob, order, ctrl, target, 0, 1, qgd.comm_unique_id(), shard=shard)
    g, _ = ev.discrete_adjoint(pcof)
    assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
    if where == "timeout":
        ev.dp.set_comm_timeout(1e-6)
    else:
        qgd_hooks.comm_debug_fail_at(ev.dp, where)
    with pytest.raises(qgd._lib.QGDError) as e:
        ev.discrete_adjoint(pcof)
    assert e.value.code == qgd._lib.QGD_ERR_COMM and "aborted" in str(e.value), str(e.value)
    assert ev.dp.comm_info()["rank"] == -1
    # argument errors are raised before anything is launched and leave a communicator alone
    ev.dp.set_comm_timeout(30000.0)
    ev.dp.comm_init(qgd.comm_unique_id(), 0, 1, shard)
    ev.dp.set_controls(ctrl); ev.dp.set_target(target)
    with pytest.raises(qgd._lib.QGDError) as e:
        ev.dp.discrete_adjoint(pcof[:-1], False, None, None, np.zeros((prob.real_system_size, 1 + prob.nsteps, prob.N_initial_conditions), order="F"))
    assert e.value.code == qgd._lib.QGD_ERR_ARGUMENT
    assert ev.dp.comm_info()["rank"] == 0
    g, o = ev.discrete_adjoint(pcof)
    assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
    assert np.abs(np.asarray(o) - o_ref).max() <= 1e-12 * max(1.0, np.abs(o_ref).max())
    ev.close()


def test_rccl_handle_never_walks_windows(qgd):
    """A handle with a communicator keeps its window resident (comm_discrete_adjoint does not walk the windows of a
    bounded-memory grid): a budget that would cut the grid into windows makes qgd_comm_init_rccl fail with QGD_ERR_MEMORY
    and leaves the handle as it was; a deferred-grid handle (QGD_CREATE_DEFER_GRID) allocates its window in comm_init."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=96, tf=96.0)
    order = 4
    g_ref, o_ref, _ = _reference(qgd, prob, ctrl, pcof, target, order)
    dp = qgd.DeviceProblem(prob, order)
    dp.set_memory_budget(dp.memory_plan()["window_bytes"] // 3)
    assert dp.memory_plan()["windows"] >= 3
    with pytest.raises(qgd._lib.QGDError) as e:
        dp.comm_init(qgd.comm_unique_id(), 0, 1, "time")
    assert e.value.code == qgd._lib.QGD_ERR_MEMORY
    assert dp.comm_info()["rank"] == -1 and dp.memory_plan()["windows"] >= 3
    dp.set_controls(ctrl); dp.set_target(target)
    g, _ = dp.discrete_adjoint(pcof)                    # (still a working single-GPU handle, in windows)
    assert np.abs(g - g_ref).max() <= 1e-11 * np.abs(g_ref).max()
    dp.close()
    dp = qgd.DeviceProblem(prob, order, defer_grid=True)
    dp.comm_init(qgd.comm_unique_id(), 0, 1, "time")
    assert dp.memory_plan()["windows"] == 1 and dp.window == (0, prob.nsteps)
    dp.set_controls(ctrl); dp.set_target(target)
    g, _ = dp.discrete_adjoint(pcof)
    assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
    dp.close()


def _fake_transport(tmp_path):
    """tests/fake_rccl/fake_rccl.cpp compiled into the test's directory: the seven nccl* entry points the library binds, carried
    over shared memory between processes that share the one GPU of the box (RCCL itself refuses two ranks on one device)."""
    import os, shutil, subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "fake_rccl.cpp")
    out = str(tmp_path / "libfake_rccl.so")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src, "-lrt"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def _run_ranks(tmp_path, lib, case, shard, world, fail_at=0, threads=False):
    """One process per rank; threads=True: all ranks as threads of ONE process (a GPU box admits six processes on its card)."""
    import os, subprocess, sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_rank_worker.py")
    env = dict(os.environ, QGD_RCCL_LIB=lib, FAKE_RCCL_TIMEOUT_MS="60000" if not fail_at else "15000")
    procs = [subprocess.Popen([sys.executable, worker, case, shard, who, str(world), str(tmp_path)] + ([str(fail_at)] if fail_at else []),
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for who in (["threads"] if threads else [str(r) for r in range(world)])]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    return [p.returncode for p in procs], outs


def _single_gpu_reference(qgd, tmp_path, case):
    import rccl_rank_worker as w
    prob, ctrl, pcof, target, order = w.problem(qgd, case)
    dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    arrays = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((shape[0], shape[2], shape[3]), order="F")]
    g, o = dp.discrete_adjoint(pcof, False, *arrays)
    f = dp.eval_forward(pcof)
    g_half, o_half = dp.discrete_adjoint(0.5 * pcof)
    dp.close()
    np.savez(tmp_path / "ref.npz", g=g, o=np.asarray(o), f=np.asarray(f), hist=arrays[0], lam=arrays[1], forc=arrays[2])
    return g_half, np.asarray(o_half)


def _check_ranks(tmp_path, world, g_half, o_half):
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in range(1, world):
        for key in res[0].files:
            assert np.array_equal(res[r][key], res[0][key]), (r, key)
    assert np.abs(res[0]["g_half"] - g_half).max() <= 1e-12 * np.abs(g_half).max()
    assert np.abs(res[0]["o_half"] - o_half).max() <= 1e-12 * max(1.0, np.abs(o_half).max())


@pytest.mark.parametrize("case,shard,world", [("cnot3", "time", 2), ("cnot3", "columns", 2), ("cnot3", "time", 3), ("guarded", "columns", 3),
                                              ("dense", "time", 2), ("dense", "columns", 2), ("cnot3", "time", 4), ("cnot3", "columns", 4)])
def test_library_protocol_between_processes(qgd, tmp_path, case, shard, world):
    """World sizes above one through the library's OWN communicator path (comm_discrete_adjoint / comm_eval_forward in
    csrc/qgd_host_comm.cpp: the in-place all-gathers at the rank's offset, the out-of-place reduction of [grad | scalars], the
    terminal condition deferred into the last rank's first adjoint launch, the zeroed scalars of column ranks other than 0,
    history_precomputed across ranks, the rank's share of the three output arrays), one PROCESS per rank on the one GPU of
    the box, with the transport replaced by tests/fake_rccl (shared memory; QGD_RCCL_LIB): every rank must return the
    single-GPU gradient and scalars to 1e-12, and all ranks the same bits."""
    g_half, o_half = _single_gpu_reference(qgd, tmp_path, case)
    lib = _fake_transport(tmp_path)
    codes, outs = _run_ranks(tmp_path, lib, case, shard, world)
    assert codes == [0] * world, "\n".join(x[-1500:] for x in outs)
    _check_ranks(tmp_path, world, g_half, o_half)


@pytest.mark.parametrize("case,shard", [("cnot3", "time"), ("cnot3_headline", "time"), ("cnot3", "columns")])
def test_library_protocol_with_eight_ranks(qgd, tmp_path, case, shard):
    """The world size of the driver's scaling run, EIGHT, through the library's own communicator path before its first
    execution over real RCCL: the ranks as eight threads of one process (a GPU box admits six processes on its card), each with
    its own handle, stream and communicator over tests/fake_rccl.  cnot3 with 120 steps: 15-step windows, ONE scan block per
    rank; the headline grid of 550 steps: 69-step windows (the last one 67); column blocks: one initial condition per rank.
    Every rank returns the single-GPU gradient and scalars to 1e-12, all ranks the same bits, the reference-shaped arrays
    its share."""
    g_half, o_half = _single_gpu_reference(qgd, tmp_path, case)
    lib = _fake_transport(tmp_path)
    codes, outs = _run_ranks(tmp_path, lib, case, shard, 8, threads=True)
    assert codes == [0], "\n".join(x[-3000:] for x in outs)
    _check_ranks(tmp_path, 8, g_half, o_half)


@pytest.mark.parametrize("shard,fail_at,world", [("time", 1, 2), ("time", 2, 2), ("time", 3, 2), ("columns", 4, 2), ("columns", 3, 2),
                                                 ("time", 2, 4), ("columns", 3, 4), ("time", 1, 8), ("columns", 4, 8)])
def test_failure_on_one_rank_ends_every_rank(qgd, tmp_path, shard, fail_at, world):
    """The failure mode between ranks: the last rank fails locally in front of one of the exchanges
    (tests/qgd_hooks.py: qgd_comm_debug_fail_at, a test-only hook outside the library).  It aborts its communicator and returns QGD_ERR_COMM; the other ranks, already waiting in the
    collective, are released by the abort and return QGD_ERR_COMM as well -- nobody hangs (exit code 7 from all).  Two and four
    ranks as processes, eight as threads of one process."""
    lib = _fake_transport(tmp_path)
    codes, outs = _run_ranks(tmp_path, lib, "cnot3", shard, world, fail_at=fail_at, threads=world > 4)
    assert codes == ([7] if world > 4 else [7] * world), "\n".join(x[-1500:] for x in outs)


def test_bench_default_multi_gpu_flow_between_processes(tmp_path):
    """`python bench.py --gpus 2` as the driver's scaling run launches it (torch.distributed.run, one process per rank,
    `--comm lib`: the collectives inside the library) on the one GPU of the box (`--oversubscribe`), the transport replaced by
    tests/fake_rccl: the headline split, the other split in the same line, the weak-in-time figure and the partitioned
    config 5 all complete, and the partitioned gradients equal the single-GPU ones."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _fake_transport(tmp_path)
    env = dict(os.environ, QGD_RCCL_LIB=lib, MASTER_PORT="29657", QGD_TINY="1")
    common = ["--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--nsteps", "120"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--no-large-n"] + common,
                         capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--oversubscribe"] + common,
                         capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and j2["value"] > 0
    assert abs(j2["grad_norm"] - j1["grad_norm"]) <= 1e-11 * j1["grad_norm"] and abs(j2["infidelity"] - j1["infidelity"]) <= 1e-12
    assert set(j2["collectives_ms"]) >= {"comm_gather_fwd", "comm_gather_adj", "comm_reduce"}, j2["collectives_ms"]
    other = j2["north_star_split"]
    assert other and "error" not in other and other["shard"] == "columns" and other["grad_rel_diff_vs_headline_split"] <= 1e-10, other
    assert set(other["collectives_ms"]) >= {"comm_reduce_scal", "comm_reduce"}
    assert j2["weak_in_time"] and "error" not in j2["weak_in_time"], j2["weak_in_time"]
    assert "error" not in j2["large_n"] and j2["large_n"]["grad_norm_rel_diff_vs_1gpu"] <= 1e-10, j2["large_n"]


def test_bench_multi_gpu_flow_with_four_ranks(tmp_path):
    """The same flow with FOUR ranks (the most a one-GPU box admits beside the test process; the driver's scaling run goes on to
    eight): the line must carry `value`, the north star's column split and both splits' collective times, and finish well
    inside the driver's 600 s."""
    import json, os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _fake_transport(tmp_path)
    env = dict(os.environ, QGD_RCCL_LIB=lib, MASTER_PORT="29671", QGD_TINY="1")
    t0 = time.time()
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--oversubscribe", "--steps", "3", "--warmup", "2",
                          "--no-cpu-baseline", "--no-large-n"], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    assert time.time() - t0 < 400
    j = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 4 and j["scaling"] == "strong" and j["value"] > 0 and j["config"]["workload"]
    assert set(j["collectives_ms"]) >= {"comm_gather_fwd", "comm_gather_adj", "comm_reduce"}, j["collectives_ms"]
    other = j["north_star_split"]
    assert other and "error" not in other and other["shard"] == "columns" and other["grad_rel_diff_vs_headline_split"] <= 1e-10, other
    assert set(other["collectives_ms"]) >= {"comm_reduce_scal", "comm_reduce"}
