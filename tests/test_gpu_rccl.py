"""GPU tests of the multi-GPU path BEHIND the C ABI: after qgd_comm_init_rccl the library itself issues the RCCL
collectives of a partitioned evaluation (include/qgd.h, "several GPUs behind ONE call").  The test box has one GPU, so
the communicator has one rank -- every collective is still an RCCL call on the handle's stream -- and the result must
equal the unpartitioned evaluation.  (Worlds of 2-8 ranks are covered through the phase hooks the protocol is made
of: test_time_partitioned_matches_single_gpu, test_bench_two_processes_share_the_gpu, tests/test_distributed_cpu.py.)
"""
import numpy as np
import pytest

import cases

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
    of each exchange of both protocols (qgd_comm_debug_fail_at), (b) a time limit that expires (qgd_set_comm_timeout).
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
        ev.dp.comm_debug_fail_at(where)
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
