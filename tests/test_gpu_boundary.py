"""GPU tests of the drop-in boundary itself: a C consumer of include/qgd.h, the CSC constructor, the
registered (pinned) output buffers of the reference-shaped discrete_adjoint! call, argument validation."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_c_consumer(tmpdir):
    csrc = os.path.join(ROOT, "quantumgatedesign.jl_amd", "csrc")
    exe = os.path.join(str(tmpdir), "rabi_consumer")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_consumer", "rabi_consumer.c"), "-o", exe,
                           "-L", csrc, "-lqgd_hip", f"-Wl,-rpath,{csrc}", "-Wl,-rpath-link,/opt/rocm/lib", "-lm"])
    return exe


def test_c_consumer_of_the_header(tmp_path):
    """tests/c_consumer/rabi_consumer.c, compiled by gcc against include/qgd.h and linked to libqgd_hip.so: Rabi SWAP
    closed form, adjoint vs centred differences, CSC constructor -- no Python, no ctypes struct in between."""
    exe = build_c_consumer(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "C_CONSUMER_OK" in res.stdout


@pytest.mark.parametrize("which,order", [("cnot2", 4), ("cnot3", 8)])
def test_csc_constructor_matches_dense(qgd, which, order):
    """qgd_create_csc (SparseMatrixCSC triples, SchrodingerProb.jl:24-31) against qgd_create on the same operators."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    out = []
    for csc in (False, True):
        dp = qgd.DeviceProblem(prob, order, csc=csc)
        dp.set_controls(ctrl); dp.set_target(target)
        out.append(dp.discrete_adjoint(pcof))
        assert dp.operator_path()[0] == "sparse"
        dp.close()
    # bitwise: the same kernels ran on the same operands, and the gradient reductions have a fixed order (k_contract adds
    # its time chunks in chunk order, the column groups' sigma planes in group order -- no atomics on the way to grad)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])      # (the scalars too, since the guard penalty is added in a fixed order)


@pytest.mark.parametrize("which,order,cols", [("cnot3", 8, 8), ("cnot3", 8, 24), ("cnot2", 8, 4), ("guarded", 6, 4), ("dense_guard", 6, 4),
                                              ("dense20", 8, 12), ("dense48", 4, 20)])
def test_gradient_is_bitwise_reproducible(qgd, which, order, cols):
    """The reference accumulates the gradient serially (eval_grad_discrete_adjoint.jl:603-643, :148-157): the same
    inputs give the same bits.  So does the device path for N <= 64: five evaluations of one handle and one of a
    fresh handle, with 1 and with 3 column groups (cnot3 with 24 initial conditions); the dense-operator kernels below
    N = 64 too (end of round 3: no atomic sum is left on the way to the gradient or the objective)."""
    if which == "cnot3":
        prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=130, tf=130.0)
        if cols != prob.N_initial_conditions:
            rng = np.random.default_rng(9)
            z = rng.standard_normal((prob.N_tot_levels, cols)) + 1j * rng.standard_normal((prob.N_tot_levels, cols))
            z /= np.linalg.norm(z, axis=0)
            prob.u0, prob.v0 = np.asfortranarray(z.real), np.asfortranarray(z.imag)
            prob.N_initial_conditions = cols
            target = rng.standard_normal((prob.N_tot_levels, cols)) + 1j * rng.standard_normal((prob.N_tot_levels, cols))
    elif which.startswith("dense") and which[5:].isdigit():      # dense random operators below N = 64: the generic kernels (k_gradsweep)
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=int(which[5:]), c=cols, n_ops=2, nsteps=30, tf=0.3)
        rng = np.random.default_rng(7)
        prob.guard_subspace_projector = np.asfortranarray(np.diag(rng.random(2 * prob.N_tot_levels)))
    else:
        prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    grads, scalars, fwd = [], [], []
    for fresh in range(2):
        dp = qgd.DeviceProblem(prob, order); dp.set_controls(ctrl); dp.set_target(target)
        for _ in range(5 if not fresh else 1):
            g, o = dp.discrete_adjoint(pcof)
            grads.append(g); scalars.append(np.asarray(o))
        for _ in range(2):
            fwd.append(np.asarray(dp.eval_forward(pcof)))
        dp.close()
    for g in grads[1:]:
        assert np.array_equal(g, grads[0])
    # ... and so do the scalars of the objective, guard penalty included (end of round 3: its partial sums are stored per
    # workgroup and added in index order by the terminal stage).  The forward-only call adds the overlaps in its own kernel:
    # bitwise equal among its own calls, equal to the gradient call's to rounding.
    for o in scalars[1:]:
        assert np.array_equal(o, scalars[0]), (o, scalars[0])
    for o in fwd[1:]:
        assert np.array_equal(o, fwd[0]), (o, fwd[0])
    assert np.abs(fwd[0] - scalars[0]).max() <= 1e-13 * max(1.0, np.abs(scalars[0]).max())


def test_registered_outputs_match_pageable(qgd, orc):
    """The reference-shaped call discrete_adjoint!(grad, history, lambda_history, adjoint_forcing, ...) as optimize_gate
    makes it (ipopt_optimal_control.jl:304-330), with the three output arrays registered (pinned: device re-layout +
    asynchronous copies beside the adjoint sweep, strided copy of the j = 0 columns of lambda_history) and not
    registered: identical arrays, and both equal to the oracle's."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=24, tf=24.0)
    order = 8
    orc.set_converged_terminal(True)
    try:
        g_ref, h_ref, lam_ref, f_ref, _ = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
    finally:
        orc.set_converged_terminal(False)
    dp = qgd.device_problem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    res = {}
    for pinned in (False, True):
        hist = np.full(h_ref.shape, np.nan, order="F"); lam = np.full(h_ref.shape, np.nan, order="F")
        forcing = np.full(f_ref.shape, np.nan, order="F")
        if pinned:
            for a in (hist, lam, forcing):
                dp.pin(a)
            lam[...] = 0.0     # (a registered lambda_history is zero-filled by the library at its first use only)
        for rep in range(2):   # the second call reuses the staging buffers and the registrations
            grad, out3 = dp.discrete_adjoint(pcof + (0.0 if rep else 1e-3), False, hist, lam, forcing)
        res[pinned] = (grad, hist, lam, forcing)
    for a, b in zip(res[False][1:], res[True][1:]):      # the three arrays: the same kernels produced them, only the transport differs
        assert np.array_equal(a, b)
    assert np.abs(res[False][0] - res[True][0]).max() <= 1e-14 * np.abs(res[True][0]).max()    # (gradient: atomics, any order)
    grad, hist, lam, forcing = res[True]
    assert np.abs(hist - h_ref).max() <= 1e-11 * max(1.0, np.abs(h_ref).max())
    assert np.abs(lam[:, 0] - lam_ref[:, 0]).max() <= 1e-11 * max(1.0, np.abs(lam_ref[:, 0]).max())
    assert not lam[:, 1:].any()
    assert np.abs(forcing - f_ref).max() <= 1e-11
    assert np.abs(grad - g_ref).max() <= 1e-10 * np.abs(g_ref).max()
    del hist, lam, forcing, res      # unpins
    qgd.clear_cache()


def test_history_precomputed_is_tied_to_its_pcof(qgd):
    """history_precomputed=True after a forward sweep with a DIFFERENT pcof must not differentiate the stale sweep
    (the reference uses the history it is given together with the pcof it is given): the device redoes the sweep."""
    prob, ctrl, pcof, target = cases.cnot2_case(qgd)
    dp = qgd.device_problem(prob, 4)
    dp.set_controls(ctrl); dp.set_target(target)
    g_a, o_a = dp.discrete_adjoint(pcof)
    pcof_b = pcof * 1.5
    g_b, o_b = dp.discrete_adjoint(pcof_b)
    dp.eval_forward(pcof_b)
    g, o = dp.discrete_adjoint(pcof, history_precomputed=True)       # the sweep on the device belongs to pcof_b
    assert np.allclose(g, g_a, rtol=0, atol=1e-14) and np.allclose(o, o_a, rtol=0, atol=1e-14)
    g, o = dp.discrete_adjoint(pcof_b, history_precomputed=True)     # now it matches: reused
    assert np.allclose(g, g_b, rtol=0, atol=1e-14) and np.allclose(o, o_b, rtol=0, atol=1e-14)
    qgd.clear_cache()


def test_output_array_validation(qgd):
    """Wrong dtype / order / shape of an output array is a ValueError before anything reaches the C ABI
    (the reference raises a DimensionMismatch); a target with the wrong row count likewise."""
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=10, tf=10.0)
    order = 4
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    grad = np.zeros(len(pcof))
    for bad in (np.zeros(shape, order="C"), np.zeros(shape, dtype=np.float32, order="F"),
                np.zeros((shape[0], shape[1], shape[2] // 2, shape[3]), order="F")):
        with pytest.raises(ValueError):
            qgd.discrete_adjoint_(grad, bad, None, None, prob, ctrl, pcof, target, order=order)
        with pytest.raises(ValueError):
            qgd.discrete_adjoint_(grad, None, bad, None, prob, ctrl, pcof, target, order=order)
    with pytest.raises(ValueError):
        qgd.discrete_adjoint_(grad, None, None, np.zeros((shape[0], shape[2], shape[3] + 1), order="F"), prob, ctrl, pcof,
                              target, order=order)
    with pytest.raises(ValueError):
        qgd.discrete_adjoint(prob, ctrl, pcof, target[:2], order=order)     # N_ess-row target
    qgd.clear_cache()


def test_handle_cache_follows_the_problem(qgd):
    """The handle cache releases a prob's device grids when the prob dies and notices in-place mutation of its
    operators / initial conditions (a stale device copy would silently answer for the old problem)."""
    import gc
    from qgd_amd import evolution
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=10, tf=10.0)
    g0 = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=4)
    assert any(k[0] == id(prob) for k in evolution._cache)
    prob.u0[0, 0] = 0.5                                  # mutate in place
    g1 = qgd.discrete_adjoint(prob, ctrl, pcof, target, order=4)
    assert np.abs(g1 - g0).max() > 1e-6
    fresh = prob.copy()
    g2 = qgd.discrete_adjoint(fresh, ctrl, pcof, target, order=4)
    assert np.allclose(g1, g2, rtol=0, atol=1e-14)
    n_before = len(evolution._cache)
    del fresh
    gc.collect()
    assert len(evolution._cache) == n_before - 1
    work = prob.copy()
    qgd.get_histories(work, ctrl, pcof, 2, orders=(2, 4), quiet=True)      # closes the handles it created
    assert len(evolution._cache) == n_before - 1
    qgd.clear_cache()


@pytest.mark.parametrize("shard", ["time", "columns"])
def test_bench_two_processes_share_the_gpu(shard):
    """`python bench.py --gpus 2` as the driver starts it (bare: it launches two ranks itself), here with two PROCESSES
    on the one GPU of the test box, through the phase hooks (--comm torch) with the gloo backend (collectives staged through
    the host: RCCL refuses two ranks on one device; the in-library RCCL route is covered with one rank below and in
    tests/test_gpu_rccl.py).  Every step of the multi-process protocol is real -- rendezvous, one handle per process, the two
    all-gathers and the all-reduce on the library's exchange buffers -- only the transport differs from the product
    path.  The partitioned gradient must equal the single-process one, for cnot3 and for C5."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--nsteps", "120"]
    env = dict(os.environ, MASTER_PORT="29631", QGD_TINY="1")
    one = subprocess.run([sys.executable, bench, "--gpus", "1", "--no-large-n"] + common, capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, bench, "--gpus", "2", "--comm", "torch", "--backend", "gloo", "--oversubscribe", "--shard", shard] + common,
                         capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["scaling"] == "strong"
    assert abs(j2["grad_norm"] - j1["grad_norm"]) <= 1e-11 * j1["grad_norm"]
    assert abs(j2["infidelity"] - j1["infidelity"]) <= 1e-12
    assert "error" not in j2["large_n"], j2["large_n"]
    assert j2["large_n"]["grad_norm_rel_diff_vs_1gpu"] <= 1e-10
    assert j2["weak_in_time"] is None or "error" not in j2["weak_in_time"], j2["weak_in_time"]
    # one run answers both questions: the other split of the same evaluation rides in the same line
    other = j2["north_star_split" if shard == "time" else "time_window_split"]
    assert other and "error" not in other, other
    assert other["shard"] == ("columns" if shard == "time" else "time") and other["value"] > 0
    assert other["grad_rel_diff_vs_headline_split"] <= 1e-10
    assert j1["settled"] and j1["settled"]["ms_per_step"] > 0 and j1["cnot2"]["roofline"]["bound"] == "launch"
    assert j1["cnot2"]["path"].startswith("small-problem") and j1["cnot2"]["general_path"]["max_rel_gradient_difference"] <= 1e-12
    # the roofline object is about the dominant kernel of the evaluation, whatever the first call's one-time costs were: the inverse
    # on this 121-point grid (the headline's 551 points take the fused front, whose kernel is then the dominant one)
    assert (j1["roofline"]["phase"], j1["roofline"]["kernel"][:9]) in (("inverse", "k_inverse"), ("front", "k_front")) and j1["roofline"]["dominant_confirmed"], j1["roofline"]
    # ... timed live (event pair) and, beside it, the average duration of that kernel in the committed rocprofv3 summary: they
    # agree up to what the event pair adds (dispatch of the grid behind a drained queue, end-of-kernel signal)
    # (only for the profiled workload, 550 steps on one GPU: this run is shorter, the fields are there and empty)
    r = j1["roofline"]
    assert r["launch_ms"] > 0 and "launch_ms_profile" in r and "traffic" in r
    if r["launch_ms_profile"] is not None:
        assert r["launch_ms_profile_source"].startswith("profiles/")
        assert 0.8 * r["launch_ms"] <= r["launch_ms_profile"] <= 1.05 * r["launch_ms"], (r["launch_ms"], r["launch_ms_profile"])


@pytest.mark.parametrize("shard", ["time", "columns"])
def test_bench_in_library_rccl_one_rank(shard):
    """`bench.py --force-dist`: the partitioned evaluation with the collectives issued INSIDE the library over an RCCL
    communicator (one rank on the one GPU of the test box); same gradient as the plain single-GPU run, and the JSON
    line carries the event-timed collectives."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-large-n", "--no-cnot2", "--no-with-history", "--nsteps", "120"]
    env = dict(os.environ, MASTER_PORT="29633")
    one = subprocess.run([sys.executable, bench] + common, capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    lib = subprocess.run([sys.executable, bench, "--force-dist", "--shard", shard] + common, capture_output=True, text=True, timeout=600, env=env)
    assert lib.returncode == 0, lib.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in lib.stdout.splitlines() if l.startswith("{")][-1])
    assert "inside libqgd_hip.so" in j2["config"]["parallelism"]
    assert abs(j2["grad_norm"] - j1["grad_norm"]) <= 1e-11 * j1["grad_norm"]
    assert abs(j2["infidelity"] - j1["infidelity"]) <= 1e-12
    assert j2["collectives_ms"] and all(v >= 0 for v in j2["collectives_ms"].values()), j2["collectives_ms"]
    other = j2["north_star_split" if shard == "time" else "time_window_split"]
    assert other and "error" not in other, other
    assert other["grad_rel_diff_vs_headline_split"] <= 1e-10 and other["collectives_ms"], other


@pytest.mark.parametrize("which,order,world", [("cnot2", 4, 2), ("cnot2", 8, 4), ("guarded", 6, 2), ("cnot3", 8, 2), ("cnot3", 8, 8),
                                               ("synthetic", 12, 2), ("synthetic", 4, 3)])
def test_column_sharded_matches_single_gpu(qgd, which, order, world):
    """The column split (the reference's own parallel axis, Threads.@threads over initial conditions,
    src/forward_evolution.jl:48,332; the form BASELINE.json's north star sketches): every rank owns a block of the
    columns of u0, v0 and the target, two small all-reduces per evaluation.  All ranks inside one process on the one
    GPU (the collectives become device sums); gradient and scalars must equal the unsharded evaluation."""
    import torch
    if which == "synthetic":
        prob, ctrl, pcof, target = cases.synthetic_case(qgd, N=100, c=24, nsteps=12, tf=0.3)
    else:
        prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd)
    dp = qgd.DeviceProblem(prob, order)
    dp.set_controls(ctrl); dp.set_target(target)
    g_ref, o_ref = dp.discrete_adjoint(pcof)
    dp.close()
    stream = torch.cuda.current_stream().cuda_stream
    backs = [qgd.ColumnBackend(prob, order, ctrl, target, r, world, device=0, stream=stream) for r in range(world)]
    assert sum(b.columns[1] - b.columns[0] for b in backs) == prob.N_initial_conditions
    for rep in range(2):
        for g, o in qgd.LocalGroup(backs).discrete_adjoint(pcof):
            assert np.abs(g - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
            assert np.abs(o - o_ref).max() <= 1e-12 * max(1.0, np.abs(o_ref).max())
    for b in backs:
        b.close()


@pytest.mark.parametrize("which,order", [("cnot3", 8), ("guarded", 6)])
def test_save_every_nsteps_in_the_c_abi(qgd, which, order):
    """qgd_set_save_every: the strided history of eval_forward(...; saveEveryNsteps) (forward_evolution.jl:104,178,
    239-241) comes out of the library's re-layout kernel directly -- every Taylor column of every stored time point
    equals the full history's, for strides that divide the grid, that do not, and that exceed it."""
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=50, tf=25.0)
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    full = np.zeros(shape, order="F")
    qgd.eval_forward_(full, prob, ctrl, pcof, order=order)
    for save in (1, 5, 7, 50, 64):
        nslots = 1 + prob.nsteps // save
        sub = np.full((shape[0], shape[1], nslots, shape[3]), np.nan, order="F")
        qgd.eval_forward_(sub, prob, ctrl, pcof, order=order, saveEveryNsteps=save)
        assert np.array_equal(sub, full[:, :, ::save, :][:, :, :nslots, :]), save
    again = np.zeros(shape, order="F")          # the stride does not stick to the handle
    qgd.eval_forward_(again, prob, ctrl, pcof, order=order)
    assert np.array_equal(again, full)
    qgd.clear_cache()


def test_result_mirror_equals_the_copy_path(qgd):
    """Round 4: the last kernel of an evaluation writes [grad | scalars | status] into pinned host memory itself and
    publishes a sequence number the host polls (no copy packet, no wait for the stream's completion signal).  Same bits
    as the copy + hipStreamSynchronize path (QGD_RESULT_MIRROR=0, read when the handle is created), for the gradient
    evaluation, the forward-only evaluation, history_precomputed reuse, the reference-shaped call with the three output
    arrays, a changing pcof (no stale sequence number) and an evaluation with event bracketing on."""
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=130, tf=130.0)
    order = 8
    shape = (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions)
    out = {}
    for label in ("copy", "mirror"):
        if label == "copy":
            os.environ["QGD_RESULT_MIRROR"] = "0"
        try:
            dp = qgd.DeviceProblem(prob, order)
        finally:
            os.environ.pop("QGD_RESULT_MIRROR", None)
        dp.set_controls(ctrl); dp.set_target(target)
        res = []
        for scale in (1.0, 0.5, 1.0, -0.25):
            g, o = dp.discrete_adjoint(scale * pcof)
            f = dp.eval_forward(scale * pcof)
            g2, o2 = dp.discrete_adjoint(scale * pcof, history_precomputed=True)
            res += [g, np.asarray(o), np.asarray(f), g2, np.asarray(o2)]
        arrays = [np.zeros(shape, order="F"), np.zeros(shape, order="F"), np.zeros((shape[0], shape[2], shape[3]), order="F")]
        g, o = dp.discrete_adjoint(pcof, False, *arrays)
        res += [g, np.asarray(o)] + arrays
        dp.set_timing(1)
        g, o = dp.discrete_adjoint(pcof)
        assert "front" in dp.timings() or "inverse" in dp.timings()
        res += [g, np.asarray(o)]
        dp.close()
        out[label] = res
    for a, b in zip(out["copy"], out["mirror"]):
        assert np.array_equal(a, b)


def test_entry_points_agree_on_random_problems():
    """scripts/fuzz_entry_points.py: on random dispersive and dense problems every quantity that two entry points (or two
    layouts of one handle: small path / general kernels / windows) return must agree, scalars included -- gradient calls with
    and without output arrays, eval_forward, the zero-forced and the forced sweep (resident and windowed), eval_adjoint fed
    with discrete_adjoint's own terminal lambda and forcing, the forced gradient, the three cost types, and the guard penalty
    and infidelity recomputed on the host from the history.  (Round 4 found the guard penalty of the forced sweep wrong this
    way.)"""
    env = dict(os.environ, QGD_TINY="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_entry_points.py"), "24", "11"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]


def test_no_state_leaks_between_calls_on_one_handle():
    """scripts/fuzz_call_sequences.py: random sequences of calls on ONE handle (resident and windowed; cnot2 with the small
    path toggled, guarded, cnot3, a dense guard matrix) -- gradient with and without output arrays, history_precomputed,
    forward-only with and without (strided) history, forced sweep, eval_adjoint, forced gradient, switches of cost type /
    event bracketing / small path / save-every, two coefficient vectors -- each result against a fresh handle's."""
    env = dict(os.environ, QGD_TINY="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_call_sequences.py"), "120", "7"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]


def test_random_partitions_match_the_unpartitioned_call():
    """scripts/fuzz_partitions.py: ONE problem over 2..8 ranks inside one process (time windows and column blocks, step counts
    the blocks do not divide, sparse / dense / N > 64): every rank's gradient and scalars against the unpartitioned call; a
    partition that would leave a rank without a step is refused with an error, not evaluated wrongly."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_partitions.py"), "16", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
