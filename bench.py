#!/usr/bin/env python3
"""bench.py -- forward+adjoint timesteps/s of the Hermite stepper on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete gradient evaluation of the workload
(discrete_adjoint!(..., history_precomputed=false): control tables, step propagators,
forward sweep, guard forcing, terminal condition, adjoint sweep, gradient), host to host
(pcof in, gradient out), through the C ABI.  value = nsteps * K / elapsed.

Workload (BASELINE.json configs[2]): 3-qubit dispersive CNOT,
subsystems (4,4,4), N=64, 8 initial-condition columns, 3 control operators, Hermite order 8,
tf=550, dt=1 (nsteps=550), fp64; 180 B-spline-with-carrier coefficients, seed 0.

The JSON line also carries
  roofline      the dominant kernel (largest share of device time): algorithmic flops or
                bytes per launch / its mean duration measured with HIP events on the
                library's stream, against the gfx950 peak;
  cpu_baseline  the oracle (CPU restatement of the reference algorithm) timed on this
                host on a bounded sample of the same workload.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP64_MATRIX_TFLOPS = 78.6   # MI355X fp64 matrix = vector peak (spec); 256 CU * 4 SIMD * 2.4 GHz * 32 flop/clk
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def workload(qgd, nsteps, tf):
    import numpy as np
    import cases
    prob, target = qgd.cnot3_problem(nsteps=nsteps, tf=tf)
    ctrl = cases.cnot3_controls(qgd, prob)
    npar = qgd.get_number_of_control_parameters(ctrl)
    pcof = (np.random.default_rng(0).random(npar) - 0.5) * 2 * np.pi * 0.005
    return prob, ctrl, pcof, target


def phase_model(N, c, m, n_ops, nt, sparse_ops=False, fused_propagator=True):
    """Algorithmic work per launch of the phases that are ONE kernel launch (DESIGN.md 'Kernels').
    The two sweeps (five launches of the chain kernel each) and the gradient (scalars + contraction)
    are several launches; they appear in the per-phase breakdown but a `roofline` object describes
    one kernel, so the dominant kernel is chosen among the single-launch phases."""
    cgemm = 8.0 * N ** 3                       # complex N x N x N product, real flops
    apply_ = 8.0 * N * N * c                   # one Hamiltonian application on all columns
    mat_b = 16.0 * N * N                       # one complex N x N matrix, bytes
    hist_b = 16.0 * N * c                      # one complex state panel, bytes
    model = {
        # inverse by Gauss-Jordan = one cgemm of flops; the fused kernel also forms P = L^-1 R
        "inverse": ("mfma", (nt - 1) * cgemm * (2 if fused_propagator else 1)),
        "propagator": ("mfma", (nt - 1) * cgemm),
        "lambda": ("hbm", (nt - 1) * (mat_b + 2 * hist_b)),
        # fused front (csrc/qgd_front.h): per time point the Gauss-Jordan inverse of L^H and S^H = L^-H R^H on the MFMA (2 cgemm) and
        # the build of L, R from the ELL operators on the fp64 vector ALU (0.76 GFLOP per 551 time points, DESIGN.md section 4:
        # k_build_LR_ell's count) -- on gfx950 the two share one fp64 arithmetic pipe, so one peak prices both;
        # psi: L^-H once, phi in, psi / f / h out
        "front": ("mfma", nt * cgemm * 2 + (0.76e9 * nt / 551.0 if sparse_ops else 0.0)),
        "psi": ("hbm", nt * (mat_b + 4 * hist_b)),
        "guard": ("hbm", nt * (3 * hist_b)),
    }
    if sparse_ops:   # ELL kernels: flops are negligible, the kernel exists to write L_n and R_n
        model["build_LR"] = ("hbm", nt * 2 * mat_b)
    else:
        model["build_LR"] = ("mfma", nt * (m * (m - 1) / 2) * cgemm)
    return model


KERNEL_OF_PHASE = {"build_LR": "k_build_LR_ell", "inverse": "k_inverse_cb", "propagator": "k_propagator",
                   "lambda": "k_lambda", "guard": "k_guard_diag", "front": "k_front", "psi": "k_psi"}
CNOT2_PROFILE = "r05_cnot2_launches.json"           # scripts/cnot2_profile.sh: launches per evaluation from rocprofv3 kernel statistics
PMC_PROFILE = "r06_pmc_fetch_write_cnot3.json"      # regenerated for this round's kernels (profiles/README.md)
PMC_MFMA_PROFILE = "r06_pmc_mfma_cnot3.json"
STATS_PROFILE = "r06_kernel_stats_cnot3.csv"         # rocprofv3 --kernel-trace --stats of this command (scripts/collect_profiles.sh)


def _lookup_kernel(table, kern):
    """Value of the first key of `table` that names kernel `kern` (keys are full demangled names with template
    arguments, e.g. 'void k_build_LR_ell<4, 8, 3>'; `kern` is the bare kernel name)."""
    for key, val in (table or {}).items():
        bare = key.replace("void ", "")
        if bare == kern or bare.startswith(kern + "<") or bare.startswith(kern + "("):
            return key, val
    return None, None


def measured_traffic(phase):
    """HBM bytes per launch of the phase's kernel from the committed rocprofv3 PMC passes
    (profiles/PMC_PROFILE: FETCH_SIZE and WRITE_SIZE in separate --pmc runs of this same command;
    on gfx950 FETCH_SIZE counts half of a wide coalesced read, so it is doubled --
    MI355X_MICROARCH.md 'HBM').  (None, None) when no profile of that kernel is committed."""
    path = os.path.join(ROOT, "profiles", PMC_PROFILE)
    kern = KERNEL_OF_PHASE.get(phase)
    if not kern or not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    _, f = _lookup_kernel(d.get("FETCH_SIZE_KB_mean_per_launch"), kern)
    _, w = _lookup_kernel(d.get("WRITE_SIZE_KB_mean_per_launch"), kern)
    if f is None or w is None:
        return None, None
    return (2.0 * f + w) * 1024.0, "profiles/" + PMC_PROFILE


def profiled_launch_ms(phase):
    """Average duration of the phase's kernel in the committed `rocprofv3 --kernel-trace --stats` summary of this same command
    (profiles/STATS_PROFILE: first wave to last wave), in ms, and the file; (None, None) without one."""
    import csv
    path = os.path.join(ROOT, "profiles", STATS_PROFILE)
    kern = KERNEL_OF_PHASE.get(phase)
    if not kern or not os.path.exists(path):
        return None, None
    for row in csv.DictReader(open(path)):
        if kern in row["Name"]:
            return float(row["AverageNs"]) * 1e-6, "profiles/" + STATS_PROFILE
    return None, None


def measured_mfma_util(phase):
    """MFMA pipe utilisation of the phase's kernel from the committed counter pass (profiles/PMC_MFMA_PROFILE:
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)), or None."""
    path = os.path.join(ROOT, "profiles", PMC_MFMA_PROFILE)
    kern = KERNEL_OF_PHASE.get(phase, "")
    if not kern or not os.path.exists(path):
        return None
    _, v = _lookup_kernel(json.load(open(path)).get("kernels"), kern)
    return None if v is None else v.get("mfma_util")


def _oracle_rate(qgd, orc, make_case, order, threads, sparse, seconds_target, probe_steps=6):
    """timesteps/s of the oracle's full discrete_adjoint on a bounded sample: `probe_steps` steps to size the sample,
    then as many steps as fit in `seconds_target` (at most the full grid)."""
    orc.set_num_threads(threads)
    orc.set_sparse_operators(sparse)
    try:
        prob, ctrl, pcof, target, full = make_case(probe_steps)
        t0 = time.time()
        orc.discrete_adjoint(prob, ctrl, pcof, target, order=order)
        per_step = (time.time() - t0) / probe_steps
        nsample = int(max(probe_steps, min(full, seconds_target / per_step)))
        prob, ctrl, pcof, target, full = make_case(nsample)
        t0 = time.time()
        _, _, _, _, st = orc.discrete_adjoint(prob, ctrl, pcof, target, order=order, return_all=True)
        el = time.time() - t0
    finally:
        orc.set_sparse_operators(False)
    return {"value": nsample / el, "unit": "timesteps/s", "cores": threads, "operators": "CSC" if sparse else "dense",
            "timesteps": nsample, "of": full, "seconds": round(el, 2),
            "gmres_iters_fwd": round(st.fwd_gmres_iters, 1), "gmres_iters_adj": round(st.adj_gmres_iters, 1)}


def cpu_baseline(qgd, orc, seconds_target=10.0):
    """Oracle ('port' of the reference CPU path; Julia is not in the image) on bounded samples of the same
    workload: cnot3 order 8 at dt=1, GMRES tolerance 1e-12 (examples/cnot3_optimize_gate.jl:12-19), columns in parallel
    as the reference's Threads.@threads (forward_evolution.jl:48,332), gradient accumulation serial
    (eval_grad_discrete_adjoint.jl:148-157).  The headline object is the 8-thread CSC-operator figure (DispersiveProblem's
    default sparse_rep=true; the faster CPU form); `variants` adds what BASELINE.md section 2 lists: dense operators
    (regression.jl:30), one thread (the author's cluster runs, cnot3_optimize_gate.sb:7), and C2 (cnot2, order 8)."""
    import numpy as np
    import cases
    cores = os.cpu_count() or 1
    threads = min(8, cores)

    def cnot3(nsteps):
        prob, ctrl, pcof, target = workload(qgd, nsteps, float(nsteps))
        prob.gmres_abstol = prob.gmres_reltol = 1e-12
        return prob, ctrl, pcof, target, 550

    def cnot2(nsteps):       # examples/cnot2_optimization.jl:10-47 (tf = nsteps: the script's dt = 1)
        prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=nsteps, tf=float(nsteps), amp=1e-2)
        prob.gmres_abstol = prob.gmres_reltol = 1e-12
        return prob, ctrl, pcof, target, 100

    # headline object: CSC operators -- what the reference's DispersiveProblem builds by default (sparse_rep=true,
    # multi_qudit_systems.jl:118-162) and the faster of the two CPU forms; the dense form of regression.jl:30 beside it
    head = _oracle_rate(qgd, orc, cnot3, 8, threads, True, seconds_target)
    variants = {}
    for name, case, th, sp, budget in (("cnot3_dense_%dthreads" % threads, cnot3, threads, False, 6.0), ("cnot3_dense_1thread", cnot3, 1, False, 4.0),
                                       ("cnot3_csc_1thread", cnot3, 1, True, 4.0), ("cnot2_order8_csc_1thread", cnot2, 1, True, 2.0),
                                       ("cnot2_order8_csc_4threads", cnot2, min(4, cores), True, 2.0)):
        try:
            variants[name] = _oracle_rate(qgd, orc, case, 8, th, sp, budget)
        except Exception as exc:          # a variant must never cost the headline line
            variants[name] = {"error": repr(exc)}
    return {
        "value": head["value"], "unit": "timesteps/s", "cores": threads, "kind": "port",
        "operators": "CSC",
        "sample": f"cnot3 order 8, dt=1, {head['timesteps']} of 550 timesteps, all 8 columns, CSC (SparseMatrixCSC) operators, GMRES tol 1e-12, "
                  f"{threads} threads over columns (gradient accumulation serial as in the reference); "
                  f"mean GMRES iterations fwd {head['gmres_iters_fwd']} adj {head['gmres_iters_adj']}",
        "seconds": head["seconds"], "host_cores": cores, "variants": variants,
    }


C5_GRAD_NORM_1GPU = 3454.7659605167805     # |grad| of the C5 workload on one GPU (profiles/r02_bench.json): what a partitioned run must reproduce


def make_partitioned(qgd, args, prob, order, ctrl, target, rank, world, local_rank, uid):
    """One rank's evaluator of a partitioned evaluation.  Default (--comm lib): RCCL inside the library, the call is
    the plain qgd_discrete_adjoint.  --comm torch: the older route, phase hooks + torch.distributed collectives."""
    import torch
    import torch.distributed as dist
    if args.comm == "lib":
        # every rank must agree on the route: a rank that cannot get its communicator (QGD_ERR_COMM: librccl not loadable,
        # ...) makes ALL ranks fall back to the phase hooks with torch.distributed's own nccl group
        ev, ok = None, 1
        try:
            ev = qgd.RcclEvaluation(prob, order, ctrl, target, rank, world, uid(), shard=args.shard, device=local_rank)
        except Exception as exc:
            ok = 0
            print(f"[bench rank {rank}] in-library RCCL unavailable ({exc!r}); falling back to --comm torch", file=sys.stderr)
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            return ev
        if ev is not None:
            ev.close()
        args.comm = "torch"
        if getattr(args, "_torch_group", None) is None:
            args._torch_group = dist.new_group(backend="nccl") if dist.get_backend() != "nccl" else None
    group = getattr(args, "_torch_group", None)
    stream = torch.cuda.current_stream().cuda_stream
    if args.shard == "columns":
        back = qgd.ColumnBackend(prob, order, ctrl, target, rank, world, device=local_rank, stream=stream)
        dp = qgd.ColumnSharded(back, qgd.TorchComm(group))
    else:
        back = qgd.DeviceBackend(prob, order, ctrl, target, rank, world, device=local_rank, stream=stream)
        dp = qgd.TimePartitioned(back, qgd.TorchComm(group))
    dp.timings, dp.close, dp.set_timing = back.timings, back.close, back.set_timing
    return dp


def large_n_partitioned(qgd, np, args, rank, world, local_rank, uid, steps=3):
    """The same C5 evaluation spread over the ranks by time windows (strong scaling: the configuration where the
    windows can pay); every rank holds the reduced gradient, rank 0 reports."""
    import torch
    import torch.distributed as dist
    N, n_ops, nsteps, order = 256, 4, 200, 12
    prob = qgd.construct_rand_prob(N, n_ops, tf=2.0, nsteps=nsteps, scale=1.0 / N)
    ctrl = [qgd.FortranBSplineControl(16, 20, prob.tf) for _ in range(n_ops)]
    pcof = np.random.default_rng(5).random(qgd.get_number_of_control_parameters(ctrl))
    target = prob.u0 + 1j * prob.v0
    shard = args.shard
    dp = make_partitioned(qgd, args, prob, order, ctrl, target, rank, world, local_rank, uid)
    dp.set_timing(0)
    dp.discrete_adjoint(pcof)
    gc.collect()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        grad, _ = dp.discrete_adjoint(pcof)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    tm = torch.tensor([time.perf_counter() - t0], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    sec = float(tm.item()) / steps
    dp.close()
    gn = float(np.linalg.norm(grad))
    return {"workload": f"C5 synthetic: N={N}, 256 columns, {n_ops} control operators, order {order}, nsteps={nsteps}, "
                        f"{'time windows' if shard == 'time' else 'column blocks'} over {world} GPUs", "scaling": "strong", "timesteps_per_s": nsteps / sec,
            "ms_per_evaluation": sec * 1e3, "grad_norm": gn, "grad_norm_rel_diff_vs_1gpu": abs(gn - C5_GRAD_NORM_1GPU) / C5_GRAD_NORM_1GPU}


def large_n_case(qgd, np, steps=3):
    """BASELINE.json configs[4] (C5: random dense SchrodingerProb, N=256, 256 columns, 4 control operators,
    order 12, tf=2, nsteps=200) -- the configuration where the MFMA roofline is the binding one.
    Reported beside the headline line, never as `value`.  Work is counted in complex N x N x N contractions ("GEMM units",
    8 N^3 flop each, the usual count for a complex matrix product) that the evaluation performs: recursion on the identity
    m(m-1)/2, four chain passes, inverse, propagator, lambda, and for the gradient 2 outer products + the reverse sweep on
    N x N matrices m(m-1)/2 + the contraction with the stored D_i m(m-1)/2 (round 2: sweep on the panels m(m-1)/2, stage
    derivatives m, inner products N_op*m) -- per time point.  `frac` = units * 8 N^3 / time / peak.  Since round 3 most units
    run on three-real-product tiles (3M: 6 N^3 flop issued per unit): `mfma_issued_*` prices the same
    time at the flops the MFMA pipe actually executes; `frac_at_round2_flop_count` at round 2's formulation."""
    import torch
    N, c, n_ops, nsteps, order = 256, 256, 4, 200, 12
    m = order // 2
    prob = qgd.construct_rand_prob(N, n_ops, tf=2.0, nsteps=nsteps, scale=1.0 / N)
    ctrl = [qgd.FortranBSplineControl(16, 20, prob.tf) for _ in range(n_ops)]
    rng = np.random.default_rng(5)
    pcof = rng.random(qgd.get_number_of_control_parameters(ctrl))
    target = prob.u0 + 1j * prob.v0
    dp = qgd.DeviceProblem(prob, order, device=0)
    dp.set_controls(ctrl); dp.set_target(target)
    dp.set_timing(0)
    dp.discrete_adjoint(pcof)
    gc.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        grad, _ = dp.discrete_adjoint(pcof)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    # the number is only worth printing if the gradient is right: centred-difference directional derivative of
    # infidelity + guard penalty (objective ~ -3e4 with the un-normalised random U0 as target: eps 1e-3)
    d = np.random.default_rng(1).standard_normal(len(pcof)); d /= np.linalg.norm(d)
    eps = 1e-3

    def obj(p):
        a, b, g = dp.eval_forward(p)
        return 1 - (a * a + b * b) / prob.N_ess_levels ** 2 + g

    fd = (obj(pcof + eps * d) - obj(pcof - eps * d)) / (2 * eps)
    fd_rel = abs(fd - grad @ d) / abs(fd)
    assert np.isfinite(grad).all() and fd_rel < 1e-6, f"C5 gradient check failed: adjoint {grad @ d}, centred difference {fd}"
    dp.close()
    # gradient: round 2 swept the seeds on the state panels (m(m-1)/2 units), formed the stage derivatives (m) and applied
    # every control operator to every one of them (n_ops*m); round 3 works on N x N matrices instead -- two outer products
    # lambda psi_0^H per time point, the same reverse sweep on the matrices Y_j = g_j psi_0^H (m(m-1)/2), and the scalars as
    # Frobenius products of sum_i (1/j) Y_j D_i^H (m(m-1)/2) with the control operators
    grad_units = min(m * (m - 1) // 2 + m + n_ops * m, 2 + m * (m - 1) // 2 + m * (m - 1) // 2)
    gemms = m * (m - 1) // 2 + 4.0 * nsteps / (nsteps + 1) + 3 + grad_units
    gemms_r02 = m * (m - 1) // 2 + m + 4.0 * nsteps / (nsteps + 1) + 3 + m * (m - 1) // 2 + n_ops * m
    tflop = 8.0 * N ** 3 * gemms * (nsteps + 1) / 1e12
    # units on three-product tiles (level recursion, sweep, D-contraction, outer products, chains) issue 6 N^3, the rest 8 N^3
    units_3m = gemms - 3 if "dense_4m" not in os.environ.get("QGD_PATHS", "") else 0.0
    issued = 8.0 * N ** 3 * (0.75 * units_3m + (gemms - units_3m)) * (nsteps + 1) / 1e12
    # `frac` is the HARDWARE fraction: flops the MFMA pipe issues (three real products per complex multiply on most
    # units) / time / peak.  The usual complex-GEMM count (8 N^3 per unit) prices the same time higher and is kept as
    # `effective_*`: an algorithmic ratio, not a utilisation.
    return {"workload": f"C5 synthetic: N={N}, {c} columns, {n_ops} control operators, order {order}, nsteps={nsteps}",
            "timesteps_per_s": nsteps / sec, "ms_per_evaluation": sec * 1e3,
            "bound": "mfma", "tflop_per_evaluation": issued, "achieved": issued / sec, "peak": PEAK_FP64_MATRIX_TFLOPS, "unit": "TFLOP/s",
            "frac": issued / sec / PEAK_FP64_MATRIX_TFLOPS,
            "frac_note": "MFMA-issued flops (3M tiles: 6 N^3 per complex GEMM unit) / time / 78.6 TFLOP/s; the card clocks ~2.1 GHz under this load "
                         "(peak assumes 2.4): ~0.78 of the clock-adjusted peak",
            "effective_tflop_per_evaluation": tflop, "effective_achieved": tflop / sec, "effective_frac": tflop / sec / PEAK_FP64_MATRIX_TFLOPS,
            "effective_note": "the same time priced at 8 N^3 per complex GEMM unit (the usual count); not a hardware utilisation",
            "gemm_units_per_time_point": gemms, "gemm_units_round2_formulation": gemms_r02,
            "frac_at_round2_flop_count": 8.0 * N ** 3 * gemms_r02 * (nsteps + 1) / 1e12 / sec / PEAK_FP64_MATRIX_TFLOPS,
            "grad_norm": float(np.linalg.norm(grad)),
            "gradient_vs_central_difference_rel": fd_rel}


def cnot2_case_gpu(qgd, np, steps=50):
    """BASELINE.json configs[1] (C2: 2-qubit CNOT of examples/cnot2_optimization.jl:10-47, N=4, 4 columns, order 8,
    tf=100, nsteps=100) on the GPU: north_star asks for the cnot2 rate beside the cnot3 one.  The library's default for a
    problem this small is the four-launch path of csrc/qgd_k_tiny.hip (fp64 vector ALU, one thread per (time point,
    column)); the general path -- a 12-launch chain of padded 16 x 16 MFMA tiles on 101 time points -- is timed beside it
    (bring-up configuration, not a throughput one)."""
    import torch
    import cases
    prob, ctrl, pcof, target = cases.cnot2_case(qgd, nsteps=100, tf=100.0, amp=1e-2)

    def timed(small):
        dp = qgd.DeviceProblem(prob, 8, device=0)
        if not small:
            dp.set_small_path(False)
        dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
        for _ in range(5):
            dp.discrete_adjoint(pcof)
        gc.collect()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            grad, out3 = dp.discrete_adjoint(pcof)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / steps
        taken = bool(dp.small_path_taken())
        dp.close()
        return sec, taken, grad, out3

    sec_gen, _, grad_gen, _ = timed(False)
    sec, small_taken, grad, out3 = timed(True)
    # roofline of this configuration: NEITHER hardware roof binds.  N = 4 pads to one 16 x 16 MFMA tile and 101 time points
    # are 101 small workgroups: the evaluation is a chain of DEPENDENT kernel launches, each costing its dispatch + one
    # workgroup's latency.  Stated as such: launches per evaluation (from the committed rocprofv3 kernel statistics of this
    # very loop, profiles/CNOT2_PROFILE; the count of the launch sequence in DESIGN.md section 7a otherwise), time per
    # launch, and -- for the record -- the algorithm's history stream against HBM and its applies against the MFMA peak.
    N_, c_, m_, n_ops_, ns_ = 4, 4, 4, 2, 100
    launches, src = (4, "csrc/qgd_k_tiny.hip (front, scan, gradient, sum)") if small_taken else (12, "DESIGN.md section 7a (launch sequence)")
    ppath = os.path.join(ROOT, "profiles", CNOT2_PROFILE)
    if small_taken and os.path.exists(ppath):       # (the committed profile is of the small-problem path: its launch count describes no other)
        try:
            d = json.load(open(ppath))
            launches, src = float(d["launches_per_evaluation"]), "profiles/" + CNOT2_PROFILE
        except Exception:
            pass
    b_step = 2 * 16 * N_ * (1 + m_) * c_ + 2 * 16 * N_ * c_
    M_ = m_ * (m_ + 1) // 2
    w_step = M_ * 8.0 * N_ * N_ * c_ * 6          # SURVEY 8(d) W_step with the propagator form's k = 0: forward M, adjoint M, gradient 4M applies
    return {"workload": "cnot2 (examples/cnot2_optimization.jl): N=4, 4 columns, 2 controls x 22 coeffs, Hermite order 8, tf=100, nsteps=100",
            "timesteps_per_s": 100 / sec, "ms_per_evaluation": sec * 1e3, "evaluations_timed": steps,
            "path": "small-problem path (qgd_k_tiny.hip, 4 launches)" if small_taken else "general path",
            "general_path": {"timesteps_per_s": 100 / sec_gen, "ms_per_evaluation": sec_gen * 1e3,
                             "max_rel_gradient_difference": float(np.abs(grad - grad_gen).max() / np.abs(grad_gen).max())},
            "roofline": {"bound": "launch", "launches_per_evaluation": launches, "launch_count_source": src,
                         "us_per_dependent_launch": sec * 1e6 / launches,
                         "hbm": {"bytes_per_timestep": b_step, "achieved": b_step * ns_ / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": b_step * ns_ / sec / 1e9 / PEAK_HBM_GBS},
                         "mfma": {"flop_per_timestep": w_step, "achieved": w_step * ns_ / sec / 1e12, "peak": PEAK_FP64_MATRIX_TFLOPS, "unit": "TFLOP/s",
                                  "frac": w_step * ns_ / sec / 1e12 / PEAK_FP64_MATRIX_TFLOPS},
                         "note": "launch-bound: a chain of dependent launches over 101 time points whose work is 2-8 us each; both hardware fractions "
                                 "are ~1e-4 by construction (N = 4 is a sixteenth of one MFMA tile; the small-problem path does not use MFMA at all) -- "
                                 "the figure of merit is us per dependent launch"},
            "infidelity": float(1 - (out3[0] ** 2 + out3[1] ** 2) / prob.N_ess_levels ** 2), "grad_norm": float(np.linalg.norm(grad))}


def executed_gflop(N, c, m, n_ops, nsteps, blocks, blen, sparse_ops, front=False):
    """Real flops one cnot3-shaped evaluation EXECUTES on the N <= 64 path: Gauss-Jordan inverse +
    L^-1 R (2 complex N^3 products per step), the matrix-matrix products of the two scan levels, the matrix-panel
    products of the history passes with their prefixes, the affine parts, lambda, and the step-matrix build and
    gradient kernels (fp64 vector ALU on the sparse path: counted from the ELL entry counts; MFMA GEMMs on the dense)."""
    cg, ap = 8.0 * N ** 3, 8.0 * N * N * max(c, 8)
    B = max(1, blocks)
    B2 = max(1, int(round((2.0 * B) ** 0.5))) if B > 8 else 1
    g = (B + B2 - 1) // B2
    inv = (nsteps + 1 if front else nsteps) * 2 * cg          # (fused front: every time point, same-point products)
    scan_mm = (nsteps - B) * cg + (B - B2 if B2 > 1 else 0) * cg
    hist_fwd = (nsteps + 3 * B * (B2 + 2)) * ap                   # own steps + prefixes of the 3-step sub-block workgroups
    hist_adj = (nsteps + B * (B2 + g)) * ap
    affine = (nsteps + B) * ap
    lam = 2 * (nsteps + 1) * ap if front else nsteps * ap       # (fused front: k_psi -- psi = L^-1 phi and h = L^-H f per time point)
    if sparse_ops:
        build, grad = 0.76e9 * nsteps / 550.0, 0.35e9 * nsteps / 550.0
    else:
        build = (nsteps + 1) * (m * (m - 1) / 2) * cg
        grad = (nsteps + 1) * (m * (m - 1) + n_ops * m) * ap
    return (inv + scan_mm + hist_fwd + hist_adj + affine + lam + build + grad) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--run-in", type=int, default=0,
                    help="extra untimed evaluations after the W warm-up steps (the card's clocks settle after ~20 ms of work, "
                         "scripts/clock_profile.py); 0 = the contract's plain 'W warm-up, K timed'; recorded as run_in_steps")
    ap.add_argument("--nsteps", type=int, default=550, help="timesteps of the workload (tf = nsteps, dt = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-n", action="store_true", help="skip the secondary C5 (N=256) measurement")
    ap.add_argument("--no-cnot2", action="store_true", help="skip the secondary C2 (cnot2) measurement")
    ap.add_argument("--no-with-history", action="store_true", help="skip the secondary measurement with the three output arrays "
                                                                   "(profiling runs: keeps the kernel statistics to the headline evaluation)")
    ap.add_argument("--force-dist", action="store_true", help="use the partitioned path even with one rank (plumbing check)")
    ap.add_argument("--comm", default="lib", choices=["lib", "torch"],
                    help="N > 1: who issues the collectives -- lib (default): RCCL inside libqgd_hip.so (qgd_comm_init_rccl; the "
                         "process group is gloo and carries only the 128-byte id, the barrier and the max over ranks of the "
                         "elapsed time); torch: the phase hooks + torch.distributed collectives (--backend)")
    ap.add_argument("--backend", default="nccl", help="--comm torch: torch.distributed backend, nccl (= RCCL) or gloo "
                                                      "(plumbing check of the multi-process protocol; collectives staged through the host)")
    ap.add_argument("--shard", default="time", choices=["time", "columns"],
                    help="N > 1: how ONE evaluation is split over the ranks -- time windows (default: 2 all-gathers + 1 all-reduce) or the "
                         "reference's thread axis, blocks of initial-condition columns (2 all-reduces; the step matrices are built on every rank)")
    ap.add_argument("--no-other-split", action="store_true", help="N > 1: do not time the other split (columns beside time windows) in the same line")
    ap.add_argument("--oversubscribe", action="store_true", help="let several ranks share a GPU (test boxes with one GPU; with --comm torch --backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started bare with --gpus N: become the launcher of N ranks (one process per GPU) BEFORE anything touches the
        # GPU in this process (device_count() does not initialise it), and pass the children's output through.
        import subprocess
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and not (args.oversubscribe and have >= 1):
            sys.exit(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible; refusing to print a "
                     f"{have}-GPU number under n_gpus={args.gpus}")
        port = os.environ.get("MASTER_PORT", "29517")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (RCCL between processes needs dmabuf IPC on this driver; before any HIP call)
    import numpy as np
    import torch   # first: so that one HIP runtime (torch's) serves both torch and libqgd_hip
    import torch.distributed as dist
    from __graft_entry__ import import_package, import_oracle

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and not args.force_dist:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    use_dist = world > 1 or args.force_dist
    if args.oversubscribe:
        local_rank %= max(torch.cuda.device_count(), 1)
    pg_backend = "gloo" if args.comm == "lib" else args.backend
    if use_dist:
        import datetime
        torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            kw = dict(device_id=torch.device("cuda", local_rank)) if pg_backend == "nccl" else {}
            dist.init_process_group(pg_backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300), **kw)
    n_gpus = max(world, 1)

    qgd = import_package()

    def uid():
        """a fresh RCCL id from rank 0 (one per communicator), carried to the other ranks by the process group"""
        t = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            t = torch.frombuffer(bytearray(qgd.comm_unique_id()), dtype=torch.uint8).clone()
        if world > 1:
            if pg_backend == "nccl":
                t = t.cuda(); dist.broadcast(t, 0); t = t.cpu()
            else:
                dist.broadcast(t, 0)
        return bytes(t.numpy().tobytes())

    def max_over_ranks(sec):
        if not use_dist:
            return sec
        t = torch.tensor([sec], device="cuda" if pg_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    prob, ctrl, pcof, target = workload(qgd, args.nsteps, float(args.nsteps))
    order = 8
    if use_dist:
        # ONE problem spread over the ranks (strong scaling), DESIGN.md "Multi-GPU"
        dp = make_partitioned(qgd, args, prob, order, ctrl, target, rank, world, local_rank, uid)
    else:
        dp = qgd.DeviceProblem(prob, order, device=0)
        dp.set_controls(ctrl)
        dp.set_target(target)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Python's cyclic garbage collector is kept out of the timed regions, as `timeit` does: a full collection of this process
    # (torch, scipy and numpy loaded) takes 40-75 ms, and one landing inside 20 reference-shaped calls turned 0.89 ms per
    # call into 2.7-3.3 (scripts/with_history_trace.py: every call 0.88-0.90 ms except the one with the collection in it).
    # Collections run between the regions instead.
    gc.disable()

    def settle():
        gc.collect()

    # The W warm-up steps.  The first one has every phase bracketed by HIP events (it pays the one-time launch costs anyway):
    # it tells which phases exist on this path; the other W-1 are the plain evaluation, exactly what the timed steps are.
    # Which launch gets the live event pair of the timed region: the inverse where the path has one (the dominant kernel of
    # every configuration measured so far) -- NOT the largest phase of this first evaluation, whose build phase also loads
    # code objects (0.4 ms) -- else the largest; the choice is checked against the per-phase breakdown taken AFTER the timed
    # region (`phases_ms_all_events`: all 13 phases bracketed, ~0.17 ms of event records per evaluation; round 3 took it from
    # the warm-up steps, which made them 1.5x as long as a timed step and mostly idle) and the line says so
    # (`roofline.dominant_confirmed`).
    merge = (("sweep_forward2", "sweep_forward"), ("sweep_adjoint2", "sweep_adjoint"))
    dp.set_timing(1)
    dp.discrete_adjoint(pcof)
    first = dp.timings()
    nbr = 1
    dp.set_timing(0)
    inner = getattr(dp, "dp", dp)
    path = inner.operator_path() if hasattr(inner, "operator_path") else ("sparse", 0, 0)
    model = phase_model(prob.N_tot_levels, prob.N_initial_conditions, order // 2, prob.N_operators, args.nsteps + 1,
                        sparse_ops=(path[0] == "sparse"), fused_propagator=("propagator" not in first))
    dom_raw = next((k for k in ("front", "inverse") if k in first and k in model), None) or max((k for k in first if k in model), key=first.get)
    for _ in range(max(args.warmup, 1) - nbr + args.run_in):
        dp.discrete_adjoint(pcof)
    phase_ms = {}
    step_times = [] if os.environ.get("QGD_BENCH_STEP_TIMES") else None      # (diagnostic: wall time of every timed step on stderr)
    settle()
    barrier()
    t0 = time.perf_counter()
    nsamp = 0
    # The live sample of the dominant kernel (an event pair on the library's stream around its launch) is not free: per-step
    # wall times (QGD_BENCH_STEP_TIMES=1) show +30 us on the bracketed evaluation (the two event records drain the queue
    # between kernels) and +30 us for reading the pair back (stream synchronisation + elapsed-time query).  Round 3
    # bracketed every 8th step and read inside the loop: 9 us per step on average over 20 steps.  Now: one bracketed step
    # per 16 (the middle one of a 20-step region), read back only when the next bracketed step is about to overwrite the
    # pair -- the last one after the timed region.
    pending = False

    def read_sample():
        for k, v in dp.timings().items():
            phase_ms[k] = phase_ms.get(k, 0.0) + v

    for i in range(args.steps):
        sample = (i % 16 == min(10, args.steps // 2))
        if sample and pending:
            read_sample(); pending = False
        if sample or (i > 0 and (i - 1) % 16 == min(10, args.steps // 2)):
            dp.set_timing(2 if sample else 0, dom_raw)
        grad, out3 = dp.discrete_adjoint(pcof)
        if step_times is not None:
            step_times.append(time.perf_counter())
        if sample:
            nsamp += 1
            pending = True
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if pending:
        read_sample(); pending = False
    # per-phase breakdown: three evaluations with every phase bracketed (untimed, after the timed region)
    breakdown = {}
    dp.set_timing(1)
    for _ in range(3):
        dp.discrete_adjoint(pcof)
        for k, v in dp.timings().items():
            breakdown[k] = breakdown.get(k, 0.0) + v / 3
    dp.set_timing(0)
    if step_times:
        print("[bench] us per timed step:", " ".join(f"{(b - a) * 1e6:.0f}" for a, b in zip([t0] + step_times[:-1], step_times)), file=sys.stderr)
    part_info = getattr(inner, "partition", None)
    # the collectives of one evaluation, timed by HIP events on the library's stream (--comm lib)
    comm_ms = None
    if use_dist and args.comm == "lib":
        dp.set_timing(1)
        acc = {}
        for _ in range(5):
            dp.discrete_adjoint(pcof)
            for k, v in dp.timings().items():
                if k.startswith("comm_"):
                    acc[k] = acc.get(k, 0.0) + v / 5
        comm_ms = {k: round(v, 4) for k, v in acc.items()}
        dp.set_timing(0)
    # N > 1, beside the strong-scaling `value`: the same evaluation on a time grid that grows with the rank count
    # (nsteps = 550 per GPU, dt unchanged), i.e. per-GPU work fixed -- where a 0.44 ms evaluation has no more
    # per-rank latency to give, this is what the time-window partition is for.  Reported as `weak_in_time`.
    weak = None
    if use_dist and args.shard == "time" and (world > 1 or os.environ.get("QGD_BENCH_WEAK")):
        try:
            dp.close()
            nsteps_w = args.nsteps * world
            prob_w, ctrl_w, pcof_w, target_w = workload(qgd, nsteps_w, float(nsteps_w))
            dpw = make_partitioned(qgd, args, prob_w, order, ctrl_w, target_w, rank, world, local_rank, uid)
            dpw.set_timing(0)
            for _ in range(max(args.warmup, 2)):
                dpw.discrete_adjoint(pcof_w)
            settle()
            barrier()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                dpw.discrete_adjoint(pcof_w)
            barrier()
            tw = max_over_ranks(time.perf_counter() - t2)
            weak = {"nsteps": nsteps_w, "value": nsteps_w * args.steps / tw, "unit": "timesteps/s",
                    "ms_per_step": tw / args.steps * 1e3, "scaling": "weak"}
            dpw.close()
        except Exception as exc:      # the secondary number must never cost the headline one
            weak = {"error": repr(exc)}
    # Beside `value` (EXACTLY the contract's K steps right behind its W warm-up steps): the same loop once the card's clocks
    # have settled.  A timed region of 20 evaluations after 5 warm-up evaluations ends ~9 ms after the GPU left idle, inside
    # its clock ramp (scripts/clock_profile.py: 345 us per evaluation in the first 7 ms of work, 328-335 afterwards); an
    # optimisation runs thousands of evaluations back to back.  Reported as `settled`, never as `value`.
    settled = None
    if not use_dist:
        try:
            dp.set_timing(0)
            for _ in range(40):
                dp.discrete_adjoint(pcof)
            settle()
            barrier()
            ts_ = time.perf_counter()
            nset = 200
            for _ in range(nset):
                dp.discrete_adjoint(pcof)
            barrier()
            sset = (time.perf_counter() - ts_) / nset
            settled = {"ms_per_step": sset * 1e3, "value": args.nsteps / sset, "unit": "timesteps/s", "evaluations": nset,
                       "note": "same loop, 40 further untimed evaluations then 200 timed ones: the rate of a long optimisation run (clocks settled)"}
        except Exception as exc:
            settled = {"error": repr(exc)}
    # forward-only (eval_forward: tables .. history, guard, overlaps), reported beside the metric
    fwd_elapsed = None
    if not use_dist:
        dp.set_timing(0)
        dp.eval_forward(pcof)
        settle()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            dp.eval_forward(pcof)
        barrier()
        fwd_elapsed = time.perf_counter() - t1

    # The reference-shaped call: optimize_gate hands discrete_adjoint! its preallocated state_history, lambda_history
    # and adjoint_forcing on every iteration (ipopt_optimal_control.jl:207-214, :304-330).  The same evaluation with
    # the three arrays downloaded in the reference layout (31.6 MB), registered (pinned) once as INTEGRATION.md's shim
    # does, and unregistered (pageable).  Reported beside `value`, never as `value`.
    with_hist = None
    if not use_dist and not args.no_with_history:
        try:
            n2, m1, ntp, cc = prob.real_system_size, 1 + order // 2, 1 + prob.nsteps, prob.N_initial_conditions
            with_hist = {"bytes_per_evaluation": 8 * (n2 * m1 * ntp * cc + 2 * n2 * ntp * cc)}
            for label, pinned in (("pinned", True), ("pageable", False)):
                hist = np.zeros((n2, m1, ntp, cc), order="F"); lam = np.zeros((n2, m1, ntp, cc), order="F")
                forc = np.zeros((n2, ntp, cc), order="F")
                if pinned:
                    for a in (hist, lam, forc):
                        dp.pin(a)
                dp.set_timing(0)
                for _ in range(max(args.warmup, 3)):     # (first calls: staging buffers, first-touch of the host pages)
                    dp.discrete_adjoint(pcof, False, hist, lam, forc)
                settle()
                barrier()
                t2 = time.perf_counter()
                for _ in range(args.steps):
                    g_h, _ = dp.discrete_adjoint(pcof, False, hist, lam, forc)
                barrier()
                with_hist[label + "_ms_per_step"] = (time.perf_counter() - t2) / args.steps * 1e3
                assert np.abs(g_h - grad).max() <= 1e-12 * np.abs(grad).max()
                del hist, lam, forc
        except Exception as exc:
            with_hist = {"error": repr(exc)}

    # N > 1: the OTHER split of the same evaluation in the same line.  `value` is the default split (time windows: the one
    # that can pay); BASELINE.json's north star sketches column blocks + all-reduce -- the reference's
    # own thread axis (src/forward_evolution.jl:48,332).  One scaling run of the driver answers both: the second split is
    # timed the same way (W warm-up, K timed steps, barrier + max over ranks, its own communicator) and reported as
    # `north_star_split` (or `time_window_split` when --shard columns made the columns the headline).
    other_split = None
    if use_dist and (world > 1 or args.force_dist) and not args.no_other_split:
        other = "columns" if args.shard == "time" else "time"
        try:
            if other == "columns" and world > prob.N_initial_conditions:
                raise ValueError(f"{world} ranks for {prob.N_initial_conditions} initial-condition columns")
            try:
                dp.close()
            except Exception:
                pass
            saved = args.shard
            args.shard = other
            try:
                dpo = make_partitioned(qgd, args, prob, order, ctrl, target, rank, world, local_rank, uid)
            finally:
                args.shard = saved
            dpo.set_timing(0)
            for _ in range(max(args.warmup, 2)):
                dpo.discrete_adjoint(pcof)
            settle()
            barrier()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                g_o, _ = dpo.discrete_adjoint(pcof)
            barrier()
            to = max_over_ranks(time.perf_counter() - t3)
            acc = {}
            if args.comm == "lib":
                dpo.set_timing(1)
                for _ in range(5):
                    dpo.discrete_adjoint(pcof)
                    for k, v in dpo.timings().items():
                        if k.startswith("comm_"):
                            acc[k] = acc.get(k, 0.0) + v / 5
                dpo.set_timing(0)
            dpo.close()
            other_split = {"shard": other, "value": args.nsteps * args.steps / to, "unit": "timesteps/s", "ms_per_step": to / args.steps * 1e3,
                           "scaling": "strong", "steps": args.steps,
                           "parallelism": (f"column blocks over {n_gpus} GPUs, 2 all-reduces per evaluation (step matrices built on every rank)" if other == "columns"
                                           else f"time windows over {n_gpus} GPUs, 2 all-gathers + 1 all-reduce per evaluation"),
                           "collectives_ms": {k: round(v, 4) for k, v in acc.items()} or None,
                           "grad_rel_diff_vs_headline_split": float(np.abs(g_o - grad).max() / np.abs(grad).max())}
        except Exception as exc:      # the secondary number must never cost the headline one
            other_split = {"shard": other, "error": repr(exc)}

    # N > 1: C5 under the same partition (every rank takes part; secondary number, must never cost the headline one)
    large_dist = None
    if use_dist and world > 1 and not args.no_large_n:
        try:
            large_dist = large_n_partitioned(qgd, np, args, rank, world, local_rank, uid)
        except Exception as exc:
            large_dist = {"error": repr(exc)}

    if rank == 0:
        for k in phase_ms:
            phase_ms[k] /= nsamp
        # the dominant kernel among the single-launch phases, timed live in the timed region
        dom = dom_raw
        timed = {dom: phase_ms[dom_raw]}
        bound, work = model[dom]
        for extra, base in merge:
            if extra in breakdown:
                breakdown[base] = breakdown.get(base, 0.0) + breakdown.pop(extra)
        if bound == "mfma":
            achieved = work / (timed[dom] * 1e-3) / 1e12
            peak, unit = PEAK_FP64_MATRIX_TFLOPS, "TFLOP/s"
        else:
            achieved = work / (timed[dom] * 1e-3) / 1e9
            peak, unit = PEAK_HBM_GBS, "GB/s"
        # N > 1: ONE evaluation is spread over the ranks (strong scaling)
        total_timesteps = args.nsteps * args.steps
        sec_eval = elapsed / args.steps
        # what the committed profiles say about the dominant kernel belongs to the profiled workload only (550 steps on one GPU)
        profiled = (args.nsteps == 550 and not use_dist)
        traffic, traffic_source = measured_traffic(dom) if profiled else (None, None)
        prof_ms, prof_src = profiled_launch_ms(dom) if profiled else (None, None)
        N_, c_, m_ = prob.N_tot_levels, prob.N_initial_conditions, order // 2
        # the only HBM-proportional term of the ALGORITHM is the history stream, written once and read
        # once: B_step = 2*16*N*(1+m)*c + 2*16*N*c bytes per timestep
        b_step = 2 * 16 * N_ * (1 + m_) * c_ + 2 * 16 * N_ * c_
        hs = b_step * args.nsteps / sec_eval / 1e9
        blocks = part_info["blocks"] if part_info else min(64, int(round(args.nsteps ** (2.0 / 3.0))))
        blen = part_info["block_len"] if part_info else -(-args.nsteps // max(blocks, 1))
        gflop = executed_gflop(N_, c_, m_, prob.N_operators, args.nsteps, -(-args.nsteps // blen), blen, path[0] == "sparse", front=("front" in first))
        if args.comm == "lib":
            how = "RCCL collectives issued inside libqgd_hip.so (qgd_comm_init_rccl)"
        else:
            how = f"phase hooks + torch.distributed ({args.backend}" + (")" if args.backend == "nccl" else ", host-staged: plumbing check, not a measurement)")
        out = {
            "metric": "forward+adjoint timesteps/sec, cnot3 order-8 fp64",
            "value": total_timesteps / elapsed, "unit": "timesteps/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "run_in_steps": args.run_in,
            "ms_per_step": sec_eval * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "python_gc": "collected between the timed regions, disabled inside them (as timeit does): one full collection is 40-75 ms",
            "config": {"workload": "cnot3 dispersive CNOT (4,4,4)/(2,2,2), N=64, 8 columns, 3 controls x 60 coeffs, "
                                   f"Hermite order 8, tf={args.nsteps}, nsteps={args.nsteps}, one full discrete_adjoint! per step",
                       "parallelism": "1 GPU" if not use_dist else
                                      ((f"time windows over {n_gpus} GPUs, 2 all-gathers + 1 all-reduce per evaluation" if args.shard == "time"
                                        else f"column blocks over {n_gpus} GPUs, 2 all-reduces per evaluation") + "; " + how)},
            "roofline": {"kernel": KERNEL_OF_PHASE.get(dom, dom), "phase": dom, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source, "mfma_pipe_busy_pmc": measured_mfma_util(dom) if profiled else None,
                         "launch_ms": timed[dom], "launches_timed": nsamp, "algorithmic_work_per_launch": work,
                         "launch_ms_profile": prof_ms, "launch_ms_profile_source": prof_src,
                         "launch_ms_note": "launch_ms: HIP event pair on the library's stream around the launch, inside the timed region -- it also covers "
                                           "the dispatch of the grid behind the drained queue and the end-of-kernel signal (5-8 us); the rocprofv3 "
                                           "figure is first wave to last wave.  frac uses launch_ms",
                         # (single-launch phases only: the sweeps are several launches each)
                         "dominant_confirmed": bool(max((k for k in breakdown if k in model and not k.startswith("sweep")), key=breakdown.get, default=dom) == dom),
                         "history_stream": {"bound": "hbm", "bytes_per_timestep": b_step, "achieved": hs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                            "frac": hs / PEAK_HBM_GBS, "note": "SURVEY 8(d) B_step * nsteps / T_eval: the algorithm's history stream "
                                                                               "(write once, read once) against HBM; the evaluation is latency-bound, not stream-bound"},
                         "whole_evaluation": {"bound": "mfma", "executed_gflop": gflop, "achieved": gflop / sec_eval / 1e3, "peak": PEAK_FP64_MATRIX_TFLOPS,
                                              "unit": "TFLOP/s", "frac": gflop / sec_eval / 1e3 / PEAK_FP64_MATRIX_TFLOPS,
                                              "note": "flops the kernels execute (bench.py executed_gflop), all phases, / wall time of one evaluation"}},
            "phases_ms_all_events": {k: round(v, 4) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1])},
            "operator_path": path[0],
            "settled": settled,
            "collectives_ms": comm_ms,
            ("north_star_split" if args.shard == "time" else "time_window_split"): other_split,
            "weak_in_time": weak,
            "with_history_ms_per_step": None if not with_hist else with_hist.get("pinned_ms_per_step"),
            "with_history": with_hist,
            "forward_only_timesteps_per_s": (total_timesteps / fwd_elapsed) if fwd_elapsed else None,
            "infidelity": float(1 - (out3[0] ** 2 + out3[1] ** 2) / prob.N_ess_levels ** 2),
            "grad_norm": float(np.linalg.norm(grad)),
        }
        if not use_dist:
            dp.close()
            # (round 4 tried these two BEFORE the headline, to start its timed region on a card that had been working: `value`
            #  did not move -- 0.3328 against 0.3314 ms -- and the pinned downloads of `with_history` behind a C5 evaluation in
            #  the same process took 3.3-3.9 ms instead of 0.87; the order of round 3 stays)
            if not args.no_cnot2:
                try:
                    out["cnot2"] = cnot2_case_gpu(qgd, np)
                except Exception as exc:
                    out["cnot2"] = {"error": repr(exc)}
            if not args.no_large_n:
                try:                      # a secondary measurement must never cost the headline line
                    out["large_n"] = large_n_case(qgd, np)
                except Exception as exc:
                    out["large_n"] = {"error": repr(exc)}
        if large_dist is not None:
            out["large_n"] = large_dist
        if not args.no_cpu_baseline:
            orc = import_oracle()
            out["cpu_baseline"] = cpu_baseline(qgd, orc)
        print(json.dumps(out))
    try:
        dp.close()
    except Exception:
        pass
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
