/*
 * qgd.h -- C ABI of libqgd_hip.so: the MI355X-native Hermite time stepper and
 * discrete-adjoint gradient that stands behind the reference's Julia methods
 *
 *   eval_forward / eval_forward!      src/forward_evolution.jl:15-70
 *   discrete_adjoint / discrete_adjoint!   src/eval_grad_discrete_adjoint.jl:83-160
 *   infidelity_real / guard_penalty_real   src/infidelity.jl:7-18, :56-96
 *   (as called by optimize_gate, src/ipopt_optimal_control.jl:243-346)
 *
 * The reference has no FFI seam on this path (it is plain Julia dispatch); its
 * only FFI precedent is the Fortran ccall of src/Controls/FortranBSpline.jl:257-265.
 * These entry points are what a Julia `ccall` shim for the path binds -- see
 * INTEGRATION.md for that shim.  Plain pointers and sizes only; every matrix
 * argument is float64, column-major (Julia layout).  The caller owns every host
 * buffer; the library copies during the call and keeps no host pointer.  Device
 * memory belongs to the handle.  One handle = one host thread = one GPU.
 *
 * Every function returns 0 on success or a QGD_ERR_* code; the message is
 * available from qgd_last_error().  All "file:line" citations are relative to
 * the reference checkout.
 */
#ifndef QGD_H
#define QGD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the last word of qgd_problem_desc, `reserved` in version 1, is the flags word (QGD_CREATE_*): a caller built against
   version 1 that left garbage there would now be misread, hence the bump (round 4 made the change without one). */
#define QGD_ABI_VERSION 2

enum {
    QGD_OK = 0,
    QGD_ERR_ARGUMENT = 1,    /* the reference's ArgumentError (SchrodingerProb.jl:73-154) */
    QGD_ERR_NO_DEVICE = 2,   /* no usable GPU / HIP failure: the path never falls back to the CPU */
    QGD_ERR_STATE = 3,       /* call order (e.g. gradient before target/controls were set) */
    QGD_ERR_UNSUPPORTED = 4, /* size outside what the kernels cover */
    QGD_ERR_NUMERIC = 5,     /* singular step matrix */
    QGD_ERR_MEMORY = 6,      /* the problem does not fit in device memory (see qgd_set_memory_budget) */
    QGD_ERR_COMM = 7         /* RCCL not loadable / a collective failed (qgd_comm_*) */
};

typedef struct qgd_handle_s *qgd_handle;

/* Replaces the SchrodingerProb container (src/SchrodingerProb.jl:25-165). */
typedef struct qgd_problem_desc {
    int32_t N;                  /* N_tot_levels (complex dimension) */
    int32_t n_cols;             /* N_initial_conditions */
    int32_t n_ops;              /* N_operators */
    int32_t n_ess;              /* N_ess_levels */
    int32_t order;              /* Hermite order, even, 2..QGD_MAX_ORDER */
    int32_t nsteps;
    double  tf;
    const double *system_sym;   /* N x N, symmetric          (real part of H) */
    const double *system_asym;  /* N x N, antisymmetric      (imag part of H) */
    const double *sym_ops;      /* n_ops consecutive N x N, symmetric */
    const double *asym_ops;     /* n_ops consecutive N x N, antisymmetric */
    const double *u0, *v0;      /* N x n_cols */
    const double *guard;        /* 2N x 2N guard_subspace_projector, or NULL (= zeros) */
    int32_t device;             /* HIP device ordinal */
    int32_t reserved;           /* flags: 0, or QGD_CREATE_DEFER_GRID */
} qgd_problem_desc;

/* qgd_problem_desc.reserved, bit 0: do not allocate the time grid in qgd_create.  For a handle whose layout is about to
 * change -- a rank of a time partition (qgd_comm_init_rccl / qgd_set_partition), a memory budget, another nsteps -- so that
 * it never holds the WHOLE grid first (the reason to shard by time is that the whole grid does not fit).  The first entry
 * point that needs the grid allocates it. */
#define QGD_CREATE_DEFER_GRID 1

#define QGD_MAX_ORDER 16
#define QGD_MAX_OPS   8

/* SchrodingerProb constructor + validation.  On failure *out is NULL and
 * qgd_last_error(NULL) holds the message.
 * Limits of this version (QGD_ERR_UNSUPPORTED / QGD_ERR_ARGUMENT beyond them): N <= 592 (the state panels are padded to
 * 16 rows), at most QGD_MAX_OPS control operators, even orders 2 .. QGD_MAX_ORDER, any number of columns and of time
 * steps (a grid longer than 65 000 steps, or larger than the memory budget, is processed in windows: qgd_set_memory_budget). */
int qgd_create(const qgd_problem_desc *desc, qgd_handle *out);

/* The same constructor for sparse operators: SchrodingerProb accepts SparseMatrixCSC
 * (src/SchrodingerProb.jl:24-31; DispersiveProblem defaults to sparse_rep=true,
 * src/ProblemConstructors/multi_qudit_systems.jl:118-162).  A qgd_csc is the three arrays of a
 * SparseMatrixCSC{Float64,Int64} (colptr: n+1 entries; rowval, nzval: nnz entries) with
 * index_base = 1 (Julia) or 0 (scipy).  desc->system_sym .. asym_ops are ignored; u0, v0, guard stay
 * dense.  The operators feed the sparse (ELL) kernels directly; stored zeros are kept out of the pattern. */
typedef struct qgd_csc {
    const int64_t *colptr, *rowval;
    const double *nzval;
    int32_t index_base, reserved;
} qgd_csc;
int qgd_create_csc(const qgd_problem_desc *desc, const qgd_csc *system_sym, const qgd_csc *system_asym,
                   const qgd_csc *sym_ops /* n_ops */, const qgd_csc *asym_ops /* n_ops */, qgd_handle *out);
void qgd_destroy(qgd_handle h);
const char *qgd_last_error(qgd_handle h);
int qgd_abi_version(void);

/* Scripts mutate prob.nsteps (examples/cnot3_optimize_gate.jl:51-52): cheap
 * re-parameterisation; invalidates control tables and histories. */
int qgd_set_nsteps(qgd_handle h, int32_t nsteps, double tf);

/* Long time grids in bounded memory.  The step matrices of a grid (~330 KB per time step at cnot3, 18 MB at N = 256)
 * stay resident when they fit; when they do not -- the reference's low-order runs have 10^4 .. 10^6 steps
 * (examples/cnot3_optimize_gate.sb:27-40) and keep no matrices at all -- the grid is processed in windows that share one
 * set of buffers: a forward pass over the windows (only the state at each window boundary is kept) and an adjoint pass
 * back over them that forms a window's matrices and forward history again before differentiating it.  Results equal the
 * resident evaluation to rounding; the price is build + inverse + forward history once more.
 * bytes = 0 (default): 70 % of the device memory that is free when the grid is allocated; a grid whose single time step
 * does not fit returns QGD_ERR_MEMORY.  Re-allocates the grid (set the control basis afterwards, for the WHOLE grid).
 * With more than one window the reference-layout outputs (uv_history, lambda_history, adjoint_forcing) are filled window
 * by window into the caller's full arrays -- uv_history with qgd_set_save_every's stride, lambda_history with its
 * derivative columns after qgd_set_lambda_derivatives, as on a resident grid; qgd_eval_adjoint walks the windows in reverse
 * with the caller's terminal condition and forcing (no forward history is formed); qgd_eval_forward_forced walks them in
 * order with its slice of the caller's forcing; qgd_set_control_tables keeps the tables of the whole grid on the host and
 * every window uploads its slice; qgd_eval_grad_forced forms every window's matrices and history again from its stored
 * start state and carries the sensitivities from window to window.  Only the diagnostics qgd_apply_hamiltonian and
 * qgd_get_intermediate return QGD_ERR_UNSUPPORTED (they address one resident grid).
 * qgd_get_partition on such a handle reports the WHOLE grid (first point 0, last point nsteps): the windows are the
 * library's business, the caller's control basis and output arrays cover every time point.
 * qgd_get_memory_plan: out4 = { windows, time steps per window, bytes of the per-window buffers, budget (0 = automatic) }. */
int qgd_set_memory_budget(qgd_handle h, size_t bytes);
int qgd_get_memory_plan(qgd_handle h, int64_t *out4);

/* Target gate in the stacked real form [Re; Im], 2N x n_cols -- what both call
 * sites build with vcat(real, imag) (eval_grad_discrete_adjoint.jl:126,
 * ipopt_optimal_control.jl:218). */
int qgd_set_target(qgd_handle h, const double *target_real);

/* Controls, linear fast path.  Every control family in scope is linear in pcof,
 * so fill_p_mat!/fill_q_mat! (src/Controls/Control.jl:125-149) over the whole
 * time grid is one matrix-vector product with a basis that depends only on the
 * grid.  For control k (k < n_ops):  Gp[k], Gq[k] are C-ordered
 * [nsteps+1][order/2+1][n_coeff[k]] with
 *     Gp[k][n][d][l] = d/dpcof_l ( p_k^(d)(t_n) / d! ),   t_n = n*tf/nsteps.
 * The basis is uploaded once; afterwards an evaluation ships only pcof. */
int qgd_set_control_basis(qgd_handle h, const int32_t *n_coeff,
                          const double *const *Gp, const double *const *Gq);

/* Controls, general path (any AbstractControl, linear in pcof or not): the tables themselves for the
 * current pcof, Julia layout [(1+order/2), n_ops, nsteps+1] (the fill_p_mat!
 * output stacked over time points).  For a gradient pair them with qgd_set_control_basis holding
 * the Jacobian of the tables AT the current pcof (eval_grad_p_derivative! / eval_grad_q_derivative! over
 * the grid), set the basis first, and call qgd_eval_forward / qgd_discrete_adjoint / qgd_eval_grad_forced /
 * qgd_eval_adjoint with pcof = NULL: the device then steps with these tables and contracts its per-time-point
 * gradient scalars with that Jacobian.  (With a non-NULL pcof the tables are basis * pcof: linear controls.) */
int qgd_set_control_tables(qgd_handle h, const double *p_tables, const double *q_tables);

/* eval_forward! (forward_evolution.jl:33-70).  pcof may be NULL when tables were
 * set directly.  uv_history (nullable) receives the reference layout
 * [2N, 1+order/2, 1+nsteps, n_cols].  out3 = { <w_N,R>, <w_N,T>, guard penalty }
 * (the first two are 0 when no target is set); infidelity =
 * 1 - (out3[0]^2 + out3[1]^2)/n_ess^2  (infidelity.jl:7-18). */
int qgd_eval_forward(qgd_handle h, const double *pcof, int32_t n_pcof,
                     double *uv_history, double *out3);

/* eval_forward's keyword saveEveryNsteps (src/forward_evolution.jl:15-19, :104, :178, :239-241): after
 * qgd_set_save_every(h, s) the uv_history of qgd_eval_forward / qgd_eval_forward_forced is
 * [2N, 1+order/2, 1 + nsteps / s, n_cols] and slot k holds time point k * s (the device keeps every time point; the
 * re-layout kernel reads them with a stride).  The histories of qgd_discrete_adjoint are not affected -- the reference's
 * discrete_adjoint! has no such keyword.  Default 1. */
int qgd_set_save_every(qgd_handle h, int32_t save_every_nsteps);

/* discrete_adjoint! (eval_grad_discrete_adjoint.jl:107-160): gradient of
 * infidelity + guard penalty (no ridge term, as the reference).  With
 * history_precomputed != 0 the forward sweep of the last evaluation is reused
 * (ipopt_optimal_control.jl:297-308) -- the device's own copy of it, and only when
 * it was computed from this same pcof (otherwise the sweep is redone); uv_history
 * is an output in both cases.  Optional outputs (nullable):
 *   uv_history      [2N, 1+m, 1+nsteps, n_cols]
 *   lambda_history  [2N, 1+m, 1+nsteps, n_cols]  (column j=0 filled: the only one
 *                   the reference consumes, eval_grad_discrete_adjoint.jl:604;
 *                   all 1+m columns after qgd_set_lambda_derivatives(h, 1))
 *   adjoint_forcing [2N, 1+nsteps, n_cols]       (:732-752)
 */
int qgd_discrete_adjoint(qgd_handle h, const double *pcof, int32_t n_pcof,
                         int32_t history_precomputed, double *grad, double *uv_history,
                         double *lambda_history, double *adjoint_forcing, double *out3);

/* cost_type of discrete_adjoint! / eval_grad_forced (src/eval_grad_discrete_adjoint.jl:26-35,
 * src/eval_grad_forced.jl:155-165): what the final state is measured by.  :Infidelity (default) is the gate
 * infidelity against the target; :Tracking is 0.5 |w_N - target|^2 and :Norm is 0.5 |w_N|^2 over the stacked real
 * states of all columns (the reference marks both untested; here they are tested like :Infidelity -- adjoint against
 * forced and centred differences, and against the oracle).  With a cost type other than :Infidelity the scalars
 * out3 of qgd_eval_forward / qgd_discrete_adjoint are { cost, 0, guard penalty }.  Anything else: QGD_ERR_ARGUMENT
 * (the reference throws "Invalid cost type"). */
#define QGD_COST_INFIDELITY 0
#define QGD_COST_TRACKING   1
#define QGD_COST_NORM       2
int qgd_set_cost_type(qgd_handle h, int32_t cost_type);

/* Host buffers of the optional outputs.  The reference's optimize_gate allocates state_history, lambda_history and
 * adjoint_forcing once and hands the same arrays to discrete_adjoint! on every iteration
 * (src/ipopt_optimal_control.jl:223-241, :304-330).  Registering such an array pins it, so that the downloads run at
 * PCIe speed beside the adjoint sweep; unregistered buffers work too (pageable copies).  A registered
 * lambda_history is zero-filled once, at its first use: the library only ever writes its j = 0 columns.
 * Unregister before the array is freed. */
int qgd_register_host_buffer(qgd_handle h, void *ptr, size_t bytes);
int qgd_unregister_host_buffer(qgd_handle h, void *ptr);

/* The derivative columns lambda_history[:, 2:end, :, :] as the reference leaves them: eval_adjoint! stores, beside
 * lambda_n, the m adjoint derivatives the explicit side of step n was built from (src/forward_evolution.jl:427-433,
 * :471-480; compute_adjoint_derivatives!, src/hermite.jl:284-305) -- evaluated with the controls at t_{n-1} for time
 * index n >= 2 and at t_1 for n = 1; index 0 is never written.  Nothing downstream reads them
 * (eval_grad_discrete_adjoint.jl:604 takes column 1 only), so they are off by default: on != 0 makes
 * qgd_discrete_adjoint and qgd_eval_adjoint fill them too (one more kernel, and 1+m times the lambda download). */
int qgd_set_lambda_derivatives(qgd_handle h, int32_t on);

/* eval_adjoint (src/forward_evolution.jl:300-315, per column :352-483): backward sweep from a
 * caller-given terminal lambda_N [2N, n_cols] with optional forcing [2N, 1+nsteps, n_cols];
 * lambda_history [2N, 1+m, 1+nsteps, n_cols] receives lambda_n in column j=0 (time index 0 stays
 * zero, as in the reference), and the derivative columns after qgd_set_lambda_derivatives(h, 1). */
int qgd_eval_adjoint(qgd_handle h, const double *pcof, int32_t n_pcof, const double *terminal_condition,
                     const double *forcing, double *lambda_history);

/* Unit-test hook for the Hamiltonian application (hermite.jl:556-588) batched
 * over columns: out = A_d(t_n) * in  (or -A_d = A_d^T with use_adjoint), using
 * the control tables currently on the device.  in/out: 2N x n_cols. */
int qgd_apply_hamiltonian(qgd_handle h, int32_t time_index, int32_t deriv_order,
                          int32_t use_adjoint, const double *in, double *out);

/* Small problems in four launches.  N <= 4 levels, <= 4 initial conditions, 1..4 control operators, order <= 12, <= 64
 * coefficients, no dense guard projector, <= 128 time points: the reference's Rabi oscillator and its two-qubit CNOT
 * (examples/cnot2_optimization.jl).  For such a problem qgd_discrete_adjoint / qgd_eval_forward WITHOUT output arrays run
 * on the fp64 vector ALU with one thread per (time point, column) -- csrc/qgd_k_tiny.hip: front (tables, step matrices,
 * inverses: one workgroup per time point), scan (both sweeps and lambda: one workgroup), gradient scalars (one workgroup
 * per time point), fixed-order sum -- instead of twelve dependent launches of padded 16 x 16 MFMA tiles: same discrete
 * quantities (1e-12 of the general path, 1e-10 of the oracle: tests/test_gpu_tiny.py; the same bits on every run).  Calls
 * with output arrays, event bracketing (qgd_set_timing), windows or a communicator take the general path; a
 * history_precomputed call after a small-path evaluation simply redoes the sweep.  on = 0 turns the path off for this
 * handle (QGD_TINY=0 in the environment: for every handle). */
int qgd_set_small_path(qgd_handle h, int32_t on);

/* Diagnostics: copy a named intermediate of the last evaluation to the host.
 * Names: "L", "R", "Linv", "P" (complex, returned as [nt][N][N][2] C-order),
 * "sigma" ([nt][n_ops][m][2]), "tables" ([nt][m][n_ops][2]), "repivoted" (1 value: how many step matrices of the last
 * evaluation were redone with full partial pivoting -- N > 64: after the block Gauss-Jordan inverse, which pivots inside its
 * 64-column diagonal blocks only, found a block multiplier above its threshold; N = 64: matrices not done by the diagonal-pivot
 * attempt + 65536 * (of these, done by the fully pivoted last resort): the low field wraps beyond 65535 such matrices),
 * "selection" (4 values: operator path -- 2 sparse ELL kernels, 1 the N > 64 GEMM-style kernels, 0 dense N <= 64 --,
 * form of the gradient scalars on the N > 64 path 0..3 or -1, block Gauss-Jordan inverse in use 0/1, windows of the time
 * grid), "small_path" (1 value: whether the last evaluation ran on the small-problem path of qgd_set_small_path), "front_path"
 * (1 value: whether the last forward evaluation took the fused front, below).  Returns the number of doubles the buffer needs
 * through *needed when out == NULL.
 *
 * The fused front (N = 64 with sparse operators, Hermite order <= 8, 513 .. 704 time points, one rank, the grid resident, a
 * diagonal guard projector or none, pcof through the control basis): full evaluations of qgd_eval_forward / qgd_discrete_adjoint use the
 * same-point step propagators S_n = R_n L_n^-1 (csrc/qgd_front.h); state history, lambda, forcing, scalars and gradient are the
 * same quantities as ever.  "L", "R", "Linv", "P" are always the two-point form's: asked for after such an evaluation they
 * make the library redo the forward evaluation on the general path first.
 *
 * Pivoting history of the N = 64 inverse (both paths): an evaluation in which the diagonal-pivot attempt was given up for more
 * than a quarter of the step matrices makes the next 32 evaluations of the handle START with pivoting inside the diagonal tiles;
 * then the diagonal is tried again.  In-tile pivoting rounds differently from diagonal pivots, so the same pcof can return
 * results that differ in the last bits (1e-13 relative) before, during and after those 32 evaluations; within one regime the
 * results are the same bits from call to call. */
int qgd_get_intermediate(qgd_handle h, const char *name, double *out, size_t capacity,
                         size_t *needed);

/* ---- time-partitioned evaluation over several GPUs (one handle per rank), phase by phase ----------------
 * (The hooks the in-library protocol of qgd_comm_init_rccl is made of; a host that brings its own transport --
 * MPI.jl, torch.distributed -- drives them itself.)
 * The reference's only parallel axis is the column loop (Threads.@threads,
 * src/forward_evolution.jl:48,332); this implementation's parallel axis is time, so ranks own
 * contiguous windows of the time grid.  With THESE entry points the library does not communicate: between the
 * phases the caller all-gathers two exchange buffers and all-reduces one (torch.distributed in
 * bench.py --comm torch; MPI.jl from Julia).  (qgd_comm_init_rccl further down is the other route: the
 * library issues the same collectives itself, over RCCL.)  Sequence per evaluation, on every rank:
 *   qgd_dist_forward_begin -> all_gather(buffer 0) -> qgd_dist_forward_end
 *   qgd_dist_adjoint_begin -> all_gather(buffer 1) -> qgd_dist_adjoint_end
 *   all_reduce_sum(buffer 2) -> qgd_dist_finish
 * After qgd_set_partition the control basis must be given for the rank's own window of time
 * points (qgd_get_partition: out8 = first, last global time point, blocks, blocks per rank,
 * block length, rank, world, global number of time points). */
int qgd_set_partition(qgd_handle h, int32_t rank, int32_t world);
int qgd_get_partition(qgd_handle h, int32_t *out8);
/* Run all device work of the handle on the caller's HIP stream (e.g. torch's current stream). */
int qgd_set_stream(qgd_handle h, void *hip_stream);
/* which: 0 window propagators, per rank [planes | panel] of the product of its step matrices
 * (all-gather, 4 N^2 doubles per rank), 1 window affine parts of the adjoint, per rank [phi | y_N]
 * (all-gather), 2 gradient + {<w,R>, <w,T>, guard, 0} (all-reduce sum), 3 the three scalars alone (column shards).
 * Sizes in doubles. */
int qgd_exchange_buffer(qgd_handle h, int32_t which, void **dev_ptr, size_t *total_doubles,
                        size_t *own_offset, size_t *own_doubles);
int qgd_dist_forward_begin(qgd_handle h, const double *pcof, int32_t n_pcof);
int qgd_dist_forward_end(qgd_handle h);
int qgd_dist_adjoint_begin(qgd_handle h);
int qgd_dist_adjoint_end(qgd_handle h);
int qgd_dist_finish(qgd_handle h, double *grad, double *out3);

/* ---- column-sharded evaluation over several GPUs (one handle per rank) -- the reference's own parallel axis
 * (Threads.@threads over initial conditions, src/forward_evolution.jl:48,332).  Each rank creates its handle from ITS
 * columns of u0, v0 and of the target, with the GLOBAL n_ess.  Only the overlaps in the terminal condition couple the
 * columns (infidelity.jl:13-17).  Sequence per evaluation, on every rank:
 *   qgd_cols_forward -> all_reduce_sum(buffer 3: <w_N,R>, <w_N,T>, guard)
 *   qgd_cols_adjoint(h, rank == 0) -> all_reduce_sum(buffer 2: gradient + scalars) -> qgd_dist_finish
 * (keep_scalars != 0 on exactly one rank: the scalars are already global when the second reduction sums them).
 * The step matrices are built on every rank; see DESIGN.md section 6 for when this split pays. */
int qgd_cols_forward(qgd_handle h, const double *pcof, int32_t n_pcof);
int qgd_cols_adjoint(qgd_handle h, int32_t keep_scalars);

/* ---- several GPUs behind ONE call: RCCL over xGMI inside the library ------------------------------------------
 * The reference spreads an evaluation over host threads (Threads.@threads over initial conditions,
 * src/forward_evolution.jl:48,332) and combines the columns through the global overlaps of the terminal condition
 * (src/infidelity.jl:13-17, src/eval_grad_discrete_adjoint.jl:26-28).  Here one process (or host thread) per GPU holds
 * one handle; after qgd_comm_init_rccl the calls
 *     qgd_discrete_adjoint   qgd_eval_forward
 * are COLLECTIVE: every rank makes them with the same pcof, the library runs its share and issues the collectives
 * itself (ncclAllGather / ncclAllReduce on the handle's stream between its phases, no host synchronisation in
 * between), and every rank receives the full gradient and the global scalars.  The host only has to carry 128 bytes
 * once: rank 0 calls qgd_comm_unique_id and hands the id to the other ranks by whatever it has (MPI.jl bcast, a
 * socket, a shared file).  An id serves ONE communicator (every handle that gets one needs a fresh id, as with
 * ncclGetUniqueId).
 *   shard = QGD_SHARD_TIME     contiguous windows of the time grid per rank (default split: 2 all-gathers + 1
 *                              all-reduce per evaluation).  Implies qgd_set_partition(rank, world):
 *                              set the control basis AFTERWARDS, for the rank's own window (qgd_get_partition).
 *   shard = QGD_SHARD_COLUMNS  the reference's axis: the handle was created from the rank's block of columns of
 *                              u0, v0 (and gets its columns of the target), with the GLOBAL n_ess; 2 all-reduces.
 * The optional outputs of qgd_discrete_adjoint / qgd_eval_forward then cover what the rank owns (its window of time
 * points / its columns).  world = 1 is allowed (the collectives still run).  librccl.so.1 is bound at run time
 * (dlopen; QGD_RCCL_LIB names another file): QGD_ERR_COMM when it cannot be loaded or a collective fails.
 * qgd_comm_info: out3 = {rank (-1 without a communicator), world, shard}. */
#define QGD_UNIQUE_ID_BYTES 128
#define QGD_SHARD_TIME    0
#define QGD_SHARD_COLUMNS 1
int qgd_comm_unique_id(void *id128);
int qgd_comm_init_rccl(qgd_handle h, const void *unique_id, int32_t rank, int32_t world, int32_t shard);
int qgd_comm_destroy(qgd_handle h);
int qgd_comm_info(qgd_handle h, int32_t *out3);
/* Failure mode of the collective calls.  A collective that one rank never enters would block the others inside an RCCL
 * kernel for good (the reference's thread loop has no such state: a failing column throws out of Threads.@threads,
 * src/forward_evolution.jl:48).  Therefore: the one host wait of a collective qgd_discrete_adjoint / qgd_eval_forward is
 * bounded (default 30 000 ms, or this setter); when it expires, when RCCL
 * reports an asynchronous error, or when THIS rank fails between two collectives (HIP / launch / memory / RCCL error),
 * the library aborts the handle's communicator (ncclCommAbort -- RCCL's kernels leave the stream), and the call returns
 * QGD_ERR_COMM with the cause in qgd_last_error.  The handle then has no communicator (qgd_comm_info: rank -1): the host
 * ends the job or calls qgd_comm_init_rccl again with a fresh id on every rank.  Argument and call-order errors are
 * raised before anything is launched, and a singular step matrix is summed into the reductions: those fail on every
 * rank alike, with their own codes, and leave the communicator intact.
 * (The tests inject a local failure in front of each exchange with a hook that is NOT part of this library:
 * tests/hooks/qgd_test_hooks.cpp.) */
int qgd_set_comm_timeout(qgd_handle h, double milliseconds);

/* Per-phase device time of the last evaluation (HIP events), milliseconds.
 * names/ms hold up to cap entries; returns the number of phases through *n. */
int qgd_get_timings(qgd_handle h, const char **names, float *ms, int32_t cap, int32_t *n);
/* Event bracketing costs ~0.17 ms per evaluation on cnot3 (26 event records): mode 0 = off,
 * (default), 1 = every phase, 2 = only the named phase.
 * Switching off keeps the times of the last bracketed evaluation readable (qgd_get_timings). */
int qgd_set_timing(qgd_handle h, int32_t mode, const char *phase);


/* eval_forward(prob, controls, pcof; forcing) (src/forward_evolution.jl:15-70, forcing path :118-129,
 * :167-206): w' = A w + F with the scaled Taylor coefficients of F given at every time point,
 * forcing[2N, order/2, 1+nsteps, n_cols] (column-major).  Any N (N <= 64: the pipelined scan kernels; larger: a plain
 * affine chain kernel, correct but not tuned -- this is a cross-check path); single GPU, resident time grid. */
int qgd_eval_forward_forced(qgd_handle h, const double *pcof, int32_t n_pcof, const double *forcing,
                            double *uv_history, double *out3);

/* eval_grad_forced (src/eval_grad_forced.jl:17-194): the same gradient by forward sensitivities --
 * one forced forward sweep per control parameter, all parameters batched as extra column groups of
 * the blocked scan.  The reference's cross-check of the discrete adjoint (agreement to rounding).
 * Needs qgd_set_control_basis and qgd_set_target; any N (see above); single GPU.  pcof may be NULL when the tables were set
 * directly (general control path: the basis then holds the Jacobian at the current pcof).  Honours qgd_set_cost_type. */
int qgd_eval_grad_forced(qgd_handle h, const double *pcof, int32_t n_pcof, double *grad);

/* Operator path of the step-matrix and gradient kernels.  mode 0: automatic (sparse when every
 * row of the assembled Hamiltonian has at most min(16, N/2) entries and N <= 64 -- the drift +
 * a_k +/- a_k^dagger operators of src/multi_qudit_systems.jl -- else dense), 1: dense fp64 MFMA
 * kernels, 2: sparse (ELL) kernels, QGD_ERR_UNSUPPORTED when the operators do not qualify.
 * qgd_get_operator_path: out3 = {path in use (1 dense, 2 sparse), entries per row of the union
 * pattern, entries per row of the widest control operator}. */
int qgd_set_operator_path(qgd_handle h, int32_t mode);
int qgd_get_operator_path(qgd_handle h, int32_t *out3);

#ifdef __cplusplus
}
#endif
#endif
