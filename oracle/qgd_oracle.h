/*
 * qgd_oracle.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C restatement of the reference's Hermite time stepper and
 * discrete-adjoint gradient (leespen1/QuantumGateDesign.jl @ 2024-12-20).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped path (libqgd_hip.so) never links or calls it.
 *
 * Parity pinning: the reference is Julia (+ un-vendored Julia packages) and
 * cannot run in the build container, and its tests hold no golden vectors.
 * The oracle is therefore pinned by (1) the closed-form known-answer matrices
 * of test/hardcoded_derivatives.jl, (2) Pade/expm and Rabi closed forms,
 * (3) the reference's own three-way gradient agreement property
 * (adjoint == forced <= 1e-14, == finite differences <= 1e-9,
 * test/GradientTests/compare_gradients.jl:47-65), (4) observed convergence
 * order, and (5) the reference's Fortran B-spline routines compiled into
 * oracle/_ref (the only part of the reference that can be built here).
 * Bitwise parity with a Julia run is UNPINNED; parity to solver tolerance is
 * pinned.  See DESIGN.md "Oracle".
 *
 * All matrices are column-major (Julia layout).  "file:line" citations are
 * relative to /root/reference.
 */
#ifndef QGD_ORACLE_H
#define QGD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- controls (src/Controls, all control files) ------------------------------------- */
enum { QO_CTRL_GRAPE = 0, QO_CTRL_BSPLINE = 1, QO_CTRL_CARRIER = 2,
       QO_CTRL_BSPLINE2 = 3,    /* hard-coded quadratic B-spline, bspline_control.jl:21-249: n_basis = D1, n_coeff = 2 D1 */
       QO_CTRL_BCARRIER2 = 4 }; /* BSplineControl = Juqbox bcarrier2 layout, bspline_control.jl:251-395 + bspline_backend.jl:381-480,
                                   :637-770, :783-955: n_basis = D1, n_freq/freqs = omega, n_coeff = 2 D1 n_freq; derivative orders 0, 1 */

typedef struct qo_control {
    int32_t kind;
    int32_t n_coeff;
    double  tf;
    /* GRAPE: grape_control.jl:18-28 */
    int32_t n_amplitudes;
    /* BSPLINE (clamped, uniform knots): FortranBSpline.jl:17-60,
       GeneralBSplineControl.jl:1-19 (same spline space) */
    int32_t degree;
    int32_t n_basis;
    /* CARRIER: CarrierControl.jl:5-24 */
    int32_t n_freq;
    const double *freqs;
    const struct qo_control *base;
} qo_control;

double qo_eval_p_derivative(const qo_control *c, double t, const double *pcof, int order);
double qo_eval_q_derivative(const qo_control *c, double t, const double *pcof, int order);
void   qo_eval_grad_p_derivative(const qo_control *c, double t, const double *pcof, int order, double *grad);
void   qo_eval_grad_q_derivative(const qo_control *c, double t, const double *pcof, int order, double *grad);
/* vals[d] = p^(d)(t)/d!, d = 0..nvals-1 (Control.jl:99-122, FortranBSpline.jl:86-147) */
void   qo_fill_p_vec(const qo_control *c, double t, const double *pcof, int nvals, double *vals);
void   qo_fill_q_vec(const qo_control *c, double t, const double *pcof, int nvals, double *vals);
/* all B-splines of order k=degree+1 not vanishing at x in [0,1], and their
   derivatives: out[i + k*d], i<k, d<nderiv.  Returns 0-based index of the
   first non-vanishing basis function. */
int    qo_bspline_basis_derivs(int degree, int n_basis, double x, int nderiv, double *out);

/* ---- problem (src/SchrodingerProb.jl:25-165) --------------------------- */
enum { QO_PRECOND_IDENTITY = 0, QO_PRECOND_DIAGONAL = 1 };

typedef struct qo_prob {
    int32_t N;         /* N_tot_levels (complex dimension) */
    int32_t n_ops;     /* N_operators */
    int32_t n_cols;    /* N_initial_conditions */
    int32_t n_ess;     /* N_ess_levels */
    int32_t nsteps;
    int32_t precond;   /* preconditioner type parameter P */
    double  tf;
    double  gmres_abstol, gmres_reltol;
    const double *system_sym, *system_asym;  /* N x N */
    const double *sym_ops, *asym_ops;        /* n_ops x (N x N) */
    const double *u0, *v0;                   /* N x n_cols */
    const double *guard;                     /* 2N x 2N */
    /* Optional sparse operators (SparseMatrixCSC, the reference's default sparse_rep=true of DispersiveProblem,
       src/ProblemConstructors/multi_qudit_systems.jl:118-162): 2 + 2 n_ops matrices in the order system_sym,
       system_asym, sym_ops[0..], asym_ops[0..], 0-based indices.  NULL = dense mul! as above. */
    const struct qo_csc *csc;
} qo_prob;

typedef struct qo_csc {
    const int64_t *colptr, *rowval;
    const double *nzval;
} qo_csc;

typedef struct qo_stats {
    double fwd_gmres_iters;   /* mean iterations per step per column */
    double adj_gmres_iters;
    int64_t applies;          /* apply_hamiltonian! calls */
} qo_stats;

/* ---- hermite.jl -------------------------------------------------------- */
double qo_coefficient(int j, int p, int q);                               /* :389-391 */
/* out += (+/-) A_d in ; tables are (1+m) x n_ops, column-major          :556-588 */
void qo_apply_hamiltonian(const qo_prob *pr, const double *pvals, const double *qvals,
                          int ld, int deriv_order, int use_adjoint,
                          const double *in, double *out);
/* uv: 2N x (1+m); forcing (nullable): 2N x m                             :56-101 */
void qo_compute_derivatives(const qo_prob *pr, const double *pvals, const double *qvals,
                            int m, const double *forcing, double *uv);
/* exponential transposed recursion                                       :225-305 */
void qo_compute_adjoint_derivatives(const qo_prob *pr, const double *pvals, const double *qvals,
                                    int m, double *uv);

/* ---- forward_evolution.jl ---------------------------------------------- */
/* history: [2N, 1+m, 1+nsteps, n_cols]; forcing (nullable): [2N, m, 1+nsteps, n_cols]
   (:33-70, :88-245) */
int qo_eval_forward(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                    int order, const double *forcing, double *history, qo_stats *st);
/* lambda_history: [2N,1+m,1+nsteps,n_cols]; terminal: [2N,n_cols];
   forcing (nullable): [2N, 1+nsteps, n_cols]   (:318-483) */
int qo_eval_adjoint(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                    int order, const double *terminal, const double *forcing,
                    double *lambda_history, qo_stats *st);

/* ---- infidelity.jl ----------------------------------------------------- */
double qo_infidelity_real(int N, int n_cols, const double *psi, const double *target, int n_ess); /* :7-18 */
double qo_guard_penalty_real(const qo_prob *pr, int m, const double *history);                    /* :56-96 */

/* ---- eval_grad_discrete_adjoint.jl ------------------------------------- */
void qo_compute_guard_forcing(const qo_prob *pr, int m, const double *history, double *forcing_out); /* :732-752 */
int  qo_compute_terminal_condition(const qo_prob *pr, const qo_control *const *controls,
                                   const double *pcof, int order, const double *target_real,
                                   const double *final_state, const double *forcing_end,
                                   double *terminal_out);                                           /* :1-67 */
/* test hook: one recursive_magic! call (eval_grad_discrete_adjoint.jl:656-726), see qgd_oracle.c */
void qo_recursive_magic(const qo_prob *pr, const qo_control *const *controls, const double *pcof, int m,
                        int control_index, double t, const double *w_mat, const double *lambda, int deriv_order,
                        double coeff, double *grad_contrib);
/* grad[P]; history/lambda_history/adjoint_forcing are caller buffers     :107-160 */
int  qo_discrete_adjoint(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                         int n_pcof, const double *target_real, int order, int history_precomputed,
                         double *grad, double *history, double *lambda_history,
                         double *adjoint_forcing, qo_stats *st);

/* ---- test oracles: eval_grad_forced.jl:17-194, eval_grad_finite_difference.jl:17-72 */
int  qo_eval_grad_forced(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                         int n_pcof, const double *target_real, int order, double *grad);
int  qo_eval_grad_finite_difference(const qo_prob *pr, const qo_control *const *controls,
                                    const double *pcof, int n_pcof, const double *target_real,
                                    int order, double dpcof, double *grad);

void qo_set_num_threads(int n);
void qo_set_converged_terminal(int on);
void qo_set_parallel_gradient(int on);   /* test switch: gradient accumulation with the columns on threads, bit-identical */
void qo_set_cost_type(int type);   /* 0 :Infidelity, 1 :Tracking, 2 :Norm */

#ifdef __cplusplus
}
#endif
#endif
