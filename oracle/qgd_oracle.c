/*
 * qgd_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 * See qgd_oracle.h for the role of this file and how it is pinned.
 *
 * Restates, function by function, the operation structure of the reference:
 * one column at a time, un-assembled Hamiltonian, matrix-free GMRES per step,
 * exponential transposed recursion in the adjoint, per-operator recursive
 * gradient accumulation.  It is therefore also the "port" CPU timing baseline.
 * Citations are file:line under /root/reference.
 */
#include "qgd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0;
void qo_set_num_threads(int n) { g_threads = n; }

/* 0 (default): the one-shot terminal solve uses IterativeSolvers' gmres! defaults as the
 * reference does (restart = min(20, 2N), maxiter = 2N; eval_grad_discrete_adjoint.jl:61-62),
 * which silently stops short of the tolerance on stiff problems (cnot3 at dt = 1).
 * 1: full-length Krylov space and 20*2N iterations, so that the terminal condition is
 * solved to the requested tolerance (used when the oracle serves as the 1e-10 yardstick). */
static int g_converged_terminal = 0;
void qo_set_converged_terminal(int on) { g_converged_terminal = on; }

/* 0 (default): the gradient accumulation runs over the columns one after the other, as the reference does
 * (eval_grad_discrete_adjoint.jl:148-157) -- this is also what the cpu_baseline times.
 * 1 (large test problems only): the columns go to the threads of qo_set_num_threads, each into its OWN zero-initialised
 * gradient vector, and the vectors are added in column order.  The serial loop does grad = grad - contrib_c per control
 * and column, and 0 - contrib_c is exact, so the result is bit-identical to the serial loop
 * (tests/test_oracle.py::test_parallel_gradient_columns_are_bit_identical). */
static int g_parallel_gradient = 0;
void qo_set_parallel_gradient(int on) { g_parallel_gradient = on; }

/* cost_type of discrete_adjoint / eval_grad_forced / eval_grad_finite_difference
 * (eval_grad_discrete_adjoint.jl:26-35, eval_grad_forced.jl:155-165, eval_grad_finite_difference.jl:48-59):
 * 0 :Infidelity (default), 1 :Tracking (0.5 |w_N - target|^2), 2 :Norm (0.5 |w_N|^2). */
static int g_cost_type = 0;
void qo_set_cost_type(int type) { g_cost_type = type; }

static double factorial_d(int n) { double f = 1.0; for (int i = 2; i <= n; i++) f *= i; return f; }
static double binomial_d(int n, int k) { return factorial_d(n) / (factorial_d(k) * factorial_d(n - k)); }

/* ======================================================================== */
/* B-splines.  Clamped uniform knot vector on [0,1] (FortranBSpline.jl:28-60).
 * Values and derivatives of the k=degree+1 basis functions that do not vanish
 * at x, by the Cox-de Boor triangle and the derivative recurrence of
 * de Boor, "A Practical Guide to Splines" ch. X (the same mathematics the
 * reference's pppack bsplvd.f implements; validated against it in
 * tests/test_oracle_bspline.py through oracle/_ref). */
/* ======================================================================== */
#define QO_MAX_ORDER 24

static double knot_at(int idx, int k, int n_distinct)
{
    /* full knot vector: (k-1) copies of 0, LinRange(0,1,n_distinct), (k-1) copies of 1 */
    int j = idx - (k - 1);
    if (j <= 0) return 0.0;
    if (j >= n_distinct - 1) return 1.0;
    return (double)j / (double)(n_distinct - 1);
}

int qo_bspline_basis_derivs(int degree, int n_basis, double x, int nderiv, double *out)
{
    const int p = degree, k = degree + 1;
    const int n_knots = n_basis + k;
    const int n_distinct = n_knots - 2 * (k - 1);
    /* interval selection: FortranBSpline.jl:268-277 (1-based 'left') */
    int left1 = (int)floor(x * (n_distinct - 1) + k);
    if (left1 > n_knots - k) left1 = n_knots - k;
    const int i = left1 - 1; /* 0-based span index: knots[i] <= x <= knots[i+1] */

    double ndu[QO_MAX_ORDER][QO_MAX_ORDER];
    double left[QO_MAX_ORDER], right[QO_MAX_ORDER];
    double a[2][QO_MAX_ORDER];

    ndu[0][0] = 1.0;
    for (int j = 1; j <= p; j++) {
        left[j]  = x - knot_at(i + 1 - j, k, n_distinct);
        right[j] = knot_at(i + j, k, n_distinct) - x;
        double saved = 0.0;
        for (int r = 0; r < j; r++) {
            ndu[j][r] = right[r + 1] + left[j - r];
            double temp = ndu[r][j - 1] / ndu[j][r];
            ndu[r][j] = saved + right[r + 1] * temp;
            saved = left[j - r] * temp;
        }
        ndu[j][j] = saved;
    }
    for (int d = 0; d < nderiv; d++)
        for (int j = 0; j < k; j++) out[j + k * d] = 0.0;
    for (int j = 0; j <= p; j++) out[j] = ndu[j][p];

    const int nd = (nderiv - 1 < p) ? nderiv - 1 : p; /* derivatives above the degree vanish */
    for (int r = 0; r <= p; r++) {
        int s1 = 0, s2 = 1;
        a[0][0] = 1.0;
        for (int d = 1; d <= nd; d++) {
            double acc = 0.0;
            const int rd = r - d, pd = p - d;
            if (r >= d) {
                a[s2][0] = a[s1][0] / ndu[pd + 1][rd];
                acc = a[s2][0] * ndu[rd][pd];
            }
            const int j1 = (rd >= -1) ? 1 : -rd;
            const int j2 = (r - 1 <= pd) ? d - 1 : p - r;
            for (int j = j1; j <= j2; j++) {
                a[s2][j] = (a[s1][j] - a[s1][j - 1]) / ndu[pd + 1][rd + j];
                acc += a[s2][j] * ndu[rd + j][pd];
            }
            if (r <= pd) {
                a[s2][d] = -a[s1][d - 1] / ndu[pd + 1][r];
                acc += a[s2][d] * ndu[r][pd];
            }
            out[r + k * d] = acc;
            int tmp = s1; s1 = s2; s2 = tmp;
        }
    }
    double fac = p;
    for (int d = 1; d <= nd; d++) {
        for (int j = 0; j <= p; j++) out[j + k * d] *= fac;
        fac *= (p - d);
    }
    return i - p; /* index of first non-vanishing basis function */
}

/* ======================================================================== */
/* Controls                                                                  */
/* ======================================================================== */
static int grape_region(const qo_control *c, double t)
{   /* grape_control.jl:82-99 */
    double width = c->tf / c->n_amplitudes;
    int idx = (int)floor(t / width);
    if (idx > c->n_amplitudes - 1) idx = c->n_amplitudes - 1;
    if (idx < 0) idx = 0;
    return idx;
}

static double bspline_eval(const qo_control *c, double t, const double *pcof, int order, int is_q)
{   /* FortranBSpline.jl:69-84 (p), :103-119 (q) */
    const int k = c->degree + 1;
    if (order >= k) return 0.0;
    double basis[QO_MAX_ORDER * QO_MAX_ORDER];
    int first = qo_bspline_basis_derivs(c->degree, c->n_basis, t / c->tf, order + 1, basis);
    const double *co = pcof + first + (is_q ? c->n_coeff / 2 : 0);
    double val = 0.0;
    for (int i = 0; i < k; i++) val += co[i] * basis[i + k * order];
    return val / pow(c->tf, order);
}

static void carrier_factors(double w, double t, int kk, int is_q, double *c1, double *c2)
{   /* CarrierControl.jl:49-61 (p) and :78-90 (q): kk-th derivative of cos/sin carriers */
    double cs = cos(w * t) * pow(w, kk), sn = sin(w * t) * pow(w, kk);
    switch (kk & 3) {
    case 0: *c1 = is_q ? sn : cs;   *c2 = is_q ? cs : -sn;  break;
    case 1: *c1 = is_q ? cs : -sn;  *c2 = is_q ? -sn : -cs; break;
    case 2: *c1 = is_q ? -sn : -cs; *c2 = is_q ? -cs : sn;  break;
    default:*c1 = is_q ? -cs : sn;  *c2 = is_q ? sn : cs;   break;
    }
}

/* The three non-zero pieces of the hard-coded quadratic B-spline at t and their first two derivatives
 * (bspline2 / gradbspline2!, bspline_control.jl:139-249; the same formulas inside bcarrier2, bspline_backend.jl:796-826).
 * idx[s] = 0-based coefficient index of piece s (nurbs k, k-1, k-2), w[s][d] = d-th derivative of its segment. */
static void bspline2_pieces(int D1, double tf, double t, int idx[3], double w[3][3])
{
    const double dtknot = tf / (D1 - 2), width = 3.0 * dtknot;
    int k = (int)ceil(t / dtknot + 2.0);          /* :149-150: t = 0 must give k = 3; protect against round-off at t = tf */
    if (k < 3) k = 3;
    if (k > D1) k = D1;
    for (int s = 0; s < 3; s++) {
        const int kk = k - s;                      /* 1-based nurb index */
        const double tc = dtknot * ((double)kk - 1.5);   /* tcenter, :39 */
        const double tau = (t - tc) / width;
        idx[s] = kk - 1;
        if (s == 0)      { w[s][0] = 9.0 / 8 + 4.5 * tau + 4.5 * tau * tau; w[s][1] = (4.5 + 9 * tau) / width;  w[s][2] = 9.0 / (width * width); }
        else if (s == 1) { w[s][0] = 0.75 - 9 * tau * tau;                  w[s][1] = (-18 * tau) / width;      w[s][2] = -18.0 / (width * width); }
        else             { w[s][0] = 9.0 / 8 - 4.5 * tau + 4.5 * tau * tau; w[s][1] = (-4.5 + 9 * tau) / width; w[s][2] = 9.0 / (width * width); }
    }
}

/* bcarrier2 / bcarrier2_dt with an explicit pcof (bspline_backend.jl:783-848, :862-955), one control (osc = 0,
 * baseIndex = 0): per frequency D1 coefficients of envelope 1 then D1 of envelope 2;
 *   p = sum_f  b1 cos(w t) - b2 sin(w t),   q = sum_f  b1 sin(w t) + b2 cos(w t),  and their time derivatives. */
static double bcarrier2_eval(const qo_control *c, double t, const double *pcof, int order, int is_q)
{
    const int D1 = c->n_basis;
    int idx[3]; double w[3][3];
    bspline2_pieces(D1, c->tf, t, idx, w);
    double f = 0.0;
    for (int fr = 0; fr < c->n_freq; fr++) {
        const double *p1 = pcof + (size_t)fr * 2 * D1, *p2 = p1 + D1;
        const double om = c->freqs[fr], cs = cos(om * t), sn = sin(om * t);
        double b1 = 0, b2 = 0, b1p = 0, b2p = 0;
        for (int s = 0; s < 3; s++) {
            b1 += p1[idx[s]] * w[s][0]; b2 += p2[idx[s]] * w[s][0];
            b1p += p1[idx[s]] * w[s][1]; b2p += p2[idx[s]] * w[s][1];
        }
        if (order == 0) f += is_q ? b1 * sn + b2 * cs : b1 * cs - b2 * sn;                      /* :840-844 */
        else f += is_q ? b1p * sn + b1 * cs * om + b2p * cs - b2 * sn * om                      /* :935-946 */
                       : b1p * cs - b1 * sn * om - b2p * sn - b2 * cs * om;
    }
    return f;
}

/* gradbcarrier2! / gradbcarrier2_dt! (bspline_backend.jl:381-440, :637-730) */
static void bcarrier2_grad(const qo_control *c, double t, int order, int is_q, double *g)
{
    const int D1 = c->n_basis;
    int idx[3]; double w[3][3];
    bspline2_pieces(D1, c->tf, t, idx, w);
    for (int fr = 0; fr < c->n_freq; fr++) {
        double *g1 = g + (size_t)fr * 2 * D1, *g2 = g1 + D1;
        const double om = c->freqs[fr], cs = cos(om * t), sn = sin(om * t);
        for (int s = 0; s < 3; s++) {
            const double bk = w[s][0], bkp = w[s][1];
            if (order == 0) {
                g1[idx[s]] = is_q ? bk * sn : bk * cs;
                g2[idx[s]] = is_q ? bk * cs : -bk * sn;
            } else {
                g1[idx[s]] = is_q ? bkp * sn + bk * om * cs : bkp * cs - bk * om * sn;
                g2[idx[s]] = is_q ? bkp * cs - bk * om * sn : -(bkp * sn + bk * om * cs);
            }
        }
    }
}

static double eval_pq_derivative(const qo_control *c, double t, const double *pcof, int order, int is_q)
{
    switch (c->kind) {
    case QO_CTRL_BSPLINE2: { /* bspline_control.jl:67-86 (p: coefficients 1..D1, q: D1+1..2 D1), bspline2 :139-205 */
        if (order > 2) return 0.0;               /* "If derivative order higher than 2, value is zero" (:202) */
        int idx[3]; double w[3][3];
        bspline2_pieces(c->n_basis, c->tf, t, idx, w);
        const double *co = pcof + (is_q ? c->n_basis : 0);
        return co[idx[0]] * w[0][order] + co[idx[1]] * w[1][order] + co[idx[2]] * w[2][order];
    }
    case QO_CTRL_BCARRIER2: /* bspline_control.jl:310-358: orders 0 and 1 only (the reference throws above that) */
        if (order > 1) return NAN;
        return bcarrier2_eval(c, t, pcof, order, is_q);
    case QO_CTRL_GRAPE: /* grape_control.jl:32-55 */
        if (order > 0) return 0.0;
        return pcof[grape_region(c, t) + (is_q ? c->n_amplitudes : 0)];
    case QO_CTRL_BSPLINE:
        return bspline_eval(c, t, pcof, order, is_q);
    case QO_CTRL_CARRIER: { /* CarrierControl.jl:42-98 */
        double val = 0.0;
        const int npf = c->base->n_coeff;
        for (int f = 0; f < c->n_freq; f++) {
            const double *lp = pcof + (size_t)f * npf;
            for (int kk = 0; kk <= order; kk++) {
                double c1, c2;
                carrier_factors(c->freqs[f], t, kk, is_q, &c1, &c2);
                double b1 = eval_pq_derivative(c->base, t, lp, order - kk, 0);
                double b2 = eval_pq_derivative(c->base, t, lp, order - kk, 1);
                val += binomial_d(order, kk) * (c1 * b1 + c2 * b2);
            }
        }
        return val;
    }
    }
    return 0.0;
}

static void eval_grad_pq_derivative(const qo_control *c, double t, const double *pcof, int order,
                                    int is_q, double *grad)
{
    for (int i = 0; i < c->n_coeff; i++) grad[i] = 0.0;
    switch (c->kind) {
    case QO_CTRL_BSPLINE2: { /* bspline_control.jl:106-131, gradbspline2! :208-270 */
        if (order > 2) return;
        int idx[3]; double w[3][3];
        bspline2_pieces(c->n_basis, c->tf, t, idx, w);
        double *g = grad + (is_q ? c->n_basis : 0);
        for (int s = 0; s < 3; s++) g[idx[s]] = w[s][order];
        return;
    }
    case QO_CTRL_BCARRIER2: /* bspline_control.jl:360-395 */
        if (order > 1) { for (int i = 0; i < c->n_coeff; i++) grad[i] = NAN; return; }
        bcarrier2_grad(c, t, order, is_q, grad);
        return;
    case QO_CTRL_GRAPE: /* grape_control.jl:57-80 */
        if (order == 0) grad[grape_region(c, t) + (is_q ? c->n_amplitudes : 0)] = 1.0;
        return;
    case QO_CTRL_BSPLINE: { /* FortranBSpline.jl:149-189 */
        const int k = c->degree + 1;
        if (order >= k) return;
        double basis[QO_MAX_ORDER * QO_MAX_ORDER];
        int first = qo_bspline_basis_derivs(c->degree, c->n_basis, t / c->tf, order + 1, basis);
        double *g = grad + first + (is_q ? c->n_coeff / 2 : 0);
        double sc = pow(c->tf, order);
        for (int i = 0; i < k; i++) g[i] = basis[i + k * order] / sc;
        return;
    }
    case QO_CTRL_CARRIER: { /* CarrierControl.jl:100-192 */
        const int npf = c->base->n_coeff;
        double *tmp = (double *)malloc(sizeof(double) * npf);
        for (int f = 0; f < c->n_freq; f++) {
            const double *lp = pcof + (size_t)f * npf;
            double *lg = grad + (size_t)f * npf;
            for (int kk = 0; kk <= order; kk++) {
                double c1, c2;
                carrier_factors(c->freqs[f], t, kk, is_q, &c1, &c2);
                double bc = binomial_d(order, kk);
                eval_grad_pq_derivative(c->base, t, lp, order - kk, 0, tmp);
                for (int i = 0; i < npf; i++) lg[i] += tmp[i] * (c1 * bc);
                eval_grad_pq_derivative(c->base, t, lp, order - kk, 1, tmp);
                for (int i = 0; i < npf; i++) lg[i] += tmp[i] * (c2 * bc);
            }
        }
        free(tmp);
        return;
    }
    }
}

double qo_eval_p_derivative(const qo_control *c, double t, const double *pcof, int order)
{ return eval_pq_derivative(c, t, pcof, order, 0); }
double qo_eval_q_derivative(const qo_control *c, double t, const double *pcof, int order)
{ return eval_pq_derivative(c, t, pcof, order, 1); }
void qo_eval_grad_p_derivative(const qo_control *c, double t, const double *pcof, int order, double *grad)
{ eval_grad_pq_derivative(c, t, pcof, order, 0, grad); }
void qo_eval_grad_q_derivative(const qo_control *c, double t, const double *pcof, int order, double *grad)
{ eval_grad_pq_derivative(c, t, pcof, order, 1, grad); }

static void fill_pq_vec(const qo_control *c, double t, const double *pcof, int nvals, double *vals, int is_q)
{
    if (c->kind == QO_CTRL_BSPLINE) {
        /* FortranBSpline.jl:86-101 / :121-147: one basis evaluation for all orders */
        const int k = c->degree + 1;
        double basis[QO_MAX_ORDER * QO_MAX_ORDER];
        int nd = nvals < k ? nvals : k;
        int first = qo_bspline_basis_derivs(c->degree, c->n_basis, t / c->tf, nd, basis);
        const double *co = pcof + first + (is_q ? c->n_coeff / 2 : 0);
        for (int d = 0; d < nvals; d++) {
            double val = 0.0;
            if (d < k) for (int i = 0; i < k; i++) val += co[i] * basis[i + k * d];
            vals[d] = val / (pow(c->tf, d) * factorial_d(d));
        }
        return;
    }
    /* Control.jl:99-122 */
    for (int d = 0; d < nvals; d++) vals[d] = eval_pq_derivative(c, t, pcof, d, is_q) / factorial_d(d);
}
void qo_fill_p_vec(const qo_control *c, double t, const double *pcof, int nvals, double *vals)
{ fill_pq_vec(c, t, pcof, nvals, vals, 0); }
void qo_fill_q_vec(const qo_control *c, double t, const double *pcof, int nvals, double *vals)
{ fill_pq_vec(c, t, pcof, nvals, vals, 1); }

/* Control.jl:125-149: tables (1+m) x n_ops, column-major */
static void fill_pq_mats(const qo_prob *pr, const qo_control *const *controls, double t,
                         const double *pcof, int m, double *pvals, double *qvals)
{
    size_t off = 0;
    for (int k = 0; k < pr->n_ops; k++) {
        qo_fill_p_vec(controls[k], t, pcof + off, 1 + m, pvals + (size_t)k * (1 + m));
        qo_fill_q_vec(controls[k], t, pcof + off, 1 + m, qvals + (size_t)k * (1 + m));
        off += controls[k]->n_coeff;
    }
}

/* ======================================================================== */
/* hermite.jl                                                                */
/* ======================================================================== */
double qo_coefficient(int j, int p, int q)
{   /* hermite.jl:389-391 */
    return factorial_d(p) * factorial_d(p + q - j) / (factorial_d(p + q) * factorial_d(p - j));
}

/* y += alpha * A x, A is n x n column-major (the reference's 5-arg mul!) */
static void gemv_acc(int n, double alpha, const double *A, const double *x, double *y)
{
    if (alpha == 0.0) return;
    for (int j = 0; j < n; j++) {
        const double ax = alpha * x[j];
        const double *col = A + (size_t)j * n;
        for (int i = 0; i < n; i++) y[i] += col[i] * ax;
    }
}

/* the same with a SparseMatrixCSC operand (SparseArrays' mul!: one pass over the stored entries, column by column) */
static void spmv_acc(int n, double alpha, const qo_csc *A, const double *x, double *y)
{
    if (alpha == 0.0) return;
    for (int j = 0; j < n; j++) {
        const double ax = alpha * x[j];
        for (int64_t e = A->colptr[j]; e < A->colptr[j + 1]; e++) y[A->rowval[e]] += A->nzval[e] * ax;
    }
}

static int64_t g_applies = 0;

void qo_apply_hamiltonian(const qo_prob *pr, const double *pvals, const double *qvals, int ld,
                          int d, int use_adjoint, const double *in, double *out)
{   /* hermite.jl:556-588 */
    const int N = pr->N;
    const double af = use_adjoint ? -1.0 : 1.0;
    const double *in_re = in, *in_im = in + N;
    double *out_re = out, *out_im = out + N;
    if (pr->csc) {      /* sparse_rep = true: the same sixteen mul! calls on SparseMatrixCSC operands */
        const qo_csc *ssym = pr->csc, *sasym = pr->csc + 1;
        if (d == 0) {
            spmv_acc(N, af, sasym, in_re, out_re);
            spmv_acc(N, af, ssym, in_im, out_re);
            spmv_acc(N, af, sasym, in_im, out_im);
            spmv_acc(N, -af, ssym, in_re, out_im);
        }
        for (int k = 0; k < pr->n_ops; k++) {
            const qo_csc *sym = pr->csc + 2 + k, *asym = pr->csc + 2 + pr->n_ops + k;
            const double p = pvals[d + (size_t)k * ld], q = qvals[d + (size_t)k * ld];
            spmv_acc(N, af * q, asym, in_re, out_re);
            spmv_acc(N, af * p, sym, in_im, out_re);
            spmv_acc(N, af * q, asym, in_im, out_im);
            spmv_acc(N, -af * p, sym, in_re, out_im);
        }
#ifdef QO_COUNT_APPLIES
        #pragma omp atomic
        g_applies++;
#endif
        return;
    }
    if (d == 0) {
        gemv_acc(N, af, pr->system_asym, in_re, out_re);
        gemv_acc(N, af, pr->system_sym, in_im, out_re);
        gemv_acc(N, af, pr->system_asym, in_im, out_im);
        gemv_acc(N, -af, pr->system_sym, in_re, out_im);
    }
    for (int k = 0; k < pr->n_ops; k++) {
        const double *sym = pr->sym_ops + (size_t)k * N * N;
        const double *asym = pr->asym_ops + (size_t)k * N * N;
        const double p = pvals[d + (size_t)k * ld], q = qvals[d + (size_t)k * ld];
        gemv_acc(N, af * q, asym, in_re, out_re);
        gemv_acc(N, af * p, sym, in_im, out_re);
        gemv_acc(N, af * q, asym, in_im, out_im);
        gemv_acc(N, -af * p, sym, in_re, out_im);
    }
#ifdef QO_COUNT_APPLIES
    #pragma omp atomic
    g_applies++;
#endif
}

void qo_compute_derivatives(const qo_prob *pr, const double *pvals, const double *qvals, int m,
                            const double *forcing, double *uv)
{   /* hermite.jl:56-101 */
    const int n2 = 2 * pr->N;
    for (int j = 0; j < m; j++) {
        double *dst = uv + (size_t)(j + 1) * n2;
        memset(dst, 0, sizeof(double) * n2);
        for (int i = j; i >= 0; i--)
            qo_apply_hamiltonian(pr, pvals, qvals, 1 + m, j - i, 0, uv + (size_t)i * n2, dst);
        if (forcing) for (int r = 0; r < n2; r++) dst[r] += forcing[r + (size_t)j * n2];
        for (int r = 0; r < n2; r++) dst[r] /= (j + 1);
    }
}

/* hermite.jl:225-275.  W, W2: 2N x (1+m) work matrices. */
static void single_adjoint_derivative(const qo_prob *pr, const double *pvals, const double *qvals,
                                      int m, const double *lambda_in, int idx, double *W, double *W2)
{
    const int n2 = 2 * pr->N;
    if (idx == 0) { memcpy(W, lambda_in, sizeof(double) * n2); return; }
    double *acc = W + (size_t)idx * n2;
    double *tmp = W2 + (size_t)idx * n2;
    memset(acc, 0, sizeof(double) * n2);
    for (int d = idx - 1; d >= 0; d--) {
        memset(tmp, 0, sizeof(double) * n2);
        qo_apply_hamiltonian(pr, pvals, qvals, 1 + m, d, 1, lambda_in, tmp);
        const int inner = idx - 1 - d;
        single_adjoint_derivative(pr, pvals, qvals, m, tmp, inner, W, W2);
        const double *src = W + (size_t)inner * n2;
        for (int r = 0; r < n2; r++) acc[r] += src[r];
    }
    for (int r = 0; r < n2; r++) acc[r] /= idx;
}

static void adjoint_derivatives_ws(const qo_prob *pr, const double *pvals, const double *qvals,
                                   int m, double *uv, double *wvec, double *W, double *W2)
{   /* hermite.jl:284-305 */
    const int n2 = 2 * pr->N;
    memcpy(wvec, uv, sizeof(double) * n2);
    for (int i = 1; i <= m; i++) {
        single_adjoint_derivative(pr, pvals, qvals, m, wvec, i, W, W2);
        memcpy(uv + (size_t)i * n2, W + (size_t)i * n2, sizeof(double) * n2);
    }
}

void qo_compute_adjoint_derivatives(const qo_prob *pr, const double *pvals, const double *qvals,
                                    int m, double *uv)
{
    const int n2 = 2 * pr->N;
    double *ws = (double *)malloc(sizeof(double) * n2 * (1 + 2 * (1 + m)));
    adjoint_derivatives_ws(pr, pvals, qvals, m, uv, ws, ws + n2, ws + n2 + (size_t)n2 * (1 + m));
    free(ws);
}

static void build_sum(int n2, int m, double dt, const double *uv, double *out)
{   /* hermite.jl:394-403 (dt>0: RHS) and :418-427 (call with -dt: LHS) */
    memset(out, 0, sizeof(double) * n2);
    for (int j = 0; j <= m; j++) {
        const double cf = pow(dt, j) * qo_coefficient(j, m, m);
        const double *src = uv + (size_t)j * n2;
        for (int r = 0; r < n2; r++) out[r] += cf * src[r];
    }
}

static void taylor_expand(int n2, int m, double dt, const double *uv, double *out)
{   /* hermite.jl:447-457 */
    memset(out, 0, sizeof(double) * n2);
    for (int j = 0; j <= m; j++) {
        const double cf = pow(dt, j) / factorial_d(j);
        const double *src = uv + (size_t)j * n2;
        for (int r = 0; r < n2; r++) out[r] += cf * src[r];
    }
}

/* ======================================================================== */
/* Left-preconditioned restarted GMRES (Saad & Schultz 1986; modified
 * Gram-Schmidt Arnoldi, Givens least squares).  Stands in for the
 * un-vendored IterativeSolvers.jl at the call sites
 * forward_evolution.jl:142-146,:404-408 and
 * eval_grad_discrete_adjoint.jl:61-62.  Results are pinned only to the
 * tolerance.                                    */
/* ======================================================================== */
typedef void (*qo_matvec)(void *ctx, const double *x, double *y);

typedef struct {
    int n;            /* real system size */
    int complex_n;
    const double *diag, *upper, *lower; /* DiagonalHamiltonianPreconditioner, preconditioners.jl:64-131 */
} qo_precond;

static void precond_solve(const qo_precond *P, double *x)
{
    if (!P) return; /* identity, preconditioners.jl:35-40 */
    const int N = P->complex_n;
    for (int i = 0; i < N; i++) { /* preconditioners.jl:107-113 */
        double ratio = P->lower[i] / P->diag[i];
        x[N + i] -= x[i] * ratio;
        x[N + i] /= (P->diag[N + i] - P->upper[i] * ratio);
    }
    for (int i = 0; i < N; i++) { /* :114-119 */
        x[i] -= P->upper[i] * x[N + i];
        x[i] /= P->diag[i];
    }
}

typedef struct {
    int n, restart;
    double *V, *H, *cs, *sn, *g, *w, *y;
} qo_gmres_ws;

static void gmres_ws_init(qo_gmres_ws *ws, int n, int restart)
{
    ws->n = n; ws->restart = restart;
    ws->V = (double *)malloc(sizeof(double) * (size_t)n * (restart + 1));
    ws->H = (double *)malloc(sizeof(double) * (size_t)(restart + 1) * restart);
    ws->cs = (double *)malloc(sizeof(double) * restart);
    ws->sn = (double *)malloc(sizeof(double) * restart);
    ws->g = (double *)malloc(sizeof(double) * (restart + 1));
    ws->w = (double *)malloc(sizeof(double) * n);
    ws->y = (double *)malloc(sizeof(double) * restart);
}
static void gmres_ws_free(qo_gmres_ws *ws)
{ free(ws->V); free(ws->H); free(ws->cs); free(ws->sn); free(ws->g); free(ws->w); free(ws->y); }

static double dotn(int n, const double *a, const double *b)
{ double s = 0.0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }

/* Solves A x = b starting from x.  tol_abs/tol_rel: stop when the
 * (preconditioned) residual norm <= max(tol_rel*|r0|, tol_abs).  Returns the
 * number of Arnoldi steps (= operator applications after the first residual). */
static int gmres_solve(qo_gmres_ws *ws, qo_matvec A, void *ctx, const qo_precond *P, double *x,
                       const double *b, double tol_abs, double tol_rel, int maxiter)
{
    const int n = ws->n, m = ws->restart, ldh = m + 1;
    int iters = 0, first = 1;
    double tol = tol_abs;
    for (;;) {
        double *r = ws->V; /* first basis vector */
        A(ctx, x, ws->w);
        for (int i = 0; i < n; i++) r[i] = b[i] - ws->w[i];
        precond_solve(P, r);
        double beta = sqrt(dotn(n, r, r));
        if (first) { tol = fmax(tol_rel * beta, tol_abs); first = 0; }
        if (beta <= tol || iters >= maxiter) break;
        for (int i = 0; i < n; i++) r[i] /= beta;
        memset(ws->g, 0, sizeof(double) * (m + 1));
        ws->g[0] = beta;
        int k = 0;
        double res = beta;
        while (k < m) {
            double *vk = ws->V + (size_t)k * n, *vn = ws->V + (size_t)(k + 1) * n;
            A(ctx, vk, vn);
            precond_solve(P, vn);
            double *h = ws->H + (size_t)k * ldh;
            for (int i = 0; i <= k; i++) {
                const double *vi = ws->V + (size_t)i * n;
                h[i] = dotn(n, vn, vi);
                for (int r2 = 0; r2 < n; r2++) vn[r2] -= h[i] * vi[r2];
            }
            h[k + 1] = sqrt(dotn(n, vn, vn));
            if (h[k + 1] > 0.0) for (int r2 = 0; r2 < n; r2++) vn[r2] /= h[k + 1];
            for (int i = 0; i < k; i++) {
                double t0 = ws->cs[i] * h[i] + ws->sn[i] * h[i + 1];
                h[i + 1] = -ws->sn[i] * h[i] + ws->cs[i] * h[i + 1];
                h[i] = t0;
            }
            double den = hypot(h[k], h[k + 1]);
            if (den == 0.0) { ws->cs[k] = 1.0; ws->sn[k] = 0.0; }
            else { ws->cs[k] = h[k] / den; ws->sn[k] = h[k + 1] / den; }
            h[k] = ws->cs[k] * h[k] + ws->sn[k] * h[k + 1];
            h[k + 1] = 0.0;
            ws->g[k + 1] = -ws->sn[k] * ws->g[k];
            ws->g[k] = ws->cs[k] * ws->g[k];
            res = fabs(ws->g[k + 1]);
            k++; iters++;
            if (res <= tol || iters >= maxiter) break;
        }
        for (int i = k - 1; i >= 0; i--) {
            double s = ws->g[i];
            for (int j = i + 1; j < k; j++) s -= ws->H[i + (size_t)j * ldh] * ws->y[j];
            ws->y[i] = s / ws->H[i + (size_t)i * ldh];
        }
        for (int j = 0; j < k; j++) {
            const double *vj = ws->V + (size_t)j * n;
            for (int i = 0; i < n; i++) x[i] += ws->y[j] * vj[i];
        }
        if (res <= tol || iters >= maxiter) break;
    }
    return iters;
}

/* ======================================================================== */
/* forward_evolution.jl                                                      */
/* ======================================================================== */
typedef struct {
    const qo_prob *pr;
    int m;
    double dt;
    double *uv;        /* 2N x (1+m) */
    double *pvals, *qvals;
    double *wvec, *W, *W2; /* adjoint recursion workspaces */
} lhs_holder;

static void lhs_apply(void *ctx, const double *x, double *y)
{   /* LHSHolder functor, forward_evolution.jl:583-592 */
    lhs_holder *h = (lhs_holder *)ctx;
    const int n2 = 2 * h->pr->N;
    memcpy(h->uv, x, sizeof(double) * n2);
    qo_compute_derivatives(h->pr, h->pvals, h->qvals, h->m, NULL, h->uv);
    build_sum(n2, h->m, -h->dt, h->uv, y);
}

static void lhs_adjoint_apply(void *ctx, const double *x, double *y)
{   /* LHSHolderAdjoint functor, forward_evolution.jl:624-633 */
    lhs_holder *h = (lhs_holder *)ctx;
    const int n2 = 2 * h->pr->N;
    memcpy(h->uv, x, sizeof(double) * n2);
    adjoint_derivatives_ws(h->pr, h->pvals, h->qvals, h->m, h->uv, h->wvec, h->W, h->W2);
    build_sum(n2, h->m, -h->dt, h->uv, y);
}

static void holder_init(lhs_holder *h, const qo_prob *pr, int m, double dt)
{
    const int n2 = 2 * pr->N;
    h->pr = pr; h->m = m; h->dt = dt;
    h->uv = (double *)calloc((size_t)n2 * (1 + m), sizeof(double));
    h->pvals = (double *)calloc((size_t)(1 + m) * pr->n_ops, sizeof(double));
    h->qvals = (double *)calloc((size_t)(1 + m) * pr->n_ops, sizeof(double));
    h->wvec = (double *)calloc(n2, sizeof(double));
    h->W = (double *)calloc((size_t)n2 * (1 + m), sizeof(double));
    h->W2 = (double *)calloc((size_t)n2 * (1 + m), sizeof(double));
}
static void holder_free(lhs_holder *h)
{ free(h->uv); free(h->pvals); free(h->qvals); free(h->wvec); free(h->W); free(h->W2); }

/* form_LHS_no_control + DiagonalHamiltonianPreconditioner ctor:
 * forward_evolution.jl:772-802, preconditioners.jl:74-94.  (The reference
 * uses A^j without the 1/j! -- kept, it only affects the iteration count.) */
static qo_precond *make_diag_precond(const qo_prob *pr, int order, int adjoint)
{
    const int N = pr->N, n2 = 2 * N, m = order / 2;
    const double dt = pr->tf / pr->nsteps;
    double *A = (double *)calloc((size_t)n2 * n2, sizeof(double));
    double *Pw = (double *)calloc((size_t)n2 * n2, sizeof(double));
    double *T = (double *)calloc((size_t)n2 * n2, sizeof(double));
    double *L = (double *)calloc((size_t)n2 * n2, sizeof(double));
    for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) {
        double K = pr->system_asym[i + (size_t)j * N], S = pr->system_sym[i + (size_t)j * N];
        A[i + (size_t)j * n2] = K;           A[i + (size_t)(j + N) * n2] = S;
        A[i + N + (size_t)j * n2] = -S;      A[i + N + (size_t)(j + N) * n2] = K;
    }
    if (adjoint) { /* A = A' */
        for (int j = 0; j < n2; j++) for (int i = 0; i < j; i++) {
            double t0 = A[i + (size_t)j * n2]; A[i + (size_t)j * n2] = A[j + (size_t)i * n2]; A[j + (size_t)i * n2] = t0;
        }
    }
    for (int i = 0; i < n2; i++) { L[i + (size_t)i * n2] = 1.0; Pw[i + (size_t)i * n2] = 1.0; }
    for (int j = 1; j <= m; j++) {
        /* Pw = Pw * A */
        memset(T, 0, sizeof(double) * n2 * n2);
        for (int c = 0; c < n2; c++) for (int k = 0; k < n2; k++) {
            double a = A[k + (size_t)c * n2];
            if (a != 0.0) for (int r = 0; r < n2; r++) T[r + (size_t)c * n2] += Pw[r + (size_t)k * n2] * a;
        }
        memcpy(Pw, T, sizeof(double) * n2 * n2);
        double cf = pow(-dt, j) * qo_coefficient(j, m, m);
        for (size_t e = 0; e < (size_t)n2 * n2; e++) L[e] += cf * Pw[e];
    }
    qo_precond *P = (qo_precond *)malloc(sizeof(qo_precond));
    double *buf = (double *)malloc(sizeof(double) * (n2 + 2 * N));
    P->n = n2; P->complex_n = N;
    for (int i = 0; i < n2; i++) buf[i] = L[i + (size_t)i * n2];
    for (int i = 0; i < N; i++) { buf[n2 + i] = L[i + (size_t)(i + N) * n2]; buf[n2 + N + i] = L[i + N + (size_t)i * n2]; }
    P->diag = buf; P->upper = buf + n2; P->lower = buf + n2 + N;
    free(A); free(Pw); free(T); free(L);
    return P;
}
static void free_precond(qo_precond *P) { if (P) { free((void *)P->diag); free(P); } }

/* per-column forward sweep: forward_evolution.jl:88-245 */
static double forward_column(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                             int order, int col, const double *forcing, double *hist)
{
    const int N = pr->N, n2 = 2 * N, m = order / 2, nsteps = pr->nsteps;
    const double dt = pr->tf / nsteps;
    const size_t slab = (size_t)n2 * (1 + m);
    lhs_holder h; holder_init(&h, pr, m, dt);
    qo_gmres_ws ws; gmres_ws_init(&ws, n2, n2);                  /* restart = real_system_size, :144 */
    qo_precond *P = pr->precond == QO_PRECOND_DIAGONAL ? make_diag_precond(pr, order, 0) : NULL; /* :140 */
    double *uv = (double *)calloc(slab, sizeof(double));
    double *uvec = (double *)calloc(n2, sizeof(double));
    double *rhs = (double *)calloc(n2, sizeof(double));
    double *fh = forcing ? (double *)calloc(slab + n2, sizeof(double)) : NULL;
    double iters = 0.0;

    memcpy(uv, pr->u0 + (size_t)col * N, sizeof(double) * N);   /* :149-151 */
    memcpy(uv + N, pr->v0 + (size_t)col * N, sizeof(double) * N);
    memcpy(hist, uv, sizeof(double) * slab);
    fill_pq_mats(pr, controls, 0.0, pcof, m, h.pvals, h.qvals); /* :158-160 */

    for (int n = 0; n < nsteps; n++) {
        const double *fn = forcing ? forcing + (size_t)n * n2 * m : NULL;
        qo_compute_derivatives(pr, h.pvals, h.qvals, m, fn, uv);      /* :172-175 */
        memcpy(hist + (size_t)n * slab, uv, sizeof(double) * slab);    /* :177-179 */
        build_sum(n2, m, dt, uv, rhs);                                 /* :181 */
        taylor_expand(n2, m, dt, uv, uvec);                            /* :183-187 */
        double t = (n + 1) * dt;                                       /* :190-193 */
        fill_pq_mats(pr, controls, t, pcof, m, h.pvals, h.qvals);
        if (forcing) {                                                 /* :196-206 */
            memset(fh, 0, sizeof(double) * slab);
            qo_compute_derivatives(pr, h.pvals, h.qvals, m, forcing + (size_t)(n + 1) * n2 * m, fh);
            build_sum(n2, m, -dt, fh, fh + slab);
            for (int r = 0; r < n2; r++) rhs[r] -= fh[slab + r];
        }
        /* :209-220; the iterable's tolerance was fixed at construction from a
           zero residual => abstol only */
        iters += gmres_solve(&ws, lhs_apply, &h, P, uvec, rhs, pr->gmres_abstol, 0.0, n2);
        memcpy(uv, uvec, sizeof(double) * n2);
    }
    /* :231-242 (final derivatives, no forcing) */
    fill_pq_mats(pr, controls, nsteps * dt, pcof, m, h.pvals, h.qvals);
    qo_compute_derivatives(pr, h.pvals, h.qvals, m, NULL, uv);
    memcpy(hist + (size_t)nsteps * slab, uv, sizeof(double) * slab);

    free(uv); free(uvec); free(rhs); free(fh);
    free_precond(P); gmres_ws_free(&ws); holder_free(&h);
    return iters / nsteps;
}

int qo_eval_forward(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                    int order, const double *forcing, double *history, qo_stats *st)
{   /* forward_evolution.jl:33-70 -- Threads.@threads over columns */
    const int n2 = 2 * pr->N, m = order / 2;
    const size_t hcol = (size_t)n2 * (1 + m) * (1 + pr->nsteps);
    const size_t fcol = (size_t)n2 * m * (1 + pr->nsteps);
    double tot = 0.0;
    int nt = g_threads;
#ifdef _OPENMP
    if (nt <= 0) nt = omp_get_max_threads();
#else
    nt = 1;
#endif
    #pragma omp parallel for num_threads(nt) reduction(+:tot) schedule(dynamic, 1)
    for (int col = 0; col < pr->n_cols; col++)
        tot += forward_column(pr, controls, pcof, order, col,
                              forcing ? forcing + (size_t)col * fcol : NULL,
                              history + (size_t)col * hcol);
    if (st) st->fwd_gmres_iters = tot / pr->n_cols;
    return 0;
}

/* per-column adjoint sweep: forward_evolution.jl:352-483 */
static double adjoint_column(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                             int order, const double *terminal, const double *forcing, double *hist)
{
    const int N = pr->N, n2 = 2 * N, m = order / 2, nsteps = pr->nsteps;
    const double dt = pr->tf / nsteps;
    const size_t slab = (size_t)n2 * (1 + m);
    lhs_holder h; holder_init(&h, pr, m, dt);
    qo_gmres_ws ws; gmres_ws_init(&ws, n2, n2);
    qo_precond *P = pr->precond == QO_PRECOND_DIAGONAL ? make_diag_precond(pr, order, 1) : NULL; /* :402 */
    double *uv = (double *)calloc(slab, sizeof(double));
    double *uvec = (double *)calloc(n2, sizeof(double));
    double *rhs = (double *)calloc(n2, sizeof(double));
    double *wvec = (double *)calloc(n2, sizeof(double));
    double *W = (double *)calloc(slab, sizeof(double)), *W2 = (double *)calloc(slab, sizeof(double));
    double iters = 0.0;

    memcpy(uv, terminal, sizeof(double) * n2);                       /* :411-414 */
    memcpy(hist + (size_t)nsteps * slab, uv, sizeof(double) * slab);
    for (int n = nsteps; n >= 2; n--) {                               /* :421 */
        double t = (n - 1) * dt;
        fill_pq_mats(pr, controls, t, pcof, m, h.pvals, h.qvals);     /* :423-425 */
        adjoint_derivatives_ws(pr, h.pvals, h.qvals, m, uv, wvec, W, W2); /* :427-432 */
        memcpy(hist + (size_t)n * slab, uv, sizeof(double) * slab);   /* :433 */
        build_sum(n2, m, dt, uv, rhs);                                /* :434 */
        if (forcing) for (int r = 0; r < n2; r++) rhs[r] += forcing[r + (size_t)(n - 1) * n2]; /* :438-440 */
        memcpy(uvec, uv, sizeof(double) * n2);                        /* :450 */
        iters += gmres_solve(&ws, lhs_adjoint_apply, &h, P, uvec, rhs, pr->gmres_abstol, 0.0, n2);
        memcpy(uv, uvec, sizeof(double) * n2);                        /* :461 */
    }
    fill_pq_mats(pr, controls, dt, pcof, m, h.pvals, h.qvals);        /* :471-480 */
    adjoint_derivatives_ws(pr, h.pvals, h.qvals, m, uv, wvec, W, W2);
    memcpy(hist + slab, uv, sizeof(double) * slab);

    free(uv); free(uvec); free(rhs); free(wvec); free(W); free(W2);
    free_precond(P); gmres_ws_free(&ws); holder_free(&h);
    return iters / nsteps;
}

int qo_eval_adjoint(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                    int order, const double *terminal, const double *forcing,
                    double *lambda_history, qo_stats *st)
{   /* forward_evolution.jl:318-350 */
    const int n2 = 2 * pr->N, m = order / 2;
    const size_t hcol = (size_t)n2 * (1 + m) * (1 + pr->nsteps);
    const size_t fcol = (size_t)n2 * (1 + pr->nsteps);
    double tot = 0.0;
    int nt = g_threads;
#ifdef _OPENMP
    if (nt <= 0) nt = omp_get_max_threads();
#else
    nt = 1;
#endif
    #pragma omp parallel for num_threads(nt) reduction(+:tot) schedule(dynamic, 1)
    for (int col = 0; col < pr->n_cols; col++)
        tot += adjoint_column(pr, controls, pcof, order, terminal + (size_t)col * n2,
                              forcing ? forcing + (size_t)col * fcol : NULL,
                              lambda_history + (size_t)col * hcol);
    if (st) st->adj_gmres_iters = tot / pr->n_cols;
    return 0;
}

/* ======================================================================== */
/* infidelity.jl                                                             */
/* ======================================================================== */
double qo_infidelity_real(int N, int n_cols, const double *psi, const double *target, int n_ess)
{   /* infidelity.jl:7-18 */
    double dR = 0.0, dT = 0.0;
    for (int c = 0; c < n_cols; c++) {
        const double *p = psi + (size_t)c * 2 * N, *R = target + (size_t)c * 2 * N;
        for (int i = 0; i < N; i++) {
            dR += p[i] * R[i] + p[N + i] * R[N + i];
            dT += p[i] * R[N + i] - p[N + i] * R[i];   /* T = [R_im; -R_re] */
        }
    }
    return 1.0 - (dR * dR + dT * dT) / ((double)n_ess * n_ess);
}

double qo_guard_penalty_real(const qo_prob *pr, int m, const double *history)
{   /* infidelity.jl:56-96 */
    const int n2 = 2 * pr->N, nt = 1 + pr->nsteps;
    const size_t slab = (size_t)n2 * (1 + m);
    const double dt = pr->tf / pr->nsteps;
    double *Ww = (double *)malloc(sizeof(double) * n2);
    double total = 0.0;
    for (int c = 0; c < pr->n_cols; c++) {
        double pen = 0.0;
        for (int i = 0; i < nt; i++) {
            const double *w = history + ((size_t)c * nt + i) * slab;
            memset(Ww, 0, sizeof(double) * n2);
            gemv_acc(n2, 1.0, pr->guard, w, Ww);
            double d = dotn(n2, w, Ww);
            pen += (i == 0 || i == nt - 1) ? 0.5 * d : d;
        }
        total += pen * dt / pr->tf;
    }
    free(Ww);
    return total;
}

/* ======================================================================== */
/* eval_grad_discrete_adjoint.jl                                             */
/* ======================================================================== */
void qo_compute_guard_forcing(const qo_prob *pr, int m, const double *history, double *f)
{   /* :732-752 ; f: [2N, 1+nsteps, n_cols] */
    const int n2 = 2 * pr->N, nt = 1 + pr->nsteps;
    const size_t slab = (size_t)n2 * (1 + m);
    const double dt = pr->tf / pr->nsteps;
    for (int c = 0; c < pr->n_cols; c++)
        for (int n = 0; n < nt; n++) {
            double *dst = f + ((size_t)c * nt + n) * n2;
            memset(dst, 0, sizeof(double) * n2);
            gemv_acc(n2, 1.0, pr->guard, history + ((size_t)c * nt + n) * slab, dst);
            double sc = -2.0 * dt / pr->tf;
            if (n == 0 || n == nt - 1) sc *= 0.5;
            for (int r = 0; r < n2; r++) dst[r] *= sc;
        }
}

int qo_compute_terminal_condition(const qo_prob *pr, const qo_control *const *controls,
                                  const double *pcof, int order, const double *R,
                                  const double *final_state, const double *forcing_end,
                                  double *terminal_out)
{   /* :1-67 */
    const int N = pr->N, n2 = 2 * N, m = order / 2, nc = pr->n_cols;
    const double dt = pr->tf / pr->nsteps;
    double dR = 0.0, dT = 0.0;
    for (int c = 0; c < nc; c++) {
        const double *p = final_state + (size_t)c * n2, *r = R + (size_t)c * n2;
        for (int i = 0; i < N; i++) {
            dR += p[i] * r[i] + p[N + i] * r[N + i];
            dT += p[i] * r[N + i] - p[N + i] * r[i];
        }
    }
    const double sc = 2.0 / ((double)pr->n_ess * pr->n_ess);
    lhs_holder h; holder_init(&h, pr, m, dt);
    fill_pq_mats(pr, controls, pr->tf, pcof, m, h.pvals, h.qvals);  /* t = prob.tf, :14 */
    int restart = n2 < 20 ? n2 : 20;                                 /* gmres! default restart */
    int maxiter = n2;                                                /* gmres! default maxiter */
    if (g_converged_terminal) { restart = n2; maxiter = 20 * n2; }
    qo_gmres_ws ws; gmres_ws_init(&ws, n2, restart);
    double *rhs = (double *)malloc(sizeof(double) * n2);
    double *x = (double *)calloc(n2, sizeof(double));                /* uv_vec persists over columns, :20,:61 */
    for (int c = 0; c < nc; c++) {
        const double *r = R + (size_t)c * n2;
        for (int i = 0; i < N; i++) {
            rhs[i]     = sc * (dR * r[i]     + dT * r[N + i]);
            rhs[N + i] = sc * (dR * r[N + i] - dT * r[i]);
        }
        if (g_cost_type == 1) for (int i = 0; i < n2; i++) rhs[i] = -(final_state[i + (size_t)c * n2] - r[i]);   /* :29-30 */
        if (g_cost_type == 2) for (int i = 0; i < n2; i++) rhs[i] = -final_state[i + (size_t)c * n2];            /* :31-32 */
        if (forcing_end) for (int i = 0; i < n2; i++) rhs[i] += forcing_end[i + (size_t)c * n2];
        gmres_solve(&ws, lhs_adjoint_apply, &h, NULL, x, rhs, pr->gmres_abstol, pr->gmres_reltol, maxiter);
        memcpy(terminal_out + (size_t)c * n2, x, sizeof(double) * n2);
    }
    free(rhs); free(x); gmres_ws_free(&ws); holder_free(&h);
    return 0;
}

/* <(dA/dq) w, lam> with K = blockdiag(asym): :764-781 */
static double inner_prod_S(int N, const double *w, const double *lam, const double *asym, double *work)
{
    memset(work, 0, sizeof(double) * 2 * N);
    gemv_acc(N, 1.0, asym, lam, work);
    gemv_acc(N, 1.0, asym, lam + N, work + N);
    return -dotn(2 * N, w, work);
}
/* <(dA/dp) w, lam>: :783-800 */
static double inner_prod_K(int N, const double *w, const double *lam, const double *sym, double *work)
{
    memset(work, 0, sizeof(double) * 2 * N);
    gemv_acc(N, 1.0, sym, lam + N, work);
    gemv_acc(N, 1.0, sym, lam, work + N);
    return -dotn(N, w, work) + dotn(N, w + N, work + N);
}

typedef struct {
    const qo_prob *pr;
    const qo_control *const *controls;
    const double *pcof;
    int m, ctrl;            /* control_index */
    size_t pcof_off;        /* offset of this control's slice */
    double *pvals, *qvals;  /* tables at the current t (all operators) */
    double *work_pcof;      /* N_coeff of this control */
    double *work_vec;       /* 2N */
} magic_ctx;

/* recursive_magic!, eval_grad_discrete_adjoint.jl:656-726 */
static void recursive_magic(magic_ctx *mc, double *grad_contrib, const double *w_mat,
                            const double *lambda, int deriv_order, double coeff, double t,
                            double *work_mat /* 2N x deriv_order-ish */)
{
    const qo_prob *pr = mc->pr;
    const int N = pr->N, n2 = 2 * N;
    const qo_control *ctl = mc->controls[mc->ctrl];
    const double *asym = pr->asym_ops + (size_t)mc->ctrl * N * N;
    const double *sym = pr->sym_ops + (size_t)mc->ctrl * N * N;
    const double *lp = mc->pcof + mc->pcof_off;
    const int j = deriv_order - 1;
    for (int i = 0; i <= j; i++) {
        double ipS = inner_prod_S(N, w_mat + (size_t)i * n2, lambda, asym, mc->work_vec);
        double ipK = inner_prod_K(N, w_mat + (size_t)i * n2, lambda, sym, mc->work_vec);
        const double den = (j + 1) * factorial_d(j - i);
        qo_eval_grad_p_derivative(ctl, t, lp, j - i, mc->work_pcof);
        for (int l = 0; l < ctl->n_coeff; l++) grad_contrib[l] += mc->work_pcof[l] * ipK * coeff / den;
        qo_eval_grad_q_derivative(ctl, t, lp, j - i, mc->work_pcof);
        for (int l = 0; l < ctl->n_coeff; l++) grad_contrib[l] += mc->work_pcof[l] * ipS * coeff / den;
    }
    for (int i = 0; i <= j; i++) {
        double *right_inner = work_mat + (size_t)i * n2;
        memset(right_inner, 0, sizeof(double) * n2);
        qo_apply_hamiltonian(pr, mc->pvals, mc->qvals, 1 + mc->m, j - i, 1, lambda, right_inner);
        recursive_magic(mc, grad_contrib, w_mat, right_inner, i, coeff / (j + 1), t, work_mat);
    }
}

/* accumulate_gradient_arbitrary_fast!, :582-647 (one column) */
static void accumulate_gradient_column(const qo_prob *pr, const qo_control *const *controls,
                                       const double *pcof, int order, const double *hist,
                                       const double *lam_hist, double *gradient)
{
    const int n2 = 2 * pr->N, m = order / 2, nsteps = pr->nsteps;
    const size_t slab = (size_t)n2 * (1 + m);
    const double dt = pr->tf / nsteps;
    magic_ctx mc;
    mc.pr = pr; mc.controls = controls; mc.pcof = pcof; mc.m = m;
    mc.pvals = (double *)malloc(sizeof(double) * (1 + m) * pr->n_ops);
    mc.qvals = (double *)malloc(sizeof(double) * (1 + m) * pr->n_ops);
    mc.work_vec = (double *)malloc(sizeof(double) * n2);
    double *work_mat = (double *)malloc(sizeof(double) * n2 * (m > 0 ? m : 1));
    size_t off = 0;
    for (int k = 0; k < pr->n_ops; k++) {
        const qo_control *ctl = controls[k];
        mc.ctrl = k; mc.pcof_off = off;
        mc.work_pcof = (double *)malloc(sizeof(double) * ctl->n_coeff);
        double *contrib = (double *)calloc(ctl->n_coeff, sizeof(double));
        for (int n = 0; n < nsteps; n++) {
            const double *lam = lam_hist + (size_t)(n + 1) * slab;   /* :604 column 1 only */
            const double *wn = hist + (size_t)n * slab, *wn1 = hist + (size_t)(n + 1) * slab;
            double tn = n * dt, tn1 = (n + 1) * dt;
            fill_pq_mats(pr, controls, tn, pcof, m, mc.pvals, mc.qvals);
            for (int kk = 0; kk <= m; kk++)                          /* explicit, :611-623 */
                recursive_magic(&mc, contrib, wn, lam, kk, pow(dt, kk) * qo_coefficient(kk, m, m), tn, work_mat);
            fill_pq_mats(pr, controls, tn1, pcof, m, mc.pvals, mc.qvals);
            for (int kk = 0; kk <= m; kk++)                          /* implicit, :626-639 */
                recursive_magic(&mc, contrib, wn1, lam, kk, -pow(-dt, kk) * qo_coefficient(kk, m, m), tn1, work_mat);
        }
        for (int l = 0; l < ctl->n_coeff; l++) gradient[off + l] -= contrib[l]; /* :642-643 */
        free(contrib); free(mc.work_pcof);
        off += ctl->n_coeff;
    }
    free(mc.pvals); free(mc.qvals); free(mc.work_vec); free(work_mat);
}

/* Test hook: ONE call of recursive_magic! (eval_grad_discrete_adjoint.jl:656-726) -- "the contribution of
 * <coeff * w_k, lambda>" to the gradient slice of control `control_index`, i.e. coeff * <d w_k / d theta_l, lambda>
 * for every coefficient l of that control, through compute_inner_prod_S!/K! (:764-800).  w_mat: 2N x (1+m) Taylor
 * coefficients at time t.  Pinned by the d/dtheta known-answer block of test/hardcoded_derivatives.jl:137-160. */
void qo_recursive_magic(const qo_prob *pr, const qo_control *const *controls, const double *pcof, int m,
                        int control_index, double t, const double *w_mat, const double *lambda, int deriv_order,
                        double coeff, double *grad_contrib)
{
    const int n2 = 2 * pr->N;
    magic_ctx mc;
    mc.pr = pr; mc.controls = controls; mc.pcof = pcof; mc.m = m; mc.ctrl = control_index;
    mc.pcof_off = 0;
    for (int k = 0; k < control_index; k++) mc.pcof_off += controls[k]->n_coeff;
    mc.pvals = (double *)malloc(sizeof(double) * (1 + m) * pr->n_ops);
    mc.qvals = (double *)malloc(sizeof(double) * (1 + m) * pr->n_ops);
    mc.work_vec = (double *)malloc(sizeof(double) * n2);
    mc.work_pcof = (double *)malloc(sizeof(double) * controls[control_index]->n_coeff);
    double *work_mat = (double *)malloc(sizeof(double) * n2 * (m > 0 ? m : 1));
    fill_pq_mats(pr, controls, t, pcof, m, mc.pvals, mc.qvals);
    recursive_magic(&mc, grad_contrib, w_mat, lambda, deriv_order, coeff, t, work_mat);
    free(mc.pvals); free(mc.qvals); free(mc.work_vec); free(mc.work_pcof); free(work_mat);
}

int qo_discrete_adjoint(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                        int n_pcof, const double *target_real, int order, int history_precomputed,
                        double *grad, double *history, double *lambda_history,
                        double *adjoint_forcing, qo_stats *st)
{   /* :107-160 */
    const int n2 = 2 * pr->N, m = order / 2, nt = 1 + pr->nsteps, nc = pr->n_cols;
    const size_t slab = (size_t)n2 * (1 + m), hcol = slab * nt;
    if (!history_precomputed) memset(history, 0, sizeof(double) * hcol * nc);
    memset(lambda_history, 0, sizeof(double) * hcol * nc);
    memset(adjoint_forcing, 0, sizeof(double) * (size_t)n2 * nt * nc);
    if (!history_precomputed) qo_eval_forward(pr, controls, pcof, order, NULL, history, st); /* :129-131 */
    qo_compute_guard_forcing(pr, m, history, adjoint_forcing);                           /* :134 */
    double *final_state = (double *)malloc(sizeof(double) * n2 * nc);
    double *forcing_end = (double *)malloc(sizeof(double) * n2 * nc);
    double *terminal = (double *)malloc(sizeof(double) * n2 * nc);
    for (int c = 0; c < nc; c++) {
        memcpy(final_state + (size_t)c * n2, history + (size_t)c * hcol + (size_t)(nt - 1) * slab, sizeof(double) * n2);
        memcpy(forcing_end + (size_t)c * n2, adjoint_forcing + ((size_t)c * nt + nt - 1) * n2, sizeof(double) * n2);
    }
    qo_compute_terminal_condition(pr, controls, pcof, order, target_real, final_state, forcing_end, terminal); /* :137-141 */
    qo_eval_adjoint(pr, controls, pcof, order, terminal, adjoint_forcing, lambda_history, st); /* :144-146 */
    for (int l = 0; l < n_pcof; l++) grad[l] = 0.0;                                       /* :149 */
    if (g_parallel_gradient && nc > 1) {       /* test switch, see qo_set_parallel_gradient: same bits, columns on threads */
        double *gc = (double *)calloc((size_t)nc * (n_pcof > 0 ? n_pcof : 1), sizeof(double));
        int nth = g_threads;
#ifdef _OPENMP
        if (nth <= 0) nth = omp_get_max_threads();
#else
        nth = 1;
#endif
        #pragma omp parallel for num_threads(nth) schedule(dynamic, 1)
        for (int c = 0; c < nc; c++)
            accumulate_gradient_column(pr, controls, pcof, order, history + (size_t)c * hcol,
                                       lambda_history + (size_t)c * hcol, gc + (size_t)c * n_pcof);
        for (int c = 0; c < nc; c++)
            for (int l = 0; l < n_pcof; l++) grad[l] += gc[(size_t)c * n_pcof + l];
        free(gc);
    } else
    for (int c = 0; c < nc; c++)                                                          /* serial, :150-157 */
        accumulate_gradient_column(pr, controls, pcof, order, history + (size_t)c * hcol,
                                   lambda_history + (size_t)c * hcol, grad);
    free(final_state); free(forcing_end); free(terminal);
    if (st) st->applies = g_applies;
    return 0;
}

/* ======================================================================== */
/* Test oracles                                                              */
/* ======================================================================== */
int qo_eval_grad_forced(const qo_prob *pr, const qo_control *const *controls, const double *pcof,
                        int n_pcof, const double *R, int order, double *gradient)
{   /* eval_grad_forced.jl:17-194 (cost_type = :Infidelity) */
    const int N = pr->N, n2 = 2 * N, m = order / 2, nt = 1 + pr->nsteps, nc = pr->n_cols;
    const size_t slab = (size_t)n2 * (1 + m), hcol = slab * nt;
    const size_t fslab = (size_t)n2 * m, fcol = fslab * nt;
    const double dt = pr->tf / pr->nsteps;
    double *history = (double *)calloc(hcol * nc, sizeof(double));
    double *hpd = (double *)calloc(hcol * nc, sizeof(double));
    double *forcing = (double *)calloc(fcol * nc, sizeof(double));
    double *z = (double *)calloc((size_t)N * nc, sizeof(double));
    double *Ww = (double *)malloc(sizeof(double) * n2);
    qo_prob diff = *pr; diff.u0 = z; diff.v0 = z;                    /* :36-39 */
    qo_eval_forward(pr, controls, pcof, order, NULL, history, NULL);  /* :55 */
    double dR = 0.0, dT = 0.0;
    for (int c = 0; c < nc; c++) {
        const double *p = history + (size_t)c * hcol + (size_t)(nt - 1) * slab, *r = R + (size_t)c * n2;
        for (int i = 0; i < N; i++) { dR += p[i] * r[i] + p[N + i] * r[N + i]; dT += p[i] * r[N + i] - p[N + i] * r[i]; }
    }
    int gidx = 0; size_t off = 0;
    for (int k = 0; k < pr->n_ops; k++) {
        const qo_control *ctl = controls[k];
        const double *asym = pr->asym_ops + (size_t)k * N * N, *sym = pr->sym_ops + (size_t)k * N * N;
        const int nl = ctl->n_coeff;
        double *pv = (double *)calloc((size_t)(1 + m) * nt * nl, sizeof(double));
        double *qv = (double *)calloc((size_t)(1 + m) * nt * nl, sizeof(double));
        double *g = (double *)malloc(sizeof(double) * nl);
        for (int n = 0; n < nt; n++) for (int d = 0; d < m; d++) {   /* :84-90 */
            qo_eval_grad_p_derivative(ctl, n * dt, pcof + off, d, g);
            for (int l = 0; l < nl; l++) pv[d + (size_t)(1 + m) * (n + (size_t)nt * l)] = g[l];
            qo_eval_grad_q_derivative(ctl, n * dt, pcof + off, d, g);
            for (int l = 0; l < nl; l++) qv[d + (size_t)(1 + m) * (n + (size_t)nt * l)] = g[l];
        }
        for (int l = 0; l < nl; l++) {
            for (int c = 0; c < nc; c++) for (int n = 0; n < nt; n++) {       /* :95-131 */
                const double *uv = history + (size_t)c * hcol + (size_t)n * slab;
                double *fm = forcing + (size_t)c * fcol + (size_t)n * fslab;
                memset(fm, 0, sizeof(double) * fslab);
                for (int j = 0; j < m; j++) for (int i = j; i >= 0; i--) {
                    double pval = pv[(j - i) + (size_t)(1 + m) * (n + (size_t)nt * l)] / factorial_d(j - i);
                    double qval = qv[(j - i) + (size_t)(1 + m) * (n + (size_t)nt * l)] / factorial_d(j - i);
                    const double *u = uv + (size_t)i * n2, *v = u + N;
                    double *ud = fm + (size_t)j * n2, *vd = ud + N;
                    gemv_acc(N, qval, asym, u, ud); gemv_acc(N, pval, sym, v, ud);
                    gemv_acc(N, qval, asym, v, vd); gemv_acc(N, -pval, sym, u, vd);
                }
            }
            qo_eval_forward(&diff, controls, pcof, order, forcing, hpd, NULL);  /* :135-142 */
            double pR = 0.0, pT = 0.0;
            for (int c = 0; c < nc; c++) {
                const double *p = hpd + (size_t)c * hcol + (size_t)(nt - 1) * slab, *r = R + (size_t)c * n2;
                for (int i = 0; i < N; i++) { pR += p[i] * r[i] + p[N + i] * r[N + i]; pT += p[i] * r[N + i] - p[N + i] * r[i]; }
            }
            double gval = -(2.0 / ((double)pr->n_ess * pr->n_ess)) * (dR * pR + dT * pT); /* :155-159 */
            if (g_cost_type) {                                                            /* :160-163 */
                gval = 0.0;
                for (int c = 0; c < nc; c++) {
                    const double *p = hpd + (size_t)c * hcol + (size_t)(nt - 1) * slab, *r = R + (size_t)c * n2;
                    const double *w = history + (size_t)c * hcol + (size_t)(nt - 1) * slab;
                    for (int i = 0; i < n2; i++) gval += p[i] * (g_cost_type == 1 ? w[i] - r[i] : w[i]);
                }
            }
            double guard = 0.0;                                                          /* :168-184 */
            for (int n = 0; n < nt; n++) {
                double val = 0.0;
                for (int c = 0; c < nc; c++) {
                    const double *w = history + (size_t)c * hcol + (size_t)n * slab;
                    const double *dw = hpd + (size_t)c * hcol + (size_t)n * slab;
                    memset(Ww, 0, sizeof(double) * n2); gemv_acc(n2, 1.0, pr->guard, w, Ww);
                    val += dotn(n2, dw, Ww);
                    memset(Ww, 0, sizeof(double) * n2); gemv_acc(n2, 1.0, pr->guard, dw, Ww);
                    val += dotn(n2, w, Ww);
                }
                guard += (n == 0 || n == nt - 1) ? 0.5 * val : val;
            }
            gradient[gidx++] = gval + guard * dt / pr->tf;
        }
        free(pv); free(qv); free(g);
        off += nl;
    }
    (void)n_pcof;
    free(history); free(hpd); free(forcing); free(z); free(Ww);
    return 0;
}

static double objective_inf_plus_guard(const qo_prob *pr, const qo_control *const *controls,
                                       const double *pcof, const double *R, int order, double *hist)
{   /* infidelity.jl:148-165 */
    const int n2 = 2 * pr->N, m = order / 2, nt = 1 + pr->nsteps, nc = pr->n_cols;
    const size_t slab = (size_t)n2 * (1 + m), hcol = slab * nt;
    qo_eval_forward(pr, controls, pcof, order, NULL, hist, NULL);
    double *fin = (double *)malloc(sizeof(double) * n2 * nc);
    for (int c = 0; c < nc; c++)
        memcpy(fin + (size_t)c * n2, hist + (size_t)c * hcol + (size_t)(nt - 1) * slab, sizeof(double) * n2);
    double v = qo_guard_penalty_real(pr, m, hist);
    if (g_cost_type == 0) v += qo_infidelity_real(pr->N, nc, fin, R, pr->n_ess);
    else {                                                  /* eval_grad_finite_difference.jl:51-56 */
        double s2 = 0.0;
        for (size_t i = 0; i < (size_t)n2 * nc; i++) { const double d = (g_cost_type == 1) ? fin[i] - R[i] : fin[i]; s2 += d * d; }
        v += 0.5 * s2;
    }
    free(fin);
    return v;
}

int qo_eval_grad_finite_difference(const qo_prob *pr, const qo_control *const *controls,
                                   const double *pcof, int n_pcof, const double *R, int order,
                                   double dpcof, double *grad)
{   /* eval_grad_finite_difference.jl:17-72 */
    const int n2 = 2 * pr->N, m = order / 2;
    double *hist = (double *)malloc(sizeof(double) * (size_t)n2 * (1 + m) * (1 + pr->nsteps) * pr->n_cols);
    double *pc = (double *)malloc(sizeof(double) * n_pcof);
    for (int i = 0; i < n_pcof; i++) {
        memcpy(pc, pcof, sizeof(double) * n_pcof); pc[i] += dpcof;
        double cr = objective_inf_plus_guard(pr, controls, pc, R, order, hist);
        memcpy(pc, pcof, sizeof(double) * n_pcof); pc[i] -= dpcof;
        double cl = objective_inf_plus_guard(pr, controls, pc, R, order, hist);
        grad[i] = (cr - cl) / (2.0 * dpcof);
    }
    free(hist); free(pc);
    return 0;
}
