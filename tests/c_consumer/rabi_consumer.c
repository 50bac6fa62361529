/* A plain C consumer of include/qgd.h -- the header, not a hand-copied ctypes/Julia struct, is the contract.
 *
 * Builds the reference's Rabi oscillator (src/ProblemConstructors/rabi_oscillator.jl:7-22: H = [[0, p+iq],[p-iq, 0]],
 * U0 = I), and checks
 *   1. the SchrodingerProb validation (a non-antisymmetric operator is an argument error with the reference's message);
 *   2. qgd_eval_forward with explicit control tables: |Omega| = 1/2, tf = pi is a SWAP
 *      (test/OptimizationTests/optimization_rabi_osc_SWAP.jl:18-39), to 1e-10;
 *   3. qgd_discrete_adjoint with a two-coefficient control basis (p = theta_0, q = theta_1) against centred
 *      differences of the infidelity from qgd_eval_forward (the reference's adjoint-vs-FD contract,
 *      test/GradientTests/compare_gradients.jl:47-65), to 1e-7 relative;
 *   4. the same problem through qgd_create_csc gives the same gradient;
 *   5. qgd_comm_unique_id / qgd_comm_init_rccl (one rank, both shard kinds): the collective qgd_discrete_adjoint gives it too;
 *   6. qgd_set_memory_budget: the grid in windows gives it too, qgd_get_partition reports the whole grid, and an
 *      impossible budget is QGD_ERR_MEMORY;
 *   7. the failure mode of the collective calls: an injected local failure and an expired time limit are QGD_ERR_COMM with
 *      the communicator aborted, and a fresh communicator on the same handle works;
 *   8. QGD_CREATE_DEFER_GRID, and a communicator on a handle that was working in windows (one resident window).
 * Exit code 0 and "C_CONSUMER_OK" on success; 3 when there is no GPU (the library has no CPU path).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qgd.h"

#define NL 2
#define NC 2
#define ORDER 8
#define M (ORDER / 2)
#define NSTEPS 20
#define NT (NSTEPS + 1)

static int die(const char *what, qgd_handle h, int rc)
{
    fprintf(stderr, "%s failed: code %d: %s\n", what, rc, qgd_last_error(h));
    return rc == QGD_ERR_NO_DEVICE ? 3 : 1;
}

static double infidelity_of(const double *out3) { return 1.0 - (out3[0] * out3[0] + out3[1] * out3[1]) / (NL * NL); }

int main(void)
{
    const double pi = 3.14159265358979323846;
    /* column-major 2x2: a + a^T and a - a^T with a = [[0,1],[0,0]] */
    double zero[4] = {0, 0, 0, 0}, sym[4] = {0, 1, 1, 0}, asym[4] = {0, -1, 1, 0}, bad[4] = {0, 1, 1, 0};
    double u0[4] = {1, 0, 0, 1}, v0[4] = {0, 0, 0, 0};
    qgd_problem_desc d;
    memset(&d, 0, sizeof d);
    d.N = NL; d.n_cols = NC; d.n_ops = 1; d.n_ess = NL; d.order = ORDER; d.nsteps = NSTEPS; d.tf = pi;
    d.system_sym = zero; d.system_asym = zero; d.sym_ops = sym; d.asym_ops = bad; d.u0 = u0; d.v0 = v0;
    d.guard = NULL; d.device = 0;
    qgd_handle h = NULL;
    int rc;

    if (qgd_abi_version() != QGD_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    /* 1. validation comes before any device work */
    rc = qgd_create(&d, &h);
    if (rc != QGD_ERR_ARGUMENT || h != NULL || !strstr(qgd_last_error(NULL), "not anti-symmetric")) {
        fprintf(stderr, "validation: expected QGD_ERR_ARGUMENT, got %d (%s)\n", rc, qgd_last_error(NULL));
        return 1;
    }
    d.asym_ops = asym;
    if ((rc = qgd_create(&d, &h))) return die("qgd_create", NULL, rc);

    /* 2. SWAP: tables[(1+M), n_ops, NT] column-major, p = 1/2, q = 0, all derivatives 0 */
    static double pt[(1 + M) * NT], qt[(1 + M) * NT];
    for (int n = 0; n < NT; n++) pt[(1 + M) * n] = 0.5;
    if ((rc = qgd_set_control_tables(h, pt, qt))) return die("qgd_set_control_tables", h, rc);
    static double hist[2 * NL * (1 + M) * NT * NC];
    double out3[3];
    if ((rc = qgd_eval_forward(h, NULL, 0, hist, out3))) return die("qgd_eval_forward", h, rc);
    for (int col = 0; col < NC; col++)
        for (int i = 0; i < NL; i++) {
            const double *w = hist + ((size_t)(col * NT + NSTEPS) * (1 + M)) * 2 * NL;   /* [2N, 1+M, NT, NC], j = 0 */
            const double p2 = w[i] * w[i] + w[NL + i] * w[NL + i], want = (i != col) ? 1.0 : 0.0;
            if (fabs(p2 - want) > 1e-10) { fprintf(stderr, "SWAP: |psi[%d,%d]|^2 = %.15g\n", i, col, p2); return 1; }
        }

    /* 3. gradient: basis Gp[n][d][l], Gq[n][d][l], n_coeff = 2: p = theta_0, q = theta_1 (constant in time) */
    static double Gp[NT * (1 + M) * 2], Gq[NT * (1 + M) * 2];
    for (int n = 0; n < NT; n++) { Gp[(n * (1 + M)) * 2 + 0] = 1.0; Gq[(n * (1 + M)) * 2 + 1] = 1.0; }
    const int32_t ncoef[1] = {2};
    const double *gp[1] = {Gp}, *gq[1] = {Gq};
    if ((rc = qgd_set_control_basis(h, ncoef, gp, gq))) return die("qgd_set_control_basis", h, rc);
    double target[2 * NL * NC] = {0, 0, 0, -1, 0, 0, -1, 0};   /* [Re; Im] of -i * SWAP, column-major 4 x 2 */
    if ((rc = qgd_set_target(h, target))) return die("qgd_set_target", h, rc);
    double theta[2] = {0.4, 0.1}, grad[2], fd[2];
    if ((rc = qgd_discrete_adjoint(h, theta, 2, 0, grad, NULL, NULL, NULL, out3))) return die("qgd_discrete_adjoint", h, rc);
    const double infid0 = infidelity_of(out3);
    for (int l = 0; l < 2; l++) {
        const double eps = 1e-5;
        double tp[2] = {theta[0], theta[1]}, tm[2] = {theta[0], theta[1]}, op[3], om[3];
        tp[l] += eps; tm[l] -= eps;
        if ((rc = qgd_eval_forward(h, tp, 2, NULL, op))) return die("qgd_eval_forward(+)", h, rc);
        if ((rc = qgd_eval_forward(h, tm, 2, NULL, om))) return die("qgd_eval_forward(-)", h, rc);
        fd[l] = (infidelity_of(op) - infidelity_of(om)) / (2 * eps);
        if (fabs(fd[l] - grad[l]) > 1e-7 * fmax(1.0, fabs(fd[l]))) {
            fprintf(stderr, "gradient[%d]: adjoint %.12g, centred difference %.12g\n", l, grad[l], fd[l]);
            return 1;
        }
    }
    /* at theta = (1/2, 0) the gate is the target: infidelity 0 */
    double opt[2] = {0.5, 0.0}, g2[2];
    if ((rc = qgd_discrete_adjoint(h, opt, 2, 0, g2, NULL, NULL, NULL, out3))) return die("qgd_discrete_adjoint(opt)", h, rc);
    if (fabs(infidelity_of(out3)) > 1e-10) { fprintf(stderr, "infidelity at the optimum: %.3e\n", infidelity_of(out3)); return 1; }

    /* 4. the CSC constructor (Julia's 1-based SparseMatrixCSC arrays) */
    const int64_t cp_zero[3] = {1, 1, 1}, cp_off[3] = {1, 2, 3}, rv_off[2] = {2, 1};
    const double nz_sym[2] = {1, 1}, nz_asym[2] = {-1, 1};
    qgd_csc zs = {cp_zero, NULL, NULL, 1, 0}, s1 = {cp_off, rv_off, nz_sym, 1, 0}, a1 = {cp_off, rv_off, nz_asym, 1, 0};
    qgd_handle hc = NULL;
    if ((rc = qgd_create_csc(&d, &zs, &zs, &s1, &a1, &hc))) return die("qgd_create_csc", NULL, rc);
    if ((rc = qgd_set_control_basis(hc, ncoef, gp, gq))) return die("qgd_set_control_basis(csc)", hc, rc);
    if ((rc = qgd_set_target(hc, target))) return die("qgd_set_target(csc)", hc, rc);
    double gc[2];
    if ((rc = qgd_discrete_adjoint(hc, theta, 2, 0, gc, NULL, NULL, NULL, out3))) return die("qgd_discrete_adjoint(csc)", hc, rc);
    if (fabs(gc[0] - grad[0]) > 1e-13 || fabs(gc[1] - grad[1]) > 1e-13 || fabs(infidelity_of(out3) - infid0) > 1e-13) {
        fprintf(stderr, "CSC handle differs: %.15g %.15g vs %.15g %.15g\n", gc[0], gc[1], grad[0], grad[1]);
        return 1;
    }
    /* 5. several GPUs behind one call, from plain C: rank 0 of a communicator of one (the box has one GPU).  The id is what
     *    a host would carry to the other ranks; after qgd_comm_init_rccl the SAME qgd_discrete_adjoint is a collective call
     *    whose all-gathers / all-reduce the library issues itself.  (QGD_ERR_COMM when librccl cannot be loaded.) */
    unsigned char id[QGD_UNIQUE_ID_BYTES];
    int32_t info[3];
    for (int shard = QGD_SHARD_TIME; shard <= QGD_SHARD_COLUMNS; shard++) {
        if ((rc = qgd_comm_unique_id(id))) return die("qgd_comm_unique_id", NULL, rc);      /* one id per communicator */
        if ((rc = qgd_comm_init_rccl(hc, id, 0, 1, shard))) return die("qgd_comm_init_rccl", hc, rc);
        if ((rc = qgd_comm_info(hc, info)) || info[0] != 0 || info[1] != 1 || info[2] != shard) { fprintf(stderr, "qgd_comm_info\n"); return 1; }
        if (shard == QGD_SHARD_TIME) {      /* a time window re-allocates the grid: basis and target again */
            if ((rc = qgd_set_control_basis(hc, ncoef, gp, gq))) return die("qgd_set_control_basis(comm)", hc, rc);
            if ((rc = qgd_set_target(hc, target))) return die("qgd_set_target(comm)", hc, rc);
        }
        double gm[2], om3[3];
        if ((rc = qgd_discrete_adjoint(hc, theta, 2, 0, gm, NULL, NULL, NULL, om3))) return die("qgd_discrete_adjoint(comm)", hc, rc);
        if (fabs(gm[0] - grad[0]) > 1e-13 || fabs(gm[1] - grad[1]) > 1e-13 || fabs(infidelity_of(om3) - infid0) > 1e-13) {
            fprintf(stderr, "communicator handle (shard %d) differs: %.15g %.15g vs %.15g %.15g\n", shard, gm[0], gm[1], grad[0], grad[1]);
            return 1;
        }
        if ((rc = qgd_comm_destroy(hc))) return die("qgd_comm_destroy", hc, rc);
    }
    /* 6. a memory budget too small for the grid: windows; the gradient stays; the plan says how many */
    int64_t plan[4];
    if ((rc = qgd_get_memory_plan(h, plan)) || plan[0] != 1) { fprintf(stderr, "memory plan: %d windows\n", (int)plan[0]); return 1; }
    if ((rc = qgd_set_memory_budget(h, (size_t)plan[2] / 2))) return die("qgd_set_memory_budget", h, rc);
    if ((rc = qgd_get_memory_plan(h, plan)) || plan[0] < 2) { fprintf(stderr, "memory plan after the budget: %d windows\n", (int)plan[0]); return 1; }
    if ((rc = qgd_set_control_basis(h, ncoef, gp, gq))) return die("qgd_set_control_basis(budget)", h, rc);
    if ((rc = qgd_set_target(h, target))) return die("qgd_set_target(budget)", h, rc);
    double gw[2];
    if ((rc = qgd_discrete_adjoint(h, theta, 2, 0, gw, NULL, NULL, NULL, out3))) return die("qgd_discrete_adjoint(budget)", h, rc);
    if (fabs(gw[0] - grad[0]) > 1e-12 || fabs(gw[1] - grad[1]) > 1e-12) {
        fprintf(stderr, "windowed grid differs: %.15g %.15g vs %.15g %.15g\n", gw[0], gw[1], grad[0], grad[1]);
        return 1;
    }
    /* (qgd_get_partition on a handle that works in windows: the caller owns the WHOLE grid -- first point 0, last point
     *  NSTEPS, one rank -- whichever window the library processed last; a shim sizes its basis and outputs by this) */
    int32_t part[8];
    if ((rc = qgd_get_partition(h, part)) || part[0] != 0 || part[1] != NSTEPS || part[6] != 1 || part[7] != NT) {
        fprintf(stderr, "qgd_get_partition on a windowed grid: [%d, %d], world %d, %d points\n", part[0], part[1], part[6], part[7]);
        return 1;
    }
    if (qgd_set_memory_budget(h, 64) != QGD_ERR_MEMORY) { fprintf(stderr, "a 64-byte budget must be QGD_ERR_MEMORY\n"); return 1; }
    /* 7. the failure mode of a collective call: a collective call that outlives its time limit aborts its communicator and
     *    reports QGD_ERR_COMM (it does not leave the call half-entered).  Afterwards the handle has no communicator (rank -1)
     *    and takes a fresh one.  (A rank that fails locally in front of an exchange does the same: the Python tests inject
     *    that with a hook outside the library, tests/hooks/qgd_test_hooks.cpp -- this consumer uses include/qgd.h only.) */
    for (int trial = 1; trial < 2; trial++) {
        if ((rc = qgd_comm_unique_id(id))) return die("qgd_comm_unique_id", NULL, rc);
        if ((rc = qgd_comm_init_rccl(hc, id, 0, 1, QGD_SHARD_TIME))) return die("qgd_comm_init_rccl(7)", hc, rc);
        if ((rc = qgd_set_control_basis(hc, ncoef, gp, gq))) return die("qgd_set_control_basis(7)", hc, rc);
        if ((rc = qgd_set_target(hc, target))) return die("qgd_set_target(7)", hc, rc);
        if ((rc = qgd_set_comm_timeout(hc, 1e-6))) return die("qgd_set_comm_timeout", hc, rc);
        double gm[2], om3[3];
        rc = qgd_discrete_adjoint(hc, theta, 2, 0, gm, NULL, NULL, NULL, om3);
        if (rc != QGD_ERR_COMM || !strstr(qgd_last_error(hc), "aborted")) {
            fprintf(stderr, "failure mode %d: expected QGD_ERR_COMM, got %d (%s)\n", trial, rc, qgd_last_error(hc));
            return 1;
        }
        if ((rc = qgd_comm_info(hc, info)) || info[0] != -1) { fprintf(stderr, "the communicator must be gone after a failure\n"); return 1; }
        if ((rc = qgd_set_comm_timeout(hc, 30000.0))) return die("qgd_set_comm_timeout", hc, rc);
    }
    /*    ... and a healthy communicator on the same handle works again */
    if ((rc = qgd_comm_unique_id(id))) return die("qgd_comm_unique_id", NULL, rc);
    if ((rc = qgd_comm_init_rccl(hc, id, 0, 1, QGD_SHARD_TIME))) return die("qgd_comm_init_rccl(7b)", hc, rc);
    if ((rc = qgd_set_control_basis(hc, ncoef, gp, gq))) return die("qgd_set_control_basis(7b)", hc, rc);
    if ((rc = qgd_set_target(hc, target))) return die("qgd_set_target(7b)", hc, rc);
    if ((rc = qgd_discrete_adjoint(hc, theta, 2, 0, gc, NULL, NULL, NULL, out3))) return die("qgd_discrete_adjoint(7b)", hc, rc);
    if (fabs(gc[0] - grad[0]) > 1e-13 || fabs(gc[1] - grad[1]) > 1e-13) { fprintf(stderr, "after a recovered communicator: gradient differs\n"); return 1; }
    /* 8. QGD_CREATE_DEFER_GRID: no grid until an entry point needs one; a communicator on a handle that had a memory
     *    budget keeps its window resident (one window), and the result is the same */
    qgd_handle hd = NULL;
    d.reserved = QGD_CREATE_DEFER_GRID;
    if ((rc = qgd_create(&d, &hd))) return die("qgd_create(deferred)", NULL, rc);
    d.reserved = 0;
    if ((rc = qgd_comm_unique_id(id))) return die("qgd_comm_unique_id", NULL, rc);
    if ((rc = qgd_comm_init_rccl(hd, id, 0, 1, QGD_SHARD_TIME))) return die("qgd_comm_init_rccl(deferred)", hd, rc);
    if ((rc = qgd_get_memory_plan(hd, plan)) || plan[0] != 1) { fprintf(stderr, "a communicator handle must be resident: %d windows\n", (int)plan[0]); return 1; }
    if ((rc = qgd_set_control_basis(hd, ncoef, gp, gq))) return die("qgd_set_control_basis(deferred)", hd, rc);
    if ((rc = qgd_set_target(hd, target))) return die("qgd_set_target(deferred)", hd, rc);
    if ((rc = qgd_discrete_adjoint(hd, theta, 2, 0, gc, NULL, NULL, NULL, out3))) return die("qgd_discrete_adjoint(deferred)", hd, rc);
    if (fabs(gc[0] - grad[0]) > 1e-13 || fabs(gc[1] - grad[1]) > 1e-13) { fprintf(stderr, "deferred-grid handle: gradient differs\n"); return 1; }
    /*    a handle that works in windows because of its memory budget cannot take a communicator (the collective protocol
     *    keeps a rank's window resident): QGD_ERR_MEMORY -- not window 0 of several, silently -- and the handle keeps the
     *    layout it had; with the budget lifted the communicator is accepted and the grid is one resident window */
    if ((rc = qgd_set_memory_budget(h, 0))) return die("qgd_set_memory_budget(0)", h, rc);
    if ((rc = qgd_get_memory_plan(h, plan)) || plan[0] != 1) { fprintf(stderr, "automatic budget: %d windows\n", (int)plan[0]); return 1; }
    if ((rc = qgd_set_memory_budget(h, (size_t)plan[2] / 2))) return die("qgd_set_memory_budget(half)", h, rc);
    if ((rc = qgd_comm_unique_id(id))) return die("qgd_comm_unique_id", NULL, rc);
    if (qgd_comm_init_rccl(h, id, 0, 1, QGD_SHARD_TIME) != QGD_ERR_MEMORY) { fprintf(stderr, "communicator on a windowed handle: %s\n", qgd_last_error(h)); return 1; }
    if ((rc = qgd_comm_info(h, info)) || info[0] != -1) { fprintf(stderr, "no communicator may be left behind\n"); return 1; }
    if ((rc = qgd_get_memory_plan(h, plan)) || plan[0] < 2) { fprintf(stderr, "the refused handle must keep its windows: %d\n", (int)plan[0]); return 1; }
    if ((rc = qgd_set_control_basis(h, ncoef, gp, gq))) return die("qgd_set_control_basis(refused)", h, rc);
    if ((rc = qgd_set_target(h, target))) return die("qgd_set_target(refused)", h, rc);
    if ((rc = qgd_discrete_adjoint(h, theta, 2, 0, gw, NULL, NULL, NULL, out3))) return die("qgd_discrete_adjoint(refused)", h, rc);
    if (fabs(gw[0] - grad[0]) > 1e-12 || fabs(gw[1] - grad[1]) > 1e-12) { fprintf(stderr, "refused handle: gradient differs\n"); return 1; }
    if ((rc = qgd_set_memory_budget(h, 0))) return die("qgd_set_memory_budget(0)", h, rc);
    if ((rc = qgd_comm_init_rccl(h, id, 0, 1, QGD_SHARD_TIME))) return die("qgd_comm_init_rccl(budget lifted)", h, rc);
    if ((rc = qgd_get_memory_plan(h, plan)) || plan[0] != 1) { fprintf(stderr, "communicator handle: %d windows\n", (int)plan[0]); return 1; }
    if ((rc = qgd_set_control_basis(h, ncoef, gp, gq))) return die("qgd_set_control_basis(comm)", h, rc);
    if ((rc = qgd_set_target(h, target))) return die("qgd_set_target(comm)", h, rc);
    if ((rc = qgd_discrete_adjoint(h, theta, 2, 0, gw, NULL, NULL, NULL, out3))) return die("qgd_discrete_adjoint(comm)", h, rc);
    if (fabs(gw[0] - grad[0]) > 1e-12 || fabs(gw[1] - grad[1]) > 1e-12) { fprintf(stderr, "communicator handle: gradient differs\n"); return 1; }
    qgd_destroy(hd);
    qgd_destroy(hc);
    qgd_destroy(h);
    printf("infidelity(0.4, 0.1) = %.12f  grad = [%.12f, %.12f]  fd = [%.12f, %.12f]\nC_CONSUMER_OK\n", infid0, grad[0], grad[1], fd[0], fd[1]);
    return 0;
}
