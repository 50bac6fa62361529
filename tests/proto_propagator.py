"""numpy statement of the DEVICE algorithm (time-parallel propagator form), used
only by tests to localise kernel bugs stage by stage.  It is not the oracle
(oracle/ restates the reference's algorithm) and not the product.

Real form w=[u;v] <-> psi=u+iv;  A=[K S;-S K] <-> K - iS;  A^T <-> (.)^H;
<w,l> = Re(psi^H chi).  See DESIGN.md "Algorithm".
"""
import math

import numpy as np


def coefficient(j, p, q):
    return math.factorial(p) * math.factorial(p + q - j) / (math.factorial(p + q) * math.factorial(p - j))


def tables(Gp, Gq, offsets, pcof, n_deriv):
    """tp[n, d, k] = p_k^(d)(t_n)/d!"""
    nt = Gp[0].shape[0]
    tp = np.zeros((nt, n_deriv + 1, len(Gp)))
    tq = np.zeros_like(tp)
    for k, (gp, gq, off) in enumerate(zip(Gp, Gq, offsets)):
        sl = pcof[off:off + gp.shape[2]]
        tp[:, :, k] = gp[:, :n_deriv + 1] @ sl
        tq[:, :, k] = gq[:, :n_deriv + 1] @ sl
    return tp, tq


def assemble(prob, tp, tq, m):
    """Ac[n, d] = K_d - i S_d (complex N x N), d = 0..m-1."""
    nt = tp.shape[0]
    N = prob.N_tot_levels
    Ac = np.zeros((nt, m, N, N), dtype=complex)
    sym = np.stack(prob.sym_operators) if prob.N_operators else np.zeros((0, N, N))
    asym = np.stack(prob.asym_operators) if prob.N_operators else np.zeros((0, N, N))
    for d in range(m):
        K = np.einsum("nk,kij->nij", tq[:, d, :], asym)
        S = np.einsum("nk,kij->nij", tp[:, d, :], sym)
        if d == 0:
            K = K + prob.system_asym
            S = S + prob.system_sym
        Ac[:, d] = K - 1j * S
    return Ac


def build_LR(Ac, m, dt):
    nt, _, N, _ = Ac.shape
    D = [np.broadcast_to(np.eye(N, dtype=complex), (nt, N, N)).copy()]
    for j in range(m):
        acc = np.zeros((nt, N, N), dtype=complex)
        for i in range(j + 1):
            acc += Ac[:, j - i] @ D[i]
        D.append(acc / (j + 1))
    L = sum(coefficient(j, m, m) * (-dt) ** j * D[j] for j in range(m + 1))
    R = sum(coefficient(j, m, m) * dt ** j * D[j] for j in range(m + 1))
    return L, R, D


def evaluate(prob, Gp, Gq, offsets, pcof, target, order, overlap=None):
    """Full forward + adjoint + gradient.  Returns dict of every intermediate.  ``overlap=(a, b)``: form the terminal
    condition from these (global) overlaps instead of the ones of prob's own columns (column-sharded evaluation)."""
    m = order // 2
    N, c, nsteps = prob.N_tot_levels, prob.N_initial_conditions, prob.nsteps
    dt = prob.tf / nsteps
    nt = nsteps + 1
    tp, tq = tables(Gp, Gq, offsets, pcof, m)
    Ac = assemble(prob, tp, tq, m)
    L, R, _ = build_LR(Ac, m, dt)
    Linv = np.linalg.inv(L)
    P = Linv[1:] @ R[:-1]
    psi = np.zeros((nt, N, c), dtype=complex)
    psi[0] = prob.u0 + 1j * prob.v0
    for n in range(nsteps):
        psi[n + 1] = P[n] @ psi[n]
    W = prob.guard_subspace_projector
    wreal = np.concatenate([psi.real, psi.imag], axis=1)           # [nt, 2N, c]
    Ww = np.einsum("ij,njc->nic", W, wreal)
    trap = np.ones(nt); trap[0] = trap[-1] = 0.5
    guard = (dt / prob.tf) * np.einsum("n,nic,nic->", trap, wreal, Ww)
    f = -(2 * dt / prob.tf) * trap[:, None, None] * Ww
    fc = f[:, :N] + 1j * f[:, N:]
    T = np.asarray(target)
    ovl = np.sum(np.conj(T) * psi[-1])                              # <w,R> + i<w,T>
    a, b = ovl.real, ovl.imag
    infid = 1 - (a * a + b * b) / prob.N_ess_levels ** 2
    ga, gb = (a, b) if overlap is None else overlap
    y = np.zeros((nt, N, c), dtype=complex)
    y[-1] = (2 / prob.N_ess_levels ** 2) * (ga + 1j * gb) * T + fc[-1]  # a*R + b*T with T=[Rim;-Rre] <-> -i*R
    for n in range(nsteps - 1, 0, -1):
        y[n] = P[n].conj().T @ y[n + 1] + fc[n]
    lam = np.zeros_like(y)
    lam[1:] = np.conj(np.transpose(Linv[1:], (0, 2, 1))) @ y[1:]
    # derivatives of the state at every time point
    ws = [psi]
    for j in range(m):
        acc = np.zeros_like(psi)
        for i in range(j + 1):
            acc += Ac[:, j - i] @ ws[i]
        ws.append(acc / (j + 1))
    # gradient seeds
    g = [None] * (m + 1)
    lam_next = np.zeros_like(lam); lam_next[:-1] = lam[1:]
    lam_here = lam.copy(); lam_here[0] = 0
    for j in range(m + 1):
        cj = coefficient(j, m, m)
        g[j] = cj * dt ** j * lam_next - cj * (-dt) ** j * lam_here
    AcH = np.conj(np.transpose(Ac, (0, 1, 3, 2)))
    for j in range(m, 1, -1):
        for i in range(1, j):
            g[i] = g[i] + (1.0 / j) * (AcH[:, j - 1 - i] @ g[j])
    nops = prob.N_operators
    sigP = np.zeros((nt, nops, m))
    sigQ = np.zeros((nt, nops, m))
    for k in range(nops):
        Sk, Ak = prob.sym_operators[k], prob.asym_operators[k]
        for j in range(1, m + 1):
            for i in range(j):
                d = j - 1 - i
                Pw = -1j * (Sk @ ws[i])
                Qw = Ak @ ws[i]
                sigP[:, k, d] += (1.0 / j) * np.einsum("nic,nic->n", np.conj(Pw), g[j]).real
                sigQ[:, k, d] += (1.0 / j) * np.einsum("nic,nic->n", np.conj(Qw), g[j]).real
    grad = np.zeros(len(pcof))
    for k, (gp, gq, off) in enumerate(zip(Gp, Gq, offsets)):
        nl = gp.shape[2]
        grad[off:off + nl] -= np.einsum("ndl,nd->l", gp[:, :m], sigP[:, k]) + np.einsum("ndl,nd->l", gq[:, :m], sigQ[:, k])
    return dict(tp=tp, tq=tq, Ac=Ac, L=L, R=R, Linv=Linv, P=P, psi=psi, ws=ws, guard=guard, infidelity=infid,
                overlap=(a, b), f=fc, y=y, lam=lam, g=g, sigP=sigP, sigQ=sigQ, grad=grad)


def evaluate_local(prob, Gp, Gq, offsets, pcof, target, order, overlap=None):
    """The same evaluation with every per-time-point quantity LOCAL to its time point (round 6, the fused front kernel
    csrc/qgd_k_front.hip): the step propagator is the same-point product S_n = R_n L_n^-1 instead of P_n = L_{n+1}^-1 R_n.
      forward in phi_n = L_n psi_n:   phi_0 = L_0 psi_0,  phi_{n+1} = S_n phi_n,  psi_n = L_n^-1 phi_n   (forward_evolution.jl:181-220)
      adjoint directly in lambda:     lambda_N = L_N^-H rhs_N,  lambda_n = S_n^H lambda_{n+1} + L_n^-H f_n  (forward_evolution.jl:421-461)
    The device eliminates [L_n^H | R_n^H] -> [L_n^-H | S_n^H]: X = L^-H and Y = S^H are what it stores; they are formed that way
    here too.  Gradient part unchanged.  Returns the dict of evaluate() plus S, X, Y, phi, h."""
    m = order // 2
    N, c, nsteps = prob.N_tot_levels, prob.N_initial_conditions, prob.nsteps
    dt = prob.tf / nsteps
    nt = nsteps + 1
    tp, tq = tables(Gp, Gq, offsets, pcof, m)
    Ac = assemble(prob, tp, tq, m)
    L, R, _ = build_LR(Ac, m, dt)
    LH = np.conj(np.transpose(L, (0, 2, 1)))
    RH = np.conj(np.transpose(R, (0, 2, 1)))
    X = np.linalg.inv(LH)                                    # L_n^-H
    Y = X @ RH                                               # S_n^H
    XH = np.conj(np.transpose(X, (0, 2, 1)))                 # L_n^-1
    YH = np.conj(np.transpose(Y, (0, 2, 1)))                 # S_n
    phi = np.zeros((nt, N, c), dtype=complex)
    phi[0] = L[0] @ (prob.u0 + 1j * prob.v0)
    for n in range(nsteps):
        phi[n + 1] = YH[n] @ phi[n]
    psi = XH @ phi
    psi[0] = prob.u0 + 1j * prob.v0                          # (the device keeps the given initial state, not L_0^-1 L_0 psi_0)
    W = prob.guard_subspace_projector
    wreal = np.concatenate([psi.real, psi.imag], axis=1)
    Ww = np.einsum("ij,njc->nic", W, wreal)
    trap = np.ones(nt); trap[0] = trap[-1] = 0.5
    guard = (dt / prob.tf) * np.einsum("n,nic,nic->", trap, wreal, Ww)
    f = -(2 * dt / prob.tf) * trap[:, None, None] * Ww
    fc = f[:, :N] + 1j * f[:, N:]
    h = X @ fc                                               # L_n^-H f_n
    T = np.asarray(target)
    ovl = np.sum(np.conj(T) * psi[-1])
    a, b = ovl.real, ovl.imag
    infid = 1 - (a * a + b * b) / prob.N_ess_levels ** 2
    ga, gb = (a, b) if overlap is None else overlap
    lam = np.zeros((nt, N, c), dtype=complex)
    lam[-1] = X[-1] @ ((2 / prob.N_ess_levels ** 2) * (ga + 1j * gb) * T) + h[-1]
    for n in range(nsteps - 1, 0, -1):
        lam[n] = Y[n] @ lam[n + 1] + h[n]
    out = gradient_from(prob, Gp, Gq, offsets, pcof, Ac, psi, lam, m, dt)
    out.update(tp=tp, tq=tq, Ac=Ac, L=L, R=R, X=X, Y=Y, S=YH, phi=phi, psi=psi, h=h, f=fc, lam=lam, guard=guard,
               infidelity=infid, overlap=(a, b))
    return out


def gradient_from(prob, Gp, Gq, offsets, pcof, Ac, psi, lam, m, dt):
    """Stage derivatives, reverse sweep and contraction with the control basis (the tail of evaluate())."""
    nt = psi.shape[0]
    ws = [psi]
    for j in range(m):
        acc = np.zeros_like(psi)
        for i in range(j + 1):
            acc += Ac[:, j - i] @ ws[i]
        ws.append(acc / (j + 1))
    g = [None] * (m + 1)
    lam_next = np.zeros_like(lam); lam_next[:-1] = lam[1:]
    lam_here = lam.copy(); lam_here[0] = 0
    for j in range(m + 1):
        cj = coefficient(j, m, m)
        g[j] = cj * dt ** j * lam_next - cj * (-dt) ** j * lam_here
    AcH = np.conj(np.transpose(Ac, (0, 1, 3, 2)))
    for j in range(m, 1, -1):
        for i in range(1, j):
            g[i] = g[i] + (1.0 / j) * (AcH[:, j - 1 - i] @ g[j])
    nops = prob.N_operators
    sigP = np.zeros((nt, nops, m))
    sigQ = np.zeros((nt, nops, m))
    for k in range(nops):
        Sk, Ak = prob.sym_operators[k], prob.asym_operators[k]
        for j in range(1, m + 1):
            for i in range(j):
                d = j - 1 - i
                Pw = -1j * (Sk @ ws[i])
                Qw = Ak @ ws[i]
                sigP[:, k, d] += (1.0 / j) * np.einsum("nic,nic->n", np.conj(Pw), g[j]).real
                sigQ[:, k, d] += (1.0 / j) * np.einsum("nic,nic->n", np.conj(Qw), g[j]).real
    grad = np.zeros(len(pcof))
    for k, (gp, gq, off) in enumerate(zip(Gp, Gq, offsets)):
        nl = gp.shape[2]
        grad[off:off + nl] -= np.einsum("ndl,nd->l", gp[:, :m], sigP[:, k]) + np.einsum("ndl,nd->l", gq[:, :m], sigQ[:, k])
    return dict(ws=ws, g=g, sigP=sigP, sigQ=sigQ, grad=grad)


def history_real(ws):
    """[2N, 1+m, nt, c] Julia-layout history from the list of complex derivatives."""
    arr = np.stack(ws, axis=0)                       # [1+m, nt, N, c]
    real = np.concatenate([arr.real, arr.imag], axis=2)  # [1+m, nt, 2N, c]
    return np.asfortranarray(np.transpose(real, (2, 0, 1, 3)))


def forced_gradient(prob, Gp, Gq, offsets, pcof, target, order, ref=None, return_all=False):
    """Forward-sensitivity ("forced") gradient in the same time-parallel form the device uses
    (reference: src/eval_grad_forced.jl:17-194 -- one forced forward sweep per control parameter).

    dA_d(t_n)/dtheta_l = Gp_l^(d)(t_n) (-i Sym_k) + Gq_l^(d)(t_n) Asym_k, so the forcing of every
    parameter of control k is a combination of 2m *basis responses* per time point: for
    (tau, d) the Taylor recursion U_0 = 0, U_{j+1} = (sum_i A_{j-i} U_i + [j >= d] Omega w_{j-d})/(j+1),
    rhoR = sum_j c_j dt^j U_j, rhoL = sum_j c_j (-dt)^j U_j.  The sensitivity obeys
        L_{n+1} s_{n+1} = R_n s_n + sum G(t_n) rhoR_n - sum G(t_{n+1}) rhoL_{n+1}.
    """
    if ref is None:
        ref = evaluate(prob, Gp, Gq, offsets, pcof, target, order)
    m = order // 2
    N, c, nsteps = prob.N_tot_levels, prob.N_initial_conditions, prob.nsteps
    dt = prob.tf / nsteps
    nt = nsteps + 1
    Ac, ws, Linv, P, psi = ref["Ac"], ref["ws"], ref["Linv"], ref["P"], ref["psi"]
    cj = [coefficient(j, m, m) for j in range(m + 1)]
    T = np.asarray(target)
    ovl = np.sum(np.conj(T) * psi[-1])
    W = prob.guard_subspace_projector
    trap = np.ones(nt); trap[0] = trap[-1] = 0.5
    wreal = np.concatenate([psi.real, psi.imag], axis=1)
    Ww = np.einsum("ij,njc->nic", W, wreal)
    Wc = Ww[:, :N] + 1j * Ww[:, N:]                       # <Ww, s>_real = Re(conj(Wc) s)
    rhoR = {}
    rhoL = {}
    for k in range(prob.N_operators):
        for tau, Om in (("p", -1j * prob.sym_operators[k]), ("q", prob.asym_operators[k] + 0j)):
            for d in range(m):
                U = [np.zeros((nt, N, c), dtype=complex)]
                for j in range(m):
                    acc = np.zeros((nt, N, c), dtype=complex)
                    for i in range(j + 1):
                        acc += Ac[:, j - i] @ U[i]
                    if j >= d:
                        acc += Om @ ws[j - d]
                    U.append(acc / (j + 1))
                rhoR[k, tau, d] = sum(cj[j] * dt ** j * U[j] for j in range(1, m + 1))
                rhoL[k, tau, d] = sum(cj[j] * (-dt) ** j * U[j] for j in range(1, m + 1))
    grad = np.zeros(len(pcof))
    sN = {}
    for k, (gp, gq, off) in enumerate(zip(Gp, Gq, offsets)):
        for l in range(gp.shape[2]):
            s = np.zeros((N, c), dtype=complex)
            gsum = 0.0                                    # guard part; s_0 = 0 contributes nothing
            for n in range(nsteps):
                r = np.zeros((N, c), dtype=complex)
                for d in range(m):
                    r += gp[n, d, l] * rhoR[k, "p", d][n] + gq[n, d, l] * rhoR[k, "q", d][n]
                    r -= gp[n + 1, d, l] * rhoL[k, "p", d][n + 1] + gq[n + 1, d, l] * rhoL[k, "q", d][n + 1]
                s = P[n] @ s + Linv[n + 1] @ r
                gsum += trap[n + 1] * np.sum(np.conj(Wc[n + 1]) * s).real
            sN[off + l] = s
            grad[off + l] = (-(2.0 / prob.N_ess_levels ** 2) * (np.conj(ovl) * np.sum(np.conj(T) * s)).real
                             + (2.0 * dt / prob.tf) * gsum)
    if return_all:
        return grad, rhoR, rhoL, sN
    return grad
