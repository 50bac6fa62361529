"""numpy stand-in for quantumgatedesign.jl_amd.distributed.DeviceBackend (TEST INFRASTRUCTURE):
the same phases, partition rule and exchange-buffer layout as the C ABI's qgd_dist_* entry
points, on the CPU, so that the multi-rank orchestration (TimePartitioned + TorchComm) can be
exercised with gloo where no GPU exists."""
import math

import numpy as np
import torch

import proto_propagator as pp


def partition(S, world, rank):
    """Same rule as alloc_grid() in csrc/qgd_host_alloc.cpp."""
    B0 = int(round(S ** (2.0 / 3.0)))
    if S < 24:
        B0 = 1
    B0 = max(1, min(64, B0))
    bpr = (B0 + world - 1) // world
    while True:   # every rank must own at least one non-empty block
        B = bpr * world
        blen = (S + B - 1) // B
        nonempty = (S + blen - 1) // blen
        if (world - 1) * bpr < nonempty or bpr == 1:
            break
        bpr -= 1
    blk_lo, blk_hi = rank * bpr, (rank + 1) * bpr
    s_lo, s_hi = blk_lo * blen, min(S, blk_hi * blen)
    assert s_lo < S, "rank would own no step"
    return dict(B=B, bpr=bpr, blen=blen, blk_lo=blk_lo, blk_hi=blk_hi, n_lo=s_lo, n_hi=s_hi)


class NumpyBackend:
    def __init__(self, qgd, prob, order, controls, target, rank, world):
        self.prob, self.m, self.rank, self.world = prob, order // 2, rank, world
        self.N, self.c, self.S = prob.N_tot_levels, prob.N_initial_conditions, prob.nsteps
        self.dt = prob.tf / prob.nsteps
        self.p = partition(self.S, world, rank)
        self.n_lo, self.n_hi = self.p["n_lo"], self.p["n_hi"]
        Gp, Gq, self.off = qgd.control_basis(controls, prob.nsteps, prob.tf, self.m)
        self.Gp = [g[self.n_lo:self.n_hi + 1] for g in Gp]
        self.Gq = [g[self.n_lo:self.n_hi + 1] for g in Gq]
        self.target = np.asarray(target, dtype=complex)
        self.n_pcof = sum(g.shape[2] for g in Gp)
        N, c, bpr = self.N, self.c, self.p["bpr"]
        self.PiX = torch.zeros(world * bpr * N * N * 2, dtype=torch.float64)
        self.phiX = torch.zeros(world * (bpr + 1) * N * c * 2, dtype=torch.float64)
        self.red = torch.zeros(self.n_pcof + 4, dtype=torch.float64)

    # views of the exchange buffers as complex numpy arrays
    def _pi(self):
        return self.PiX.numpy().view(np.complex128).reshape(self.world * self.p["bpr"], self.N, self.N)

    def _phi(self):
        return self.phiX.numpy().view(np.complex128).reshape(self.world, self.p["bpr"] + 1, self.N, self.c)

    def exchange_buffer(self, which):
        if which == 0:
            n = self.p["bpr"] * self.N * self.N * 2
            return self.PiX, self.PiX[self.rank * n:(self.rank + 1) * n]
        if which == 1:
            n = (self.p["bpr"] + 1) * self.N * self.c * 2
            return self.phiX, self.phiX[self.rank * n:(self.rank + 1) * n]
        return self.red, self.red

    def _block_range(self, b):
        """local step range of global block b (owned)"""
        s = b * self.p["blen"] - self.n_lo
        e = min(self.S, (b + 1) * self.p["blen"]) - self.n_lo
        return s, max(s, e)

    def forward_begin(self, pcof):
        self.pcof = np.asarray(pcof, float)
        m, dt = self.m, self.dt
        tp, tq = pp.tables(self.Gp, self.Gq, self.off, self.pcof, m)
        self.Ac = pp.assemble(self.prob, tp, tq, m)
        L, R, _ = pp.build_LR(self.Ac, m, dt)
        self.Linv = np.linalg.inv(L)
        self.P = self.Linv[1:] @ R[:-1]                      # local steps
        pi = self._pi()
        for b in range(self.p["blk_lo"], self.p["blk_hi"]):
            s, e = self._block_range(b)
            M = np.eye(self.N, dtype=complex)
            for n in range(s, e):
                M = self.P[n] @ M
            pi[b] = M

    def forward_end(self):
        pi, B, N, c = self._pi(), self.p["B"], self.N, self.c
        bnd = np.zeros((B + 1, N, c), dtype=complex)
        bnd[0] = self.prob.u0 + 1j * self.prob.v0
        for b in range(B):
            bnd[b + 1] = pi[b] @ bnd[b]
        self.bnd = bnd
        nt = self.n_hi - self.n_lo + 1
        psi = np.zeros((nt, N, c), dtype=complex)
        psi[0] = bnd[self.p["blk_lo"]]
        for n in range(nt - 1):
            psi[n + 1] = self.P[n] @ psi[n]
        self.psi = psi
        W = self.prob.guard_subspace_projector
        w = np.concatenate([psi.real, psi.imag], axis=1)
        Ww = np.einsum("ij,njc->nic", W, w)
        ng = np.arange(self.n_lo, self.n_hi + 1)
        trap = np.where((ng == 0) | (ng == self.S), 0.5, 1.0)
        count = trap.copy()
        if self.n_lo != 0:
            count[0] = 0.0                                    # shared with the previous rank
        self.guard = (self.dt / self.prob.tf) * np.einsum("n,nic,nic->", count, w, Ww)
        f = -(2 * self.dt / self.prob.tf) * trap[:, None, None] * Ww
        self.f = f[:, :N] + 1j * f[:, N:]
        self.a = self.b_ = 0.0
        self.yN = np.zeros((N, c), dtype=complex)
        if self.rank == self.world - 1:
            ovl = np.sum(np.conj(self.target) * psi[-1])
            self.a, self.b_ = ovl.real, -ovl.imag
            self.yN = (2 / self.prob.N_ess_levels ** 2) * ovl * self.target + self.f[-1]

    def adjoint_begin(self):
        phi = self._phi()
        for j, b in enumerate(range(self.p["blk_lo"], self.p["blk_hi"])):
            s, e = self._block_range(b)
            y = np.zeros((self.N, self.c), dtype=complex)
            for n in range(e - 1, s - 1, -1):
                y = self.P[n].conj().T @ y + self.f[n]
            phi[self.rank, j] = y
        phi[self.rank, self.p["bpr"]] = self.yN

    def adjoint_end(self):
        pi, phi, B, bpr = self._pi(), self._phi(), self.p["B"], self.p["bpr"]
        bndY = np.zeros((B + 1, self.N, self.c), dtype=complex)
        bndY[B] = phi[self.world - 1, bpr]
        for b in range(B - 1, -1, -1):
            bndY[b] = pi[b].conj().T @ bndY[b + 1] + phi[b // bpr, b % bpr]
        nt = self.n_hi - self.n_lo + 1
        y = np.zeros((nt, self.N, self.c), dtype=complex)
        y[nt - 1] = bndY[self.p["blk_hi"]]
        for n in range(nt - 2, -1, -1):
            y[n] = self.P[n].conj().T @ y[n + 1] + self.f[n]
        lam = np.zeros_like(y)
        lam[1:] = np.conj(np.transpose(self.Linv[1:], (0, 2, 1))) @ y[1:]
        m, dt, psi, Ac = self.m, self.dt, self.psi, self.Ac
        ws = [psi]
        for j in range(m):
            acc = np.zeros_like(psi)
            for i in range(j + 1):
                acc += Ac[:, j - i] @ ws[i]
            ws.append(acc / (j + 1))
        lam_next = np.zeros_like(lam); lam_next[:-1] = lam[1:]
        lam_here = lam.copy(); lam_here[0] = 0
        g = [None] * (m + 1)
        for j in range(m + 1):
            cj = pp.coefficient(j, m, m)
            g[j] = cj * dt ** j * lam_next - cj * (-dt) ** j * lam_here
        AcH = np.conj(np.transpose(Ac, (0, 1, 3, 2)))
        for j in range(m, 1, -1):
            for i in range(1, j):
                g[i] = g[i] + (1.0 / j) * (AcH[:, j - 1 - i] @ g[j])
        grad = np.zeros(self.n_pcof)
        for k in range(self.prob.N_operators):
            Sk, Ak = self.prob.sym_operators[k], self.prob.asym_operators[k]
            sigP = np.zeros((nt, m)); sigQ = np.zeros((nt, m))
            for j in range(1, m + 1):
                for i in range(j):
                    d = j - 1 - i
                    sigP[:, d] += (1.0 / j) * np.einsum("nic,nic->n", np.conj(-1j * (Sk @ ws[i])), g[j]).real
                    sigQ[:, d] += (1.0 / j) * np.einsum("nic,nic->n", np.conj(Ak @ ws[i]), g[j]).real
            nl = self.Gp[k].shape[2]
            grad[self.off[k]:self.off[k] + nl] -= (np.einsum("ndl,nd->l", self.Gp[k][:, :m], sigP)
                                                  + np.einsum("ndl,nd->l", self.Gq[k][:, :m], sigQ))
        r = self.red.numpy()
        r[:self.n_pcof] = grad
        r[self.n_pcof:self.n_pcof + 4] = [self.a, self.b_, self.guard, 0.0]

    def finish(self):
        r = self.red.numpy()
        return r[:self.n_pcof].copy(), r[self.n_pcof:self.n_pcof + 3].copy()



class NumpyColumnBackend:
    """numpy stand-in for quantumgatedesign.jl_amd.distributed.ColumnBackend (TEST INFRASTRUCTURE): a block of the
    initial-condition columns per rank, the same two exchange buffers (3: the overlap scalars, 2: [grad | scalars])."""

    def __init__(self, qgd, prob, order, controls, target, rank, world):
        c = prob.N_initial_conditions
        lo, hi = rank * c // world, (rank + 1) * c // world
        self.sub = prob.copy()
        self.sub.u0 = np.asfortranarray(prob.u0[:, lo:hi]); self.sub.v0 = np.asfortranarray(prob.v0[:, lo:hi])
        self.sub.N_initial_conditions = hi - lo
        self.order, self.rank = order, rank
        self.target = np.asarray(target, dtype=complex)[:, lo:hi]
        self.Gp, self.Gq, self.off = qgd.control_basis(controls, prob.nsteps, prob.tf, order // 2)
        self.n_pcof = sum(g.shape[2] for g in self.Gp)
        self.red = torch.zeros(self.n_pcof + 4, dtype=torch.float64)

    def exchange_buffer(self, which):
        whole = self.red[self.n_pcof:self.n_pcof + 3] if which == 3 else self.red
        return whole, whole

    def forward(self, pcof):
        self.pcof = np.asarray(pcof, dtype=float)
        ref = pp.evaluate(self.sub, self.Gp, self.Gq, self.off, self.pcof, self.target, self.order)
        self.red[self.n_pcof:self.n_pcof + 3] = torch.tensor([ref["overlap"][0], ref["overlap"][1], ref["guard"]])

    def adjoint(self):
        a, b = float(self.red[self.n_pcof]), float(self.red[self.n_pcof + 1])
        ref = pp.evaluate(self.sub, self.Gp, self.Gq, self.off, self.pcof, self.target, self.order, overlap=(a, b))
        self.red[:self.n_pcof] = torch.from_numpy(ref["grad"])
        if self.rank != 0:
            self.red[self.n_pcof:self.n_pcof + 3] = 0.0

    def finish(self):
        r = self.red.numpy()
        return r[:self.n_pcof].copy(), r[self.n_pcof:self.n_pcof + 3].copy()
