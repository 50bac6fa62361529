"""numpy statement of the column-block Gauss-Jordan elimination of [L | R] that k_inverse_cb runs (development aid of
scripts/ubench/inverse_cb_bench.hip; not the oracle and not the product).

Block step k (16 columns): the owner of column block k eliminates its 64 x 16 panel in four 4-pivot sub-panels with
partial pivoting CONFINED to the rows of diagonal block k (implicit: rows never move), which leaves
    rows of block k:   D^-1        other rows:  -F D^-1
in place up to the pivot permutation of block k.  Every other column block J (of L and of R) then takes ONE rank-16
update  M[:, J] += (G - I_K) M[K, J]  whose right operand is its own row block k.  No permutation survives a block step.
Run as a script it applies this to the 550 step matrices of the cnot3 headline and prints the error against
numpy.linalg.inv and the largest multiplier modulus (the growth flag's input).
"""
import sys

import numpy as np


def panel_gj(Pn, k, nb=16, sub=4):
    """In-place pivoted Gauss-Jordan on the N x nb panel Pn (copy), pivots among rows of block k only.  Returns the panel in
    NATURAL order (permutation resolved), i.e. G with G[K] = D^-1, G[I] = -F D^-1, and the largest multiplier modulus."""
    N = Pn.shape[0]
    Pn = Pn.copy()
    K = slice(k * nb, (k + 1) * nb)
    used = np.ones(N, bool); used[K] = False
    rho = np.zeros(nb, int)
    growth = 0.0
    for sp in range(nb // sub):
        cols = slice(sp * sub, (sp + 1) * sub)
        X = Pn[:, cols].copy()                       # the chain's registers, lane = row
        prow = []
        for s in range(sub):
            mag = np.where(used, -1.0, np.abs(X[:, s]))
            pr = int(np.argmax(mag))
            used[pr] = True
            rho[sp * sub + s] = pr
            prow.append(pr)
            y = X[pr].copy()
            inv = 1.0 / y[s]
            r = y * inv; r[s] = inv
            f = X[:, s].copy()
            b = X.copy(); b[:, s] = 0
            Xn = b - np.outer(f, r)
            Xn[pr] = r
            X = Xn
        # rank-`sub` update of the rest of the panel: M += A M[P, :], A = multipliers minus identity on the pivot rows
        A = X.copy()
        for s, pr in enumerate(prow):
            A[pr, s] -= 1.0
        growth = max(growth, np.abs(X).max())
        B = Pn[prow, :].copy()
        Pn = Pn + A @ B
        Pn[:, cols] = X
    # natural order: G[rinv'(x)][rho_local(j)] = Pn[x][j]
    G = np.zeros_like(Pn)
    rowmap = np.arange(N)
    for p in range(nb):
        rowmap[rho[p]] = k * nb + p
    colmap = rho - k * nb
    G[np.ix_(rowmap, colmap)] = Pn
    return G, growth


def block_inverse(L, R, nb=16):
    N = L.shape[0]
    M = np.concatenate([L, R], axis=1).astype(complex)
    growth = 0.0
    for k in range(N // nb):
        K = slice(k * nb, (k + 1) * nb)
        G, g = panel_gj(M[:, K], k, nb)
        growth = max(growth, g)
        A = G.copy(); A[K] -= np.eye(nb)
        B = M[K, :].copy()
        M = M + A @ B
        M[:, K] = G
    return M[:, :N], M[:, N:], growth


if __name__ == "__main__":
    sys.path.insert(0, "tests"); sys.path.insert(0, ".")
    from __graft_entry__ import import_package
    qgd = import_package()
    import cases
    import proto_propagator as pp
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 550
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=float(nsteps))
    m = 4
    Gp, Gq, offsets = qgd.control_basis(ctrl, prob.nsteps, prob.tf, m)[:3]
    tp, tq = pp.tables(Gp, Gq, offsets, pcof, m)
    Ac = pp.assemble(prob, tp, tq, m)
    L, R, _ = pp.build_LR(Ac, m, prob.tf / prob.nsteps)
    worst = 0.0; gmax = 0.0; dmin = 1e9
    for n in range(1, nsteps + 1, max(1, nsteps // 50)):
        Li, P, g = block_inverse(L[n], R[n - 1])
        ref = np.linalg.inv(L[n])
        worst = max(worst, np.abs(Li - ref).max() / np.abs(ref).max(), np.abs(P - ref @ R[n - 1]).max())
        gmax = max(gmax, g)
        dmin = min(dmin, np.abs(np.diag(L[n])).min())
    print(f"cnot3 {nsteps} steps: max rel err {worst:.2e}, largest multiplier {gmax:.3f}, smallest |L_ii| {dmin:.3f}, cond {np.linalg.cond(L[1]):.2f}")
    rng = np.random.default_rng(0)
    for trial in range(3):
        A = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
        Rr = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
        Li, P, g = block_inverse(A, Rr)
        ref = np.linalg.inv(A)
        print(f"random dense: rel err {np.abs(Li - ref).max() / np.abs(ref).max():.2e}, multiplier {g:.1f}, cond {np.linalg.cond(A):.1f}")
