"""Problem constructors -- host-side mirror of
src/ProblemConstructors/{multi_qudit_systems,rabi_oscillator,random_problem}.jl.

These are setup code (run once); they reproduce the reference's *outputs*
(matrices, initial conditions, guard projector, bit-string ordering) so that a
problem built here is the problem the reference would build.
"""
from __future__ import annotations

import itertools

import numpy as np

from .schrodinger_prob import SchrodingerProb


def _julia_product(ranges):
    """Iterate like Julia's Iterators.product: FIRST range fastest."""
    for tup in itertools.product(*[list(r) for r in reversed(list(ranges))]):
        yield tuple(reversed(tup))


def lowering_operator_subsystem(subsystem_size: int) -> np.ndarray:
    """multi_qudit_systems.jl:354-358."""
    return np.diag(np.sqrt(np.arange(1, subsystem_size, dtype=float)), k=1)


def lowering_operators_system(subsystem_sizes, bitstring_ordered=True):
    """Lowering operator of each subsystem embedded in the full system by
    Kronecker products, first subsystem = leftmost bit (multi_qudit_systems.jl:364-389)."""
    if not bitstring_ordered:
        raise ValueError("bitstring_ordered=false not yet supported.")
    ops = []
    for i, n in enumerate(subsystem_sizes):
        mats = [np.eye(s) for s in subsystem_sizes]
        mats[i] = lowering_operator_subsystem(n)
        full = mats[0]
        for mtx in mats[1:]:
            full = np.kron(full, mtx)
        ops.append(full)
    return ops


def multi_qudit_hamiltonian_dispersive(subsystem_sizes, transition_freqs, rotation_freqs, kerr_coeffs):
    """multi_qudit_systems.jl:26-58 (complex Hermitian, here real diagonal-ish)."""
    kerr = np.asarray(kerr_coeffs, float)
    Q = len(subsystem_sizes)
    assert len(transition_freqs) == kerr.shape[0] == kerr.shape[1]
    assert np.array_equal(kerr, kerr.T)
    n = int(np.prod(subsystem_sizes))
    H = np.zeros((n, n), dtype=complex)
    low = lowering_operators_system(subsystem_sizes)
    for q in range(Q):
        a = low[q]
        H += (transition_freqs[q] - rotation_freqs[q]) * (a.T @ a)
        H -= 0.5 * kerr[q, q] * (a.T @ a.T @ a @ a)
        for p in range(q + 1, Q):
            ap = low[p]
            H -= kerr[p, q] * (ap.T @ ap @ a.T @ a)
    return H


def control_ops(subsystem_sizes):
    """a + a', a - a' per subsystem (multi_qudit_systems.jl:60-71)."""
    low = lowering_operators_system(subsystem_sizes)
    return [a + a.T for a in low], [a - a.T for a in low]


def basis_state(subsystem_sizes, subsystem_indices, bitstring_ordered=True):
    """multi_qudit_systems.jl:229-253: |n0 n1 ...> with the last subsystem fastest."""
    sizes = list(subsystem_sizes)
    idx = list(subsystem_indices)
    if any(i + 1 > s for i, s in zip(idx, sizes)):
        raise ValueError(f"Subsystem indices {tuple(idx)} are invalid for subsystem sizes {tuple(sizes)}.")
    if bitstring_ordered:
        sizes, idx = sizes[::-1], idx[::-1]
    tensor = np.zeros(sizes, order="F")
    tensor[tuple(idx)] = 1.0
    return tensor.reshape(-1, order="F")


def create_initial_conditions(subsystem_sizes, essential_subsystem_sizes, bitstring_ordered=True):
    """Essential basis states as columns (multi_qudit_systems.jl:255-279)."""
    n = int(np.prod(subsystem_sizes))
    ne = int(np.prod(essential_subsystem_sizes))
    U0 = np.zeros((n, ne), dtype=complex)
    ranges = [range(e) for e in essential_subsystem_sizes]
    if bitstring_ordered:
        ranges = ranges[::-1]
    for i, sub in enumerate(_julia_product(ranges)):
        if bitstring_ordered:
            sub = sub[::-1]
        U0[:, i] = basis_state(subsystem_sizes, sub, bitstring_ordered)
    return U0


def guard_projector(subsystem_sizes, essential_subsystem_sizes, bitstring_ordered=True):
    """Real-valued projector onto the guard levels, ``[G 0; 0 G]``
    (multi_qudit_systems.jl:316-349).  The essential test compares the indices in
    iteration order with ``essential_subsystem_sizes`` exactly as the reference
    does (:335), which its doc examples (:291-314) pin."""
    n = int(np.prod(subsystem_sizes))
    G = np.zeros((n, n))
    ranges = [range(s) for s in subsystem_sizes]
    if bitstring_ordered:
        ranges = ranges[::-1]
    ess = list(essential_subsystem_sizes)
    for i, sub in enumerate(_julia_product(ranges)):
        if all(s < e for s, e in zip(sub, ess)):
            continue
        if bitstring_ordered:
            sub = sub[::-1]
        G[:, i] = basis_state(subsystem_sizes, sub, bitstring_ordered)
    Z = np.zeros((n, n))
    return np.block([[G, Z], [Z, G]])


def create_gate(subsystem_sizes, essential_subsystem_sizes, initial_final_pairs, bitstring_ordered=True):
    """Target gate as columns: identity on the essential basis except for the listed
    (initial -> final) pairs (multi_qudit_systems.jl:391-410)."""
    G = create_initial_conditions(subsystem_sizes, essential_subsystem_sizes, bitstring_ordered)
    ranges = [range(e) for e in essential_subsystem_sizes]
    if bitstring_ordered:
        ranges = ranges[::-1]
    ordered = list(_julia_product(ranges))
    for first, second in initial_final_pairs:
        i = ordered.index(tuple(reversed(tuple(first))))
        G[:, i] = basis_state(subsystem_sizes, second)
    return G


def DispersiveProblem(subsystem_sizes, essential_subsystem_sizes, transition_freqs, rotation_freqs,
                      kerr_coeffs, tf, nsteps, sparse_rep=True, bitstring_ordered=True,
                      gmres_abstol=1e-10, gmres_reltol=1e-10,
                      preconditioner_type="DiagonalHamiltonianPreconditioner"):
    """multi_qudit_systems.jl:118-162.  ``sparse_rep`` is accepted for signature
    compatibility; device storage is always dense tiles."""
    H = multi_qudit_hamiltonian_dispersive(subsystem_sizes, transition_freqs, rotation_freqs, kerr_coeffs)
    sym_ops, asym_ops = control_ops(subsystem_sizes)
    guard = guard_projector(subsystem_sizes, essential_subsystem_sizes)
    U0 = create_initial_conditions(subsystem_sizes, essential_subsystem_sizes, bitstring_ordered)
    return SchrodingerProb.from_hamiltonian(H, sym_ops, asym_ops, U0, tf, nsteps,
                                            int(np.prod(essential_subsystem_sizes)), guard,
                                            gmres_abstol=gmres_abstol, gmres_reltol=gmres_reltol,
                                            preconditioner_type=preconditioner_type)


def multi_qudit_hamiltonian_jayne(subsystem_sizes, transition_freqs, rotation_freq, kerr_coeffs, jayne_cummings_coeffs):
    """multi_qudit_systems.jl:81-116: one rotation frequency for all subsystems, self- and cross-Kerr
    terms, and the exchange coupling g_pq (a_q' a_p + a_q a_p')."""
    kerr = np.asarray(kerr_coeffs, float)
    jc = np.asarray(jayne_cummings_coeffs, float)
    assert np.array_equal(kerr, kerr.T) and np.array_equal(jc, jc.T) and not np.any(np.diag(jc))
    Q = len(subsystem_sizes)
    n = int(np.prod(subsystem_sizes))
    H = np.zeros((n, n), dtype=complex)
    low = lowering_operators_system(subsystem_sizes)
    for q in range(Q):
        a = low[q]
        H += (transition_freqs[q] - rotation_freq) * (a.T @ a)
        H -= 0.5 * kerr[q, q] * (a.T @ a.T @ a @ a)
        for p in range(q + 1, Q):
            ap = low[p]
            H -= kerr[p, q] * (ap.T @ ap @ a.T @ a)
            H += jc[p, q] * (a.T @ ap + a @ ap.T)
    return H


def JaynesCummingsProblem(subsystem_sizes, essential_subsystem_sizes, transition_freqs, rotation_freq, kerr_coeffs,
                          jayne_cummings_coeffs, tf, nsteps, sparse_rep=True, bitstring_ordered=True,
                          gmres_abstol=1e-10, gmres_reltol=1e-10, preconditioner_type="LUPreconditioner"):
    """multi_qudit_systems.jl:169-217 (the reference body refers to undefined ``u0, v0``; the evident
    intent -- the initial conditions it builds two lines earlier -- is used)."""
    H = multi_qudit_hamiltonian_jayne(subsystem_sizes, transition_freqs, rotation_freq, kerr_coeffs, jayne_cummings_coeffs)
    sym_ops, asym_ops = control_ops(subsystem_sizes)
    guard = guard_projector(subsystem_sizes, essential_subsystem_sizes)
    U0 = create_initial_conditions(subsystem_sizes, essential_subsystem_sizes, bitstring_ordered)
    return SchrodingerProb.from_hamiltonian(H, sym_ops, asym_ops, U0, tf, nsteps,
                                            int(np.prod(essential_subsystem_sizes)), guard,
                                            gmres_abstol=gmres_abstol, gmres_reltol=gmres_reltol,
                                            preconditioner_type=preconditioner_type)


def rotating_frame_qubit(N_ess_levels, N_guard_levels, tf=1.0, nsteps=10, detuning_frequency=1.0,
                         self_kerr_coefficient=1.0):
    """Single qubit in the rotating frame (rotating_frame_qubit.jl:8-41): frequencies in GHz, multiplied
    by 2 pi; controls a + a', a - a'; initial conditions = the essential basis states."""
    n = int(N_ess_levels) + int(N_guard_levels)
    a = np.diag(np.sqrt(np.arange(1.0, n)), 1)
    S = 2 * np.pi * detuning_frequency * (a.T @ a) - 0.5 * 2 * np.pi * self_kerr_coefficient * (a.T @ a.T @ a @ a)
    u0 = np.zeros((n, N_ess_levels)); u0[np.arange(N_ess_levels), np.arange(N_ess_levels)] = 1.0
    return SchrodingerProb(S, np.zeros((n, n)), [a + a.T], [a - a.T], u0, np.zeros_like(u0), None, tf, nsteps, N_ess_levels)


def dahlquist_problem(lam, initial_condition=1.0, with_control=False):
    """The scalar test equation y' = lambda y (dahlquist_problem.jl:1-47): H = i lambda must be real
    (lambda purely imaginary); tf = 1, 10 steps; optionally one symmetric control operator [1]."""
    Hs = 1j * complex(lam)
    if abs(Hs.imag) > 0:
        raise ValueError("lambda must be purely imaginary (the scalar Hamiltonian i*lambda must be Hermitian)")
    ic = complex(initial_condition)
    sym_ops = [np.ones((1, 1))] if with_control else []
    asym_ops = [np.zeros((1, 1))] if with_control else []
    return SchrodingerProb(np.array([[Hs.real]]), np.array([[0.0]]), sym_ops, asym_ops,
                           np.array([[ic.real]]), np.array([[ic.imag]]), None, 1.0, 10, 1)


def construct_rabi_prob(tf=np.pi, gmres_abstol=1e-10, gmres_reltol=1e-10, nsteps=100):
    """Two-level Rabi oscillator (rabi_oscillator.jl:7-22): |Omega| = 1/2 for
    tf = pi gives a SWAP."""
    a = np.array([[0.0, 1.0], [0.0, 0.0]])
    return SchrodingerProb.from_hamiltonian(np.zeros((2, 2)), [a + a.T], [a - a.T], np.eye(2),
                                            tf, nsteps, 2, gmres_abstol=gmres_abstol,
                                            gmres_reltol=gmres_reltol)


def construct_rand_prob(complex_system_size, N_operators, tf=2.0, nsteps=100, gmres_abstol=1e-10,
                        gmres_reltol=1e-10, scale=1.0):
    """Random dense problem of the reference's shape (random_problem.jl:15-35).

    Julia's MersenneTwister streams are not reproducible outside Julia, so the
    entries come from numpy's PCG64 with the reference's seeds (0, 2, 3, 100+i,
    200+i); ``scale`` multiplies every Hamiltonian entry (BASELINE.md uses 1/N
    for the N=256 synthetic problem)."""
    N = complex_system_size
    rng = np.random.default_rng(0)
    U0 = rng.random((N, N)) + 1j * rng.random((N, N))

    def sym(seed):
        r = np.random.default_rng(seed).random((N, N))
        return (r + r.T) * scale

    def asym(seed):
        r = np.random.default_rng(seed).random((N, N))
        return (r - r.T) * scale

    H = sym(2) + 1j * asym(3)
    sym_ops = [sym(100 + i) for i in range(1, N_operators + 1)]
    asym_ops = [asym(200 + i) for i in range(1, N_operators + 1)]
    return SchrodingerProb.from_hamiltonian(H, sym_ops, asym_ops, U0, tf, nsteps, N,
                                            gmres_abstol=gmres_abstol, gmres_reltol=gmres_reltol)


# ---------------------------------------------------------------------------
# The BASELINE.json configurations
# ---------------------------------------------------------------------------
def cnot2_problem(nsteps=100, tf=100.0):
    """examples/cnot2_optimization.jl:10-37: returns (prob, target)."""
    freqs = 2 * np.pi * np.array([4.10595, 4.81526])
    xa, xb, xab = 2 * 0.1099, 2 * 0.1126, 1e-2
    kerr = 2 * np.pi * np.array([[xa, xab], [xab, xb]])
    prob = DispersiveProblem((2, 2), (2, 2), freqs, freqs, kerr, tf, nsteps)
    target = np.eye(prob.N_tot_levels, prob.N_initial_conditions, dtype=complex)
    return prob, target


def cnot3_problem(nsteps=550, tf=550.0):
    """Hamiltonian of examples/regression.jl:6-31 (subsystems (4,4,4), essential
    (2,2,2)); target = CNOT on qubits a,b, identity on s, built with create_gate.
    tf/nsteps default to the headline grid of BASELINE.md (dt = 1.0,
    examples/cnot3_optimize_gate.sb:36)."""
    freqs = 2 * np.pi * np.array([4.10595, 4.81526, 7.8447])
    xa, xb = 2 * 0.1099, 2 * 0.1126
    xs = 0.002494 ** 2 / xa
    xab = 1e-6
    xas, xbs = np.sqrt(xa * xs), np.sqrt(xb * xs)
    kerr = 2 * np.pi * np.array([[xa, xab, xas], [xab, xb, xbs], [xas, xbs, xs]])
    sizes, ess = (4, 4, 4), (2, 2, 2)
    prob = DispersiveProblem(sizes, ess, freqs, freqs, kerr, tf, nsteps, sparse_rep=False)
    pairs = [((1, 0, s), (1, 1, s)) for s in (0, 1)] + [((1, 1, s), (1, 0, s)) for s in (0, 1)]
    target = create_gate(sizes, ess, pairs)
    return prob, target
