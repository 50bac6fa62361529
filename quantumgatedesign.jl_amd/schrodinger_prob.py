"""``SchrodingerProb`` -- the problem container that is the drop-in boundary
(reference: src/SchrodingerProb.jl:25-233).

Same fields, same validation and the same error behaviour (Julia's
``ArgumentError`` becomes ``ValueError``).  Matrices are kept as dense
float64 numpy arrays; the reference's sparse/dense choice is a CPU storage
detail -- on the device every operator is a dense, padded, column-blocked
tile (see DESIGN.md "Data layout").
"""
from __future__ import annotations

import numpy as np


def _dense(a) -> np.ndarray:
    if hasattr(a, "toarray"):
        a = a.toarray()
    return np.array(a, dtype=np.float64, order="F")


class SchrodingerProb:
    """SchrodingerProb(system_sym, system_asym, sym_operators, asym_operators,
    u0, v0, guard_subspace_projector, tf, nsteps, N_ess_levels,
    gmres_abstol, gmres_reltol, preconditioner_type)

    or, as the reference's outer constructor (SchrodingerProb.jl:167-233),
    :meth:`from_hamiltonian` with a complex Hermitian system Hamiltonian.
    ``nsteps``, ``gmres_abstol`` and ``gmres_reltol`` are mutable, as scripts
    rely on (examples/cnot3_optimize_gate.jl:51-55).
    """

    def __init__(self, system_sym, system_asym, sym_operators, asym_operators, u0, v0,
                 guard_subspace_projector, tf, nsteps, N_ess_levels,
                 gmres_abstol=1e-10, gmres_reltol=1e-10, preconditioner_type="IdentityPreconditioner"):
        system_sym = _dense(system_sym)
        system_asym = _dense(system_asym)
        sym_operators = [_dense(o) for o in sym_operators]
        asym_operators = [_dense(o) for o in asym_operators]
        u0 = np.array(u0, dtype=np.float64, order="F")
        v0 = np.array(v0, dtype=np.float64, order="F")
        if u0.ndim == 1:
            u0 = u0.reshape(-1, 1, order="F")
        if v0.ndim == 1:
            v0 = v0.reshape(-1, 1, order="F")

        # SchrodingerProb.jl:73-101 -- shape / symmetry checks
        if system_sym.ndim != 2 or system_sym.shape[0] != system_sym.shape[1]:
            raise ValueError("Real part of system Hamiltonian is not square.")
        csize = system_sym.shape
        N = csize[0]
        if not np.array_equal(system_sym, system_sym.T):
            raise ValueError("Real part of system Hamiltonian is not symmetric.")
        for i, op in enumerate(sym_operators):
            if op.shape != csize:
                raise ValueError(f"Size {op.shape} of symmetric operator {i+1} does match size {csize} of system Hamiltonian.")
            if not np.array_equal(op, op.T):
                raise ValueError(f"Symmetric operator {i+1} is not symmetric.")
        if system_asym.shape != csize:
            raise ValueError(f"Size {system_asym.shape} of imaginary part of Hamiltonian does not match size {csize} real part of Hamiltonian.")
        if not np.array_equal(system_asym, -system_asym.T):
            raise ValueError("Imaginary part of system Hamiltonian is not anti-symmetric.")
        for i, op in enumerate(asym_operators):
            if op.shape != csize:
                raise ValueError(f"Size {op.shape} of anti-symmetric operator {i+1} does match size {csize} of system Hamiltonian.")
            if not np.array_equal(op, -op.T):
                raise ValueError(f"Anti-symmetric operator {i+1} is not anti-symmetric.")
        # :118-135
        if u0.shape != v0.shape:
            raise ValueError(f"Size {u0.shape} of the real part of the initial condition does not match the size {v0.shape} of the imaginary part of the initial condition.")
        if u0.shape[0] != N:
            raise ValueError(f"Number of levels {N} in initial condition is inconsistent with the size {csize} of system Hamiltonian.")
        if len(sym_operators) != len(asym_operators):
            raise ValueError(f"Number of symmetric operators {len(sym_operators)} does not match number of anti-symmetric operators {len(asym_operators)}.")
        # :142-150
        if guard_subspace_projector is None:
            guard_subspace_projector = np.zeros((2 * N, 2 * N))
        guard_subspace_projector = _dense(guard_subspace_projector)
        if guard_subspace_projector.shape != (2 * N, 2 * N):
            raise ValueError(f"Guard subspace projector size {guard_subspace_projector.shape} should be twice the size {csize} of the complex-valued system.")
        if N_ess_levels > N:
            raise ValueError(f"Number of essential levels {N_ess_levels} cannot be greater than the total number of levels {N}.")
        if preconditioner_type not in ("IdentityPreconditioner", "LUPreconditioner", "DiagonalHamiltonianPreconditioner"):
            raise ValueError("preconditioner_type is not an AbstractQGDPreconditioner.")

        self.system_sym = system_sym
        self.system_asym = system_asym
        self.sym_operators = sym_operators
        self.asym_operators = asym_operators
        self.u0 = u0
        self.v0 = v0
        self.guard_subspace_projector = guard_subspace_projector
        self.tf = float(tf)
        self.nsteps = int(nsteps)
        self.N_initial_conditions = u0.shape[1]
        self.N_ess_levels = int(N_ess_levels)
        self.N_tot_levels = N
        self.N_operators = len(sym_operators)
        self.real_system_size = 2 * N
        self.gmres_abstol = float(gmres_abstol)
        self.gmres_reltol = float(gmres_reltol)
        self.preconditioner_type = preconditioner_type

    @classmethod
    def from_hamiltonian(cls, system_hamiltonian, sym_operators, asym_operators, U0, tf, nsteps,
                         N_ess_levels, guard_subspace_projector=None, gmres_abstol=1e-10,
                         gmres_reltol=1e-10, preconditioner_type="IdentityPreconditioner"):
        """Outer constructor, SchrodingerProb.jl:167-233."""
        H = np.asarray(system_hamiltonian.toarray() if hasattr(system_hamiltonian, "toarray") else system_hamiltonian)
        if not np.array_equal(H, H.conj().T):
            raise ValueError("System Hamiltonian is not Hermitian.")
        U0 = np.asarray(U0)
        return cls(np.real(H), np.imag(H), sym_operators, asym_operators, np.real(U0), np.imag(U0),
                   guard_subspace_projector, tf, nsteps, N_ess_levels, gmres_abstol, gmres_reltol,
                   preconditioner_type)

    def copy(self):
        """SchrodingerProb.jl:237-252."""
        return SchrodingerProb(self.system_sym.copy(), self.system_asym.copy(),
                               [o.copy() for o in self.sym_operators], [o.copy() for o in self.asym_operators],
                               self.u0.copy(), self.v0.copy(), self.guard_subspace_projector.copy(),
                               self.tf, self.nsteps, self.N_ess_levels, self.gmres_abstol, self.gmres_reltol,
                               self.preconditioner_type)

    def vector_prob(self, initial_condition_index: int):
        """VectorSchrodingerProb, SchrodingerProb.jl:257-272 (0-based index)."""
        j = initial_condition_index
        return SchrodingerProb(self.system_sym, self.system_asym, self.sym_operators, self.asym_operators,
                               self.u0[:, j], self.v0[:, j], self.guard_subspace_projector, self.tf,
                               self.nsteps, self.N_ess_levels, self.gmres_abstol, self.gmres_reltol,
                               self.preconditioner_type)

    def __repr__(self):
        return (f"SchrodingerProb(N_tot_levels={self.N_tot_levels}, N_ess_levels={self.N_ess_levels}, "
                f"N_operators={self.N_operators}, N_initial_conditions={self.N_initial_conditions}, "
                f"tf={self.tf}, nsteps={self.nsteps})")
