"""qgd-hip: MI355X-native Hermite time stepper and discrete-adjoint gradient
behind the reference's SchrodingerProb / eval_forward / discrete_adjoint /
optimize_gate interface (leespen1/QuantumGateDesign.jl).

The directory name contains a dot, so the package is loaded through
``__graft_entry__.import_package()`` (registers it as ``qgd_amd``).
"""
from .schrodinger_prob import SchrodingerProb
from .controls import (AbstractControl, GRAPEControl, FortranBSplineControl, GeneralBSplineControl,
                       CarrierControl, BSpline2Control, BSplineControl, ZeroControl, GeneralGRAPEControl,
                       SinCosControl, SinControl, CosControl, SingleSymCosControl, HermiteControl, HermiteCarrierControl, get_number_of_control_parameters, get_control_vector_slice,
                       fill_p_mat, fill_q_mat, control_basis, bspline_basis_derivatives)
from .problems import (DispersiveProblem, construct_rabi_prob, construct_rand_prob, guard_projector,
                       create_initial_conditions, create_gate, basis_state, lowering_operators_system,
                       control_ops, multi_qudit_hamiltonian_dispersive, cnot2_problem, cnot3_problem,
                       multi_qudit_hamiltonian_jayne, JaynesCummingsProblem, rotating_frame_qubit, dahlquist_problem)
from .evolution import (DeviceProblem, device_problem, clear_cache, release, eval_forward, eval_forward_, eval_adjoint, eval_grad_forced, eval_grad_finite_difference, discrete_adjoint,
                        discrete_adjoint_, infidelity, infidelity_real, guard_penalty_real, complex_to_real,
                        real_to_complex)
from .distributed import (DeviceBackend, TimePartitioned, TorchComm, LocalGroup, ColumnBackend, ColumnSharded,
                          RcclEvaluation, comm_unique_id)
from .optimize import optimize_gate, OptimizationHistory, read_optimization_history
from .convergence import (get_histories, richardson_extrap_rel_err, richardson_extrap_sol, observed_orders,
                          save_histories, load_histories)
from . import _lib, jld2io
