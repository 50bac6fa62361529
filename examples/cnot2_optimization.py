"""Mirror of the reference's examples/cnot2_optimization.jl on the device path:
2-qubit dispersive CNOT-frame problem, degree-2 B-spline controls with 10 knots (22 coefficients
per control), target = identity on the 4 essential states, order 4, 70 L-BFGS iterations,
|pcof| <= 0.5."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime for the process)
from __graft_entry__ import import_package

qgd = import_package()
prob, target = qgd.cnot2_problem(nsteps=100, tf=100.0)                    # cnot2_optimization.jl:10-37
controls = [qgd.GeneralBSplineControl(2, 10, prob.tf) for _ in range(prob.N_operators)]
pcof = 1e-2 * (0.5 - np.random.default_rng(1).random(qgd.get_number_of_control_parameters(controls)))

history = qgd.eval_forward(prob, controls, pcof, order=2)                 # :52
grad = qgd.discrete_adjoint(prob, controls, pcof, target, order=2)        # :58
print("forward history", history.shape, " |grad|", np.linalg.norm(grad))
t0 = time.time()
ret = qgd.optimize_gate(prob, controls, pcof, target, order=4, maxIter=70, pcof_L=-0.5, pcof_U=0.5,
                        print_level=0)                                    # :66-67
print(ret)
print(f"final infidelity {ret.infidelity[-1]:.3e}, guard {ret.guard_penalty[-1]:.3e}, "
      f"{len(ret)} iterations in {time.time() - t0:.2f} s")
