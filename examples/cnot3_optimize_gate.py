"""Mirror of the reference's examples/cnot3_optimize_gate.jl on the device path: the 3-qubit dispersive
CNOT problem of the benchmark (subsystems (4,4,4), 8 essential states, order 8 by default) handed to
``optimize_gate`` (L-BFGS-B standing in for Ipopt).

    python examples/cnot3_optimize_gate.py ORDER STEPSIZE [--max_iter 50]

The reference's run of this script (cnot3_optimize_gate.sb: one CPU core, order 8, stepsize 1.0) is the
workload BASELINE.json quotes its metric on; here one optimiser iteration is one forward+adjoint
evaluation on the GPU (~0.31 ms) plus scipy's bookkeeping.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime for the process)
from __graft_entry__ import import_package

ap = argparse.ArgumentParser()
ap.add_argument("order", type=int)
ap.add_argument("stepsize", type=float)
ap.add_argument("--max_iter", "-m", type=int, default=50)
ap.add_argument("--tf", type=float, default=550.0)
args = ap.parse_args()

qgd = import_package()
import cases  # the benchmark's controls: 3 carriers x degree-2 B-spline with 10 basis functions per control

nsteps = int(np.ceil(args.tf / args.stepsize))                                 # cnot3_optimize_gate.jl:48
prob, target = qgd.cnot3_problem(nsteps=nsteps, tf=args.tf)
controls = cases.cnot3_controls(qgd, prob)
npar = qgd.get_number_of_control_parameters(controls)
pcof0 = (np.random.default_rng(0).random(npar) - 0.5) * 2 * np.pi * 0.005
amax = 2 * np.pi * 0.04 / 3                                                    # amplitude bound per carrier
t0 = time.time()
hist = qgd.optimize_gate(prob, controls, pcof0, target, order=args.order, maxIter=args.max_iter,
                         pcof_L=-amax, pcof_U=amax, print_level=0)
el = time.time() - t0
print(hist)
print(f"order {args.order}, nsteps {nsteps}, {npar} parameters: infidelity {hist.infidelity[0]:.4f} -> {min(hist.infidelity):.3e}, "
      f"guard {hist.guard_penalty[-1]:.2e}, {len(hist)} iterations in {el:.2f} s")
