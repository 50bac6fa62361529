"""One gradient evaluation of the cnot3 benchmark problem spread over the GPUs of a node, with the RCCL collectives
issued INSIDE libqgd_hip.so (include/qgd.h: qgd_comm_unique_id / qgd_comm_init_rccl).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/multi_gpu_rccl.py [time|columns]

torch.distributed (gloo) is used for one thing only: carrying the 128-byte communicator id from rank 0 to the other
ranks.  A Julia host does the same with MPI.jl.  After `RcclEvaluation(...)` every rank calls
the ordinary `discrete_adjoint(pcof)` -- a collective call -- and receives the full gradient.

The reference spreads this evaluation over host threads (Threads.@threads over initial conditions,
src/forward_evolution.jl:48,332); `columns` is that split, `time` (default) gives every rank a window of the time grid.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch                      # noqa: E402  (first: one HIP runtime for torch and the library)
import torch.distributed as dist  # noqa: E402
from __graft_entry__ import import_package  # noqa: E402
import cases                      # noqa: E402


def main():
    shard = sys.argv[1] if len(sys.argv) > 1 else "time"
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    qgd = import_package()
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=550, tf=550.0)
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        uid = torch.frombuffer(bytearray(qgd.comm_unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(uid, 0)
    ev = qgd.RcclEvaluation(prob, 8, ctrl, target, rank, world, bytes(uid.numpy().tobytes()), shard=shard, device=local)
    grad, (a, b, guard) = ev.discrete_adjoint(pcof)
    infidelity = 1 - (a * a + b * b) / prob.N_ess_levels ** 2
    print(f"rank {rank}/{world} ({shard}): infidelity {infidelity:.12f}  guard {guard:.3e}  |grad| {np.linalg.norm(grad):.12e}")
    ev.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
