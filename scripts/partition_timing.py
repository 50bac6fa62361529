"""Per-rank device time of the time-partitioned evaluation, all ranks in one process on one GPU
(collectives = device copies, not timed): what each rank would spend computing at world = 1, 2, 4, 8.
    python scripts/partition_timing.py          cnot3 headline (N=64, 8 columns, order 8, 550 steps)
    python scripts/partition_timing.py c5       BASELINE.json configs[4] (N=256, 256 columns, order 12, 200 steps)
    python scripts/partition_timing.py [c5] columns    the same under the column split (ColumnBackend)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, numpy as np
from __graft_entry__ import import_package
qgd = import_package()
import cases
C5 = "c5" in sys.argv[1:]
COLS = "columns" in sys.argv[1:]          # the column split instead of time windows
if C5:
    prob, ctrl, pcof, _ = cases.synthetic_case(qgd, N=256, c=256, n_ops=4, nsteps=200, tf=2.0)
    target, order = prob.u0 + 1j * prob.v0, 12
else:
    prob, target = qgd.cnot3_problem(nsteps=550, tf=550.0)
    ctrl = cases.cnot3_controls(qgd, prob)
    pcof = (np.random.default_rng(0).random(qgd.get_number_of_control_parameters(ctrl)) - 0.5) * 2 * np.pi * 0.005
    order = 8
for world in (1, 2, 4, 8):
    backs = [(qgd.ColumnBackend if COLS else qgd.DeviceBackend)(prob, order, ctrl, target, r, world) for r in range(world)]
    grp = qgd.LocalGroup(backs)
    for b in backs:   # serialise the ranks (they share this one GPU): per-rank event times then mean what they say
        for name in (("forward", "adjoint") if COLS else ("forward_begin", "forward_end", "adjoint_begin", "adjoint_end")):
            fn = getattr(b, name)
            setattr(b, name, (lambda f: (lambda *a: (f(*a), torch.cuda.synchronize())[0]))(fn))
    for b in backs: b.set_timing(1)
    for _ in range(3): res = grp.discrete_adjoint(pcof)
    per_rank = []
    for b in backs:
        t = b.timings()
        per_rank.append(sum(t.values()))
    t0 = backs[0].timings()
    if COLS:
        print(f"world {world} (column blocks): per-rank device ms max {max(per_rank):.3f} min {min(per_rank):.3f}; two all-reduces of 24 B and {backs[0].exchange_buffer(2)[0].numel() * 8} B")
    else:
        sizes = [backs[0].exchange_buffer(w)[0].numel() * 8 / 1e6 for w in (0, 1, 2)]
        print(f"world {world}: per-rank device ms max {max(per_rank):.3f} min {min(per_rank):.3f}; exchange MB (all-gather RX, all-gather phiRX, all-reduce) {sizes[0]:.2f} {sizes[1]:.3f} {sizes[2]:.4f}")
    print("   rank 0 phases:", {k: round(v, 3) for k, v in sorted(t0.items(), key=lambda kv: -kv[1])})
    for b in backs: b.close()
