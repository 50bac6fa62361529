"""CPU coverage of the N>1 path: the time-partitioned orchestration
(quantumgatedesign.jl_amd.distributed.TimePartitioned + TorchComm) under torch.distributed/gloo
with world_size 2 and 3, driving the numpy stand-in of the device backend.  The same
orchestration class, partition rule and exchange-buffer layout run on the GPUs with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import proto_propagator as pp
from numpy_backend import NumpyBackend, NumpyColumnBackend, partition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, which, nsteps, order, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import import_package
        qgd = import_package()
        prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=nsteps / 2.0)
        back = NumpyBackend(qgd, prob, order, ctrl, target, rank, world)
        grad, out3 = qgd.TimePartitioned(back, qgd.TorchComm()).discrete_adjoint(pcof)
        out[rank] = (grad, out3)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which,nsteps,order,world", [("cnot2", 60, 4, 2), ("guarded", 48, 6, 2), ("cnot2", 90, 8, 3)])
def test_gloo_time_partition(qgd, which, nsteps, order, world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), which, nsteps, order, out), nprocs=world, join=True)
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=nsteps / 2.0)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    a, b = ref["overlap"]
    for r in range(world):
        grad, out3 = out[r]
        assert np.abs(grad - ref["grad"]).max() <= 1e-12 * np.abs(ref["grad"]).max()
        assert abs(out3[0] ** 2 + out3[1] ** 2 - (a * a + b * b)) < 1e-12
        assert abs(out3[2] - ref["guard"]) < 1e-12


def _worker_cols(rank, world, port, which, nsteps, order, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import import_package
        qgd = import_package()
        prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=nsteps / 2.0)
        back = NumpyColumnBackend(qgd, prob, order, ctrl, target, rank, world)
        out[rank] = qgd.ColumnSharded(back, qgd.TorchComm()).discrete_adjoint(pcof)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which,nsteps,order,world", [("cnot2", 30, 4, 2), ("guarded", 24, 6, 2), ("cnot2", 20, 8, 4)])
def test_gloo_column_shards(qgd, which, nsteps, order, world):
    """The column split (ColumnSharded + TorchComm: all-reduce of the three overlap scalars at the turnaround, all-reduce
    of [grad | scalars] at the end) under gloo, against the unsharded numpy statement."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_cols, args=(world, _free_port(), which, nsteps, order, out), nprocs=world, join=True)
    prob, ctrl, pcof, target = getattr(cases, which + "_case")(qgd, nsteps=nsteps, tf=nsteps / 2.0)
    Gp, Gq, off = qgd.control_basis(ctrl, prob.nsteps, prob.tf, order // 2)
    ref = pp.evaluate(prob, Gp, Gq, off, pcof, target, order)
    a, b = ref["overlap"]
    for r in range(world):
        grad, out3 = out[r]
        assert np.abs(grad - ref["grad"]).max() <= 1e-12 * np.abs(ref["grad"]).max()
        assert abs(out3[0] - a) < 1e-12 and abs(out3[1] - b) < 1e-12 and abs(out3[2] - ref["guard"]) < 1e-12


def test_partition_rule_covers_the_grid():
    for S in (24, 100, 550, 551, 1000, 4096):
        for world in (1, 2, 3, 4, 8):
            parts = [partition(S, world, r) for r in range(world)]
            assert parts[0]["n_lo"] == 0 and parts[-1]["n_hi"] == S
            for p0, p1 in zip(parts, parts[1:]):
                assert p0["n_hi"] == p1["n_lo"]            # neighbouring windows share one time point
            assert all(p["B"] == parts[0]["bpr"] * world for p in parts)
