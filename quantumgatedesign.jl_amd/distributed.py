"""Time-partitioned evaluation over several GPUs (one process per GPU, torch.distributed over
RCCL/xGMI as plumbing).  See DESIGN.md "Multi-GPU" and include/qgd.h.

The reference's only parallelism is a thread loop over initial-condition columns
(src/forward_evolution.jl:48,332).  Here the parallel axis is time: rank r owns a contiguous
window of the time grid, builds its own step propagators, and the blocked scan of the sweeps
is completed with two all-gathers (block propagators; adjoint affine parts) and one
all-reduce (gradient + the three objective scalars) per evaluation.

``TimePartitioned`` is the orchestration; it is backend-agnostic so that the exchange pattern
is also exercised on CPU with gloo (tests/test_distributed_cpu.py drives it with a numpy
backend).  ``DeviceBackend`` is the product backend (C ABI).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .controls import as_control_list, control_basis
from .evolution import DeviceProblem, _vp


def comm_unique_id() -> bytes:
    """The 128-byte RCCL id rank 0 makes and every rank passes to ``comm_init`` (qgd_comm_unique_id)."""
    buf = C.create_string_buffer(_lib.QGD_UNIQUE_ID_BYTES)
    _lib.check(None, _lib.lib().qgd_comm_unique_id(buf))
    return buf.raw


class RcclEvaluation:
    """discrete_adjoint! of ONE SchrodingerProb over the ranks of an RCCL communicator that lives INSIDE the library
    (include/qgd.h, qgd_comm_init_rccl): this class only builds the rank's handle -- its window of the time grid, or its
    block of initial-condition columns (the reference's thread axis, src/forward_evolution.jl:48,332) -- and then
    calls the same two entry points a single-GPU host calls.  No collective is issued from Python."""

    def __init__(self, prob, order, controls, target, rank, world, unique_id, shard="time", device=0):
        self.rank, self.world, self.shard = rank, world, shard
        target = np.asarray(target)
        if shard == "columns":
            c = prob.N_initial_conditions
            if world > c:
                raise ValueError(f"{world} ranks for {c} initial conditions: a rank would own no column")
            lo, hi = rank * c // world, (rank + 1) * c // world
            sub = prob.copy()
            sub.u0 = np.asfortranarray(prob.u0[:, lo:hi]); sub.v0 = np.asfortranarray(prob.v0[:, lo:hi])
            sub.N_initial_conditions = hi - lo            # N_ess_levels stays the global one
            self.columns = (lo, hi)
            self.dp = DeviceProblem(sub, order, device)
            self.dp.comm_init(unique_id, rank, world, "columns")
            self.dp.set_controls(controls)
            self.dp.set_target(target[:, lo:hi])
        else:
            # (the grid is allocated by comm_init, for the rank's window only: a problem that is sharded by time because
            #  its whole grid does not fit one GPU must not be allocated whole first)
            self.dp = DeviceProblem(prob, order, device, defer_grid=True)
            self.dp.comm_init(unique_id, rank, world, "time")
            self.partition = self.dp.partition
            self.dp.set_controls(controls)
            self.dp.set_target(target)
        self.n_pcof = self.dp.n_pcof

    def discrete_adjoint(self, pcof, history_precomputed=False, uv_history=None, lambda_history=None, adjoint_forcing=None):
        return self.dp.discrete_adjoint(pcof, history_precomputed, uv_history, lambda_history, adjoint_forcing)

    def eval_forward(self, pcof, uv_history=None):
        return self.dp.eval_forward(pcof, uv_history)

    def timings(self):
        return self.dp.timings()

    def set_timing(self, mode, phase=None):
        self.dp.set_timing(mode, phase)

    def close(self):
        self.dp.close()


class _DevArray:
    """Expose raw device memory to torch through the CUDA array interface."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class DeviceBackend:
    """One rank's share of a SchrodingerProb on its GPU."""

    def __init__(self, prob, order, controls, target, rank, world, device=0, stream=None):
        self.dp = DeviceProblem(prob, order, device)
        self.lib, self.h = self.dp.lib, self.dp.h
        if stream is not None:
            _lib.check(self.h, self.lib.qgd_set_stream(self.h, C.c_void_p(stream)))
        _lib.check(self.h, self.lib.qgd_set_partition(self.h, rank, world))
        part = np.zeros(8, dtype=np.int32)
        _lib.check(self.h, self.lib.qgd_get_partition(self.h, _vp(part)))
        self.n_lo, self.n_hi = int(part[0]), int(part[1])
        self.partition = dict(n_lo=self.n_lo, n_hi=self.n_hi, blocks=int(part[2]), blocks_per_rank=int(part[3]),
                              block_len=int(part[4]), rank=int(part[5]), world=int(part[6]), nt=int(part[7]))
        cl = as_control_list(controls)
        m = order // 2
        Gp, Gq, _ = control_basis(cl, prob.nsteps, prob.tf, m)
        Gp = [np.ascontiguousarray(g[self.n_lo:self.n_hi + 1]) for g in Gp]
        Gq = [np.ascontiguousarray(g[self.n_lo:self.n_hi + 1]) for g in Gq]
        nco = np.array([c.N_coeff for c in cl], dtype=np.int32)
        gp = (C.c_void_p * max(len(cl), 1))(*[_vp(g) for g in Gp])
        gq = (C.c_void_p * max(len(cl), 1))(*[_vp(g) for g in Gq])
        _lib.check(self.h, self.lib.qgd_set_control_basis(self.h, _vp(nco), gp, gq))
        self.n_pcof = int(nco.sum())
        self.dp.set_target(target)

    def exchange_buffer(self, which):
        """(whole buffer, this rank's chunk) as torch CUDA tensors over the library's memory."""
        import torch
        cache = self.__dict__.setdefault("_xbuf", {})      # the buffers live as long as the control basis
        if which not in cache:
            ptr, tot, off, own = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_size_t()
            _lib.check(self.h, self.lib.qgd_exchange_buffer(self.h, which, C.byref(ptr), C.byref(tot), C.byref(off), C.byref(own)))
            whole = torch.as_tensor(_DevArray(ptr.value, tot.value), device="cuda")
            cache[which] = (whole, whole[off.value:off.value + own.value])
        return cache[which]

    def forward_begin(self, pcof):
        pc = np.ascontiguousarray(pcof, dtype=np.float64)
        _lib.check(self.h, self.lib.qgd_dist_forward_begin(self.h, _vp(pc), len(pc)))

    def forward_end(self):
        _lib.check(self.h, self.lib.qgd_dist_forward_end(self.h))

    def adjoint_begin(self):
        _lib.check(self.h, self.lib.qgd_dist_adjoint_begin(self.h))

    def adjoint_end(self):
        _lib.check(self.h, self.lib.qgd_dist_adjoint_end(self.h))

    def finish(self):
        grad, out3 = np.zeros(self.n_pcof), np.zeros(3)
        _lib.check(self.h, self.lib.qgd_dist_finish(self.h, _vp(grad), _vp(out3)))
        return grad, out3

    def timings(self):
        return self.dp.timings()

    def set_timing(self, mode, phase=None):
        self.dp.set_timing(mode, phase)

    def close(self):
        self.dp.close()


class ColumnBackend:
    """One rank's share of a SchrodingerProb under the COLUMN split (the reference's thread axis,
    src/forward_evolution.jl:48,332): a contiguous block of the initial-condition columns of u0, v0 and of the
    target; operators, controls and the whole time grid are replicated."""

    def __init__(self, prob, order, controls, target, rank, world, device=0, stream=None):
        c = prob.N_initial_conditions
        if world > c:
            raise ValueError(f"{world} ranks for {c} initial conditions: a rank would own no column")
        lo, hi = rank * c // world, (rank + 1) * c // world
        sub = prob.copy()
        sub.u0 = np.asfortranarray(prob.u0[:, lo:hi]); sub.v0 = np.asfortranarray(prob.v0[:, lo:hi])
        sub.N_initial_conditions = hi - lo          # N_ess_levels stays the global one
        self.columns, self.rank, self.world = (lo, hi), rank, world
        self.dp = DeviceProblem(sub, order, device)
        self.lib, self.h = self.dp.lib, self.dp.h
        if stream is not None:
            _lib.check(self.h, self.lib.qgd_set_stream(self.h, C.c_void_p(stream)))
        self.dp.set_controls(controls)
        target = np.asarray(target)
        self.dp.set_target(target[:, lo:hi])
        self.n_pcof = self.dp.n_pcof

    exchange_buffer = DeviceBackend.exchange_buffer

    def forward(self, pcof):
        pc = np.ascontiguousarray(pcof, dtype=np.float64)
        _lib.check(self.h, self.lib.qgd_cols_forward(self.h, _vp(pc), len(pc)))

    def adjoint(self):
        _lib.check(self.h, self.lib.qgd_cols_adjoint(self.h, 1 if self.rank == 0 else 0))

    finish, timings, set_timing, close = DeviceBackend.finish, DeviceBackend.timings, DeviceBackend.set_timing, DeviceBackend.close


class ColumnSharded:
    """discrete_adjoint! with the columns spread over the ranks: two small all-reduces per evaluation."""

    def __init__(self, backend, comm):
        self.b, self.comm = backend, comm

    def discrete_adjoint(self, pcof):
        b = self.b
        b.forward(pcof)
        self.comm.all_reduce(b.exchange_buffer(3)[0])       # <w_N,R>, <w_N,T>, guard: global before the terminal condition
        b.adjoint()
        self.comm.all_reduce(b.exchange_buffer(2)[0])       # gradient (+ the scalars, kept by rank 0 only)
        return b.finish()


class TorchComm:
    """Collectives of one evaluation through torch.distributed (backend "nccl" = RCCL on ROCm,
    "gloo" on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group

    def _host_staged(self, t):
        # gloo has no device collectives for every op: a plumbing-check backend stages through the host
        return self.dist.get_backend(self.group) == "gloo" and getattr(t, "is_cuda", False)

    def all_gather(self, whole, own):
        # in place: `own` is this rank's chunk of `whole` (NCCL/RCCL in-place all-gather layout)
        if self._host_staged(whole):
            import torch
            host = torch.empty(whole.shape, dtype=whole.dtype)
            self.dist.all_gather_into_tensor(host, own.cpu(), group=self.group)
            whole.copy_(host)
            return
        self.dist.all_gather_into_tensor(whole, own, group=self.group)

    def all_reduce(self, whole):
        if self._host_staged(whole):
            host = whole.cpu()
            self.dist.all_reduce(host, group=self.group)
            whole.copy_(host)
            return
        self.dist.all_reduce(whole, group=self.group)


class TimePartitioned:
    """discrete_adjoint! of one SchrodingerProb spread over the ranks of a communicator."""

    def __init__(self, backend, comm):
        self.b, self.comm = backend, comm

    def discrete_adjoint(self, pcof):
        b = self.b
        b.forward_begin(pcof)
        self.comm.all_gather(*b.exchange_buffer(0))
        b.forward_end()
        b.adjoint_begin()
        self.comm.all_gather(*b.exchange_buffer(1))
        b.adjoint_end()
        self.comm.all_reduce(b.exchange_buffer(2)[0])
        return b.finish()


class LocalGroup:
    """All ranks of a partition inside ONE process (every backend on the same GPU, or numpy
    backends on the CPU): the collectives become copies.  Used to validate the partitioned
    algorithm where only one GPU is available."""

    def __init__(self, backends):
        self.backends = backends

    def _gather(self, which):
        bufs = [b.exchange_buffer(which) for b in self.backends]
        for _, (whole_src, own_src) in enumerate(bufs):
            off = own_src.storage_offset() - whole_src.storage_offset()
            for whole_dst, _ in bufs:
                if whole_dst.data_ptr() != whole_src.data_ptr():
                    whole_dst[off:off + own_src.numel()].copy_(own_src)

    def _reduce(self, which):
        wholes = [b.exchange_buffer(which)[0] for b in self.backends]
        total = wholes[0].clone()
        for w in wholes[1:]:
            total += w
        for w in wholes:
            w.copy_(total)

    def discrete_adjoint(self, pcof):
        bs = self.backends
        if isinstance(bs[0], ColumnBackend):
            for b in bs:
                b.forward(pcof)
            self._reduce(3)
            for b in bs:
                b.adjoint()
            self._reduce(2)
            return [b.finish() for b in bs]
        for b in bs:
            b.forward_begin(pcof)
        self._gather(0)
        for b in bs:
            b.forward_end()
        for b in bs:
            b.adjoint_begin()
        self._gather(1)
        for b in bs:
            b.adjoint_end()
        wholes = [b.exchange_buffer(2)[0] for b in bs]
        total = wholes[0].clone()
        for w in wholes[1:]:
            total += w
        for w in wholes:
            w.copy_(total)
        return [b.finish() for b in bs]
