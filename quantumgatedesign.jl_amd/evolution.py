"""Host-side mirror of the reference's stepper / gradient entry points, calling the
C ABI (include/qgd.h) through ctypes.  Same names, argument meaning and error
behaviour as the reference:

  eval_forward(prob, controls, pcof; order, saveEveryNsteps)      forward_evolution.jl:15-29
  eval_forward_(uv_history, prob, controls, pcof; order)          forward_evolution.jl:33-70   (Julia eval_forward!)
  discrete_adjoint(prob, controls, pcof, target; order)           eval_grad_discrete_adjoint.jl:83-102
  discrete_adjoint_(grad, history, lambda_history, adjoint_forcing, prob, controls, pcof, target;
                    order, history_precomputed)                   eval_grad_discrete_adjoint.jl:107-160
  infidelity(...), infidelity_real, guard_penalty_real            infidelity.jl:7-96

Julia's trailing ``!`` is spelled as a trailing underscore.  Arrays use the
reference's column-major layouts (numpy ``order="F"``).
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _lib
from .controls import as_control_list, control_basis, get_number_of_control_parameters


def _f(a):
    return np.asfortranarray(np.array(a, dtype=np.float64))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _check_out(a, shape, name):
    """Output arrays go to the C ABI as bare pointers: the shape, dtype and memory order the library will
    write must be exactly what it is given (the reference raises a DimensionMismatch here)."""
    if a is None:
        return None
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or a.shape != tuple(shape) or not a.flags.f_contiguous \
            or not a.flags.writeable:
        raise ValueError(f"{name} must be a writable float64 Fortran-ordered array of shape {tuple(shape)}; got "
                         f"{getattr(a, 'dtype', type(a))} {getattr(a, 'shape', None)}")
    return a


def complex_to_real(x):
    """state_vector_helpers.jl:72-74."""
    x = np.asarray(x)
    return np.asfortranarray(np.vstack([np.real(x), np.imag(x)]).astype(np.float64))


def real_to_complex(x):
    """state_vector_helpers.jl:78-87."""
    N = x.shape[0] // 2
    return x[:N] + 1j * x[N:]


COST_TYPES = {"Infidelity": 0, ":Infidelity": 0, "Tracking": 1, ":Tracking": 1, "Norm": 2, ":Norm": 2}


class DeviceProblem:
    """One qgd handle: a SchrodingerProb resident on one GPU for one Hermite order."""

    def __init__(self, prob, order: int, device: int = 0, csc: bool = False, defer_grid: bool = False):
        """``csc=True`` hands the operators to the library as SparseMatrixCSC triples (qgd_create_csc), the
        form DispersiveProblem(sparse_rep=true) produces in the reference.  ``defer_grid=True`` (QGD_CREATE_DEFER_GRID):
        the time grid is not allocated by the constructor -- for a handle that is about to become a rank of a time
        partition (``comm_init``) and must never hold the whole grid first."""
        self.lib = _lib.lib()
        self.N = prob.N_tot_levels
        self.c = prob.N_initial_conditions
        self.n_ops = prob.N_operators
        self.order = int(order)
        self.m = self.order // 2
        self.nsteps = prob.nsteps
        self.tf = prob.tf
        N = self.N
        bufs = dict(
            ssym=_f(prob.system_sym), sasym=_f(prob.system_asym),
            sym=np.ascontiguousarray(np.stack([_f(o).T for o in prob.sym_operators])) if self.n_ops else np.zeros(1),
            asym=np.ascontiguousarray(np.stack([_f(o).T for o in prob.asym_operators])) if self.n_ops else np.zeros(1),
            u0=_f(prob.u0), v0=_f(prob.v0), guard=_f(prob.guard_subspace_projector))
        d = _lib.ProblemDesc(N, self.c, self.n_ops, prob.N_ess_levels, self.order, prob.nsteps, prob.tf,
                             _vp(bufs["ssym"]), _vp(bufs["sasym"]), _vp(bufs["sym"]), _vp(bufs["asym"]),
                             _vp(bufs["u0"]), _vp(bufs["v0"]), _vp(bufs["guard"]), device,
                             _lib.QGD_CREATE_DEFER_GRID if defer_grid else 0)
        h = C.c_void_p()
        if csc:
            from scipy.sparse import csc_matrix
            keep = []

            def triple(a):
                sp = csc_matrix(np.asarray(a))
                arrs = (sp.indptr.astype(np.int64), sp.indices.astype(np.int64), sp.data.astype(np.float64))
                keep.append(arrs)
                return _lib.CSC(_vp(arrs[0]), _vp(arrs[1]), _vp(arrs[2]), 0, 0)

            ssym, sasym = triple(prob.system_sym), triple(prob.system_asym)
            syms = (_lib.CSC * max(self.n_ops, 1))(*[triple(o) for o in prob.sym_operators])
            asyms = (_lib.CSC * max(self.n_ops, 1))(*[triple(o) for o in prob.asym_operators])
            rc = self.lib.qgd_create_csc(C.byref(d), C.byref(ssym), C.byref(sasym), syms, asyms, C.byref(h))
        else:
            rc = self.lib.qgd_create(C.byref(d), C.byref(h))
        _lib.check(None, rc)
        self.h = h
        self._finalizer = weakref.finalize(self, self.lib.qgd_destroy, h)
        self._basis_key = None
        self._target_key = None
        self.n_pcof = 0

    def close(self):
        self._finalizer()

    # -- setup -------------------------------------------------------------
    def set_nsteps(self, nsteps, tf):
        _lib.check(self.h, self.lib.qgd_set_nsteps(self.h, int(nsteps), float(tf)))
        self.nsteps, self.tf = int(nsteps), float(tf)
        self._basis_key = None

    def set_target(self, target):
        target = np.asarray(target)
        if target.ndim != 2 or target.shape[1] != self.c or target.shape[0] not in (self.N, 2 * self.N) \
                or (np.iscomplexobj(target) and target.shape[0] != self.N):
            raise ValueError(f"target must be [{self.N}, {self.c}] (complex or real) or the stacked real form "
                             f"[{2 * self.N}, {self.c}]; got {target.shape}")
        t = complex_to_real(target) if target.shape[0] == self.N else _f(target)
        key = t.tobytes()
        if key != self._target_key:
            _lib.check(self.h, self.lib.qgd_set_target(self.h, _vp(t)))
            self._target_key = key

    def set_controls(self, controls):
        cl = as_control_list(controls)
        if len(cl) != self.n_ops:
            raise ValueError(f"{len(cl)} controls for {self.n_ops} control operators")
        key = (tuple(id(c) for c in cl), self.nsteps, self.tf)
        if key == self._basis_key:
            return
        if not all(getattr(c, "is_linear", False) for c in cl):
            # general path (any AbstractControl): tables and their Jacobian at the current pcof are formed on the host
            # from the pointwise protocol at every evaluation (_upload_general)
            self._general = cl
            self._general_pcof = None
            self._basis_key = key
            self._controls_keepalive = cl
            self.n_pcof = int(sum(c.N_coeff for c in cl))
            return
        self._general = None
        Gp, Gq, _ = control_basis(cl, self.nsteps, self.tf, self.m)
        win = getattr(self, "window", None)
        if win is not None:                               # time-partitioned handle: the basis of the rank's own time points
            Gp = [np.ascontiguousarray(g[win[0]:win[1] + 1]) for g in Gp]
            Gq = [np.ascontiguousarray(g[win[0]:win[1] + 1]) for g in Gq]
        nco = np.array([c.N_coeff for c in cl], dtype=np.int32)
        gp_ptrs = (C.c_void_p * max(self.n_ops, 1))(*[_vp(g) for g in Gp])
        gq_ptrs = (C.c_void_p * max(self.n_ops, 1))(*[_vp(g) for g in Gq])
        _lib.check(self.h, self.lib.qgd_set_control_basis(self.h, _vp(nco), gp_ptrs, gq_ptrs))
        self._basis_key = key
        self._controls_keepalive = cl
        self.n_pcof = int(nco.sum())

    def _upload_general(self, pcof):
        """Non-linear controls: values and Jacobian of the control tables at this pcof (qgd.h: "Pair with
        qgd_set_control_basis holding the Jacobian at the current pcof when a gradient is wanted")."""
        pc = np.ascontiguousarray(pcof, dtype=np.float64)
        if len(pc) != self.n_pcof:
            raise ValueError("length of pcof does not match the controls")
        if self._general_pcof is not None and np.array_equal(pc, self._general_pcof):
            return
        from .controls import control_tables_general
        cl = self._general
        p, q, Gp, Gq = control_tables_general(cl, pc, self.nsteps, self.tf, self.m)
        nco = np.array([c.N_coeff for c in cl], dtype=np.int32)
        gp_ptrs = (C.c_void_p * max(self.n_ops, 1))(*[_vp(g) for g in Gp])
        gq_ptrs = (C.c_void_p * max(self.n_ops, 1))(*[_vp(g) for g in Gq])
        _lib.check(self.h, self.lib.qgd_set_control_basis(self.h, _vp(nco), gp_ptrs, gq_ptrs))
        _lib.check(self.h, self.lib.qgd_set_control_tables(self.h, _vp(p), _vp(q)))
        self._general_pcof = pc.copy()

    def set_control_tables(self, p_tables, q_tables):
        p, q = _f(p_tables), _f(q_tables)
        shape = (self.m + 1, self.n_ops, self.nsteps + 1)
        if p.shape != shape or q.shape != shape:
            raise ValueError(f"control tables must have shape {shape}")
        _lib.check(self.h, self.lib.qgd_set_control_tables(self.h, _vp(p), _vp(q)))

    # -- long time grids in bounded memory -----------------------------------
    def set_memory_budget(self, nbytes=0):
        """qgd_set_memory_budget: device bytes the per-time-point buffers may take (0 = automatic).  A grid that does not
        fit is processed in windows (same results to rounding, build + inverse done twice).  Re-allocates the grid."""
        _lib.check(self.h, self.lib.qgd_set_memory_budget(self.h, int(nbytes)))
        self._basis_key = None

    def memory_plan(self):
        out = np.zeros(4, dtype=np.int64)
        _lib.check(self.h, self.lib.qgd_get_memory_plan(self.h, _vp(out)))
        return dict(windows=int(out[0]), steps_per_window=int(out[1]), window_bytes=int(out[2]), budget=int(out[3]))

    # -- several GPUs behind one call (qgd_comm_init_rccl) -------------------
    def comm_init(self, unique_id: bytes, rank: int, world: int, shard: str = "time"):
        """Give the handle its RCCL communicator: discrete_adjoint / eval_forward become collective calls whose
        exchanges the library issues itself.  ``shard="time"``: this rank's window of the time grid (set the controls
        afterwards); ``"columns"``: the handle must have been created from this rank's columns."""
        if len(unique_id) != _lib.QGD_UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        code = {"time": _lib.QGD_SHARD_TIME, "columns": _lib.QGD_SHARD_COLUMNS}[shard]
        buf = C.create_string_buffer(bytes(unique_id), _lib.QGD_UNIQUE_ID_BYTES)
        _lib.check(self.h, self.lib.qgd_comm_init_rccl(self.h, buf, int(rank), int(world), code))
        self._basis_key = None
        self.window = None
        if shard == "time":
            part = np.zeros(8, dtype=np.int32)
            _lib.check(self.h, self.lib.qgd_get_partition(self.h, _vp(part)))
            self.window = (int(part[0]), int(part[1]))
            self.partition = dict(n_lo=int(part[0]), n_hi=int(part[1]), blocks=int(part[2]), blocks_per_rank=int(part[3]),
                                  block_len=int(part[4]), rank=int(part[5]), world=int(part[6]), nt=int(part[7]))

    def comm_destroy(self):
        _lib.check(self.h, self.lib.qgd_comm_destroy(self.h))

    def set_comm_timeout(self, milliseconds):
        """qgd_set_comm_timeout: the bound on the host wait of a collective evaluation; past it the communicator is
        aborted and the call raises QGDError(QGD_ERR_COMM)."""
        _lib.check(self.h, self.lib.qgd_set_comm_timeout(self.h, float(milliseconds)))

    def comm_info(self):
        out = (C.c_int32 * 3)()
        _lib.check(self.h, self.lib.qgd_comm_info(self.h, out))
        return dict(rank=out[0], world=out[1], shard="columns" if out[2] == _lib.QGD_SHARD_COLUMNS else "time")

    # -- evaluation --------------------------------------------------------
    def _hist_shape(self, save=1):
        win = getattr(self, "window", None)               # (a time-partitioned handle returns its own window)
        nt = self.nsteps + 1 if win is None else win[1] - win[0] + 1
        return (2 * self.N, self.m + 1, 1 + (nt - 1) // save, self.c)

    def set_save_every(self, save_every_nsteps=1):
        """eval_forward's saveEveryNsteps (forward_evolution.jl:104,239-241): uv_history of eval_forward then holds
        the time points 0, s, 2s, ... (qgd_set_save_every)."""
        save = int(save_every_nsteps)
        if save != getattr(self, "_save_every", 1):
            _lib.check(self.h, self.lib.qgd_set_save_every(self.h, save))
            self._save_every = save

    def pin(self, array):
        """Register (pin) an output array that will be handed to discrete_adjoint / eval_forward repeatedly
        (qgd_register_host_buffer): its downloads then run at PCIe speed.  Unpinned when the array dies."""
        if not isinstance(array, np.ndarray) or array.dtype != np.float64 or not array.flags.f_contiguous:
            raise ValueError("only float64 Fortran-ordered arrays can be pinned")
        key = (array.ctypes.data, array.nbytes)
        pins = self.__dict__.setdefault("_pins", {})
        if key in pins:
            return array
        _lib.check(self.h, self.lib.qgd_register_host_buffer(self.h, C.c_void_p(key[0]), key[1]))
        lib, h, fin = self.lib, self.h, self._finalizer

        def unpin(ptr=key[0]):
            pins.pop(key, None)
            if fin.alive:
                lib.qgd_unregister_host_buffer(h, C.c_void_p(ptr))
        pins[key] = weakref.finalize(array, unpin)
        return array

    def eval_forward(self, pcof=None, uv_history=None):
        _check_out(uv_history, self._hist_shape(getattr(self, "_save_every", 1)), "uv_history")
        out3 = np.zeros(3)
        if pcof is not None and getattr(self, "_general", None):
            self._upload_general(pcof)
            pcof = None                                   # (the tables are on the device)
        pc = None if pcof is None else np.ascontiguousarray(pcof, dtype=np.float64)
        _lib.check(self.h, self.lib.qgd_eval_forward(
            self.h, None if pc is None else _vp(pc), 0 if pc is None else len(pc),
            None if uv_history is None else _vp(uv_history), _vp(out3)))
        return out3

    def discrete_adjoint(self, pcof, history_precomputed=False, uv_history=None, lambda_history=None,
                         adjoint_forcing=None):
        if getattr(self, "_general", None):
            self._upload_general(pcof)
            _check_out(uv_history, self._hist_shape(), "history")
            _check_out(lambda_history, self._hist_shape(), "lambda_history")
            _check_out(adjoint_forcing, (2 * self.N, self.nsteps + 1, self.c), "adjoint_forcing")
            grad = np.zeros(self.n_pcof)
            out3 = np.zeros(3)
            _lib.check(self.h, self.lib.qgd_discrete_adjoint(
                self.h, None, 0, 1 if history_precomputed else 0, _vp(grad),
                None if uv_history is None else _vp(uv_history),
                None if lambda_history is None else _vp(lambda_history),
                None if adjoint_forcing is None else _vp(adjoint_forcing), _vp(out3)))
            return grad, out3
        if uv_history is None and lambda_history is None and adjoint_forcing is None and len(pcof) == self.n_pcof:
            # the optimiser's call: persistent host buffers with cached pointers (three ndarray.ctypes look-ups
            # cost 6 us, 1.5 % of a cnot3 evaluation)
            io = self.__dict__.get("_io")
            if io is None or len(io[0]) != self.n_pcof:
                bufs = (np.zeros(self.n_pcof), np.zeros(self.n_pcof), np.zeros(3))
                io = self._io = bufs + tuple(_vp(b) for b in bufs)
            io[0][:] = pcof
            rc = self.lib.qgd_discrete_adjoint(self.h, io[3], self.n_pcof, 1 if history_precomputed else 0, io[4],
                                               None, None, None, io[5])
            if rc:
                _lib.check(self.h, rc)
            return io[1].copy(), io[2].copy()
        _check_out(uv_history, self._hist_shape(), "history")
        _check_out(lambda_history, self._hist_shape(), "lambda_history")
        _check_out(adjoint_forcing, (2 * self.N, self._hist_shape()[2], self.c), "adjoint_forcing")
        pc = np.ascontiguousarray(pcof, dtype=np.float64)
        grad = np.zeros(len(pc))
        out3 = np.zeros(3)
        _lib.check(self.h, self.lib.qgd_discrete_adjoint(
            self.h, _vp(pc), len(pc), 1 if history_precomputed else 0, _vp(grad),
            None if uv_history is None else _vp(uv_history),
            None if lambda_history is None else _vp(lambda_history),
            None if adjoint_forcing is None else _vp(adjoint_forcing), _vp(out3)))
        return grad, out3

    def eval_adjoint(self, pcof, terminal_condition, forcing=None):
        if getattr(self, "_general", None):
            self._upload_general(pcof)
            pcof = np.zeros(0)                            # (NULL pcof: the tables are on the device)
        pc = np.ascontiguousarray(pcof, dtype=np.float64)
        term = _f(terminal_condition).reshape(2 * self.N, self.c, order="F")
        fo = None if forcing is None else _f(forcing)
        if fo is not None and fo.shape != (2 * self.N, self.nsteps + 1, self.c):
            raise ValueError(f"forcing must have shape {(2 * self.N, self.nsteps + 1, self.c)}")
        lam = np.zeros((2 * self.N, self.m + 1, self.nsteps + 1, self.c), order="F")
        _lib.check(self.h, self.lib.qgd_eval_adjoint(self.h, _vp(pc) if len(pc) else None, len(pc), _vp(term),
                                                     None if fo is None else _vp(fo), _vp(lam)))
        return lam

    def set_cost_type(self, cost_type="Infidelity"):
        """cost_type of discrete_adjoint / eval_grad_forced (eval_grad_discrete_adjoint.jl:26-35): ``Infidelity``,
        ``Tracking`` (0.5 |w_N - target|^2) or ``Norm`` (0.5 |w_N|^2); a leading ':' as in Julia is accepted.
        Anything else raises like the reference ("Invalid cost type")."""
        code = COST_TYPES.get(str(cost_type))
        if code is None:
            raise ValueError(f"Invalid cost type: {cost_type}")
        if code != getattr(self, "_cost_type", 0):
            _lib.check(self.h, self.lib.qgd_set_cost_type(self.h, code))
            self._cost_type = code

    def set_small_path(self, on=True):
        """qgd_set_small_path: the four-launch evaluation of small problems (N <= 4: Rabi, cnot2) on / off for this handle."""
        _lib.check(self.h, self.lib.qgd_set_small_path(self.h, 1 if on else 0))

    def small_path_taken(self):
        """Whether the last evaluation of this handle ran on the small-problem path."""
        return bool(self.intermediate("small_path")[0])

    def front_path_taken(self):
        """Did the last forward evaluation take the fused front (csrc/qgd_front.h: same-point step propagators)?"""
        return bool(self.intermediate("front_path")[0])

    def set_lambda_derivatives(self, on=True):
        """Fill ``lambda_history[:, 1:, :, :]`` as the reference leaves it (forward_evolution.jl:427-433, :471-480):
        the adjoint derivatives of lambda_n with the controls at t_{n-1} (t_1 for n = 1).  Off by default -- nothing
        reads these columns, and they make the lambda download 1+m times as large."""
        _lib.check(self.h, self.lib.qgd_set_lambda_derivatives(self.h, 1 if on else 0))

    def apply_hamiltonian(self, w, time_index=0, derivative_order=0, use_adjoint=False):
        w = _f(w).reshape(2 * self.N, self.c, order="F")
        out = np.zeros_like(w, order="F")
        _lib.check(self.h, self.lib.qgd_apply_hamiltonian(self.h, time_index, derivative_order,
                                                          1 if use_adjoint else 0, _vp(w), _vp(out)))
        return out

    def intermediate(self, name):
        need = C.c_size_t()
        _lib.check(self.h, self.lib.qgd_get_intermediate(self.h, name.encode(), None, 0, C.byref(need)))
        out = np.zeros(need.value)
        _lib.check(self.h, self.lib.qgd_get_intermediate(self.h, name.encode(), _vp(out), need.value, C.byref(need)))
        nt = self.nsteps + 1
        if name in ("L", "R", "Linv", "P"):
            z = out.reshape(nt, self.N, self.N, 2)
            return z[..., 0] + 1j * z[..., 1]
        if name == "repivoted":
            return int(out[0])
        if name in ("selection", "small_path", "front_path"):
            return out
        if name == "sigma":
            return out.reshape(nt, self.n_ops, self.m, 2)
        return out.reshape(nt, self.m + 1, self.n_ops, 2)

    def set_timing(self, mode, phase=None):
        """0: no events, 1: every phase (default), 2: only ``phase``."""
        _lib.check(self.h, self.lib.qgd_set_timing(self.h, int(mode), None if phase is None else phase.encode()))

    def eval_forward_forced(self, pcof, forcing, uv_history=None):
        """eval_forward with a forcing array ``[2N, order/2, 1+nsteps, n_cols]`` (Fortran order): the scaled
        Taylor coefficients of the forcing at every time point (forward_evolution.jl:118-129)."""
        if getattr(self, "_general", None):
            self._upload_general(pcof)
            pcof = np.zeros(0)
        pcof = np.ascontiguousarray(pcof, dtype=np.float64)
        forcing = np.asfortranarray(forcing, dtype=np.float64)
        want = (2 * self.N, self.m, self.nsteps + 1, self.c)
        if forcing.shape != want:
            raise ValueError(f"forcing must have shape {want}")
        _check_out(uv_history, self._hist_shape(getattr(self, "_save_every", 1)), "uv_history")
        out3 = np.zeros(3)
        _lib.check(self.h, self.lib.qgd_eval_forward_forced(self.h, _vp(pcof) if len(pcof) else None, len(pcof), _vp(forcing),
                                                             None if uv_history is None else _vp(uv_history), _vp(out3)))
        return out3

    def eval_grad_forced(self, pcof):
        """Gradient by forward sensitivities (eval_grad_forced.jl:17-194); needs controls and target."""
        if getattr(self, "_general", None):               # tables and Jacobian at this pcof are uploaded; NULL pcof
            self._upload_general(pcof)
            grad = np.zeros(self.n_pcof)
            _lib.check(self.h, self.lib.qgd_eval_grad_forced(self.h, None, 0, _vp(grad)))
            return grad
        pcof = np.ascontiguousarray(pcof, dtype=np.float64)
        grad = np.zeros(len(pcof))
        _lib.check(self.h, self.lib.qgd_eval_grad_forced(self.h, _vp(pcof), len(pcof), _vp(grad)))
        return grad

    def set_operator_path(self, mode):
        """"auto" | "dense" (fp64 MFMA kernels) | "sparse" (ELL kernels; raises if the operators do not qualify)."""
        code = {"auto": 0, "dense": 1, "sparse": 2}[mode]
        _lib.check(self.h, self.lib.qgd_set_operator_path(self.h, code))

    def operator_path(self):
        """(path in use, entries per row of the union pattern, entries per row of the widest control)."""
        out = (C.c_int32 * 3)()
        _lib.check(self.h, self.lib.qgd_get_operator_path(self.h, out))
        return ("sparse" if out[0] == 2 else "dense", out[1], out[2])

    def timings(self):
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = C.c_int32()
        _lib.check(self.h, self.lib.qgd_get_timings(self.h, names, ms, cap, C.byref(n)))
        return {names[i].decode(): ms[i] for i in range(min(n.value, cap))}


# ---------------------------------------------------------------------------
# handle cache: one DeviceProblem per (prob object, order)
# ---------------------------------------------------------------------------
_cache: dict = {}


try:
    from xxhash import xxh3_64_intdigest as _digest          # ~10 GB/s
except ImportError:                                           # pragma: no cover
    from zlib import crc32 as _digest


def _fingerprint(prob):
    """Digest of the fields a handle copies to the device at creation (operators, initial conditions,
    guard): a prob mutated in place gets a fresh handle instead of a stale device copy.  Arrays up to 1 MB are
    hashed whole, larger ones through 65536 evenly spaced elements (the digest runs on every reference-shaped
    call: ~40 us for a cnot3 problem)."""
    h = 0
    for a in (prob.system_sym, prob.system_asym, prob.u0, prob.v0, prob.guard_subspace_projector,
              *prob.sym_operators, *prob.asym_operators):
        a = np.asarray(a)
        flat = a.reshape(-1, order="A") if (a.flags.f_contiguous or a.flags.c_contiguous) else np.ascontiguousarray(a).reshape(-1)
        if flat.nbytes > (1 << 20):
            flat = np.ascontiguousarray(flat[:: max(1, flat.size // 65536)])
        h = (h * 1000003) ^ _digest(flat.view(np.uint8).data) ^ hash(a.shape)
    return (h, prob.N_tot_levels, prob.N_initial_conditions, prob.N_ess_levels, prob.N_operators)


def _evict(key):
    ent = _cache.pop(key, None)
    if ent is not None:
        ent[1].close()


def device_problem(prob, order: int, device: int = 0) -> DeviceProblem:
    """The handle of (prob, order) -- created on first use, released when ``prob`` is garbage-collected
    (or by release(prob) / clear_cache()), re-created when prob's operators or initial conditions changed."""
    key = (id(prob), int(order), device)
    ent = _cache.get(key)
    fp = _fingerprint(prob)
    if ent is not None and ent[0]() is prob and ent[2] == fp:
        dp = ent[1]
        if dp.nsteps != prob.nsteps or dp.tf != prob.tf:
            dp.set_nsteps(prob.nsteps, prob.tf)
        return dp
    if ent is not None:
        _evict(key)
    dp = DeviceProblem(prob, order, device)
    _cache[key] = (weakref.ref(prob), dp, fp)
    weakref.finalize(prob, _evict, key)
    return dp


def release(prob):
    """Close every cached handle of ``prob`` (all orders, all devices)."""
    for key in [k for k, ent in _cache.items() if k[0] == id(prob) and ent[0]() is prob]:
        _evict(key)


def clear_cache():
    for key in list(_cache):
        _evict(key)


# ---------------------------------------------------------------------------
# reference-shaped API
# ---------------------------------------------------------------------------
def _history_shape(prob, order, saveEveryNsteps=1):
    return (prob.real_system_size, 1 + order // 2, 1 + prob.nsteps // saveEveryNsteps, prob.N_initial_conditions)


def eval_forward_(uv_history, prob, controls, pcof, order=2, saveEveryNsteps=1, forcing=None):
    """eval_forward! (forward_evolution.jl:33-70): fills ``uv_history``
    ``[2N, 1+order/2, 1+nsteps, N_initial_conditions]`` (Fortran order) in place."""
    save = int(saveEveryNsteps)
    if save < 1:
        raise ValueError("saveEveryNsteps must be a positive integer")
    shape = _history_shape(prob, order, save)
    if uv_history.shape != shape or not uv_history.flags.f_contiguous:
        raise ValueError(f"uv_history must be Fortran-ordered with shape {shape}")
    dp = device_problem(prob, order)
    dp.set_controls(controls)
    run = dp.eval_forward if forcing is None else (lambda p, hist: dp.eval_forward_forced(p, forcing, hist))
    # the device keeps every time point; the stored ones are n = 0, save, 2 save, ... <= nsteps
    # (forward_evolution.jl:104,178,239-241: slot 1 + div(n, saveEveryNsteps) when n % saveEveryNsteps == 0): the
    # library's re-layout kernel reads them with that stride (qgd_set_save_every)
    dp.set_save_every(save)
    try:
        run(pcof, uv_history)
    finally:
        dp.set_save_every(1)
    return None


def eval_forward(prob, controls, pcof, order=2, saveEveryNsteps=1, forcing=None):
    """forward_evolution.jl:15-29: complex state history ``[N, 1+nsteps, N_initial_conditions]``."""
    hist = np.zeros(_history_shape(prob, order, int(saveEveryNsteps)), order="F")
    eval_forward_(hist, prob, controls, pcof, order=order, saveEveryNsteps=saveEveryNsteps, forcing=forcing)
    return real_to_complex(hist[:, 0, :, :])


def eval_grad_forced(prob, controls, pcof, target, order=2, cost_type="Infidelity"):
    """eval_grad_forced(prob, controls, pcof, target; order, cost_type) (src/eval_grad_forced.jl:17-60):
    the gradient of infidelity + guard penalty by differentiating the forward sweep (one forced sweep
    per control parameter).  On the device all parameters run at once as extra columns of the scan."""
    dp = device_problem(prob, order)
    dp.set_controls(controls)
    dp.set_target(target)
    dp.set_cost_type(cost_type)
    try:
        return dp.eval_grad_forced(pcof)
    finally:
        dp.set_cost_type("Infidelity")


def eval_grad_finite_difference(prob, controls, pcof, target, dpcof=1e-5, order=2, cost_type="Infidelity"):
    """eval_grad_finite_difference (src/eval_grad_finite_difference.jl:1-72): centred differences of
    infidelity + guard penalty, two device forward evaluations per control parameter (the third leg of
    the reference's adjoint / forced / finite-difference contract, compare_gradients.jl:47-65)."""
    dp = device_problem(prob, order)
    dp.set_controls(controls)
    dp.set_target(target)
    dp.set_cost_type(cost_type)
    pcof = np.asarray(pcof, dtype=np.float64)
    plain = COST_TYPES[str(cost_type)] == 0

    def cost(p):
        a, b, guard = dp.eval_forward(p)       # (:Tracking / :Norm: a is the cost itself, b = 0)
        return (1.0 - (a * a + b * b) / prob.N_ess_levels ** 2 if plain else a) + guard

    grad = np.zeros(len(pcof))
    try:
        for i in range(len(pcof)):
            e = np.zeros(len(pcof)); e[i] = dpcof
            grad[i] = (cost(pcof + e) - cost(pcof - e)) / (2 * dpcof)
    finally:
        dp.set_cost_type("Infidelity")
    return grad


def eval_adjoint(prob, controls, pcof, terminal_condition, order=2, forcing=None, lambda_derivatives=False):
    """QuantumGateDesign.eval_adjoint (forward_evolution.jl:300-315), used by the reference's scripts
    (examples/cnot2_optimization.jl:56, regression.jl:49): lambda history ``[2N, 1+order/2, 1+nsteps, c]``
    from a given terminal condition; column j=0 holds lambda_n (the only column the package consumes);
    ``lambda_derivatives=True`` also fills the derivative columns the reference leaves there."""
    dp = device_problem(prob, order)
    dp.set_controls(controls)
    dp.set_lambda_derivatives(lambda_derivatives)
    try:
        return dp.eval_adjoint(pcof, terminal_condition, forcing)
    finally:
        if lambda_derivatives:
            dp.set_lambda_derivatives(False)


def infidelity_real(psi, target, N_ess):
    """infidelity.jl:7-18 (host arithmetic on two small matrices)."""
    R = np.asarray(target, float)
    N = R.shape[0] // 2
    T = np.vstack([R[N:], -R[:N]])
    psi = np.asarray(psi, float)
    return 1 - (np.sum(psi * R) ** 2 + np.sum(psi * T) ** 2) / N_ess ** 2


def infidelity(*args, order=2):
    """infidelity(psi, target, N_ess)  or  infidelity(prob, controls, pcof, target; order)
    (infidelity.jl:20-47).  The second form runs the forward sweep on the device."""
    if len(args) == 3:
        psi, target, N_ess = args
        return infidelity_real(complex_to_real(psi), complex_to_real(target), N_ess)
    prob, controls, pcof, target = args
    dp = device_problem(prob, order)
    dp.set_controls(controls)
    dp.set_target(target)
    a, b, _ = dp.eval_forward(pcof)
    return 1 - (a * a + b * b) / prob.N_ess_levels ** 2


def guard_penalty_real(history, dt, T, W):
    """infidelity.jl:56-96 (host; the device computes the same sum inside eval_forward)."""
    h = np.asarray(history)
    if h.ndim == 3:
        h = h[..., None]
    w = h[:, 0, :, :]
    Ww = np.einsum("ij,jnc->inc", np.asarray(W), w)
    trap = np.ones(w.shape[1]); trap[0] = trap[-1] = 0.5
    return float(np.einsum("n,inc,inc->", trap, w, Ww) * dt / T)


def discrete_adjoint_(grad, history, lambda_history, adjoint_forcing, prob, controls, pcof, target,
                      order=2, cost_type="Infidelity", history_precomputed=False, lambda_derivatives=False):
    """discrete_adjoint! (eval_grad_discrete_adjoint.jl:107-160).  ``lambda_derivatives=True``: columns 1..m of
    ``lambda_history`` as the reference leaves them (default: zeros; only column 0 is ever consumed)."""
    dp = device_problem(prob, order)
    dp.set_controls(controls)
    dp.set_target(target)
    dp.set_cost_type(cost_type)
    if lambda_derivatives:
        dp.set_lambda_derivatives(True)
    try:
        g, _ = dp.discrete_adjoint(pcof, history_precomputed, history, lambda_history, adjoint_forcing)
    finally:
        dp.set_cost_type("Infidelity")
        if lambda_derivatives:
            dp.set_lambda_derivatives(False)
    grad[:] = g
    return grad


def discrete_adjoint(prob, controls, pcof, target, order=2, cost_type="Infidelity"):
    """eval_grad_discrete_adjoint.jl:83-102: returns the gradient."""
    grad = np.zeros(get_number_of_control_parameters(controls))
    return discrete_adjoint_(grad, None, None, None, prob, controls, pcof, target, order=order, cost_type=cost_type)
