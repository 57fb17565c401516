"""BatchProblem: B independent separable problems (or B problems x S right-hand sides) on one GPU.

This is the Python face of the ``vp_batch`` handle (include/varpro_hip.h).  It holds what the
reference's ``SeparableProblem`` + ``CachedCalculations`` hold (src/problem.rs:57-107) for a whole
batch, resident in HBM, and exposes the ``LeastSquaresProblem`` surface
(src/solvers/levmar/mod.rs:22-202) batch-wise.

Array conventions (row index fastest, i.e. each problem's slice is the reference's column-major
matrix):  Y, R: (B, S, m) or (B, m) when S == 1;  alpha: (B, q);  C: (B, S, n) / (B, n);
J: (B, q, S, m) / (B, q, m).

Inputs may be numpy arrays (host-pointer mode: the library stages through HBM) or torch CUDA
tensors (device-pointer mode: zero copies, results are torch tensors on the same device and the
work is enqueued on torch's current stream).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

REPORT_DTYPE = np.dtype([("termination", np.int32), ("n_evals", np.int32), ("objective", np.float64)])

STATUS_OK, STATUS_NONFINITE, STATUS_NOT_EVALUATED = 0, 1, 2


def _is_torch(a):
    return torch is not None and isinstance(a, torch.Tensor)


class LevenbergMarquardt:
    """mirrors ``levenberg_marquardt::LevenbergMarquardt`` builder knobs (reached in the reference via
    ``LevMarSolver::with_solver``, src/solvers/levmar/mod.rs:221-223)."""

    def __init__(self, dtype=np.float64):
        eps = float(np.finfo(dtype).eps)
        self.ftol = self.xtol = self.gtol = 30.0 * eps
        self.stepbound = 100.0
        self.patience = 100
        self.scale_diag = True

    @classmethod
    def new(cls, dtype=np.float64):
        return cls(dtype)

    def with_ftol(self, v):
        self.ftol = float(v)
        return self

    def with_xtol(self, v):
        self.xtol = float(v)
        return self

    def with_gtol(self, v):
        self.gtol = float(v)
        return self

    def with_tol(self, v):
        self.ftol = self.xtol = self.gtol = float(v)
        return self

    def with_stepbound(self, v):
        self.stepbound = float(v)
        return self

    def with_patience(self, v):
        self.patience = int(v)
        return self

    def with_scale_diag(self, v):
        self.scale_diag = bool(v)
        return self

    def _c(self):
        o = _lib.LmOpts()
        o.ftol, o.xtol, o.gtol, o.stepbound = self.ftol, self.xtol, self.gtol, self.stepbound
        o.patience, o.scale_diag = self.patience, int(self.scale_diag)
        return o


def _tensors_in(obj):
    if _is_torch(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_in(v)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            yield from _tensors_in(v)


def _device_entry(fn):
    """Stream discipline of device-pointer mode.  A handle enqueues all its work on ONE HIP stream: the stream that
    was current when the handle was created (one handle <-> one stream, include/varpro_hip.h).  If the caller's current
    stream is a different one, the call is bracketed: the handle's stream waits for the caller's stream (inputs
    ready), the body runs with the handle's stream current (so torch allocates the outputs there), the caller's stream
    waits for the handle's stream (outputs ready), and every tensor that crosses is recorded on the other stream so
    that the caching allocator does not recycle its memory while that stream still uses it."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        if not self.device_mode:
            return fn(self, *args, **kwargs)
        cur = torch.cuda.current_stream(self._tdev)
        hs = self._tstream
        if cur.cuda_stream == hs.cuda_stream:
            return fn(self, *args, **kwargs)
        hs.wait_stream(cur)
        for t in _tensors_in((args, kwargs)):
            if t.is_cuda:
                t.record_stream(hs)
        with torch.cuda.stream(hs):
            out = fn(self, *args, **kwargs)
        cur.wait_stream(hs)
        for t in _tensors_in(out):
            if t.is_cuda:
                t.record_stream(cur)
        return out

    return wrapper


class BatchProblem:
    def __init__(self, model, Y, x=None, weights=None, epsilon=None, device=0, grid_recurrence=True, stream_rows=False):
        """model: varpro_amd.SeparableModel; Y: (B, m) or (B, S, m); x: (m,) shared grid or (B, m)
        per-problem grids (default: model.x); weights: None (unit), (m,) or (B, m).
        grid_recurrence=False sets VP_FLAG_NO_GRID_RECURRENCE (per-row exponentials even on uniform grids)."""
        self.lib = _lib.load()
        self.model = model
        self.n = model.base_function_count()
        self.q = model.parameter_count()
        self.p = len(model.pairs)
        self.device_mode = _is_torch(Y)
        if self.device_mode:
            self._tdev = Y.device
        self.np_dtype = np.dtype(model.dtype)
        self.vp_dtype = _lib.VP_F64 if self.np_dtype == np.float64 else _lib.VP_F32
        self.external = hasattr(model, "ext_pairs")  # model.ExternalModel: the caller evaluates Phi / dPhi
        x = model.x if x is None else x
        Y = self._as_array(Y)
        if Y.ndim == 2:
            self.single_rhs = True
            B, m = Y.shape
            S = 1
        elif Y.ndim == 3:
            self.single_rhs = False
            B, S, m = Y.shape
        else:
            raise ValueError("Y must be (B, m) or (B, S, m)")
        self.B, self.S, self.m = int(B), int(S), int(m)
        if self.external:
            x = None
        else:
            x = self._as_array(x)
        flags = 0 if grid_recurrence else _lib.VP_FLAG_NO_GRID_RECURRENCE
        if stream_rows:  # VP_FLAG_STREAM_ROWS: the length-agnostic fit kernels even where a resident set covers m
            flags |= _lib.VP_FLAG_STREAM_ROWS
        if self.device_mode:
            flags |= _lib.VP_FLAG_DEVICE_PTRS  # work is enqueued on torch's current stream
        else:
            flags |= _lib.VP_FLAG_OWN_STREAM
        if x is None:
            pass  # a caller-evaluated model has no grid
        elif x.ndim == 2:
            flags |= _lib.VP_FLAG_T_PER_PROBLEM
            if tuple(x.shape) != (self.B, self.m):
                raise ValueError("per-problem grid must be (B, m)")
        elif int(x.shape[0]) != self.m:
            # SeparableProblemBuilderError::InvalidLengthOfData (src/problem/builder.rs:294-299)
            raise ValueError("InvalidLengthOfData: x length = %d and y length = %d" % (int(x.shape[0]), self.m))
        w = None
        if weights is not None:
            w = self._as_array(weights)
            if w.ndim == 2:
                flags |= _lib.VP_FLAG_W_PER_PROBLEM
                if tuple(w.shape) != (self.B, self.m):
                    raise ValueError("InvalidLengthOfWeights")
            elif int(w.shape[0]) != self.m:
                raise ValueError("InvalidLengthOfWeights")  # src/problem/builder.rs:300-303
        self._keep = (x, Y, w)
        stream = None
        if self.device_mode:
            device = Y.device.index if Y.device.index is not None else torch.cuda.current_device()
            self._tstream = torch.cuda.current_stream(device)  # the handle's stream for its whole life
            stream = C.c_void_p(self._tstream.cuda_stream)
        self.device = int(device)
        h = C.c_void_p()
        eps = -1.0 if epsilon is None else abs(float(epsilon))
        if self.external:
            npairs = len(model.ext_pairs)
            pb = (C.c_int32 * max(1, npairs))(*[j for j, _k in model.ext_pairs])
            pp = (C.c_int32 * max(1, npairs))(*[k for _j, k in model.ext_pairs])
            check(self.lib.vp_batch_create_external(C.byref(h), self.n, self.q, npairs, pb, pp, self.vp_dtype, self.m,
                                                    self.S, self.B, self._ptr(Y), self._ptr(w), eps, flags, self.device,
                                                    stream))
        else:
            desc = model.desc()
            check(self.lib.vp_batch_create(C.byref(h), C.byref(desc), self.vp_dtype, self.m, self.S, self.B,
                                           self._ptr(x), self._ptr(Y), self._ptr(w), eps, flags, self.device, stream))
        self._h = h
        self._have_params = False  # mirrors the handle: set_params / evaluate / fit has run on the current data
        self._keep = None  # the handle owns copies (Y_w = W*Y, t, w)

    # ---- plumbing ----
    def _as_array(self, a):
        if self.device_mode:
            if not _is_torch(a):
                a = torch.as_tensor(np.asarray(a), device=self._torch_device())
            tdt = torch.float64 if self.np_dtype == np.float64 else torch.float32
            return a.to(dtype=tdt).contiguous()
        return np.ascontiguousarray(a, dtype=self.np_dtype)

    def _torch_device(self):
        return getattr(self, "_tdev", None) or "cuda"

    def _ptr(self, a):
        if a is None:
            return None
        if _is_torch(a):
            self._tdev = a.device
            return C.c_void_p(a.data_ptr())
        return C.c_void_p(a.ctypes.data)

    def _empty(self, shape, dtype=None):
        if self.device_mode:
            if dtype is None:
                tdt = torch.float64 if self.np_dtype == np.float64 else torch.float32
            else:
                tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.int32): torch.int32,
                       np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)]
            return torch.empty(shape, dtype=tdt, device=self._torch_device())
        return np.empty(shape, dtype=self.np_dtype if dtype is None else dtype)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vp_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _shape_rhs(self, *lead_trail):
        return lead_trail

    # ---- LeastSquaresProblem surface, batch-wise ----
    @_device_entry
    def set_params(self, alpha):
        """== SeparableProblem::set_params (src/solvers/levmar/mod.rs:42-73)"""
        a = self._as_array(alpha).reshape(self.B, self.q)
        check(self.lib.vp_set_params(self._h, self._ptr(a)))
        self._have_params = True

    # ---- caller-evaluated models (model.ExternalModel; vp_batch_create_external) ----
    @_device_entry
    def set_params_with_basis(self, alpha, Phi, dPhi=None):
        """== SeparableProblem::set_params (src/solvers/levmar/mod.rs:42-73) with `model.eval()` replaced by its result:
        Phi (B, n, m) and optionally the derivative columns dPhi (B, p, m), both UNWEIGHTED.  Device tensors are kept by
        POINTER until the next call (keep them alive and unchanged); numpy arrays are copied."""
        a = self._as_array(alpha).reshape(self.B, self.q)
        Phi = self._as_array(Phi).reshape(self.B, self.n, self.m)
        dPhi = None if dPhi is None else self._as_array(dPhi).reshape(self.B, self.p, self.m)
        self._ext_keep = (Phi, dPhi)
        check(self.lib.vp_set_params_with_basis(self._h, self._ptr(a), self._ptr(Phi), self._ptr(dPhi)))
        self._have_params = True

    @_device_entry
    def jacobian_with_derivatives(self, dPhi, with_status=False):
        """== jacobian() (src/solvers/levmar/mod.rs:101-201) with `model.eval_partial_deriv(k)` (:141) replaced by its
        non-zero columns dPhi (B, p, m) at the current parameters"""
        dPhi = self._as_array(dPhi).reshape(self.B, self.p, self.m)
        self._ext_keep = (getattr(self, "_ext_keep", (None, None))[0], dPhi)
        J = self._empty((self.B, self.q, self.S * self.m))
        st = self._empty((self.B,), np.int32)
        check(self.lib.vp_jacobian_with_derivatives(self._h, self._ptr(dPhi), self._ptr(J), self._ptr(st)))
        if not self._have_params:
            J = None
        return (J, st) if with_status else J

    @_device_entry
    def evaluate_with_basis(self, alpha, Phi, dPhi=None, want_residuals=True, want_jacobian=True):
        """fused form (vp_evaluate_with_basis): Phi, dPhi in -> r, J, C, cost, status out in one pass"""
        a = self._as_array(alpha).reshape(self.B, self.q)
        Phi = self._as_array(Phi).reshape(self.B, self.n, self.m)
        dPhi = None if dPhi is None else self._as_array(dPhi).reshape(self.B, self.p, self.m)
        self._ext_keep = (Phi, dPhi)
        want_jacobian = want_jacobian and (dPhi is not None or self.p == 0)
        r = self._empty((self.B, self.S * self.m)) if want_residuals else None
        J = self._empty((self.B, self.q, self.S * self.m)) if want_jacobian else None
        Cm = self._empty((self.B, self.S, self.n))
        cost = self._empty((self.B,), np.float64)
        st = self._empty((self.B,), np.int32)
        check(self.lib.vp_evaluate_with_basis(self._h, self._ptr(a), self._ptr(Phi), self._ptr(dPhi), self._ptr(r),
                                              self._ptr(J), self._ptr(Cm), self._ptr(cost), self._ptr(st)))
        self._have_params = True
        return dict(r=r, J=J, C=Cm.reshape(self.B, self.n) if self.single_rhs else Cm, cost=cost, status=st)

    @_device_entry
    def params(self):
        out = self._empty((self.B, self.q))
        check(self.lib.vp_params(self._h, self._ptr(out)))
        return out

    @_device_entry
    def status(self):
        st = self._empty((self.B,), np.int32)
        check(self.lib.vp_linear_coeffs(self._h, None, self._ptr(st)))
        return st

    @_device_entry
    def residuals(self, with_status=False):
        """== residuals() (src/solvers/levmar/mod.rs:91-95): (B, S*m) column-stacked per problem"""
        r = self._empty((self.B, self.S * self.m))
        st = self._empty((self.B,), np.int32)
        check(self.lib.vp_residuals(self._h, self._ptr(r), self._ptr(st)))
        if not self._have_params:
            r = None  # residuals() before any set_params(): the reference returns None (cached is None)
        return (r, st) if with_status else r

    @_device_entry
    def jacobian(self, with_status=False):
        """== jacobian() (src/solvers/levmar/mod.rs:101-201): (B, q, S*m); J[b, k] is column k"""
        J = self._empty((self.B, self.q, self.S * self.m))
        st = self._empty((self.B,), np.int32)
        check(self.lib.vp_jacobian(self._h, self._ptr(J), self._ptr(st)))
        if not self._have_params:
            J = None  # jacobian() before any set_params(): None, not an uninitialised buffer
        return (J, st) if with_status else J

    @_device_entry
    def linear_coefficients(self):
        """(B, n) for single RHS, (B, S, n) for MRHS (each [b] is the reference's n x S matrix, column-major)"""
        Cm = self._empty((self.B, self.S, self.n))
        check(self.lib.vp_linear_coeffs(self._h, self._ptr(Cm), None))
        return Cm.reshape(self.B, self.n) if self.single_rhs else Cm

    @_device_entry
    def weighted_data(self):
        Yw = self._empty((self.B, self.S, self.m))
        check(self.lib.vp_weighted_data(self._h, self._ptr(Yw)))
        return Yw.reshape(self.B, self.m) if self.single_rhs else Yw

    @_device_entry
    def cost(self):
        c = self._empty((self.B,), np.float64)
        check(self.lib.vp_cost(self._h, self._ptr(c)))
        return c

    @_device_entry
    def evaluate(self, alpha, want_residuals=True, want_jacobian=True):
        """fused set_params + residuals + jacobian + coefficients + cost in one launch (vp_evaluate)"""
        a = self._as_array(alpha).reshape(self.B, self.q)
        r = self._empty((self.B, self.S * self.m)) if want_residuals else None
        J = self._empty((self.B, self.q, self.S * self.m)) if want_jacobian else None
        Cm = self._empty((self.B, self.S, self.n))
        cost = self._empty((self.B,), np.float64)
        st = self._empty((self.B,), np.int32)
        check(self.lib.vp_evaluate(self._h, self._ptr(a), self._ptr(r), self._ptr(J), self._ptr(Cm), self._ptr(cost),
                                   self._ptr(st)))
        self._have_params = True
        return dict(r=r, J=J, C=Cm.reshape(self.B, self.n) if self.single_rhs else Cm, cost=cost, status=st)

    # ---- model surface ----
    @_device_entry
    def basis(self, alpha, skip_invariant=False, want_phi=True, want_dphi=True, out_phi=None, out_dphi=None):
        """== eval / eval_partial_deriv for the batch, UNWEIGHTED (vp_basis): Phi (B, n, m), dPhi (B, p, m)"""
        a = self._as_array(alpha).reshape(self.B, self.q)
        ncols = self.n - (sum(1 for k in self.model.kinds if k == 0) if skip_invariant else 0)
        phi = out_phi if out_phi is not None else (self._empty((self.B, ncols, self.m)) if want_phi else None)
        dphi = out_dphi if out_dphi is not None else (self._empty((self.B, self.p, self.m)) if want_dphi else None)
        check(self.lib.vp_basis(self._h, self._ptr(a), self._ptr(phi), self._ptr(dphi),
                                _lib.VP_BASIS_SKIP_INVARIANT if skip_invariant else 0))
        return phi, dphi

    # ---- solver surface ----
    @_device_entry
    def fit(self, alpha0, solver=None, want_coefficients=True):
        """== LevMarSolver::fit for every problem (vp_fit).  Returns (alpha, C, report) where report
        is a structured numpy array (termination, n_evals, objective); termination > 0 <=> Ok."""
        solver = solver or LevenbergMarquardt(self.np_dtype)
        opts = solver._c()
        a = self._as_array(alpha0).reshape(self.B, self.q)
        a = a.clone() if _is_torch(a) else a.copy()
        Cm = self._empty((self.B, self.S, self.n)) if want_coefficients else None
        if self.device_mode:
            rep_t = torch.empty((self.B, 16), dtype=torch.uint8, device=self._torch_device())
            check(self.lib.vp_fit(self._h, C.byref(opts), self._ptr(a), self._ptr(Cm), self._ptr(rep_t)))
            self._have_params = True  # (only once the call has succeeded: a failed fit leaves the handle without parameters)
            rep = rep_t  # raw bytes on device; use report_to_numpy() to decode
            if Cm is not None and self.single_rhs:
                Cm = Cm.reshape(self.B, self.n)
            return a, Cm, rep
        else:
            rep = np.zeros(self.B, dtype=REPORT_DTYPE)
            check(self.lib.vp_fit(self._h, C.byref(opts), self._ptr(a), self._ptr(Cm), C.c_void_p(rep.ctypes.data)))
            self._have_params = True
        if Cm is not None and self.single_rhs:
            Cm = Cm.reshape(self.B, self.n)
        return a, Cm, rep

    # ---- batched fit of a caller-evaluated model by reverse communication (vp_fit_begin / _step_with_basis / _end) ----
    @_device_entry
    def fit_begin(self, alpha0, solver=None, derivatives_on_accept=False):
        """== LevMarSolver::fit (src/solvers/levmar/mod.rs:238-254) for a batch whose model the CALLER evaluates: the
        device keeps one LM driver per problem, the model's columns enter step by step (``fit_step_with_basis``).
        The first step takes the columns at alpha0."""
        solver = solver or LevenbergMarquardt(self.np_dtype)
        opts = solver._c()
        a = self._as_array(alpha0).reshape(self.B, self.q)
        flags = _lib.VP_FIT_DERIVATIVES_ON_ACCEPT if derivatives_on_accept else 0
        check(self.lib.vp_fit_begin(self._h, C.byref(opts), self._ptr(a), flags))
        self._have_params = False
        self._xf_trial = self._empty((self.B, self.q))
        self._xf_want = self._empty((self.B,), np.int32)

    @_device_entry
    def fit_step_with_basis(self, Phi, dPhi=None, want_count=True):
        """one LM iteration of every active problem (vp_fit_step_with_basis): Phi (B, n, m) / dPhi (B, p, m) at the trial
        points of the previous step -> (alpha_trial (B, q), want (B,) int32 of VP_WANT_* bits, n_active or None).
        The returned arrays are REUSED by the next step."""
        Phi = self._as_array(Phi).reshape(self.B, self.n, self.m)
        dPhi = None if dPhi is None else self._as_array(dPhi).reshape(self.B, self.p, self.m)
        self._ext_keep = (Phi, dPhi)
        nact = C.c_int64(0)
        check(self.lib.vp_fit_step_with_basis(self._h, self._ptr(Phi), self._ptr(dPhi), self._ptr(self._xf_trial),
                                              self._ptr(self._xf_want), C.byref(nact) if want_count else None))
        return self._xf_trial, self._xf_want, (int(nact.value) if want_count else None)

    @_device_entry
    def fit_active_set(self, index_out=None, count_out=None):
        """the compacted active set of the running stepped fit (vp_fit_active_set): ``(index (B,) int32, count (1,) int32)`` --
        index[:count] are the problems still running; entries beyond are stale but valid indices.  Device-pointer handles:
        torch tensors filled asynchronously (no synchronisation); pass the previous call's tensors to reuse them."""
        idx = index_out if index_out is not None else self._empty((self.B,), np.int32)
        cnt = count_out if count_out is not None else self._empty((1,), np.int32)
        check(self.lib.vp_fit_active_set(self._h, self._ptr(idx), self._ptr(cnt)))
        return idx, cnt

    @_device_entry
    def fit_end(self, want_coefficients=True):
        """(alpha, C, report) of the stepped fit (vp_fit_end); report as ``fit`` returns it"""
        a = self._empty((self.B, self.q))
        Cm = self._empty((self.B, self.n) if self.single_rhs else (self.B, self.S, self.n)) if want_coefficients else None
        if self.device_mode:
            rep = torch.empty((self.B, 16), dtype=torch.uint8, device=self._torch_device())
            check(self.lib.vp_fit_end(self._h, self._ptr(a), self._ptr(Cm), self._ptr(rep)))
        else:
            rep = np.zeros(self.B, dtype=REPORT_DTYPE)
            check(self.lib.vp_fit_end(self._h, self._ptr(a), self._ptr(Cm), C.c_void_p(rep.ctypes.data)))
        self._have_params = True
        return a, Cm, rep

    def fit_with_model(self, evaluate, alpha0, solver=None, derivatives_on_accept=False, max_steps=None, check_every=1):
        """The whole stepped fit: ``evaluate(alpha (B, q), want (B,) or None) -> (Phi, dPhi)`` is the caller's model (numpy
        on host-pointer handles, torch on device-pointer handles; ``want`` is None for the first call, afterwards the
        VP_WANT_* bits per problem -- entries of problems that want nothing are never read, a model may skip them).
        Returns (alpha, C, report, steps)."""
        self.fit_begin(alpha0, solver, derivatives_on_accept)
        solver = solver or LevenbergMarquardt(self.np_dtype)
        limit = max_steps if max_steps is not None else 2 * (solver.patience * (self.q + 1) + 2)
        alpha, want = self._as_array(alpha0).reshape(self.B, self.q), None
        steps = 0
        while steps < limit:
            Phi, dPhi = evaluate(alpha, want)
            steps += 1
            look = (steps % check_every == 0) or steps == limit
            alpha, want, nact = self.fit_step_with_basis(Phi, dPhi, want_count=look)
            if look and nact == 0:
                break
        a, Cm, rep = self.fit_end()
        return a, Cm, rep, steps

    def fit_trace(self, alpha0, solver=None, max_rows=512):
        """diagnostics (host mode only): fit + per-evaluation trace (B, max_rows, q+4) with rows
        [alpha_trial, ||r||, ratio, delta, par]; unused rows are NaN"""
        assert not self.device_mode
        solver = solver or LevenbergMarquardt(self.np_dtype)
        opts = solver._c()
        a = self._as_array(alpha0).reshape(self.B, self.q).copy()
        Cm = self._empty((self.B, self.S, self.n))
        rep = np.zeros(self.B, dtype=REPORT_DTYPE)
        tr = np.zeros((self.B, max_rows, self.q + 4))
        check(self.lib.vp_fit_trace(self._h, C.byref(opts), self._ptr(a), self._ptr(Cm), C.c_void_p(rep.ctypes.data),
                                    C.c_void_p(tr.ctypes.data), int(max_rows)))
        self._have_params = True
        if self.single_rhs:
            Cm = Cm.reshape(self.B, self.n)
        return a, Cm, rep, tr

    @staticmethod
    def report_to_numpy(rep):
        if _is_torch(rep):
            return rep.cpu().numpy().view(REPORT_DTYPE).reshape(-1)
        return rep

    @_device_entry
    def debug_gram_evaluate(self, alpha):
        """diagnostics (vp_debug_gram_evaluate): the fp64-Gram fit kernel's formulation evaluated once at alpha ->
        dict(cost (B,), C (B, n), Jtr (B, q), JtJ (B, q, q)), all float64"""
        a = self._as_array(alpha).reshape(self.B, self.q)
        per = 1 + self.n + self.q + self.q * self.q
        out = self._empty((self.B, per), np.float64)
        check(self.lib.vp_debug_gram_evaluate(self._h, self._ptr(a), self._ptr(out)))
        n, q = self.n, self.q
        return dict(cost=out[:, 0], C=out[:, 1:1 + n], Jtr=out[:, 1 + n:1 + n + q],
                    JtJ=out[:, 1 + n + q:].reshape(self.B, q, q))

    @_device_entry
    def best_fit(self):
        """== FitResult::best_fit (src/fit.rs:55-59, 87-91)"""
        f = self._empty((self.B, self.S, self.m))
        check(self.lib.vp_best_fit(self._h, self._ptr(f)))
        return f.reshape(self.B, self.m) if self.single_rhs else f

    @_device_entry
    def statistics(self, want_confidence_sigma=True):
        """== FitStatistics::try_calculate for every problem (vp_statistics): dict(cov (B,k,k),
        reduced_chi2 (B,), conf_sigma (B,m) or None, status (B,), dof)"""
        k = self.n + self.q
        cov = self._empty((self.B, k, k))
        chi2 = self._empty((self.B,), np.float64)
        sig = self._empty((self.B, self.m)) if want_confidence_sigma else None
        st = self._empty((self.B,), np.int32)
        check(self.lib.vp_statistics(self._h, self._ptr(cov), self._ptr(chi2), self._ptr(sig), self._ptr(st)))
        return dict(cov=cov, reduced_chi2=chi2, conf_sigma=sig, status=st, dof=self.m - k)

    @_device_entry
    def set_observations(self, Y):
        """replace the data of this handle by another batch of the same shape (vp_set_observations): the next frame
        of a stream of same-shaped problems without re-allocating the device state"""
        Y = self._as_array(Y)
        want = (self.B, self.m) if self.single_rhs else (self.B, self.S, self.m)
        if tuple(Y.shape) != want:
            raise ValueError("observations must have shape %r" % (want,))
        if self.device_mode and not Y.is_contiguous():
            Y = Y.contiguous()
        check(self.lib.vp_set_observations(self._h, self._ptr(Y)))
        self._have_params = False

    def set_rhs_allreduce(self, global_rhs_count, group=None):
        """Shard ONE global fit over ranks by right-hand sides (vp_set_rhs_allreduce): this handle
        holds the local block of the S columns (torch device tensors); per LM evaluation the B*(1+n*n+p) reduced
        sums are all-reduced over `group` (torch.distributed: backend "nccl" == RCCL over xGMI; "gloo" is bounced
        through the host and only meant for tests).  Pass global_rhs_count=None to switch back."""
        if global_rhs_count is None:
            self._rhs_cb = None
            check(self.lib.vp_set_rhs_allreduce(self._h, _lib.ALLREDUCE_FN(0), None, 0))
            return
        if not self.device_mode:
            raise ValueError("right-hand-side sharding needs device tensors (the collective runs on the stream)")
        import torch
        import torch.distributed as dist

        dev = self._tdev

        class _Raw:  # zero-copy view of the library's device buffer for torch
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                                 "version": 2}

        def _cb(ptr, count, stream, user):
            try:
                # the library hands over the stream its kernels run on: the collective must be ordered on THAT
                # stream (after the partial-sum kernel, before the LM step), whatever torch's current stream is
                sptr = int(stream) if stream else 0
                lib_stream = torch.cuda.ExternalStream(sptr, device=dev) if sptr else torch.cuda.default_stream(dev)
                with torch.cuda.stream(lib_stream):
                    t = torch.as_tensor(_Raw(ptr, count), device=dev)
                    if dist.get_backend(group) == "gloo":
                        h = t.cpu()
                        dist.all_reduce(h, group=group)
                        t.copy_(h)
                    else:
                        dist.all_reduce(t, group=group)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                self._rhs_cb_error = e
                return -1

        self._rhs_cb = _lib.ALLREDUCE_FN(_cb)  # keep alive as long as the handle uses it
        check(self.lib.vp_set_rhs_allreduce(self._h, self._rhs_cb, None, int(global_rhs_count)))

    @_device_entry
    def summary(self):
        """local {sum cost, #successful, #failed, sum n_evals} after fit (vp_summary)"""
        out = (C.c_double * 4)()
        check(self.lib.vp_summary(self._h, out))
        return np.array(list(out))

    @_device_entry
    def global_fit_condition(self):
        """(B,) largest estimate of cond(J D^-1) the Gram-based LM steps of the last GLOBAL fit (S > 1) have seen
        (vp_global_fit_condition): the step is exact to ~10 cond^2 eps -- beyond ~1e5 it is at rounding level"""
        out = self._empty((self.B,), np.float64)
        check(self.lib.vp_global_fit_condition(self._h, self._ptr(out)))
        return out

    @_device_entry
    def summary_device(self, out):
        """the same 4 aggregates into a CUDA float64 tensor of 4 elements, asynchronously on the handle's stream
        (no host synchronisation): ready to be all-reduced over RCCL"""
        assert _is_torch(out) and out.is_cuda and out.numel() == 4 and out.dtype == torch.float64
        check(self.lib.vp_summary_device(self._h, C.c_void_p(out.data_ptr())))
        return out

    def set_fit_kernel(self, which):
        """single-RHS fit kernel selection: "auto" | "wave" (one wavefront per problem) | "slots" (persistent
        slot kernel wherever it covers the problem) -- see include/varpro_hip.h:vp_set_fit_kernel"""
        code = {"auto": _lib.VP_FIT_KERNEL_AUTO, "wave": _lib.VP_FIT_KERNEL_WAVE, "slots": _lib.VP_FIT_KERNEL_SLOTS}[which]
        check(self.lib.vp_set_fit_kernel(self._h, code))

    def set_refit(self, enable=True):
        """diagnostics (include/varpro_hip_debug.h:vp_debug_set_refit): False returns what the fit kernels themselves report
        for a problem whose Jacobian factor is not representable column by column (Numerical, parameters = the guess)
        instead of re-fitting it with scaled columns in vp_fit's second launch"""
        check(self.lib.vp_debug_set_refit(self._h, int(enable)))

    def set_timing(self, enable=True):
        check(self.lib.vp_set_timing(self._h, int(enable)))

    def last_kernel_ms(self, which):
        ms = C.c_float(-1.0)
        check(self.lib.vp_last_kernel_ms(self._h, int(which), C.byref(ms)))
        return float(ms.value)

    def synchronize(self):
        check(self.lib.vp_synchronize(self._h))
