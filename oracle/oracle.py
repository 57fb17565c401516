"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE -- see oracle/varpro_oracle.h).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import
this module.  The product package ``varpro_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvarpro_oracle.so")

VP_MAX_BASIS = 8
VP_MAX_PARAMS = 8
VP_MAX_BASIS_PARAMS = 2

CONST, EXP_DECAY, EXP_RATE, EXP_COS, SIN_PHASE = 0, 1, 2, 3, 4


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_basis", C.c_int32),
        ("n_params", C.c_int32),
        ("kind", C.c_int32 * VP_MAX_BASIS),
        ("param", (C.c_int32 * VP_MAX_BASIS_PARAMS) * VP_MAX_BASIS),
    ]


class LmOpts(C.Structure):
    _fields_ = [
        ("ftol", C.c_double),
        ("xtol", C.c_double),
        ("gtol", C.c_double),
        ("stepbound", C.c_double),
        ("patience", C.c_int32),
        ("scale_diag", C.c_int32),
    ]


class Report(C.Structure):
    _fields_ = [("termination", C.c_int32), ("n_evals", C.c_int32), ("objective", C.c_double)]


REPORT_DTYPE = np.dtype([("termination", np.int32), ("n_evals", np.int32), ("objective", np.float64)])


# callbacks of a model outside the descriptor language (vpo_problem_set_external_model): eval() and eval_partial_deriv(k)
EXT_EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))
EXT_DPHI_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double))


def build():
    """(re)build liboracle with the committed Makefile"""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    lib.vpo_problem_create.restype = C.c_void_p
    lib.vpo_problem_create.argtypes = [C.POINTER(ModelDesc), C.c_int, C.c_int, dp, dp, dp, C.c_double,
                                       C.POINTER(C.c_int)]
    lib.vpo_problem_destroy.argtypes = [C.c_void_p]
    lib.vpo_eval_phi.argtypes = [C.POINTER(ModelDesc), C.c_int, dp, dp, dp]
    lib.vpo_eval_dphi.argtypes = [C.POINTER(ModelDesc), C.c_int, dp, dp, C.c_int, dp]
    lib.vpo_set_params.argtypes = [C.c_void_p, dp]
    lib.vpo_residuals.argtypes = [C.c_void_p, dp]
    lib.vpo_residuals.restype = C.c_int
    lib.vpo_jacobian.argtypes = [C.c_void_p, dp]
    lib.vpo_jacobian.restype = C.c_int
    lib.vpo_best_fit.argtypes = [C.c_void_p, dp]
    lib.vpo_best_fit.restype = C.c_int
    lib.vpo_fit.argtypes = [C.c_void_p, C.POINTER(LmOpts), C.POINTER(Report)]
    lib.vpo_fit_trace.argtypes = [C.c_void_p, C.POINTER(LmOpts), C.POINTER(Report), dp, C.c_int]
    lib.vpo_fit_trace.restype = C.c_int
    lib.vpo_statistics.argtypes = [C.c_void_p, dp, dp, dp]
    lib.vpo_statistics.restype = C.c_int
    lib.vpo_thin_svd.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp]
    lib.vpo_enorm.argtypes = [C.c_int, dp]
    lib.vpo_enorm.restype = C.c_double
    lib.vpo_fit_batch.restype = C.c_double
    lib.vpo_fit_batch.argtypes = [C.POINTER(ModelDesc), C.c_int, C.c_int64, dp, dp, dp, C.c_double,
                                  C.POINTER(LmOpts), dp, dp, C.c_void_p, C.c_int]
    lib.vpo_evaluate_batch.argtypes = [C.POINTER(ModelDesc), C.c_int, C.c_int64, dp, dp, dp, C.c_double, dp, dp,
                                       dp, dp, dp, C.POINTER(C.c_int32), C.c_int]
    lib.vpo_lm_opts_default.argtypes = [C.POINTER(LmOpts)]
    lib.vpo_max_threads.restype = C.c_int
    lib.vpo_problem_set_external_model.argtypes = [C.c_void_p, EXT_EVAL_FN, EXT_DPHI_FN, C.c_void_p]
    lib.vpo_problem_set_external_model.restype = None
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def make_desc(kinds, params, n_params):
    """kinds: list of basis kinds; params: list of tuples of alpha indices per basis"""
    d = ModelDesc()
    d.n_basis = len(kinds)
    d.n_params = n_params
    for j in range(VP_MAX_BASIS):
        d.kind[j] = 0
        for a in range(VP_MAX_BASIS_PARAMS):
            d.param[j][a] = -1
    for j, (k, ps) in enumerate(zip(kinds, params)):
        d.kind[j] = int(k)
        for a, pi in enumerate(ps):
            d.param[j][a] = int(pi)
    return d


def desc_of(model):
    """accepts an oracle ModelDesc, or any object with .kinds/.param_indices/.n_params (duck-typed model)"""
    if isinstance(model, ModelDesc):
        return model
    return make_desc(list(model.kinds), [tuple(p) for p in model.param_indices], int(model.n_params))


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def default_opts(**kw):
    o = LmOpts()
    lib().vpo_lm_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def make_shape_desc(n_basis, n_params):
    """descriptor that only carries the shape of a caller-evaluated model (kinds unused)"""
    d = make_desc([CONST] * n_basis, [()] * n_basis, n_params)
    return d


def multiexp_desc(n_exp, offset=True):
    kinds = [EXP_DECAY] * n_exp + ([CONST] if offset else [])
    params = [(i,) for i in range(n_exp)] + ([()] if offset else [])
    return make_desc(kinds, params, n_exp)


class Problem:
    """one SeparableProblem on the CPU oracle (Y: (m,) or (S, m) i.e. [s][i])"""

    def __init__(self, model, t, Y, w=None, eps=-1.0, external=None):
        """external = (eval, dphi): a model outside the descriptor language, i.e. ANY SeparableNonlinearModel
        (src/model/mod.rs:239-363) -- eval(alpha) -> Phi (n, m), dphi(alpha, k) -> D_k (n, m) with zero rows for
        basis functions that do not depend on alpha_k; `model` then only carries the shape (make_shape_desc)."""
        self.desc = desc_of(model)
        self.t = np.ascontiguousarray(np.zeros(np.asarray(Y).shape[-1]) if t is None else t, dtype=np.float64)
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        self.single = Y.ndim == 1
        Y2 = Y.reshape(1, -1) if self.single else Y
        self.S, self.m = Y2.shape
        self.n, self.q = self.desc.n_basis, self.desc.n_params
        self.w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        err = C.c_int(0)
        self._h = lib().vpo_problem_create(C.byref(self.desc), self.m, self.S, _dp(self.t), _dp(Y2), _dp(self.w),
                                           float(eps), C.byref(err))
        if not self._h:
            raise ValueError("oracle problem build error %d" % err.value)
        if external is not None:
            ev, dv = external
            n, m, q = self.n, self.m, self.q

            def _eval(_user, a_ptr, out_ptr):
                a = np.ctypeslib.as_array(a_ptr, shape=(max(q, 1),))[:q]
                np.ctypeslib.as_array(out_ptr, shape=(n, m))[:] = np.asarray(ev(a.copy()), dtype=np.float64).reshape(n, m)

            def _dphi(_user, a_ptr, k, out_ptr):
                a = np.ctypeslib.as_array(a_ptr, shape=(max(q, 1),))[:q]
                np.ctypeslib.as_array(out_ptr, shape=(n, m))[:] = np.asarray(dv(a.copy(), int(k)), dtype=np.float64).reshape(n, m)

            self._ext = (EXT_EVAL_FN(_eval), EXT_DPHI_FN(_dphi))  # keep alive
            lib().vpo_problem_set_external_model(self._h, self._ext[0], self._ext[1], None)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vpo_problem_destroy(self._h)
            except Exception:  # interpreter shutdown: module globals may already be gone
                pass
            self._h = None

    def set_params(self, alpha):
        a = np.ascontiguousarray(alpha, dtype=np.float64)
        assert a.size == self.q
        lib().vpo_set_params(self._h, _dp(a))

    def _struct(self):
        class P(C.Structure):
            _fields_ = [("model", ModelDesc), ("m", C.c_int), ("S", C.c_int), ("t", C.c_void_p), ("w", C.c_void_p),
                        ("Yw", C.c_void_p), ("eps", C.c_double), ("alpha", C.POINTER(C.c_double)),
                        ("cached", C.c_int), ("U", C.POINTER(C.c_double)), ("sigma", C.POINTER(C.c_double)),
                        ("V", C.POINTER(C.c_double)), ("C", C.POINTER(C.c_double)), ("R", C.POINTER(C.c_double)),
                        ("n_set_params", C.c_long), ("n_jacobians", C.c_long)]
        return C.cast(self._h, C.POINTER(P)).contents

    def params(self):
        st = self._struct()
        return np.array([st.alpha[i] for i in range(self.q)])

    def cached(self):
        return bool(self._struct().cached)

    def linear_coefficients(self):
        st = self._struct()
        if not st.cached:
            return None
        c = np.array([st.C[i] for i in range(self.n * self.S)]).reshape(self.S, self.n)
        return c[0] if self.single else c

    def singular_values(self):
        st = self._struct()
        return np.array([st.sigma[i] for i in range(self.n)])

    def residuals(self):
        r = np.empty(self.m * self.S)
        if not lib().vpo_residuals(self._h, _dp(r)):
            return None
        return r

    def jacobian(self):
        """(q, S*m): row k is Jacobian column k (memory order of the reference's column-major matrix)"""
        J = np.empty((self.q, self.m * self.S))
        if not lib().vpo_jacobian(self._h, _dp(J)):
            return None
        return J

    def best_fit(self):
        f = np.empty((self.S, self.m))
        if not lib().vpo_best_fit(self._h, _dp(f)):
            return None
        return f[0] if self.single else f

    def fit(self, opts=None):
        opts = opts or default_opts()
        rep = Report()
        lib().vpo_fit(self._h, C.byref(opts), C.byref(rep))
        return rep

    def fit_trace(self, opts=None, max_rows=512):
        """fit and return (report, trace[rows, q+4]) with rows [x_trial, ||r||, ratio, delta, par]"""
        opts = opts or default_opts()
        rep = Report()
        tr = np.zeros((max_rows, self.q + 4))
        n = lib().vpo_fit_trace(self._h, C.byref(opts), C.byref(rep), _dp(tr), max_rows)
        return rep, tr[:n]

    def statistics(self):
        """== FitStatistics::try_calculate: dict(cov (k,k), reduced_chi2, conf_sigma (m,)) or None"""
        k = self.n + self.q
        cov = np.zeros((k, k))
        chi2 = np.zeros(1)
        sig = np.zeros(self.m)
        if not lib().vpo_statistics(self._h, _dp(cov), _dp(chi2), _dp(sig)):
            return None
        return dict(cov=cov.T.copy(), reduced_chi2=float(chi2[0]), conf_sigma=sig, dof=self.m - k)

    def counters(self):
        st = self._struct()
        return int(st.n_set_params), int(st.n_jacobians)


def eval_phi(model, t, alpha):
    d = desc_of(model)
    t = np.ascontiguousarray(t, dtype=np.float64)
    a = np.ascontiguousarray(alpha, dtype=np.float64)
    Phi = np.empty((d.n_basis, t.size))
    lib().vpo_eval_phi(C.byref(d), t.size, _dp(t), _dp(a), _dp(Phi))
    return Phi  # [j][i]


def eval_dphi(model, t, alpha, k):
    d = desc_of(model)
    t = np.ascontiguousarray(t, dtype=np.float64)
    a = np.ascontiguousarray(alpha, dtype=np.float64)
    D = np.empty((d.n_basis, t.size))
    lib().vpo_eval_dphi(C.byref(d), t.size, _dp(t), _dp(a), int(k), _dp(D))
    return D


def fit_batch(model, t, Y, alpha0, w=None, eps=-1.0, opts=None, n_threads=1):
    """returns (alpha[B,q], C[B,n], report[B] structured array, seconds inside the fits)"""
    d = desc_of(model)
    t = np.ascontiguousarray(t, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    B, m = Y.shape
    alpha = np.array(alpha0, dtype=np.float64, order="C", copy=True).reshape(B, d.n_params)
    Cout = np.empty((B, d.n_basis))
    rep = np.zeros(B, dtype=REPORT_DTYPE)
    opts = opts or default_opts()
    w_ = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    secs = lib().vpo_fit_batch(C.byref(d), m, B, _dp(t), _dp(Y), _dp(w_), float(eps), C.byref(opts), _dp(alpha),
                               _dp(Cout), rep.ctypes.data_as(C.c_void_p), int(n_threads))
    return alpha, Cout, rep, secs


def evaluate_batch(model, t, Y, alpha, w=None, eps=-1.0, n_threads=1, want_jac=True):
    """returns dict(r[B,m], J[B,q,m], C[B,n], cost[B], status[B])"""
    d = desc_of(model)
    t = np.ascontiguousarray(t, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    B, m = Y.shape
    a = np.ascontiguousarray(alpha, dtype=np.float64).reshape(B, d.n_params)
    r = np.empty((B, m))
    J = np.empty((B, d.n_params, m)) if want_jac else None
    Cc = np.empty((B, d.n_basis))
    cost = np.empty(B)
    st = np.zeros(B, dtype=np.int32)
    w_ = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    lib().vpo_evaluate_batch(C.byref(d), m, B, _dp(t), _dp(Y), _dp(w_), float(eps), _dp(a), _dp(r), _dp(J), _dp(Cc),
                             _dp(cost), st.ctypes.data_as(C.POINTER(C.c_int32)), int(n_threads))
    return dict(r=r, J=J, C=Cc, cost=cost, status=st)


def lmpar(R, ipvt, diag, qtb, delta, par):
    """MINPACK lmpar of the restatement on an upper-triangular factor R (n x n) -> (par, step, ||diag * step||)"""
    R = np.asarray(R, dtype=np.float64)
    n = R.shape[0]
    rc = np.asfortranarray(R).ravel(order="F").copy()
    ip = np.ascontiguousarray(ipvt, dtype=np.int32)
    dg = np.ascontiguousarray(diag, dtype=np.float64)
    qb = np.ascontiguousarray(qtb, dtype=np.float64)
    x = np.empty(n)
    dx = C.c_double(0.0)
    f = lib().vpo_lmpar
    f.restype = C.c_double
    f.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                  C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    out = f(n, _dp(rc), ip.ctypes.data_as(C.POINTER(C.c_int32)), _dp(dg), _dp(qb), float(delta), float(par), _dp(x), C.byref(dx))
    return float(out), x, float(dx.value)


def max_threads():
    return lib().vpo_max_threads()


# ---- the fp32-storage build of the same restatement (oracle/Makefile: libvarpro_oracle_f32.so) -------------------------
# Yardstick for the fp32 device path (BASELINE configs[4]): "what does the reference ALGORITHM do in single precision on
# this problem" -- same entry points, float buffers.  Slightly optimistic (mixed expressions are evaluated in double
# before rounding to float), see the Makefile.
_lib32 = None


def lib_f32():
    global _lib32
    if _lib32 is None:
        path = os.path.join(_HERE, "libvarpro_oracle_f32.so")
        if not os.path.exists(path):
            build()
        l32 = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        l32.vpo_fit_batch.restype = C.c_float
        l32.vpo_fit_batch.argtypes = [C.POINTER(ModelDesc), C.c_int, C.c_int64, fp, fp, fp, C.c_float, C.POINTER(LmOpts), fp,
                                      fp, C.c_void_p, C.c_int]
        l32.vpo_evaluate_batch.argtypes = [C.POINTER(ModelDesc), C.c_int, C.c_int64, fp, fp, fp, C.c_float, fp, fp, fp, fp,
                                           fp, C.POINTER(C.c_int32), C.c_int]
        l32.vpo_lm_opts_default.argtypes = [C.POINTER(LmOpts)]
        _lib32 = l32
    return _lib32


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def default_opts_f32(**kw):
    o = LmOpts()
    lib_f32().vpo_lm_opts_default(C.byref(o))  # 30 * FLT_EPSILON tolerances
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def fit_batch_f32(model, t, Y, alpha0, w=None, eps=-1.0, opts=None, n_threads=1):
    """fp32-storage oracle: returns (alpha[B,q] float32, C[B,n] float32, report[B], seconds inside the fits)"""
    d = desc_of(model)
    t = np.ascontiguousarray(t, dtype=np.float32)
    Y = np.ascontiguousarray(Y, dtype=np.float32)
    B, m = Y.shape
    alpha = np.array(alpha0, dtype=np.float32, order="C", copy=True).reshape(B, d.n_params)
    Cout = np.empty((B, d.n_basis), dtype=np.float32)
    rep = np.zeros(B, dtype=REPORT_DTYPE)
    opts = opts or default_opts_f32()
    w_ = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    secs = lib_f32().vpo_fit_batch(C.byref(d), m, B, _fp(t), _fp(Y), _fp(w_), float(eps), C.byref(opts), _fp(alpha),
                                   _fp(Cout), rep.ctypes.data_as(C.c_void_p), int(n_threads))
    return alpha, Cout, rep, float(secs)
