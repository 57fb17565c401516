"""Host-side mirror of the reference's model layer for the GPU path.

Reference: ``SeparableModelBuilder`` (src/model/builder/mod.rs:252-272, methods :338-525, validity
check :535-571), ``SeparableModel`` (src/model/mod.rs:367-517) and the plugin trait
``SeparableNonlinearModel`` (src/model/mod.rs:239-363).

The reference hands the solver opaque Rust closures; those cannot run on a GPU.  The
drop-in replaces "closure" by "basis kind" from a closed descriptor language (``basis.*``), keeping
the builder's call sequence, argument meaning and error variants: ``function(params, kind)`` /
``partial_deriv(param)`` / ``invariant_function(kind)`` / ``independent_variable(x)`` /
``initial_parameters(p)`` / ``build()``.  Derivatives are built into each kind; ``partial_deriv``
only declares (and validates) that the derivative is provided, exactly where the reference takes
the derivative closure.
"""
import numpy as np

from . import _lib


class basis:
    """basis-function kinds (include/varpro_hip.h): f(t, p0[, p1])"""
    CONST = 0       # 1                     (invariant_function)
    EXP_DECAY = 1   # exp(-t/p0)            shared_test_code/src/lib.rs:101-114
    EXP_RATE = 2    # exp(-p0 t)
    EXP_COS = 3     # exp(-p0 t) cos(p1 t)  shared_test_code/src/models.rs:310-372
    SIN_PHASE = 4   # sin(p0 t + p1)        src/test_helpers/mod.rs:28-52
    ARITY = {CONST: 0, EXP_DECAY: 1, EXP_RATE: 1, EXP_COS: 2, SIN_PHASE: 2}
    NAME = {CONST: "const", EXP_DECAY: "exp_decay", EXP_RATE: "exp_rate", EXP_COS: "exp_cos", SIN_PHASE: "sin_phase"}


class ModelBuildError(ValueError):
    """mirrors varpro::model::builder::error::ModelBuildError (src/model/builder/error.rs)"""

    def __init__(self, variant, message):
        super().__init__("%s: %s" % (variant, message))
        self.variant = variant


class ModelError(RuntimeError):
    """mirrors varpro::model::errors::ModelError (derivative index out of bounds etc.)"""


class SeparableModel:
    """A separable model  f(x, alpha, c) = sum_j c_j phi_j(x, alpha)  described by basis kinds.

    Implements the ``SeparableNonlinearModel`` surface: ``parameter_count``, ``base_function_count``,
    ``output_len``, ``set_params``, ``params``, ``eval``, ``eval_partial_deriv`` (the last two run
    the stand-alone Phi kernel, ``vp_basis``).
    """

    def __init__(self, parameter_names, kinds, params, x, initial_parameters, dtype=np.float64):
        self.parameter_names = list(parameter_names)
        self.kinds = [int(k) for k in kinds]
        self.param_indices = [tuple(int(i) for i in p) for p in params]
        self.n_params = len(self.parameter_names)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError("ScalarType must be float64 or float32")
        self.x = np.ascontiguousarray(x, dtype=self.dtype).reshape(-1)
        self._alpha = np.ascontiguousarray(initial_parameters, dtype=self.dtype).reshape(-1)
        # dependency pairs in model order (basis-major): the layout of vp_basis' dPhi output
        self.pairs = [(j, a, pi) for j, ps in enumerate(self.param_indices) for a, pi in enumerate(ps)]

    # -- SeparableNonlinearModel surface (src/model/mod.rs:256-271) --
    def parameter_count(self):
        return self.n_params

    def base_function_count(self):
        return len(self.kinds)

    def output_len(self):
        return int(self.x.size)

    def set_params(self, parameters):
        p = np.ascontiguousarray(parameters, dtype=self.dtype).reshape(-1)
        if p.size != self.n_params:
            raise ModelError("IncorrectParameterCount: expected %d, got %d" % (self.n_params, p.size))
        self._alpha = p.copy()

    def params(self):
        return self._alpha.copy()

    def desc(self):
        d = _lib.ModelDesc()
        d.n_basis = len(self.kinds)
        d.n_params = self.n_params
        for j in range(_lib.VP_MAX_BASIS):
            d.kind[j] = 0
            for a in range(_lib.VP_MAX_BASIS_PARAMS):
                d.param[j][a] = -1
        for j, (k, ps) in enumerate(zip(self.kinds, self.param_indices)):
            d.kind[j] = k
            for a, pi in enumerate(ps):
                d.param[j][a] = pi
        return d

    def _basis(self):
        from .batch import BatchProblem
        # a data-free handle is enough for the Phi kernel; y is a dummy column
        bp = BatchProblem(self, np.zeros((1, self.x.size), dtype=self.dtype), x=self.x)
        return bp.basis(self._alpha.reshape(1, -1))

    def eval(self):
        """Phi(alpha): (m, n) matrix, column j = basis j  (src/model/mod.rs:308)"""
        phi, _ = self._basis()
        return np.ascontiguousarray(phi[0].T)

    def eval_partial_deriv(self, derivative_index):
        """dPhi/dalpha_k: (m, n), zero columns for independent basis functions (src/model/mod.rs:359-362)"""
        k = int(derivative_index)
        if k < 0 or k >= self.n_params:
            raise ModelError("DerivativeIndexOutOfBounds: %d" % k)
        _, dphi = self._basis()
        out = np.zeros((self.x.size, len(self.kinds)), dtype=self.dtype)
        for p, (j, _a, pi) in enumerate(self.pairs):
            if pi == k:
                out[:, j] += dphi[0, p]
        return out


class ExternalModel:
    """Shape of a model the CALLER evaluates: any ``SeparableNonlinearModel`` (src/model/mod.rs:239-363), e.g. the
    reference's closure-based ``SeparableModel`` (src/model/mod.rs:441-512), whose basis functions the closed descriptor
    language above cannot express.  Only what the device needs: n basis functions, q parameters and the
    (basis j, parameter k) pairs with a non-zero derivative -- the columns ``eval_partial_deriv(k)`` fills
    (src/model/mod.rs:473-512).  ``BatchProblem(ExternalModel(...), Y)`` makes a handle with
    ``vp_batch_create_external``; Phi / dPhi then enter through ``set_params_with_basis`` /
    ``jacobian_with_derivatives`` / ``evaluate_with_basis``.  ``ClosureModel`` builds one from Python callables."""

    def __init__(self, n_basis, n_params, pairs, dtype=np.float64):
        self.n_basis, self.n_params = int(n_basis), int(n_params)
        self.ext_pairs = [(int(j), int(k)) for j, k in pairs]
        self.pairs = self.ext_pairs  # len() == number of derivative columns
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError("ScalarType must be float64 or float32")
        self.x = None

    def parameter_count(self):
        return self.n_params

    def base_function_count(self):
        return self.n_basis


class ClosureModel:
    """== the reference's closure-based ``SeparableModel`` (src/model/mod.rs:367-517, built by
    ``SeparableModelBuilder::function(params, f).partial_deriv(param, df)``, src/model/builder/mod.rs:338-525) with REAL
    Python callables: ``f(x, *params) -> (m,)`` evaluated on the host with numpy for a whole batch of parameter sets.
    ``eval_batch(alpha (B, q)) -> Phi (B, n, m)`` and ``derivs_batch(alpha) -> dPhi (B, p, m)`` produce exactly what
    ``BatchProblem.set_params_with_basis`` takes; ``shape()`` is the matching ``ExternalModel``."""

    def __init__(self, parameter_names, x, dtype=np.float64):
        self.names = list(parameter_names)
        # the builder's checks on the model parameter list (src/model/builder/mod.rs:338-370, error.rs variant names)
        if len(self.names) == 0:
            raise ModelBuildError("EmptyParameters", "A function or model parameter list is empty!")
        if len(set(self.names)) != len(self.names):
            raise ModelBuildError("DuplicateParameterNames", "Parameter list %r contains duplicates!" % (self.names,))
        if any("," in n for n in self.names):
            raise ModelBuildError("CommaInParameterNameNotAllowed", "Parameter names may not contain comma separator")
        self.x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        self.dtype = np.dtype(dtype)
        self._functions = []  # (param indices, f, {param index: df})

    def invariant_function(self, f):
        self._functions.append(((), f, {}))
        return self

    def function(self, function_params, f):
        fp = list(function_params)
        if len(fp) == 0:
            raise ModelBuildError("EmptyParameters", "A function or model parameter list is empty!")
        if len(set(fp)) != len(fp):
            raise ModelBuildError("DuplicateParameterNames", "Parameter list %r contains duplicates!" % (fp,))
        for p in fp:
            if p not in self.names:
                raise ModelBuildError("FunctionParameterNotInModel",
                                      "Function parameter '%s' is not part of the model parameters." % p)
        idx = tuple(self.names.index(p) for p in fp)
        self._functions.append((idx, f, {}))
        return self

    def partial_deriv(self, parameter, df):
        if not self._functions or not self._functions[-1][0]:
            # (the reference's builder has no partial_deriv in the state that follows an invariant function: a type error there)
            raise ModelBuildError("InvalidDerivative", "partial_deriv needs a preceding function with parameters")
        idx, _f, derivs = self._functions[-1]
        if parameter not in self.names:
            raise ModelBuildError("InvalidDerivative", "Parameter '%s' is not in the function's parameter list" % parameter)
        k = self.names.index(parameter)
        if k not in idx:
            raise ModelBuildError("InvalidDerivative", "Parameter '%s' is not in the function's parameter list" % parameter)
        if k in derivs:
            raise ModelBuildError("DuplicateDerivative", "Derivative for parameter '%s' was already provided!" % parameter)
        derivs[k] = df
        return self

    def pairs(self):
        return [(j, k) for j, (idx, _f, _d) in enumerate(self._functions) for k in idx]

    def shape(self):
        # == SeparableModelBuilder::build (src/model/builder/mod.rs:527-553): EmptyModel, MissingDerivative, UnusedParameter
        if not self._functions:
            raise ModelBuildError("EmptyModel", "A model must contain at least one function")
        for idx, _f, derivs in self._functions:
            for k in idx:
                if k not in derivs:
                    raise ModelBuildError("MissingDerivative", "missing derivative for parameter '%s'" % self.names[k])
        used = {k for idx, _f, _d in self._functions for k in idx}
        for k, name in enumerate(self.names):
            if k not in used:
                raise ModelBuildError("UnusedParameter", "Parameter '%s' is not used by any function of the model" % name)
        return ExternalModel(len(self._functions), len(self.names), self.pairs(), dtype=self.dtype)

    def eval_batch(self, alpha):
        """== eval() (src/model/mod.rs:441-471) for every parameter set of the batch: (B, n, m), UNWEIGHTED"""
        a = np.asarray(alpha, dtype=np.float64).reshape(-1, len(self.names))
        out = np.empty((a.shape[0], len(self._functions), self.x.size), dtype=self.dtype)
        for b in range(a.shape[0]):
            for j, (idx, f, _d) in enumerate(self._functions):
                out[b, j] = f(self.x, *[a[b, k] for k in idx])
        return out

    def derivs_batch(self, alpha):
        """== the non-zero columns of eval_partial_deriv(k) (src/model/mod.rs:473-512) in pair order: (B, p, m)"""
        a = np.asarray(alpha, dtype=np.float64).reshape(-1, len(self.names))
        prs = self.pairs()
        out = np.empty((a.shape[0], len(prs), self.x.size), dtype=self.dtype)
        for b in range(a.shape[0]):
            for p, (j, k) in enumerate(prs):
                idx, _f, derivs = self._functions[j]
                out[b, p] = derivs[k](self.x, *[a[b, kk] for kk in idx])
        return out


class SeparableModelBuilder:
    """mirrors ``SeparableModelBuilder`` (src/model/builder/mod.rs:338-525)"""

    def __init__(self, parameter_names, dtype=np.float64):
        self._error = None
        self._names = list(parameter_names)
        self._dtype = dtype
        self._functions = []  # dicts: params(list of names), kind, derivs(set)
        self._x = None
        self._initial = None
        self._building = False  # state FunctionBuilding
        if len(self._names) == 0:
            self._fail("EmptyParameters", "A function or model parameter list is empty!")
        elif len(set(self._names)) != len(self._names):
            self._fail("DuplicateParameterNames", "Parameter list %r contains duplicates!" % (self._names,))
        elif any("," in n for n in self._names):
            self._fail("CommaInParameterNameNotAllowed", "Parameter names may not contain comma separator")

    @classmethod
    def new(cls, parameter_names, dtype=np.float64):
        return cls(parameter_names, dtype)

    def _fail(self, variant, msg):
        if self._error is None:
            self._error = ModelBuildError(variant, msg)
        return self

    def invariant_function(self, kind=basis.CONST):
        if self._error:
            return self
        if basis.ARITY.get(kind) != 0:
            return self._fail("IncorrectParameterCount", "invariant function takes no parameters")
        self._functions.append(dict(params=[], kind=kind, derivs=set()))
        self._building = False
        return self

    def function(self, function_params, kind):
        if self._error:
            return self
        fp = list(function_params)
        if len(fp) == 0:
            return self._fail("EmptyParameters", "A function or model parameter list is empty!")
        if len(set(fp)) != len(fp):
            return self._fail("DuplicateParameterNames", "Parameter list %r contains duplicates!" % (fp,))
        for p in fp:
            if p not in self._names:
                return self._fail("FunctionParameterNotInModel",
                                  "Function parameter '%s' is not part of the model parameters." % p)
        if kind not in basis.ARITY:
            return self._fail("IncorrectParameterCount", "unknown basis kind %r" % (kind,))
        if basis.ARITY[kind] != len(fp):
            return self._fail("IncorrectParameterCount", "Incorrect number of parameters for function: expected %d, "
                              "got %d" % (basis.ARITY[kind], len(fp)))
        self._functions.append(dict(params=fp, kind=kind, derivs=set()))
        self._building = True
        return self

    def partial_deriv(self, parameter, _derivative=None):
        if self._error:
            return self
        if not self._building:
            return self._fail("IllegalCallToPartialDeriv", "a call to this function can only follow a call to "
                              "'function' or another call to 'partial_deriv'")
        f = self._functions[-1]
        if parameter not in f["params"]:
            return self._fail("InvalidDerivative", "Parameter '%s' given for partial derivative does not exist in "
                              "parameter list '%r'." % (parameter, f["params"]))
        if parameter in f["derivs"]:
            return self._fail("DuplicateDerivative", "Derivative for parameter '%s' was already provided!" % parameter)
        f["derivs"].add(parameter)
        return self

    def independent_variable(self, x):
        if self._error:
            return self
        self._x = np.asarray(x)
        self._building = False
        return self

    def initial_parameters(self, initial):
        if self._error:
            return self
        init = np.asarray(initial, dtype=np.float64).reshape(-1)
        if init.size != len(self._names):
            return self._fail("IncorrectParameterCount", "Incorrect number of parameters for function: expected %d, "
                              "got %d" % (len(self._names), init.size))
        self._initial = init
        self._building = False
        return self

    def build(self):
        if self._error:
            raise self._error
        if not self._functions:
            raise ModelBuildError("EmptyModel", "Tried to construct model with no functions.")
        for f in self._functions:
            for p in f["params"]:
                if p not in f["derivs"]:
                    raise ModelBuildError("MissingDerivative", "Function with paramter list %r is missing derivative "
                                          "for parametr '%s'." % (f["params"], p))
        used = set(p for f in self._functions for p in f["params"])
        for n in self._names:
            if n not in used:
                raise ModelBuildError("UnusedParameter", "Model depends on parameter '%s', but none of its functions "
                                      "use it." % n)
        if self._x is None:
            raise ModelBuildError("MissingX", "Missing vector for independent variable x")
        if self._initial is None:
            raise ModelBuildError("MissingInitialParameters", "Missing initial guesses for model parameters")
        if len(self._functions) > _lib.VP_MAX_BASIS or len(self._names) > _lib.VP_MAX_PARAMS:
            raise ModelBuildError("IncorrectParameterCount", "model exceeds VP_MAX_BASIS/VP_MAX_PARAMS")
        kinds = [f["kind"] for f in self._functions]
        params = [tuple(self._names.index(p) for p in f["params"]) for f in self._functions]
        return SeparableModel(self._names, kinds, params, self._x, self._initial, dtype=self._dtype)


def multi_exponential_model(x, initial_taus, offset=True, dtype=np.float64):
    """sum of exponential decays exp(-x/tau_j) (+ constant): the model family of every bench in the
    reference (shared_test_code/src/lib.rs:119-135, shared_test_code/src/models.rs:16-156)."""
    taus = list(np.asarray(initial_taus, dtype=np.float64).reshape(-1))
    names = ["tau%d" % (i + 1) for i in range(len(taus))]
    b = SeparableModelBuilder(names, dtype=dtype)
    for nme in names:
        b = b.function([nme], basis.EXP_DECAY).partial_deriv(nme)
    if offset:
        b = b.invariant_function(basis.CONST)
    return b.independent_variable(x).initial_parameters(taus).build()
