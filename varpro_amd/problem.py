"""Host-side mirror of ``SeparableProblemBuilder`` / ``SeparableProblem`` (single problem, single or
multiple right-hand sides) on top of the GPU batch handle.

Reference: src/problem/builder.rs:116-324 (builder + its 5 error variants :15-46),
src/problem.rs:57-212 (problem state and accessors), src/solvers/levmar/mod.rs:22-202
(``impl LeastSquaresProblem for SeparableProblem``).  Method names, argument meaning and the
``None``-on-failure convention are the reference's; matrices are returned in the reference's shapes
(m x S, n x S, (m*S) x q).
"""
import numpy as np

from .batch import BatchProblem, STATUS_OK


class SeparableProblemBuilderError(ValueError):
    """mirrors ``SeparableProblemBuilderError`` (src/problem/builder.rs:15-46)"""

    def __init__(self, variant, message):
        super().__init__("%s: %s" % (variant, message))
        self.variant = variant


class SeparableProblem:
    """One separable least-squares problem living on the GPU (B = 1 batch handle)."""

    def __init__(self, model, Y, weights, epsilon, mrhs):
        self._model = model
        self._mrhs = mrhs
        self._weights = weights
        Yb = Y.T[None, :, :] if mrhs else Y[None, :]  # (1, S, m) / (1, m)
        self._batch = BatchProblem(model, np.ascontiguousarray(Yb), x=model.x, weights=weights, epsilon=epsilon)
        self.m, self.S, self.n, self.q = self._batch.m, self._batch.S, self._batch.n, self._batch.q
        # build(): initial set_params with the model's current parameters (src/problem/builder.rs:321)
        self.set_params(model.params())

    # -- LeastSquaresProblem (src/solvers/levmar/mod.rs:42-201) --
    def set_params(self, params):
        p = np.ascontiguousarray(params, dtype=self._model.dtype).reshape(-1)
        self._model.set_params(p)
        self._batch.set_params(p.reshape(1, -1))

    def params(self):
        return self._model.params()

    def _ok(self):
        return int(self._batch.status()[0]) == STATUS_OK

    def residuals(self):
        """vector of length m*S (column-stacked), or None if the last set_params failed"""
        r, st = self._batch.residuals(with_status=True)
        return r[0].copy() if int(st[0]) == STATUS_OK else None

    def jacobian(self):
        """(m*S) x q matrix, or None"""
        J, st = self._batch.jacobian(with_status=True)
        return np.ascontiguousarray(J[0].T) if int(st[0]) == STATUS_OK else None

    # -- accessors (src/problem.rs:142-212) --
    def linear_coefficients(self):
        """vector n (single RHS) or n x S matrix (MRHS); None if the last evaluation failed"""
        if not self._ok():
            return None
        c = self._batch.linear_coefficients()
        return np.ascontiguousarray(c[0].T) if self._mrhs else c[0].copy()

    def weighted_data(self):
        yw = self._batch.weighted_data()
        return np.ascontiguousarray(yw[0].T) if self._mrhs else yw[0].copy()

    def model(self):
        return self._model

    def weights(self):
        return self._weights

    def close(self):
        self._batch.close()


class SeparableProblemBuilder:
    """mirrors ``SeparableProblemBuilder::{new, mrhs, observations, weights, epsilon, build}``"""

    def __init__(self, model, _mrhs=False):
        self._model = model
        self._mrhs = _mrhs
        self._Y = None
        self._weights = None
        self._epsilon = None

    @classmethod
    def new(cls, model):
        return cls(model, False)

    @classmethod
    def mrhs(cls, model):
        return cls(model, True)

    def observations(self, Y):
        self._Y = np.asarray(Y, dtype=self._model.dtype)
        return self

    def weights(self, w):
        self._weights = np.asarray(w, dtype=self._model.dtype).reshape(-1)
        return self

    def epsilon(self, eps):
        self._epsilon = abs(float(eps))  # src/problem/builder.rs:246-251
        return self

    def build(self):
        if self._Y is None:
            raise SeparableProblemBuilderError("YDataMissing", "Right hand side(s) not provided")
        Y = self._Y
        if self._mrhs:
            if Y.ndim != 2:
                raise SeparableProblemBuilderError("InvalidLengthOfData", "MRHS observations must be an m x S matrix")
        else:
            Y = Y.reshape(-1)
        x_len = self._model.output_len()
        if x_len == 0 or Y.size == 0:
            raise SeparableProblemBuilderError("ZeroLengthVector", "x or y must have nonzero number of elements.")
        if x_len != Y.shape[0]:
            raise SeparableProblemBuilderError(
                "InvalidLengthOfData",
                "Vectors x and y must have same lengths. Given x length = %d and y length = %d" % (x_len, Y.shape[0]))
        if self._weights is not None and self._weights.size != Y.shape[0]:
            raise SeparableProblemBuilderError("InvalidLengthOfWeights",
                                               "The weights must have the same length as the data y.")
        return SeparableProblem(self._model, Y, self._weights, self._epsilon, self._mrhs)
