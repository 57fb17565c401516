"""Deterministic synthetic inputs for the BASELINE.json configurations.

The reference has no batch generator (it fits one problem per call); its bench inputs are
reproduced exactly where they exist (configs[0]: benches/double_exponential_without_noise.rs:97-112
incl. the ``linspace`` sign quirk of shared_test_code/src/lib.rs:20-34) and extended to batches with
a counter-based SplitMix64 stream per problem so that host and device paths, and every rank of a
multi-GPU run, see bit-identical inputs.  Pure numpy input generation -- no solver arithmetic here.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class SplitMix64:
    """vectorised SplitMix64: one independent stream per entry of ``seeds``"""

    def __init__(self, seeds):
        self.state = np.asarray(seeds, dtype=np.uint64).copy()

    def next_u64(self):
        with np.errstate(over="ignore"):
            self.state = self.state + _GOLDEN
            z = self.state.copy()
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            return z ^ (z >> np.uint64(31))

    def uniform(self, lo=0.0, hi=1.0):
        u = (self.next_u64() >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def normal(self, count):
        """(len(seeds), count) standard normals (Box-Muller, two uniforms per pair)"""
        n = self.state.size
        out = np.empty((n, count))
        for k in range(0, count, 2):
            u1 = np.maximum(self.uniform(), 2.0 ** -53)
            u2 = self.uniform()
            r = np.sqrt(-2.0 * np.log(u1))
            out[:, k] = r * np.cos(2.0 * np.pi * u2)
            if k + 1 < count:
                out[:, k + 1] = r * np.sin(2.0 * np.pi * u2)
        return out


def linspace_reference(first, last, count):
    """shared_test_code::linspace incl. its sign quirk: first + (first - last)/(count-1) * n
    (shared_test_code/src/lib.rs:20-34).  linspace_reference(0, 12.5, 1024) runs 0 -> -12.5."""
    n = np.arange(count, dtype=np.float64)
    return first + (first - last) / float(count - 1) * n


def config0():
    """BASELINE configs[0] == bench "Handcrafted Model" (benches/double_exponential_without_noise.rs:97-169):
    returns dict(x, y, tau_true, c_true, tau_guess)."""
    x = linspace_reference(0.0, 12.5, 1024)
    tau = np.array([1.0, 3.0])
    c = np.array([4.0, 2.5, 1.0])
    y = c[0] * np.exp(-x / tau[0]) + c[1] * np.exp(-x / tau[1]) + c[2]
    return dict(x=x, y=y, tau_true=tau, c_true=c, tau_guess=np.array([2.0, 6.5]))


def double_exp_batch(B, m=1024, first_problem=0, noise=1e-3, quirk_grid=False, seed_base=0x5EED0000):
    """BASELINE configs[1]/[3]: B independent double-exponential fits, shared grid.

    Problem b (global index first_problem + b) draws from SplitMix64(seed_base + index):
    tau1~U(0.5,2), tau2~U(2.5,8), c1,c2,c3~U(0,100), y = Phi c + sigma N(0,1) with
    sigma = noise*max|y|, guess = truth*(1+U(-0.3,0.3)) clipped to >= 0.05.
    Returns dict(x (m,), Y (B,m), tau_true (B,2), c_true (B,3), tau_guess (B,2)).
    """
    idx = np.arange(first_problem, first_problem + B, dtype=np.uint64)
    rng = SplitMix64(np.uint64(seed_base) + idx)
    x = linspace_reference(0.0, 12.5, m) if quirk_grid else 12.5 * np.arange(m, dtype=np.float64) / float(m - 1)
    tau = np.stack([rng.uniform(0.5, 2.0), rng.uniform(2.5, 8.0)], axis=1)
    c = np.stack([rng.uniform(0.0, 100.0) for _ in range(3)], axis=1)
    g = np.stack([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3)], axis=1)
    guess = np.maximum(tau * (1.0 + g), 0.05)
    Y = (c[:, 0:1] * np.exp(-x[None, :] / tau[:, 0:1]) + c[:, 1:2] * np.exp(-x[None, :] / tau[:, 1:2]) + c[:, 2:3])
    if noise > 0:
        sigma = noise * np.abs(Y).max(axis=1, keepdims=True)
        Y = Y + sigma * rng.normal(m)
    return dict(x=x, Y=np.ascontiguousarray(Y), tau_true=tau, c_true=c, tau_guess=guess)


def multi_exp_batch(B, n_exp, m, taus, first_problem=0, noise=1e-3, spread=0.1, guess_spread=0.2,
                    seed_base=0x5EED1000, dtype=np.float64):
    """generic multi-exponential batch (configs[4]-style): tau_j = taus[j]*(1+U(-spread,spread)), c~U(1,100)"""
    idx = np.arange(first_problem, first_problem + B, dtype=np.uint64)
    rng = SplitMix64(np.uint64(seed_base) + idx)
    x = 12.5 * np.arange(m, dtype=np.float64) / float(m - 1)
    taus = np.asarray(taus, dtype=np.float64)
    tau = np.stack([taus[j] * (1.0 + rng.uniform(-spread, spread)) for j in range(n_exp)], axis=1)
    c = np.stack([rng.uniform(1.0, 100.0) for _ in range(n_exp + 1)], axis=1)
    guess = np.stack([tau[:, j] * (1.0 + rng.uniform(-guess_spread, guess_spread)) for j in range(n_exp)], axis=1)
    Y = np.tile(c[:, n_exp:n_exp + 1], (1, m))
    for j in range(n_exp):
        Y = Y + c[:, j:j + 1] * np.exp(-x[None, :] / tau[:, j:j + 1])
    if noise > 0:
        sigma = noise * np.abs(Y).max(axis=1, keepdims=True)
        Y = Y + sigma * rng.normal(m)
    return dict(x=x.astype(dtype), Y=np.ascontiguousarray(Y.astype(dtype)), tau_true=tau, c_true=c,
                tau_guess=guess.astype(dtype))


def mrhs_triple_exp(S=16384, m=2048, seed=2314093240213841123):
    """BASELINE configs[2]: one alpha shared by S right-hand sides, triple-exponential + offset.
    C ~ U(0,100)^{4 x S} from SplitMix64(seed) (the literal of benches/multiple_right_hand_sides.rs:67)."""
    x = 12.5 * np.arange(m, dtype=np.float64) / float(m - 1)
    tau = np.array([1.0, 3.0, 7.0])
    rng = SplitMix64(np.uint64(seed) + np.arange(S, dtype=np.uint64))
    Cm = np.stack([rng.uniform(0.0, 100.0) for _ in range(4)], axis=1)  # (S, 4)
    Phi = np.stack([np.exp(-x / tau[0]), np.exp(-x / tau[1]), np.exp(-x / tau[2]), np.ones_like(x)], axis=0)
    Y = Cm @ Phi  # (S, m): RHS s in row s
    return dict(x=x, Y=np.ascontiguousarray(Y), tau_true=tau, C_true=Cm, tau_guess=np.array([1.5, 4.0, 9.0]))
