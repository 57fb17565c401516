"""The persistent slot kernel (vp_fit2.hpp: a wavefront owns several problems, lane-parallel LM bookkeeping, problems
handed out from a device-side queue) against the one-wavefront-per-problem kernel and against the CPU oracle.
Both kernels implement LevenbergMarquardt::minimize (src/solvers/levmar/mod.rs:247) with the same arithmetic per
problem, so their reports must agree exactly in the termination code and evaluation count and to rounding in the
numbers."""
import numpy as np
import pytest

import varpro_amd as vp
from models import double_exp_builder_model
from oracle import oracle as O
from test_gpu_parity import _check_fit
from varpro_amd import synth

pytestmark = pytest.mark.gpu


def _fit_with(kernel, mdl, Y, x, guess, trace_rows=0):
    bp = vp.BatchProblem(mdl, Y, x=x)
    bp.set_fit_kernel(kernel)
    if trace_rows:
        out = bp.fit_trace(guess, max_rows=trace_rows)
    else:
        out = bp.fit(guess)
    bp.close()
    return out


# m -> rows per lane R -> slots per wavefront: 1024 -> 2, 1000 -> 2 (padding in the last register pair),
# 512 -> 4, 300 -> 4 (general padding), 100 -> 8; 2048 / 2000 -> R = 32 at one wave per SIMD, 2 slots; 4096 / 4000 ->
# a group of four waves per problem (scalar phase on wave 0 only)
@pytest.mark.parametrize("m,B", [(1024, 301), (1000, 77), (512, 203), (300, 64), (100, 517), (2048, 45), (2000, 33),
                                 (4096, 37), (4000, 21)])
def test_slot_kernel_equals_wave_kernel(m, B):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    a0, c0, r0, t0 = _fit_with("wave", mdl, d["Y"], d["x"], d["tau_guess"], trace_rows=8)
    a1, c1, r1, t1 = _fit_with("slots", mdl, d["Y"], d["x"], d["tau_guess"], trace_rows=8)
    assert np.array_equal(r0["termination"], r1["termination"])
    assert np.array_equal(r0["n_evals"], r1["n_evals"])
    assert np.array_equal(r0["objective"], r1["objective"], equal_nan=True)
    assert np.array_equal(a0, a1)
    assert np.array_equal(np.isnan(c0), np.isnan(c1))
    assert np.nanmax(np.abs(c0 - c1)) <= 1e-13 * np.nanmax(np.abs(c0)), np.nanmax(np.abs(c0 - c1))
    assert np.array_equal(t0, t1, equal_nan=True)


def test_slot_kernel_queue_refill_and_order_independence():
    # more problems than the persistent grid holds at once (256 CUs x 8 wavefronts x 2 slots = 4096): finished slots
    # pull the rest from the queue; a problem's result must not depend on which wavefront / slot / time it ran
    B, m = 6000, 1024
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    a0, c0, r0 = _fit_with("wave", mdl, d["Y"], d["x"], d["tau_guess"])
    a1, c1, r1 = _fit_with("slots", mdl, d["Y"], d["x"], d["tau_guess"])
    assert np.array_equal(r0["termination"], r1["termination"]) and np.array_equal(r0["n_evals"], r1["n_evals"])
    assert np.array_equal(a0, a1) and np.array_equal(c0, c1, equal_nan=True)
    assert np.array_equal(r0["objective"], r1["objective"], equal_nan=True)
    perm = np.random.default_rng(3).permutation(B)
    a2, c2, r2 = _fit_with("slots", mdl, d["Y"][perm], d["x"], d["tau_guess"][perm])
    assert np.array_equal(a2, a1[perm]) and np.array_equal(c2, c1[perm], equal_nan=True)
    assert np.array_equal(r2["n_evals"], r1["n_evals"][perm])
    # the automatic choice picks the slot kernel at this size; same numbers again
    a3, c3, r3 = _fit_with("auto", mdl, d["Y"], d["x"], d["tau_guess"])
    assert np.array_equal(a3, a1) and np.array_equal(r3["n_evals"], r1["n_evals"])


def test_slot_kernel_short_problems_queue_refill():
    # m = 100: 8 slots per wavefront, the persistent grid holds 16384 problems
    B, m = 20000, 100
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    a0, c0, r0 = _fit_with("wave", mdl, d["Y"], d["x"], d["tau_guess"])
    a1, c1, r1 = _fit_with("slots", mdl, d["Y"], d["x"], d["tau_guess"])
    assert np.array_equal(r0["termination"], r1["termination"]) and np.array_equal(r0["n_evals"], r1["n_evals"])
    assert np.array_equal(a0, a1) and np.array_equal(c0, c1, equal_nan=True)


@pytest.mark.parametrize("m,noise", [(1024, 1e-3), (1024, 0.0), (1000, 1e-3), (256, 1e-3)])
def test_slot_kernel_matches_oracle(m, noise, monkeypatch):
    # the full oracle-parity check of test_gpu_parity (trajectory, success flags, minimum, handle state) with the
    # slot kernel forced for every handle
    orig = vp.BatchProblem.__init__

    def patched(self, *a, **k):
        orig(self, *a, **k)
        self.set_fit_kernel("slots")

    monkeypatch.setattr(vp.BatchProblem, "__init__", patched)
    d = synth.double_exp_batch(48, m=m, noise=noise)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    _check_fit(mdl, d["x"], d["Y"], d["tau_guess"], noise_free=(noise == 0.0))


def test_slot_kernel_failed_evaluations_and_fallbacks():
    # exp(-t/0): non-finite basis column -> TerminationReason::User on both kernels; weighted problems and
    # per-problem grids are outside the slot kernel's coverage and silently run one wavefront per problem
    d = synth.double_exp_batch(40, m=1024, noise=1e-3)
    mdl = double_exp_builder_model(d["x"], d["tau_guess"][0])
    g = d["tau_guess"].copy()
    g[5] = [0.0, 3.0]
    g[17] = [np.nan, 3.0]
    a0, c0, r0 = _fit_with("wave", mdl, d["Y"], d["x"], g)
    a1, c1, r1 = _fit_with("slots", mdl, d["Y"], d["x"], g)
    assert r1["termination"][5] == -1 and r1["termination"][17] == -1
    assert np.array_equal(r0["termination"], r1["termination"]) and np.array_equal(r0["n_evals"], r1["n_evals"])
    assert np.array_equal(a0, a1, equal_nan=True) and np.array_equal(c0, c1, equal_nan=True)
    w = 0.5 + np.random.default_rng(0).random(1024)
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"], weights=w)
    bp.set_fit_kernel("slots")
    aw, cw, rw = bp.fit(d["tau_guess"])
    bp.close()
    _aw, _cw, rw_ref, _ = O.fit_batch(mdl, d["x"], d["Y"], d["tau_guess"], w=w, n_threads=4)
    assert ((rw["termination"] > 0) == (rw_ref["termination"] > 0)).all()
