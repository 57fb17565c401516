"""CPU tier: the C-ABI library loads and exports every symbol include/varpro_hip.h declares, the
host-side mirror validates like the reference's builders, and nothing computes without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import varpro_amd as vp
from varpro_amd import _lib, basis

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = vp.load_library()
    header = open(os.path.join(ROOT, "include", "varpro_hip.h")).read()
    declared = set(re.findall(r"\b(vp_[a-z0-9_]+)\s*\(", header))
    declared -= {"vp_model_desc", "vp_lm_opts", "vp_report", "vp_batch"}
    assert declared, "no declarations found"
    assert not any(name.startswith("vp_debug") for name in declared)  # test hooks live in varpro_hip_debug.h
    debug = set(re.findall(r"\b(vp_debug_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "varpro_hip_debug.h")).read()))
    assert debug == set(_lib.DEBUG_SYMBOLS)
    assert declared == set(_lib.ABI_SYMBOLS)
    declared |= debug
    for name in sorted(declared):
        assert hasattr(lib, name), "libvarpro_hip.so does not export %s" % name
    assert b"gfx950" in lib.vp_version()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.ModelDesc) == 4 * (2 + 8 + 16)
    assert ctypes.sizeof(_lib.LmOpts) == 4 * 8 + 8
    assert ctypes.sizeof(_lib.Report) == 16
    assert vp.REPORT_DTYPE.itemsize == 16


def test_lm_defaults_match_crate_defaults():
    lib = vp.load_library()
    o = _lib.LmOpts()
    lib.vp_lm_opts_default(ctypes.byref(o), _lib.VP_F64)
    eps = np.finfo(np.float64).eps
    assert (o.ftol, o.xtol, o.gtol) == (30 * eps, 30 * eps, 30 * eps)
    assert o.stepbound == 100.0 and o.patience == 100 and o.scale_diag == 1
    py = vp.LevenbergMarquardt()
    assert (py.ftol, py.stepbound, py.patience) == (o.ftol, 100.0, 100)


def test_model_builder_errors_mirror_reference():
    # src/model/builder/test.rs: error variants of ModelBuildError
    x = np.linspace(0, 1, 8)
    with pytest.raises(vp.ModelBuildError) as e:
        vp.SeparableModelBuilder(["a", "a"]).build()
    assert e.value.variant == "DuplicateParameterNames"
    with pytest.raises(vp.ModelBuildError) as e:
        vp.SeparableModelBuilder([]).build()
    assert e.value.variant == "EmptyParameters"
    with pytest.raises(vp.ModelBuildError) as e:
        vp.SeparableModelBuilder(["a"]).function(["b"], basis.EXP_DECAY).build()
    assert e.value.variant == "FunctionParameterNotInModel"
    with pytest.raises(vp.ModelBuildError) as e:
        (vp.SeparableModelBuilder(["a"]).function(["a"], basis.EXP_DECAY).independent_variable(x)
         .initial_parameters([1.0]).build())
    assert e.value.variant == "MissingDerivative"
    with pytest.raises(vp.ModelBuildError) as e:
        vp.SeparableModelBuilder(["a"]).partial_deriv("a").build()
    assert e.value.variant == "IllegalCallToPartialDeriv"
    with pytest.raises(vp.ModelBuildError) as e:
        (vp.SeparableModelBuilder(["a", "b"]).function(["a"], basis.EXP_DECAY).partial_deriv("a")
         .independent_variable(x).initial_parameters([1.0, 2.0]).build())
    assert e.value.variant == "UnusedParameter"
    with pytest.raises(vp.ModelBuildError) as e:
        (vp.SeparableModelBuilder(["a"]).function(["a"], basis.EXP_DECAY).partial_deriv("a")
         .initial_parameters([1.0]).build())
    assert e.value.variant == "MissingX"
    with pytest.raises(vp.ModelBuildError) as e:
        vp.SeparableModelBuilder(["a"]).function(["a"], basis.EXP_DECAY).partial_deriv("a").partial_deriv("a").build()
    assert e.value.variant == "DuplicateDerivative"
    with pytest.raises(vp.ModelBuildError) as e:
        vp.SeparableModelBuilder(["a"]).function(["a"], basis.EXP_COS).build()
    assert e.value.variant == "IncorrectParameterCount"
    m = vp.multi_exponential_model(x, [1.0, 2.0])
    assert (m.parameter_count(), m.base_function_count(), m.output_len()) == (2, 3, 8)
    assert m.pairs == [(0, 0, 0), (1, 0, 1)]
    m.set_params([3.0, 4.0])
    assert np.array_equal(m.params(), [3.0, 4.0])


def test_problem_builder_errors_mirror_reference():
    # src/problem/builder.rs:15-46 / src/problem/builder/test.rs
    x = np.linspace(0, 1, 8)
    m = vp.multi_exponential_model(x, [1.0, 2.0])
    with pytest.raises(vp.SeparableProblemBuilderError) as e:
        vp.SeparableProblemBuilder(m).build()
    assert e.value.variant == "YDataMissing"
    with pytest.raises(vp.SeparableProblemBuilderError) as e:
        vp.SeparableProblemBuilder(m).observations(np.ones(7)).build()
    assert e.value.variant == "InvalidLengthOfData"
    with pytest.raises(vp.SeparableProblemBuilderError) as e:
        vp.SeparableProblemBuilder(m).observations(np.ones(8)).weights(np.ones(3)).build()
    assert e.value.variant == "InvalidLengthOfWeights"
    with pytest.raises(vp.SeparableProblemBuilderError) as e:
        vp.SeparableProblemBuilder(m).observations(np.ones(0)).build()
    assert e.value.variant in ("ZeroLengthVector", "InvalidLengthOfData")


def test_no_cpu_fallback_without_device():
    """with no GPU visible every compute entry point must fail loudly, never compute on the CPU"""
    if vp.device_count() > 0:
        pytest.skip("a GPU is visible")
    x = np.linspace(0, 1, 8)
    m = vp.multi_exponential_model(x, [1.0, 2.0])
    with pytest.raises(vp.VarproHipError) as e:
        vp.BatchProblem(m, np.ones((2, 8)))
    assert e.value.code == _lib.VP_ERR_NO_DEVICE
    with pytest.raises(vp.VarproHipError):
        vp.SeparableProblemBuilder(m).observations(np.ones(8)).build()
    with pytest.raises(vp.VarproHipError):
        m.eval()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "varpro_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), "%s mentions the oracle" % f


def test_synthetic_generators_are_deterministic():
    from varpro_amd import synth
    a = synth.double_exp_batch(8, m=64)
    b = synth.double_exp_batch(4, m=64, first_problem=4)
    assert np.array_equal(a["Y"][4:], b["Y"]) and np.array_equal(a["tau_guess"][4:], b["tau_guess"])
    x = synth.linspace_reference(0., 12.5, 1024)
    assert x[0] == 0.0 and abs(x[-1] + 12.5) < 1e-12  # the reference's sign quirk
    c0 = synth.config0()
    assert abs(c0["y"][0] - 7.5) < 1e-12 and c0["y"].max() > 1e6
