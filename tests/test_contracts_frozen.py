"""CPU tier: the census contracts (tests/contracts.py) may be tightened, never loosened.  The table
below is the state at the end of round 6; a value that moves in the permissive direction fails here -- changing it means
changing this table and INTEGRATION.md section 8 in the same commit, with the measurement that justifies it."""
import os
import re

import contracts as K

# name -> (frozen value, direction): "max" = an upper bound the device must stay under (loosening = a LARGER value),
# "min" = a lower bound it must reach (loosening = a SMALLER value)
FROZEN = {
    "FP64.success_class_disagreements": (0, "max"),
    "FP64.objective_rel_median_max": (1e-12, "max"),
    "FP64.objective_rel_max_max": (1e-6, "max"),
    "FP64.share_evals_within_3_min": (0.95, "min"),
    "FP64.max_evals_slack": (0.05, "max"),
    "FP64.sum_evals_slack": (0.02, "max"),
    "STREAMED_M10000.max_evals_slack": (0.10, "max"),
    "STREAMED_M10000.sum_evals_slack": (0.03, "max"),
    "OLEARY_M5000.beyond_1e-6_max_problems": (4, "max"),
    "CFG4_ALL.same_success_class_min": (0.95, "min"),
    "CFG4_ALL.failed_device_max_share": (0.03, "max"),
    "CFG4_ALL.failed_oracle_max_share": (0.05, "max"),
    "CFG4_ALL.numerical_failures_max": (2, "max"),
    "CFG4_ALL.objective_rel_median_max": (1e-4, "max"),
    "CFG4_ALL.share_objective_within_1e-3_min": (0.9, "min"),
    "CFG4_ALL.sum_evals_slack": (0.1, "max"),
    "CFG4_ALL.reported_objective_share_above_1e-3_max": (0.003, "max"),
    "CFG4_ALL.reported_objective_share_above_1e-2_max": (0.0005, "max"),
    "CFG4_ALL.reported_objective_median_max": (1e-8, "max"),
    "CFG4_SAMPLE.same_success_class_min": (0.93, "min"),
    "CFG4_SAMPLE.failed_device_max_share": (0.04, "max"),
    "CFG4_SAMPLE.failed_oracle_max_share": (0.05, "max"),
    "REFIT.objective_rel_max": (1e-6, "max"),
    "REFIT.evals_abs_slack": (8, "max"),
    "REFIT.evals_rel_slack": (0.35, "max"),
    "EXTFIT.objective_rel_median_max": (1e-12, "max"),
    "EXTFIT.objective_rel_max_max": (1e-6, "max"),
    "EXTFIT.share_evals_within_3_min": (0.95, "min"),
}


def _current():
    out = {}
    for group in ("FP64", "STREAMED_M10000", "OLEARY_M5000", "CFG4_ALL", "CFG4_SAMPLE", "REFIT", "EXTFIT"):
        for k, v in getattr(K, group).items():
            out["%s.%s" % (group, k)] = v
    return out


def test_no_contract_is_loosened_and_none_is_dropped():
    cur = _current()
    assert set(cur) == set(FROZEN), (set(cur) ^ set(FROZEN))
    for name, (val, direction) in FROZEN.items():
        if direction == "max":
            assert cur[name] <= val, "%s loosened: %r > frozen %r" % (name, cur[name], val)
        else:
            assert cur[name] >= val, "%s loosened: %r < frozen %r" % (name, cur[name], val)
    assert K.EVALUATION_REL_TOL <= 1e-10  # north_star's fp64 tolerance on c, residual norm and Jacobian entries


def test_the_census_tests_read_their_numbers_from_the_contract_file():
    """no literal slack overrides the contract at a call site of the fp64 census"""
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_census.py")).read()
    for m in re.finditer(r"_assert_fp64_contract\(([^)]*)\)", src):
        args = m.group(1)
        for kw in re.findall(r"(max_evals_slack|sum_evals_slack)\s*=\s*([^,)]+)", args):
            assert kw[1].strip() == "None" or kw[1].strip().startswith("K."), m.group(0)
    assert "import contracts as K" in src


def test_integration_md_states_every_contract():
    doc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    assert "tests/contracts.py" in doc
    for name in FROZEN:
        assert "`%s`" % name in doc, "INTEGRATION.md section 8 does not list %s" % name
