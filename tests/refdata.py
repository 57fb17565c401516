"""Inputs and known answers held by the reference's own tests for the hot path (data only)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

# src/solvers/levmar/test.rs:21-29, 111-163, 166-208 (octave: t = linspace(0,10,11); y = 2*exp(-t/2)+exp(-t/4)+1)
T11 = np.arange(11.0)
Y11 = np.array([4.0000, 2.9919, 2.3423, 1.9186, 1.6386, 1.4507, 1.3227, 1.2342, 1.1720, 1.1276, 1.0956])
RES_UNWEIGHTED_05_65 = np.array([-0.032243, 0.236772, 0.028277, -0.105709, -0.149393, -0.136205, -0.092002,
                                 -0.032946, 0.031394, 0.095542, 0.156511])
RES_WEIGHTED_05_65 = np.array([-0.307187, 0.493658, 0.286886, -0.150538, -0.346541, -0.342850, -0.235283, -0.084548,
                               0.077943, 0.237072, 0.385972])

# tests/integration_tests/main.rs:713-778 (O'Leary & Rust example; matlab/examples/varpro_example.m:26-43)
OLEARY_T = np.array([0., 0.1, 0.22, 0.31, 0.46, 0.50, 0.63, 0.78, 0.85, 0.97])
OLEARY_Y = np.array([6.9842, 5.1851, 2.8907, 1.4199, -0.2473, -0.5243, -1.0156, -1.0260, -0.9165, -0.6805])
OLEARY_W = np.array([1.0, 1.0, 1.0, 0.5, 0.5, 1.0, 0.5, 1.0, 0.5, 0.5])
OLEARY_GUESS = np.array([0.5, 2., 3.])
OLEARY_ALPHA = np.array([1.0132255e+00, 2.4968675e+00, 4.0625148e+00])
OLEARY_C = np.array([5.8416357e+00, 1.1436854e+00])
OLEARY_WRES = np.array([-1.1211e-03, 3.1751e-03, -2.7656e-03, -1.4600e-03, 1.2081e-03, 2.2586e-03, -1.1101e-03,
                        -2.2554e-03, 1.3257e-03, 1.4716e-03])

# tests/integration_tests/main.rs:554-598 and :616-668 (python lmfit output; inputs test_assets/*)
LMFIT_UNWEIGHTED = dict(c=np.array([2.19344628, 6.80462652, 1.59995673]), tau=np.array([2.40392137, 5.99571068]))
LMFIT_WEIGHTED = dict(c=np.array([2.24275841, 6.75609070, 1.59790007]), tau=np.array([2.43119160, 6.02052311]))


def read_raw_f64(name):
    """little-endian f64 fixture (the reference's read_vec_f64, tests/integration_tests/main.rs:691-709)"""
    return np.fromfile(os.path.join(GOLDEN, name), dtype="<f8")
