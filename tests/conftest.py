import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # process(robust=True) -- what FunctionalMapping.preprocess always asks for, like the reference -- needs the robust_laplacian wheel or
    # an explicit opt-in to this package's restatement of it (it fails closed otherwise: tests/test_gpu_laplacian.py); the tests opt in
    from densematcher_amd.pyFM.mesh import laplacian
    laplacian.set_robust_backend("restated")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def fx_cfg1():
    return load_golden("fx_cfg1.npz")


@pytest.fixture(scope="session")
def fx_ties():
    return load_golden("fx_ties.npz")


@pytest.fixture(scope="session")
def fx_cfg2():
    fx = load_golden("fx_cfg2.npz")
    from densematcher_amd import synth
    n = fx["Phi1"].shape[0]
    s1, s2 = (int(x) for x in fx["feat_seeds"])
    F1, F2, _ = synth.feature_pair(n, n, int(fx["D"]), s1, s2, sigma=float(fx["feat_sigma"]), perm="identity")
    assert synth.sha256_of(F1, F2) == str(fx["feat_sha256"]), "regenerated descriptors differ from the fixture's"
    fx["F1"], fx["F2"] = F1, F2
    return fx


@pytest.fixture(scope="session")
def fx_cfg5():
    """BASELINE config-5 size (N = 8192, D = 384, k = 200), one pair, run through the reference (tools/make_golden_r05.py)"""
    fx = load_golden("fx_cfg5.npz")
    from densematcher_amd import synth
    n = fx["Phi1"].shape[0]
    s1, s2 = (int(x) for x in fx["feat_seeds"])
    F1, F2, _ = synth.feature_pair(n, n, int(fx["D"]), s1, s2, sigma=float(fx["feat_sigma"]), perm="identity")
    assert synth.sha256_of(F1, F2) == str(fx["feat_sha256"]), "regenerated descriptors differ from the fixture's"
    fx["F1"], fx["F2"] = F1, F2
    return fx


@pytest.fixture(scope="session")
def fx_cfg4():
    """BASELINE config 4 at full length (ZoomOut 50 -> 200, 150 iterations, N = 2048), one pair, run through the reference
    (tools/make_golden_r05.py cfg4)"""
    return load_golden("fx_cfg4.npz")


@pytest.fixture(scope="session")
def fx_cfg2_icp():
    return load_golden("fx_cfg2_icp.npz")


@pytest.fixture(scope="session")
def fx_cfg1_terms():
    return load_golden("fx_cfg1_terms.npz")


@pytest.fixture(scope="session")
def fx_cfg1_shape_terms():
    """area / conformal / orientation terms of the reference at a fixed map (tools/make_golden_r04.py)"""
    return load_golden("fx_cfg1_shape_terms.npz")


@pytest.fixture(scope="session")
def oracle_cfg1_fits():
    """float64 minimisers computed by the ORACLE (tools/make_oracle_vectors.py), committed to save test time"""
    return load_golden("oracle_cfg1_fits.npz")


@pytest.fixture(scope="session")
def fx_cfg1_precise():
    return load_golden("fx_cfg1_precise.npz")


@pytest.fixture(scope="session")
def fx_cfg1_notebook_call():
    return load_golden("fx_cfg1_notebook_call.npz")


@pytest.fixture(scope="session")
def fx_cfg2_f64():
    """config-2 shape on the reference's un-rounded float64 spectrum (tools/make_golden_r03.py)"""
    fx = load_golden("fx_cfg2_f64.npz")
    from densematcher_amd import synth
    n = fx["Phi1"].shape[0]
    s1, s2 = (int(x) for x in fx["feat_seeds"])
    F1, F2, _ = synth.feature_pair(n, n, int(fx["D"]), s1, s2, sigma=float(fx["feat_sigma"]), perm="identity")
    assert synth.sha256_of(F1, F2) == str(fx["feat_sha256"]), "regenerated descriptors differ from the fixture's"
    fx["F1"], fx["F2"] = F1, F2
    return fx
