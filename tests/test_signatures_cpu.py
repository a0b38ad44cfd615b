"""HKS / WKS descriptors of the host mirror (densematcher_amd/pyFM/signatures.py) against vectors produced by the
reference's own pyFM/signatures (tools/make_golden.py: case_signatures -> tests/golden/fx_sig.npz).  CPU only."""
import os
import types

import numpy as np
import pytest

from densematcher_amd.pyFM import signatures as sg

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fx_sig.npz")


@pytest.fixture(scope="module")
def fx():
    return np.load(GOLD)


def _mesh(fx, which, k=None):
    phi, lam = fx[f"Phi{which}"].astype(np.float64), fx[f"lam{which}"]
    if k is not None:
        phi, lam = phi[:, :k], lam[:k]
    return types.SimpleNamespace(eigenvalues=lam, eigenvectors=phi)


def _close(a, b, rel=1e-12):
    """equal up to rounding; non-finite entries (the reference's 2048-energy WKS divides 0 by 0 where every weight of an
    energy underflows) must sit at the same places with the same kind"""
    assert a.shape == b.shape
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin) and np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(a[~fin & ~np.isnan(b)], b[~fin & ~np.isnan(b)])      # +-inf
    if fin.any():
        assert np.abs(a[fin] - b[fin]).max() <= rel * np.abs(b[fin]).max()


def test_hks_wks_match_reference(fx):
    k = int(fx["k"])
    m1 = _mesh(fx, 1)
    _close(sg.mesh_HKS(m1, 16, k=k), fx["hks"])
    _close(sg.mesh_WKS(m1, 24, k=k), fx["wks"])
    _close(sg.mesh_HKS(_mesh(fx, 2), 9), fx["hks_allk"])                       # k=None: every stored eigenpair
    big = sg.mesh_WKS(m1, 2048, k=k)                                           # compute_surface_map's WKS size
    _close(big[:, ::64], fx["wks_big_cols"])
    _close(big.sum(axis=1), fx["wks_big_sum"], rel=1e-11)


def test_landmark_signatures_match_reference(fx):
    k = int(fx["k"])
    m1 = _mesh(fx, 1)
    lm = fx["landmarks"]
    _close(sg.mesh_HKS(m1, 5, landmarks=lm, k=k), fx["hks_lm"])
    _close(sg.mesh_WKS(m1, 7, landmarks=lm, k=k), fx["wks_lm"])


def test_preprocess_descriptor_assembly(fx):
    """FunctionalMapping.preprocess with HKS + two-column landmarks + subsample_step = 2 (functional.py:308-334)"""
    from densematcher_amd.pyFM.functional import FunctionalMapping
    k = int(fx["k"])

    class _M(types.SimpleNamespace):
        def process(self, *a, **kw):
            return self

    model = FunctionalMapping(_M(**vars(_mesh(fx, 1, k))), _M(**vars(_mesh(fx, 2, k))))
    model.preprocess(n_ev=(k, k), n_descr=16, descr_type="HKS", landmarks=fx["landmarks2"], subsample_step=2)
    _close(model.descr1, fx["pre_descr1"])
    _close(model.descr2, fx["pre_descr2"])
    with pytest.raises(ValueError):
        model.preprocess(n_ev=(k, k), descr_type="SHOT")
