"""
GPU (-m gpu): the one-pass fp16-split projection (csrc/dm_project.hip: proj_onepass_kernel -- running power-of-two scale per
workgroup, no maxima pass) against the float64 product of the fp32-rounded operands (what the reference's fit projects,
pyFM/functional.py:410-414, base_functions.py:526-532) and against the two launches it replaces (dm_set_option "proj_onepass" = 0).
Tolerance: 2e-6 of sum_n |mass Phi| |F| per output -- the fp32 accumulation both kernels share; the split itself is exact to 2^-22.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _engine():
    from densematcher_amd.engine import MatchEngine
    return MatchEngine()


@pytest.fixture
def eng(_engine):
    yield _engine
    _engine.reset_options()


def _ref(Phi, a, F, k):
    X = Phi[:, :, :k].astype(np.float32).astype(np.float64) * a.astype(np.float32).astype(np.float64)[:, :, None]
    Fd = F.astype(np.float64)
    return np.einsum("bnk,bnd->bkd", X, Fd), np.einsum("bnk,bnd->bkd", np.abs(X), np.abs(Fd))


PROFILES = {
    "flat": lambda N: np.ones(N),
    "growing": lambda N: np.exp2(np.linspace(-20, 20, N)),          # the running scale shrinks stage after stage
    "shrinking": lambda N: np.exp2(np.linspace(20, -20, N)),        # set once, values sink into the fp16 subnormals
    "zeros then data": lambda N: np.concatenate([np.zeros(N // 2 + 7), np.ones(N - N // 2 - 7)]),
    "spike at the end": lambda N: np.concatenate([np.ones(N - 1), [1e6]]),
    "all zero": lambda N: np.zeros(N),
}


@pytest.mark.parametrize("profile", list(PROFILES))
@pytest.mark.parametrize("N,k,pad,D,dt", [(2500, 77, 4, 200, np.float64), (1024, 128, 0, 384, np.float32), (700, 5, 1, 16, np.float64),
                                          (3000, 200, 3, 770, np.float32)])
def test_onepass_projection(eng, profile, N, k, pad, D, dt):
    """row strides that are odd, k not a multiple of 8 or 32, D below / above one tile and not a multiple of 8, chunk and stage
    remainders; whatever lies behind column k (NaN here) must not reach the result or the scale"""
    rng = np.random.default_rng(N + k)
    B = 2
    Phi = rng.standard_normal((B, N, k + pad)) * PROFILES[profile](N)[None, :, None]
    Phi[:, :, k:] = np.nan
    Phi = Phi.astype(dt)
    a = rng.uniform(0.5, 1.5, (B, N)).astype(dt)
    F = rng.standard_normal((B, N, D)).astype(np.float16)
    R, S = _ref(Phi, a, F, k)
    S = np.maximum(S, 1e-300)
    got = {}
    for opt in (1, 0):
        eng.set_option("proj_onepass", opt)
        A = eng.project(Phi, a, F, k).cpu().numpy().astype(np.float64)
        assert np.isfinite(A).all(), (profile, opt)
        assert (np.abs(A - R) / S).max() <= 2e-6, (profile, opt, (np.abs(A - R) / S).max())
        got[opt] = A
    # a pair's projection does not depend on the batch it is in
    eng.set_option("proj_onepass", 1)
    A1 = eng.project(Phi[1:], a[1:], F[1:], k).cpu().numpy()
    assert np.array_equal(A1[0], got[1][1].astype(np.float32))


def test_onepass_projection_inside_fit(eng, fx_cfg2):
    """dm_fmap_fit (projections + Gram + solve) gives the same map to 1e-6 with either projection; both far inside the 1e-4 bar"""
    fx = fx_cfg2
    b = lambda x: np.ascontiguousarray(x)[None]
    k = int(fx["k"])
    C = {}
    for opt in (1, 0):
        eng.set_option("proj_onepass", opt)
        C[opt] = eng.fmap_fit(b(fx["Phi1"]), b(fx["Phi2"]), b(fx["a1"]), b(fx["a2"]), b(fx["F1"]), b(fx["F2"]), b(fx["lam1"][:k]), b(fx["lam2"][:k]),
                              float(fx["w_descr"]), float(fx["w_lap"]), k, k).cpu().numpy()[0]
    assert np.abs(C[1] - C[0]).max() <= 1e-6
    assert np.abs(C[1] - fx["C_f64"]).max() <= 1e-5 and np.abs(C[0] - fx["C_f64"]).max() <= 1e-5
