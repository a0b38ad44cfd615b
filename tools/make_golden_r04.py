#!/usr/bin/env python
"""
Round-4 golden fixture, produced by IMPORTING THE REFERENCE (/root/reference) in the build container (recipe and stubs:
tools/make_golden.py, which this script re-uses).

    fx_cfg1_shape_terms.npz   on the config-1 pair (N = 500, k = 30, tests/golden/fx_cfg1.npz) at a fixed map C:
        area_E / area_G             reference area(C) and its torch autograd gradient         (base_functions.py:228-255)
        conf_E / conf_G             reference conformal(C, evals1, evals2) and its gradient   (base_functions.py:257-294)
        orient_np_op1 / _op2        FunctionalMapping.compute_orientation_op() of the first NDESC descriptors: the NumPy form the
                                    reference uses for the w_orient rescale (functional.py:686-728, geometry.py:919-985)
        orient_t_op1 / _op2         the same operators as energy_func_std builds them for the optimisation
                                    (base_functions.py:430-478, :567-597: orientation_op_torch, float32 inside)
        orient_E / orient_G         oplist_commutation(C, orient_t ops) and its autograd gradient (base_functions.py:176-203)
        fit_orient_C                FunctionalMapping.fit(w_descr=1e4, w_lap=1e3, w_dcomm=0, w_orient=1, w_area=1e2, w_conformal=1e2)
                                    on the NDESC descriptors, with the rescaled orientation weight it reports
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402
from make_golden_r02 import mesh_from_fixture  # noqa: E402

OUT = mg.OUT
NDESC = 6


def main():
    fx = dict(np.load(os.path.join(OUT, "fx_cfg1.npz"), allow_pickle=False))
    k = int(fx["k"])
    bf = mg.ref_bf
    rng = np.random.default_rng(42)
    Ctest = fx["C_f64"] + 0.05 * rng.standard_normal((k, k))
    m1, m2 = mg.truncated(mesh_from_fixture(fx, 1), k), mg.truncated(mesh_from_fixture(fx, 2), k)
    out = {"C": Ctest, "ndesc": NDESC}

    def val_grad(fn):
        Ct = torch.tensor(Ctest, dtype=torch.float64, requires_grad=True)
        e = fn(Ct)
        e.backward()
        return float(e), Ct.grad.numpy().copy()
    ev1, ev2 = torch.tensor(m1.eigenvalues, dtype=torch.float64), torch.tensor(m2.eigenvalues, dtype=torch.float64)
    out["area_E"], out["area_G"] = val_grad(lambda C: bf.area(C))
    out["conf_E"], out["conf_G"] = val_grad(lambda C: bf.conformal(C, ev1, ev2))
    # the hand-written gradient functions of the reference agree with autograd (they are what grad_energy_std adds)
    Ct = torch.tensor(Ctest, dtype=torch.float64)
    assert np.abs(bf.area_grad(Ct).numpy() - out["area_G"]).max() < 1e-10
    assert np.abs(bf.conformal_grad(Ct, ev1, ev2).numpy() - out["conf_G"]).max() < 1e-10

    model = mg.FunctionalMapping(m1, m2, partial=False, optimizer="L-BFGS-B")
    model.k1, model.k2 = k, k
    model.descr1 = fx["F1"][:, :NDESC].astype(np.float64)
    model.descr2 = fx["F2"][:, :NDESC].astype(np.float64)
    ops = model.compute_orientation_op(reversing=False)
    out["orient_np_op1"] = np.stack([np.asarray(a) for a, _ in ops])
    out["orient_np_op2"] = np.stack([np.asarray(b) for _, b in ops])
    out["vertex_areas1"], out["vertex_areas2"] = model.mesh1.vertex_areas, model.mesh2.vertex_areas

    # the torch form of energy_func_std (base_functions.py:567-597), restated call by call on the reference's functions
    def torch_ops(mesh, descr):
        n = mesh.n_vertices
        gm = mg.ref_bf  # noqa: F841
        from densematcher.pyFM.mesh import geometry
        gradmat = torch.tensor(geometry.grad_mat(mesh.vertlist, mesh.facelist, mesh.normals).todense()).to(torch.float64).to_sparse()
        d = torch.tensor(descr, dtype=torch.float64)
        f = mesh.facelist.shape[0]
        grads = (gradmat.float() @ d.float()).reshape(f, 3, descr.shape[1]).to(d.dtype)
        A = torch.tensor(mesh.A.toarray(), dtype=torch.float64)
        op_hat = bf.orientation_op_torch(grads, torch.tensor(mesh.vertlist, dtype=torch.float64), torch.tensor(mesh.facelist, dtype=int),
                                         torch.tensor(mesh.normals, dtype=torch.float64), A[torch.arange(n), torch.arange(n)])
        ev = torch.tensor(mesh.eigenvectors, dtype=torch.float64)
        pinv = ev.T @ A
        dense = op_hat.to_dense().permute(2, 0, 1)
        return torch.bmm(torch.bmm(pinv.unsqueeze(0).expand(descr.shape[1], -1, -1), dense), ev.unsqueeze(0).expand(descr.shape[1], -1, -1))
    t1, t2 = torch_ops(model.mesh1, model.descr1), torch_ops(model.mesh2, model.descr2)
    out["orient_t_op1"], out["orient_t_op2"] = t1.numpy(), t2.numpy()
    Ct = torch.tensor(Ctest, dtype=torch.float64, requires_grad=True)
    e, g, _ = bf.oplist_commutation(Ct, None, [(t1[i], t2[i]) for i in range(NDESC)])
    out["orient_E"], out["orient_G"] = float(e), g.numpy().copy()

    # the reference's fit with the three terms on
    bf.can_op1 = None
    bf.can_op2 = None
    model.fit(w_descr=1e4, w_lap=1e3, w_dcomm=0, w_orient=1, w_area=1e2, w_conformal=1e2, optinit="zeros", verbose=False, device=mg.CPU)
    out["fit_orient_C"] = model.FM.copy()
    np.savez_compressed(os.path.join(OUT, "fx_cfg1_shape_terms.npz"), **out)
    print({n: (np.asarray(v).shape if hasattr(v, "shape") else v) for n, v in out.items()})
    print("area E", out["area_E"], "conf E", out["conf_E"], "orient E", out["orient_E"],
          "np vs torch ops", np.abs(out["orient_np_op1"] - out["orient_t_op1"]).max(), np.abs(out["orient_t_op1"]).max())


if __name__ == "__main__":
    main()
