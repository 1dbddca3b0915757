"""CPU tests of the host-side logic that needs no GPU: layer geometry vs the reference's parameter dump, architecture
constants, the seeded weight generator, the body-model container, OBJ / sampling helpers, eval metrics."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_build_params_reproduces_reference_dump_exactly():
    from etch_amd.models.so3net import build_params
    ref = json.load(open(os.path.join(GOLDEN, "epn_model_setting.json")))
    mine = build_params(mlps=[[32, 32], [64, 64]], strides=[2, 2])
    assert json.dumps(mine) == json.dumps(ref)          # same keys, same order, same floats
    # independent restatement in the oracle agrees for other depths / radii too
    from oracle import stage1 as S
    for radius, mlps, strides, n in ((0.4, [[32, 32], [64, 64], [128, 128]], [2, 2, 2], 1024), (0.55, [[32, 32]], [2], 1024), (0.4, [[32, 32], [64, 64]], [2, 2], 4096)):
        a = build_params(input_radius=radius, input_num=n, mlps=mlps, strides=strides)["backbone"]
        b = S.build_layer_table(radius, mlps, strides, n)
        for blk_a, blk_b in zip(a, b):
            for ca, cb in zip(blk_a, blk_b):
                for k, v in cb.items():
                    assert ca["args"][k] == v, (radius, k)


def test_constants_match_golden(golden):
    from etch_amd import constants as K
    g = golden("constants.npz")
    assert np.array_equal(K.get_anchors(), g["anchors"]) and np.array_equal(K.get_intra_idx(), g["intra_idx"])
    for tag, rad in (("b0c0", 0.08000000000000002), ("b0c1", 0.11313708498984763), ("b1c0", 0.16000000000000003)):
        assert np.array_equal(K.get_kernel_points(rad), g[f"kernels_{tag}"])
    ms = K.default_markerset()
    assert list(ms.keys()) == json.load(open(os.path.join(GOLDEN, "marker_names.json"))) and list(ms.values()) == g["marker_vids"].tolist()


def test_seeded_weights_are_name_keyed_and_deterministic():
    import torch
    from etch_amd.utils.weights import seeded_tensor
    a = seeded_tensor("encoder.x.W", (4, 6), torch.float32, 1)
    assert torch.equal(a, seeded_tensor("encoder.x.W", (4, 6), torch.float32, 1))
    assert not torch.equal(a, seeded_tensor("encoder.y.W", (4, 6), torch.float32, 1))
    assert not torch.equal(a, seeded_tensor("encoder.x.W", (4, 6), torch.float32, 2))
    assert float(a.abs().max()) <= np.sqrt(6.0 / 10) + 1e-6                       # Xavier-uniform bound
    v = seeded_tensor("bn.running_var", (8,), torch.float32, 1)
    assert float(v.min()) >= 0.5


def test_synthetic_body_model_shapes():
    from etch_amd.utils.body_model import SMPL_PARENTS, SyntheticSMPL
    bm = SyntheticSMPL(7)
    assert bm.v_template.shape == (6890, 3) and bm.shapedirs.shape == (6890, 3, 10) and bm.posedirs.shape == (207, 20670)
    assert bm.J_regressor.shape == (24, 6890) and bm.lbs_weights.shape == (6890, 24) and bm.faces.shape[1] == 3
    assert np.allclose(bm.lbs_weights.sum(1), 1, atol=1e-6) and (np.count_nonzero(bm.lbs_weights, axis=1) <= 4).all()
    assert (SMPL_PARENTS[1:] < np.arange(1, 24)).all()                            # parents precede children (LM kernel relies on it)
    assert np.array_equal(SyntheticSMPL(7).v_template, bm.v_template)


def test_obj_roundtrip_and_surface_sampler(tmp_path):
    from etch_amd.inference_demo import load_obj, preprocess_scan, sample_points_from_mesh
    from etch_amd.models.fit_SMPL import Mesh
    v = np.array([[0, 0, 0], [2, 0, 0], [0, 4, 0], [0, 0, 6]], np.float64)
    f = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    p = tmp_path / "t.obj"
    Mesh(v, f).export(str(p))
    m = load_obj(str(p))
    assert np.allclose(m.vertices, v) and np.array_equal(m.faces, f)
    centred, centre = preprocess_scan(str(p))
    assert np.allclose(centre, [1, 2, 3]) and np.allclose(centred.vertices, v - centre)   # bbox mid-point (inference_demo.py:25-28)
    pts = sample_points_from_mesh(m, 4000, seed=3)
    assert pts.shape == (4000, 3) and np.array_equal(pts, sample_points_from_mesh(m, 4000, seed=3))
    assert (pts >= -1e-9).all() and (pts[:, 0] / 2 + pts[:, 1] / 4 + pts[:, 2] / 6 <= 1 + 1e-9).all()   # inside the tetrahedron hull
    # area weighting: the largest face (1,2,3) receives the largest share
    on_big = np.isclose(pts[:, 0] / 2 + pts[:, 1] / 4 + pts[:, 2] / 6, 1.0, atol=1e-9).mean()
    assert on_big > 0.35


def test_eval_metrics():
    from etch_amd.eval import mpjpe, v2v
    assert v2v(np.ones((5, 3)), np.zeros((5, 3))) == np.sqrt(3.0)
    j = np.zeros((45, 3)); k = j.copy(); k[30:] = 9.0                             # joints beyond the first 22 are ignored
    assert mpjpe(j, k) == 0.0
