"""Body-model containers for stage 2 (SMPL marker fit).

`SyntheticSMPL(seed)` builds a seeded SMPL-SHAPED model (V = 6890, J = 24 with the public SMPL parent table,
10 betas, 207 pose-basis rows, <= 4 non-zero skinning weights per vertex): the licensed SMPL .pkl files the
reference loads (src/models/fit_SMPL.py:92-101) are not redistributable and absent here.  `load_smpl_pkl`
reads a real chumpy-free SMPL pickle when the user has one.  Array names follow smplx.SMPL buffers
(v_template, shapedirs, posedirs, J_regressor, lbs_weights, parents, faces) [upstream smplx, not in the tree].
"""
import pickle

import numpy as np

SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int32)
# 21 vertex-picked joints appended by smplx's VertexJointSelector for SMPL (nose, eyes, ears, feet, finger tips)
SMPL_EXTRA_JOINT_VIDS = np.array([332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                                  2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133], np.int32)


# SMPL-X kinematic tree (55 joints: 22 body, jaw, 2 eyes, 2 x 15 finger joints) [upstream smplx, public]
SMPLX_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
                          20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
                          21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], np.int32)


class BodyModel:
    """Plain numpy container (fp32)."""

    def __init__(self, v_template, shapedirs, posedirs, J_regressor, lbs_weights, parents, faces, extra_vids=SMPL_EXTRA_JOINT_VIDS):
        self.v_template = np.ascontiguousarray(v_template, np.float32)          # (V,3)
        self.shapedirs = np.ascontiguousarray(shapedirs, np.float32)            # (V,3,NB)
        self.posedirs = np.ascontiguousarray(posedirs, np.float32)              # (9*(J-1), V*3)
        self.J_regressor = np.ascontiguousarray(J_regressor, np.float32)        # (J,V)
        self.lbs_weights = np.ascontiguousarray(lbs_weights, np.float32)        # (V,J)
        self.parents = np.ascontiguousarray(parents, np.int32)
        self.faces = np.ascontiguousarray(faces, np.int32)
        self.extra_vids = np.ascontiguousarray(extra_vids, np.int32)
        self.num_betas = self.shapedirs.shape[2]
        self.NUM_BODY_JOINTS = len(self.parents) - 1

    @property
    def num_verts(self):
        return self.v_template.shape[0]

    @property
    def num_joints(self):
        return len(self.parents)


def _synthetic(seed, V, NB, parents, extra_vids, shape_scale=0.01, pose_scale=0.001):
    rng = np.random.default_rng(seed)
    J = len(parents)
    v_t = rng.standard_normal((V, 3)) * np.array([0.15, 0.45, 0.10])
    # a crude skeleton: joint centres = random vertices; weights = softmax over the 4 nearest joints
    cent = v_t[rng.choice(V, J, replace=False)]
    d2 = ((v_t[:, None] - cent[None]) ** 2).sum(-1)
    near = np.argsort(d2, 1)[:, :4]
    w = np.exp(-np.take_along_axis(d2, near, 1) / 0.01)
    w /= w.sum(1, keepdims=True)
    W = np.zeros((V, J))
    np.put_along_axis(W, near, w, 1)
    Jreg = np.zeros((J, V))
    for j in range(J):
        Jreg[j, np.argsort(d2[:, j])[:50]] = 1.0 / 50
    S = rng.standard_normal((V, 3, NB)) * shape_scale
    P = rng.standard_normal((9 * (J - 1), V * 3)) * pose_scale
    faces = np.stack([np.arange(V - 2), np.arange(1, V - 1), np.arange(2, V)], 1)[: 2 * V - 4 - (V - 2)]  # a strip, only for OBJ export
    faces = np.concatenate([faces, faces[:, ::-1]])[:2 * V - 4]
    return BodyModel(v_t, S, P, Jreg, W, parents, faces, extra_vids)


def SyntheticSMPL(seed=7, V=6890, NB=10, shape_scale=0.01, pose_scale=0.001):
    """shape_scale / pose_scale: standard deviation of the shape / pose blend-shape entries (real SMPL: ~3e-2 / ~1e-2 m; the defaults
    are the gentler values every round-1/2 fixture was generated with)."""
    return _synthetic(seed, V, NB, SMPL_PARENTS, SMPL_EXTRA_JOINT_VIDS, shape_scale, pose_scale)


def SyntheticSMPLX(seed=7, V=10475, NB=20):
    """Seeded SMPL-X-SIZED model (V = 10 475, J = 55 with the public SMPL-X tree, 10 shape + 10 expression coefficients as
    one 20-vector, 486 pose-basis rows) for BASELINE configs[4].  The 21 vertex-picked extra joints reuse the SMPL ids (< V)."""
    return _synthetic(seed, V, NB, SMPLX_PARENTS, SMPL_EXTRA_JOINT_VIDS)


def load_smpl_pkl(path, num_betas=10):
    """Chumpy-free SMPL pickle (e.g. SMPL_NEUTRAL_10pc_rmchumpy.pkl): keys v_template, shapedirs, posedirs (V,3,207),
    J_regressor (sparse or dense), weights, kintree_table, f."""
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    Jr = d["J_regressor"]
    Jr = np.asarray(Jr.todense()) if hasattr(Jr, "todense") else np.asarray(Jr)
    posedirs = np.asarray(d["posedirs"])
    Vn = posedirs.shape[0]
    posedirs = posedirs.reshape(Vn * 3, -1).T                                  # smplx: (207, V*3)
    parents = np.asarray(d["kintree_table"])[0].astype(np.int64).copy()
    parents[0] = -1
    return BodyModel(np.asarray(d["v_template"]), np.asarray(d["shapedirs"])[:, :, :num_betas], posedirs, Jr, np.asarray(d["weights"]),
                     parents, np.asarray(d["f"]))


# vertex ids of the 21 tip / face / foot joints smplx's VertexJointSelector appends for SMPL-X (smplx/vertex_ids.py, 'smplx' table, in the
# selector's order: nose, reye, leye, rear, lear, LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel, then thumb..pinky tips left, right)
# [upstream smplx, public]
SMPLX_EXTRA_JOINT_VIDS = np.array([9120, 9929, 9448, 616, 6, 5770, 5780, 8846, 8463, 8474, 8635,
                                   5361, 4933, 5058, 5169, 5286, 8079, 7669, 7794, 7905, 8022], np.int32)


def load_smplx(path, num_betas=10, num_expression_coeffs=10):
    """A real SMPL-X model file as distributed by smpl-x.is.tue.mpg.de (SMPLX_{NEUTRAL,MALE,FEMALE}.npz or the chumpy-free .pkl): keys
    v_template (10475,3), shapedirs (10475,3,400: 300 shape + 100 expression components), posedirs (10475,3,486), J_regressor (55,10475),
    weights (10475,55), kintree_table (2,55), f.  BASELINE configs[4] names SMPL-X; the reference itself only loads SMPL
    (src/models/fit_SMPL.py:92-101).  Returns a BodyModel whose coefficient vector is [betas[:num_betas] | expression[:num_expression_coeffs]]
    (the LM kernel is instantiated for 55 joints x 20 coefficients: the defaults), pose = 54 x 3 (body 21, jaw, eyes 2, hands 2 x 15, full
    axis-angle hands: no PCA), joints = 55 regressed + the 21 vertex-picked ones (the 51 + 17 face landmarks of smplx's full output are not
    appended)."""
    if str(path).endswith(".npz"):
        d = dict(np.load(path, allow_pickle=True))
    else:
        with open(path, "rb") as f:
            d = pickle.load(f, encoding="latin1")
    sd = np.asarray(d["shapedirs"])
    n_shape_total = 300 if sd.shape[2] >= 400 else sd.shape[2] - min(100, sd.shape[2] // 4)
    shape = sd[:, :, :num_betas]
    expr = sd[:, :, n_shape_total:n_shape_total + num_expression_coeffs]
    if expr.shape[2] != num_expression_coeffs or shape.shape[2] != num_betas:
        raise ValueError(f"{path}: shapedirs has {sd.shape[2]} components, cannot take {num_betas} shape + {num_expression_coeffs} expression")
    Jr = d["J_regressor"]
    Jr = np.asarray(Jr.todense()) if hasattr(Jr, "todense") else np.asarray(Jr)
    posedirs = np.asarray(d["posedirs"])
    V = posedirs.shape[0]
    posedirs = posedirs.reshape(V * 3, -1).T                                   # smplx: (9 * 54, V * 3)
    parents = np.asarray(d["kintree_table"])[0].astype(np.int64).copy()
    parents[0] = -1
    if len(parents) != 55 or not np.array_equal(parents[1:], SMPLX_PARENTS[1:]):
        raise ValueError(f"{path}: not the 55-joint SMPL-X kinematic tree")
    extra = SMPLX_EXTRA_JOINT_VIDS if V > int(SMPLX_EXTRA_JOINT_VIDS.max()) else SMPL_EXTRA_JOINT_VIDS
    return BodyModel(np.asarray(d["v_template"]), np.concatenate([shape, expr], 2), posedirs, Jr, np.asarray(d["weights"]), parents,
                     np.asarray(d["f"]), extra)
