"""CPU restatement (PyTorch fp32) of ETCH stage 2: get_markers + fit_smpl.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for the LM / LBS part: `theseus` (unpinned git HEAD, README.md:48), `smplx` (unpinned pip,
environment.yml:26) and the licensed SMPL .pkl (README.md:64) are all absent from /root/reference and from this
image, and the reference holds no tests or golden vectors for this boundary.  What IS pinned:
  get_markers          src/models/fit_SMPL.py:17-62   <- tests/golden/markers.npz (emitted by the reference's own function)
  residual / schedule  src/models/fit_SMPL.py:111-152,161-249 (variable order pose|shape|orient|transl, 258 residuals,
                       stage 0: 77 DoF, 30 it, step 0.5, damping 0.01; stage 1: 85 DoF, 50 it, step 0.2, damping 1e-3)
  batch_rodrigues      the in-tree verbatim copy src/data_utils/GT_dataloader_mixed.py:29-64 (angle = |theta + 1e-8|)
Restated from the published upstream algorithms [not in the tree]:
  lbs                  smplx.lbs.lbs / batch_rigid_transform / vertices2joints / VertexJointSelector
  lm                   theseus.LevenbergMarquardt with a dense Cholesky solver, fixed damping, no step rejection,
                       error 0.5*|r|^2, per-sample freeze when |d err| < 1e-10 or |d err|/err_prev < 1e-8.
The Jacobian here is torch.func.jacrev through the FULL 6890-vertex LBS, exactly the reference's formulation
(AutoDiffCostFunction); the HIP kernel's analytic marker-restricted Jacobian is validated against it.
"""
import numpy as np
import torch


def get_markers(num_markers, inner_points, part_labels, confidences):
    """fit_SMPL.py:17-62."""
    B = inner_points.shape[0]
    valid = torch.zeros(B, num_markers, dtype=torch.bool)
    pos = torch.zeros(B, num_markers, 3)
    for b in range(B):
        for label in range(num_markers):
            mask = part_labels[b] == label
            cnt = int(mask.sum())
            if cnt == 0:
                continue
            pts, conf = inner_points[b][mask], confidences[b][mask]
            _, ind = torch.topk(conf, min(cnt, 3), dim=0, largest=True)
            ind = ind.squeeze()
            if ind.dim() == 0:
                ind = ind.unsqueeze(0)
            w = conf[ind] ** 20
            valid[b, label] = True
            pos[b, label] = (pts[ind] * w).sum(dim=0) / w.sum()
    return pos, valid


class TorchBody:
    def __init__(self, bm, dtype=torch.float32):
        t = lambda a: torch.from_numpy(a).to(dtype) if a.dtype.kind == "f" else torch.from_numpy(a)
        self.dtype = dtype
        self.v_t, self.S, self.P = t(bm.v_template), t(bm.shapedirs), t(bm.posedirs)
        self.Jreg, self.W = t(bm.J_regressor), t(bm.lbs_weights)
        self.parents = [int(p) for p in bm.parents]
        self.extra = t(bm.extra_vids).long()
        self.J = len(self.parents)


def rodrigues(r):
    """GT_dataloader_mixed.py:29-64 (= smplx.lbs.batch_rodrigues)."""
    angle = torch.norm(r + 1e-8, dim=1, keepdim=True)
    d = r / angle
    c, s = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    z = torch.zeros_like(rx)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).view(-1, 3, 3)
    return torch.eye(3, dtype=r.dtype)[None] + s * K + (1 - c) * torch.bmm(K, K)


def lbs(tb, betas, pose72, transl):
    """[upstream] smplx.lbs.lbs.  betas [B,NB], pose72 [B,3J] = orient|body_pose (SMPL: 72), transl [B,3] -> verts [B,V,3], joints [B,J,3]."""
    B, J = betas.shape[0], tb.J
    v_s = tb.v_t[None] + torch.einsum("bl,mkl->bmk", betas, tb.S)
    Jl = torch.einsum("bik,ji->bjk", v_s, tb.Jreg)
    R = rodrigues(pose72.reshape(-1, 3)).view(B, J, 3, 3)
    pf = (R[:, 1:] - torch.eye(3, dtype=R.dtype)).reshape(B, -1)
    v_p = v_s + (pf @ tb.P).view(B, -1, 3)
    par = torch.tensor(tb.parents[1:])
    rel = torch.cat([Jl[:, :1], Jl[:, 1:] - Jl[:, par]], 1)
    T = torch.cat([torch.cat([R, rel[..., None]], -1), torch.tensor([0, 0, 0, 1.0], dtype=R.dtype).expand(B, J, 1, 4)], -2)
    chain = [T[:, 0]]
    for i in range(1, J):
        chain.append(chain[tb.parents[i]] @ T[:, i])
    G = torch.stack(chain, 1)
    Jh = torch.cat([Jl, torch.zeros(B, J, 1, dtype=R.dtype)], -1)[..., None]
    A = G - torch.nn.functional.pad(G @ Jh, (3, 0))
    Tm = (tb.W[None].expand(B, -1, -1) @ A.view(B, J, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_p, torch.ones(B, v_p.shape[1], 1, dtype=R.dtype)], -1)[..., None]
    verts = (Tm @ vh)[:, :, :3, 0]
    return verts + transl[:, None], G[:, :, :3, 3] + transl[:, None]


def smpl_forward(tb, betas, body_pose, orient, transl):
    """smplx.SMPL.forward: vertices [B,V,3], joints [B,J+21,3] (J regressed + 21 vertex-picked; SMPL: 45)."""
    v, j = lbs(tb, betas, torch.cat([orient, body_pose], 1), transl)
    return v, torch.cat([j, v[:, tb.extra]], 1)


def residual_fn(tb, marker_vids, nb_opt):
    nb_total = tb.S.shape[2]

    def f(x, target, mask):
        npose = 3 * (tb.J - 1)
        pose, b, go, t = x[:npose], x[npose:npose + nb_opt], x[npose + nb_opt:npose + nb_opt + 3], x[npose + nb_opt + 3:npose + nb_opt + 6]
        betas = torch.cat([b, torch.zeros(nb_total - nb_opt, dtype=x.dtype)])[None]
        v, _ = lbs(tb, betas, torch.cat([go, pose])[None], t[None])
        return ((target - v[0][marker_vids]) * mask[:, None]).reshape(-1)      # fit_SMPL.py:127-131

    return f


def lm(f, x, target, mask, iters, step, damping, trace=None):
    """[upstream] theseus LevenbergMarquardt, dense Cholesky, fixed damping."""
    B, dof = x.shape
    jac = torch.func.vmap(torch.func.jacrev(f))
    rf = torch.func.vmap(f)
    conv = torch.zeros(B, dtype=torch.bool)
    last = 0.5 * (rf(x, target, mask) ** 2).sum(1)
    if trace is not None:
        trace.append(last.clone())
    for _ in range(iters):
        r = rf(x, target, mask)
        Jm = jac(x, target, mask)
        AtA = Jm.transpose(1, 2) @ Jm + damping * torch.eye(dof, dtype=x.dtype)
        Atb = -(Jm.transpose(1, 2) @ r[..., None])
        delta = torch.cholesky_solve(Atb, torch.linalg.cholesky(AtA))[..., 0]
        x = torch.where(conv[:, None], x, x + step * delta)
        err = 0.5 * (rf(x, target, mask) ** 2).sum(1)
        if trace is not None:
            trace.append(err.clone())
        a = (last - err).abs()
        conv = (a < 1e-10) | (a / last < 1e-8)
        last = err
        if bool(conv.all()):
            break
    return x


def fit_smpl(bm, marker_vids, markers, valid, steps_stage0=30, steps_stage1=50, lr_stage0=0.5, lr_stage1=0.2, trace=None, dtype=torch.float32):
    """fit_SMPL.py:68-269 after get_markers.  Returns dict(pose [B,69], betas [B,10], orient, transl, verts, joints).
    dtype = float32 is the reference's arithmetic (Theseus / smplx run in fp32); float64 is the same algorithm as a
    rounding-free yardstick: tests use |fp32 run - fp64 run| as the error any fp32 implementation is entitled to on the
    weakly observed parameters."""
    with torch.no_grad():
        tb = TorchBody(bm, dtype)
        markers = markers.to(dtype)
        mv = torch.as_tensor(np.asarray(marker_vids)).long()
        B = markers.shape[0]
        mask = valid.to(dtype)
        t0 = [] if trace is not None else None
        npose, nbt = 3 * (tb.J - 1), tb.S.shape[2]          # SMPL: 69 pose variables, 10 betas -> 77 / 85 DoF
        x0 = lm(residual_fn(tb, mv, 2), torch.zeros(B, npose + 2 + 6, dtype=dtype), markers, mask, steps_stage0, lr_stage0, 0.01, t0)
        x1 = torch.cat([x0[:, :npose], x0[:, npose:npose + 2], torch.zeros(B, nbt - 2, dtype=dtype), x0[:, npose + 2:]], 1)
        t1 = [] if trace is not None else None
        x1 = lm(residual_fn(tb, mv, nbt), x1, markers, mask, steps_stage1, lr_stage1, 1e-3, t1)
        if trace is not None:
            trace.extend([t0, t1])
        pose, betas, orient, transl = x1[:, :npose], x1[:, npose:npose + nbt], x1[:, npose + nbt:npose + nbt + 3], x1[:, npose + nbt + 3:]
        v, j = smpl_forward(tb, betas, pose, orient, transl)
        return dict(pose=pose, betas=betas, orient=orient, transl=transl, verts=v, joints=j, x_stage0=x0)


def fit_smpl_adam(bm, marker_vids, markers, valid, steps_stage0=400, steps_stage1=800, lr=1e-2, trace=None, dtype=torch.float32):
    """src/models/fit_SMPL_Adam.py:68-225 after get_markers, restated literally: torch.optim.Adam on
    mse_loss(markers(x)[valid], target[valid]) (mean over the valid coordinates of the whole batch), autograd through the full-mesh
    LBS; stage 0 on betas[:2], stage 1 with a fresh optimizer on all betas.  `verts` are those of the LAST forward pass (the
    parameters before the final step), as the reference returns them (:221-225)."""
    tb = TorchBody(bm, dtype)
    mv = torch.as_tensor(np.asarray(marker_vids)).long()
    B = markers.shape[0]
    npose, nbt = 3 * (tb.J - 1), tb.S.shape[2]
    markers = markers.to(dtype)
    pose = torch.nn.Parameter(torch.zeros(B, npose, dtype=dtype))
    shape_opt = torch.nn.Parameter(torch.zeros(B, 2, dtype=dtype))
    shape_frozen = torch.zeros(B, nbt - 2, dtype=dtype)
    orient = torch.nn.Parameter(torch.zeros(B, 3, dtype=dtype))
    transl = torch.nn.Parameter(torch.zeros(B, 3, dtype=dtype))

    def run(params, betas_fn, steps):
        opt = torch.optim.Adam(params, lr=lr)
        verts = None
        for _ in range(steps):
            opt.zero_grad()
            verts, _ = lbs(tb, betas_fn(), torch.cat([orient, pose], 1), transl)
            loss = torch.nn.functional.mse_loss(verts[:, mv][valid], markers[valid])
            if trace is not None:
                trace.append(float(loss.detach()))
            loss.backward()
            opt.step()
        return verts

    verts = run([shape_opt, orient, pose, transl], lambda: torch.cat([shape_opt, shape_frozen], 1), steps_stage0)
    shape = torch.nn.Parameter(torch.cat([shape_opt, shape_frozen], 1).detach())
    v1 = run([shape, orient, pose, transl], lambda: shape, steps_stage1)
    verts = v1 if v1 is not None else verts
    return dict(pose=pose.detach(), betas=shape.detach(), orient=orient.detach(), transl=transl.detach(), verts=None if verts is None else verts.detach())
