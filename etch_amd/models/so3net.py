"""EPN backbone assembly for MI355X: counterpart of /root/reference/src/models/so3net.py.

`build_model(opt, mlps, strides, to_file)` keeps the reference's call signature (so3net.py:36-48) and returns an
`EquivBackbone` with the reference's module tree (`backbone.<i>.blocks.<j>...`).  The per-layer geometry (search
radius, kernel width sigma, neighbour budget, stride) is derived by `layer_geometry` below -- a table-driven
restatement of the arithmetic of so3net.py:58-132 that reproduces the reference's `EPN_model_setting_json` bit for bit
(checked against tests/golden/epn_model_setting.json).
"""
import json

import torch
import torch.nn as nn

from .. import vgtk_so3conv as L
from . import so3conv as M

_BLOCK_DEFAULTS = dict(kernel_size=1, multiplier=2, activation="leaky_relu")


def layer_geometry(widths, strides, search_radius, reference_points=1024, radius0=0.2, fill=0.8, density=0.5, sigma0=0.5):
    """Geometry of every conv of every block.

    Level l (0 = input resolution) has 2**l fewer centres than the input.  With s_l = 2**l:
      radius ratio  rho_l   = radius0 * s_l ** density              radius_l = rho_l * search_radius
      sigma_0 = sigma0 * radius_0**2,  sigma_{l+1} = sigma_l * stride_l
      neighbour budget of block i = int(fill * (reference_points / s_i) * rho_i ** (1 / density)), doubled for the
      (strided) first conv of a block, and scaled by reference_points / 1024 for the very first conv.
    The first conv of block i > 0 already searches at the radius of level i + 1; block 0's first conv at level 0.
    Yields one dict per conv, grouped per block.
    """
    strides = list(strides)
    if reference_points > 1024:
        fill /= reference_points / 1024
        strides[0] = int(2 * (reference_points / 1024))
    scale = [2 ** l for l in range(len(widths) + 1)]
    rho = [radius0 * s ** density for s in scale]
    radius = [r * search_radius for r in rho]
    sigma = [sigma0 * radius[0] ** 2]
    for st in strides:
        sigma.append(sigma[-1] * st)
    centres = [int(reference_points / s) for s in scale]
    c_prev = 1
    plan = []
    for i, block in enumerate(widths):
        convs = []
        for j, c_out in enumerate(block):
            budget = int(fill * centres[i] * rho[i] ** (1 / density))
            if (i, j) == (0, 0):
                budget *= int(reference_points / 1024)
            first = j == 0
            if first:
                budget *= 2
            level = (i if i == 0 else i + 1) if first else i + 1
            convs.append(dict(dim_in=c_prev, dim_out=c_out, stride=strides[i] if first else 1, radius=radius[level], sigma=sigma[level],
                              n_neighbor=budget, lazy_sample=(i, j) != (0, 0)))
            c_prev = c_out
        plan.append(convs)
    return plan


def build_params(input_radius=0.4, input_num=1024, dropout_rate=0.0, kanchor=60, kpconv=False,
                 mlps=((32, 32), (64, 64), (128, 128), (256, 256)), strides=(2, 2, 2, 2), initial_radius_ratio=0.2,
                 sampling_ratio=0.8, sampling_density=0.5, kernel_multiplier=2, sigma_ratio=0.5, xyz_pooling=None):
    """The parameter dictionary the reference dumps to `EPN_model_setting_json` (same keys, same key order, same floats)."""
    na = 1 if kpconv else kanchor
    kind = "separable_block" if na == 60 else "inter_block"
    geo = layer_geometry(mlps, strides, input_radius, input_num, initial_radius_ratio, sampling_ratio, sampling_density, sigma_ratio)
    backbone = []
    for convs in geo:
        backbone.append([{"type": kind, "args": {
            "dim_in": g["dim_in"], "dim_out": g["dim_out"], "kernel_size": _BLOCK_DEFAULTS["kernel_size"], "stride": g["stride"],
            "radius": g["radius"], "sigma": g["sigma"], "n_neighbor": g["n_neighbor"], "lazy_sample": g["lazy_sample"],
            "dropout_rate": dropout_rate, "multiplier": kernel_multiplier, "activation": _BLOCK_DEFAULTS["activation"],
            "pooling": xyz_pooling, "kanchor": na}} for g in convs])
    return {"name": "Invariant SPConv Model", "backbone": backbone, "na": na}


class EquivBackbone(nn.Module):
    """so3net.py:10-33: a list of BasicSO3ConvBlocks; forward returns (SphericalPointCloud, per-block sample-index lists)."""

    def __init__(self, params, config=None):
        super().__init__()
        self.backbone = nn.ModuleList(M.BasicSO3ConvBlock(bp) for bp in params["backbone"])
        self.na_in = params["na"]
        self.config = config
        self.anchors = torch.from_numpy(L.get_anchors(60))   # plain attribute like the reference (:21): not in the state dict
        # a separable block whose successor gathers bf16 planes (etch_inter_so3conv_planes) writes its output split as well
        convs = [c for blk in self.backbone for c in blk.blocks]
        for cur, nxt in zip(convs[:-1], convs[1:]):
            if isinstance(cur, M.SeparableSO3ConvBlock) and isinstance(nxt, M.SeparableSO3ConvBlock):
                cur.emit_planes = nxt.inter_conv.conv.wants_planes()

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.anchors = fn(self.anchors)
        return self

    def forward(self, x):
        if x.shape[-1] > 3:
            x = x.permute(0, 2, 1).contiguous()
        cloud = M.preprocess_input(x, self.na_in, False)
        picked = []
        for block in self.backbone:
            cloud, idx_list = block(cloud)
            picked.append(idx_list)
        return cloud, picked


def build_model(opt, mlps=[[32, 32], [64, 64], [128, 128], [256, 256]], out_mlps=[128, 128], strides=[2, 2, 2, 2],
                initial_radius_ratio=0.2, sampling_ratio=0.8, sampling_density=0.5, kernel_multiplier=2, sigma_ratio=0.5,
                xyz_pooling=None, to_file=None):
    """Same signature as the reference.  `opt` = EPN cfg with attribute access (opt.model.{input_num, dropout_rate, search_radius,
    kpconv, kanchor}); writes the parameter JSON to `to_file` as the reference does (:147-149)."""
    m = opt.model
    params = build_params(m.search_radius, m.input_num, m.dropout_rate, m.kanchor, m.kpconv, mlps, strides, initial_radius_ratio,
                          sampling_ratio, sampling_density, kernel_multiplier, sigma_ratio, xyz_pooling)
    if to_file is not None:
        with open(to_file, "w") as fh:
            json.dump(params, fh)
    return EquivBackbone(params, config=opt)
