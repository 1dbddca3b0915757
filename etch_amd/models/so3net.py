"""MI355X counterpart of /root/reference/src/models/so3net.py: `build_model` derives the per-layer
radii / sigma / neighbour counts (:36-152), `EquivBackbone` runs the blocks (:10-33)."""
import json

import torch
import torch.nn as nn

from .. import vgtk_so3conv as L
from . import so3conv as M


class EquivBackbone(nn.Module):
    def __init__(self, params, config=None):
        super().__init__()
        self.backbone = nn.ModuleList()
        for block_param in params["backbone"]:
            self.backbone.append(M.BasicSO3ConvBlock(block_param))
        self.na_in = params["na"]
        self.config = config
        # plain attribute (not a buffer), like the reference (so3net.py:21): absent from the state dict
        self.anchors = torch.from_numpy(L.get_anchors(60))

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.anchors = fn(self.anchors)
        return self

    def forward(self, x):
        sample_idx_lists = []
        if x.shape[-1] > 3:
            x = x.permute(0, 2, 1).contiguous()
        x = M.preprocess_input(x, self.na_in, False)
        for block in self.backbone:
            x, sample_idx_list = block(x)
            sample_idx_lists.append(sample_idx_list)
        return x, sample_idx_lists


def build_params(input_radius=0.4, input_num=1024, dropout_rate=0.0, kanchor=60, kpconv=False,
                 mlps=((32, 32), (64, 64), (128, 128), (256, 256)), strides=(2, 2, 2, 2), initial_radius_ratio=0.2,
                 sampling_ratio=0.8, sampling_density=0.5, kernel_multiplier=2, sigma_ratio=0.5, xyz_pooling=None):
    """The parameter dictionary of so3net.py:58-132 (what the reference dumps to EPN_model_setting_json)."""
    strides = list(strides)
    na = 1 if kpconv else kanchor
    if input_num > 1024:
        sampling_ratio /= input_num / 1024
        strides[0] = int(2 * (input_num / 1024))
    params = {"name": "Invariant SPConv Model", "backbone": [], "na": na}
    dim_in = 1
    n_layer = len(mlps)
    stride_current = 1
    stride_multipliers = [stride_current]
    for _ in range(n_layer):
        stride_current *= 2
        stride_multipliers += [stride_current]
    num_centers = [int(input_num / m) for m in stride_multipliers]
    radius_ratio = [initial_radius_ratio * m ** sampling_density for m in stride_multipliers]
    radii = [r * input_radius for r in radius_ratio]
    weighted_sigma = [sigma_ratio * radii[0] ** 2]
    for idx, s in enumerate(strides):
        weighted_sigma.append(weighted_sigma[idx] * s)
    for i, block in enumerate(mlps):
        block_param = []
        for j, dim_out in enumerate(block):
            lazy_sample = i != 0 or j != 0
            stride_conv = i == 0 or xyz_pooling != "stride"
            neighbor = int(sampling_ratio * num_centers[i] * radius_ratio[i] ** (1 / sampling_density))
            if i == 0 and j == 0:
                neighbor *= int(input_num / 1024)
            kernel_size = 1
            if j == 0:
                inter_stride = strides[i]
                nidx = i if i == 0 else i + 1
                if stride_conv:
                    neighbor *= 2
                    kernel_size = 1
            else:
                inter_stride = 1
                nidx = i + 1
            block_type = "inter_block" if na != 60 else "separable_block"
            block_param.append({"type": block_type, "args": {
                "dim_in": dim_in, "dim_out": dim_out, "kernel_size": kernel_size, "stride": inter_stride, "radius": radii[nidx],
                "sigma": weighted_sigma[nidx], "n_neighbor": neighbor, "lazy_sample": lazy_sample, "dropout_rate": dropout_rate,
                "multiplier": kernel_multiplier, "activation": "leaky_relu", "pooling": xyz_pooling, "kanchor": na}})
            dim_in = dim_out
        params["backbone"].append(block_param)
    return params


def build_model(opt, mlps=[[32, 32], [64, 64], [128, 128], [256, 256]], out_mlps=[128, 128], strides=[2, 2, 2, 2],
                initial_radius_ratio=0.2, sampling_ratio=0.8, sampling_density=0.5, kernel_multiplier=2, sigma_ratio=0.5,
                xyz_pooling=None, to_file=None):
    """so3net.py:36-152.  `opt` is the EPN cfg (attribute access: opt.model.input_num, .dropout_rate, .search_radius, .kpconv, .kanchor)."""
    params = build_params(opt.model.search_radius, opt.model.input_num, opt.model.dropout_rate, opt.model.kanchor, opt.model.kpconv,
                          mlps, strides, initial_radius_ratio, sampling_ratio, sampling_density, kernel_multiplier, sigma_ratio, xyz_pooling)
    if to_file is not None:
        with open(to_file, "w") as outfile:
            json.dump(params, outfile)
    return EquivBackbone(params, config=opt)
