"""MI355X counterparts of the vgtk.so3conv operator API used by the ETCH encoder
(/root/reference/external/vgtk/vgtk/so3conv/{base,modules,functional}.py, vgtk/pc/sample.py).

Same class names, constructor arguments, parameter / buffer names and tensor shapes at the API
surface; inside, activations are kept channels-last ([b, p, 60, c]) and every op is a HIP kernel.
"""
import math

import torch
import torch.nn as nn

from . import constants as K
from . import ops
from . import vgtk_functional as functional  # noqa: F401  (the reference's `vgtk.so3conv.functional`)


class SphericalPointCloud:
    """base.py:4-20.  `feats` has the reference layout [b, c, p, a] (a view of the channels-last storage)."""

    def __init__(self, xyz, feats, anchors, feats_cl=None):
        self._xyz = xyz
        self._anchors = anchors
        if feats_cl is None:
            feats_cl = feats.permute(0, 2, 3, 1).contiguous()
        self._fcl = feats_cl

    @property
    def xyz(self):
        return self._xyz

    @property
    def feats(self):
        return self._fcl.permute(0, 3, 1, 2)

    @property
    def feats_cl(self):
        return self._fcl

    @property
    def anchors(self):
        return self._anchors


# ------------------------------------------------------------------ vgtk.pc.sample
def batch_gather(x, idx, dim=1):
    return ops.gather_points_forward(x, idx.int())


def group_nd(pc, idx):
    """pc/sample.py:50-54."""
    b = idx.shape[0]
    out = batch_gather(pc, idx.reshape(b, -1).contiguous(), dim=2)
    return out.view(b, -1, *idx.shape[1:])


def ball_query_index(query_points, support_points, radius, n_sample):
    """pc/sample.py:58-71 (without the debug prints)."""
    return ops.ball_query(query_points, support_points, radius, n_sample)


_IDENTITY_SAMPLES = {}


def furthest_sample_index(pc, n_sample, lazy_sample):
    """pc/sample.py:75-85."""
    if pc.shape[2] == n_sample or lazy_sample:
        nb = pc.shape[0]
        key = (nb, int(n_sample), str(pc.device))
        v = _IDENTITY_SAMPLES.get(key)
        if v is None:            # (a per-shape constant: two launches per conv and forward otherwise; read-only by contract)
            if len(_IDENTITY_SAMPLES) > 64:
                _IDENTITY_SAMPLES.clear()
            v = _IDENTITY_SAMPLES[key] = torch.arange(n_sample, device=pc.device, dtype=torch.int32).view(1, -1).expand(nb, -1).contiguous()
        return v
    return ops.furthest_point_sampling(pc, n_sample)


def furthest_sample(pc, n_sample, lazy_sample=True):
    """pc/sample.py:88-90."""
    idx = furthest_sample_index(pc, n_sample, lazy_sample)
    return idx, group_nd(pc, idx)


# ------------------------------------------------------------------ modules
class _Derived:
    """Device-side derived constants (permuted weights, rotated kernels), rebuilt when a parameter changes."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, tensors, builder):
        key = tuple((t.data_ptr(), t._version, t.device) for t in tensors)
        if key != self._key:
            with torch.no_grad():
                self._val = builder()
            self._key = key
        return self._val


class BasicSO3Conv(nn.Module):
    """modules.py:19-39: W [c_out, c_in*ks], bias [1, c_out, 1]; forward on a materialised [b, c1, k, p, a]."""

    def __init__(self, dim_in, dim_out, kernel_size):
        super().__init__()
        self.dim_in, self.dim_out, self.kernel_size = dim_in, dim_out, kernel_size
        W = torch.empty(dim_out, dim_in, kernel_size)
        nn.init.xavier_normal_(W, gain=nn.init.calculate_gain("relu"))
        self.register_parameter("W", nn.Parameter(W.view(dim_out, dim_in * kernel_size)))
        self.register_parameter("bias", nn.Parameter((torch.zeros(dim_out) + 1e-3).view(1, dim_out, 1)))

    def forward(self, x):
        bs, npnt, na = x.shape[0], x.shape[3], x.shape[4]
        xr = x.reshape(bs, self.dim_in * self.kernel_size, npnt * na).permute(0, 2, 1).contiguous()
        y = ops.linear(xr, self.W.detach(), bias=self.bias.detach().reshape(-1))
        return y.view(bs, npnt, na, self.dim_out).permute(0, 3, 1, 2)


class InterSO3Conv(nn.Module):
    """modules.py:92-128.  forward(x, inter_idx=None, inter_w=None) -> (inter_idx, inter_w, sample_idx, SphericalPointCloud).

    The kernel-weight tensor `inter_w` is never materialised (it is regenerated inside the fused
    kernel), so the second return value is always None; the reference never reuses it across convs in
    the ETCH configuration (so3conv.py:126,132-133)."""

    def __init__(self, dim_in, dim_out, kernel_size, stride, radius, sigma, n_neighbor, lazy_sample=True, pooling=None, kanchor=60):
        super().__init__()
        assert kanchor == 60 and pooling is None, "ETCH uses kanchor=60, pooling=None (so3net.py:109, models_pointcloud.py:46-48)"
        kernels = K.get_kernel_points(radius, kernel_size)
        self.dim_in, self.dim_out = dim_in, dim_out
        self.kernel_size = kernels.shape[0]
        self.stride, self.radius, self.sigma = stride, radius, sigma
        self.n_neighbor, self.lazy_sample, self.pooling = n_neighbor, lazy_sample, pooling
        self.basic_conv = BasicSO3Conv(dim_in, dim_out, self.kernel_size)
        self.register_buffer("anchors", torch.from_numpy(K.get_anchors(kanchor)))
        self.register_buffer("kernels", torch.from_numpy(kernels))
        self._d = _Derived()
        self._drk = _Derived()
        self._d32 = _Derived()
        self._dq = _Derived()
        self._dqn = _Derived()
        self._dq32 = _Derived()
        self._dkq = _Derived()
        self._dqh = _Derived()

    def rotated_kernels(self):
        """Rotated kernel points exactly as functional.py:296 (CPU matmul, then uploaded) [60, 24, 3]; constants of the module: not rebuilt when W changes
        (a training step would pay a device -> host round trip per conv, and could not be captured into a graph)."""
        def build_rk():
            rk = torch.matmul(self.anchors.cpu(), self.kernels.cpu().transpose(0, 1)).permute(0, 2, 1).contiguous()
            return rk.to(self.anchors.device)

        return self._drk.get((self.anchors, self.kernels), build_rk)

    def _derived(self):
        W, bias = self.basic_conv.W, self.basic_conv.bias

        def build():
            Wd = W.detach().contiguous()
            Wp = ops.inter_weight_frag(Wd, self.dim_in, self.kernel_size) if self.dim_in % 16 == 0 else None
            return Wd, Wp, bias.detach().reshape(-1).contiguous()

        return (self.rotated_kernels(),) + self._d.get((W, bias), build)

    def _wp32(self):
        """Weight in the fragment order of the 32x32x2 kernel (csrc/so3conv32.hip), for the widths it covers; else None."""
        if (self.dim_in, self.dim_out) not in ops.INTER_MFMA32_SHAPES or self.kernel_size != 24:
            return None
        W = self.basic_conv.W
        return self._d32.get((W,), lambda: ops.inter_weight_frag32(W.detach().contiguous(), self.dim_in, self.kernel_size))

    def _wq(self):
        """Weight as three bf16 planes in the fragment order of the split-operand kernel (etch_inter_so3conv_split); None for cin < 16."""
        if self.dim_in % 16 != 0 or self.dim_out % 16 != 0 or self.kernel_size != 24:
            return None
        W = self.basic_conv.W
        return self._dq.get((W,), lambda: ops.inter_weight_split(W.detach().contiguous(), self.dim_in, self.kernel_size))

    def _wqn(self):
        """The same planes in W's natural column order: etch_inter_so3conv_planes (both contractions on the bf16 matrix cores); None where
        that kernel has no instantiation."""
        if not ops.inter_planes_supported(self.dim_in, self.dim_out, self.n_neighbor) or self.kernel_size != 24:
            return None
        W = self.basic_conv.W
        return self._dqn.get((W,), lambda: ops.inter_weight_split(W.detach().contiguous(), self.dim_in, self.kernel_size, natural=True))

    def _wq32(self):
        """The planes in the physical contraction order of the 32x32x16 kernel (etch_inter_so3conv_planes32); None where it has no instantiation."""
        if not (self.wants_planes() and ops.inter_planes_form(self.dim_in) == 32):
            return None
        W = self.basic_conv.W
        return self._dq32.get((W,), lambda: ops.inter_weight_split32(W.detach().contiguous(), self.dim_in, self.kernel_size))

    def _wqh(self):
        """The two fp16 planes of 2^6 W in the physical contraction order of the 32x32x16 kernels (etch_inter_so3conv_planes_kq); None where unused."""
        if not (ops.INTER_KQ and self.wants_planes()):
            return None
        W = self.basic_conv.W
        return self._dqh.get((W,), lambda: ops.inter_weight_split32_f16(W.detach().contiguous(), self.dim_in, self.kernel_size))

    def _kq(self):
        """Kernel-point factor of the weights' pre-activation (etch_inter_so3conv_planes_kq); None where that kernel is not used."""
        if not (ops.INTER_KQ and self.wants_planes()):
            return None
        rk = self.rotated_kernels()
        return self._dkq.get((self.anchors, self.kernels), lambda: ops.inter_kpoint_operand(rk, self.sigma))

    def wants_planes(self):
        """Falsy, or the plane format this conv gathers its input in (its producer should emit it: SeparableSO3ConvBlock.emit_planes):
        "f16" (two fp16 planes, etch_inter_so3conv_planes_kq) / "bf16" (three bf16 planes, the round-4 kernels)."""
        if not (ops.inter_planes_supported(self.dim_in, self.dim_out, self.n_neighbor) and self.kernel_size == 24):
            return False
        if ops.INTER_KQ:
            return "f16"
        from . import _lib
        return "bf16" if _lib.has_experiments() else False        # the round-4 planes kernels exist only in an ETCH_BUILD_EXPERIMENTS library (round 6)

    def group(self, xyz):
        """functional.py:176-185 inter_spconv_grouping_ball (index part): -> ball_idx, sample_idx, new_xyz.
        Coordinates only, so the model may have issued it ahead of time on its index stream: inside an active
        pointops.knn_scope the result is memoised per (conv, xyz tensor)."""
        from .models import pointops

        def compute():
            n_sample = math.ceil(xyz.shape[2] / self.stride)
            sidx, new_xyz = furthest_sample(xyz, n_sample, self.lazy_sample)
            ball = ball_query_index(new_xyz, xyz, self.radius, self.n_neighbor)
            return ball, sidx, new_xyz
        return pointops._memo(("epn_group", id(self), xyz.data_ptr(), tuple(xyz.shape)), (xyz,), compute)

    spatial_schedule = True   # walk the output points along a Morton curve (scheduling only: 3x less HBM traffic, same results)

    def order(self, new_xyz):
        """Processing order of the output points for the fused kernel; coordinates only, memoised like group() (and issued by
        the model's index stream together with it)."""
        if not self.spatial_schedule or self.dim_in < 16:
            return None
        from .models import pointops
        return pointops._memo(("epn_order", new_xyz.data_ptr(), tuple(new_xyz.shape)), (new_xyz,), lambda: ops.spatial_order(new_xyz))

    def forward(self, x, inter_idx=None, inter_w=None):
        xyz = x.xyz
        if inter_idx is None:
            inter_idx, sample_idx, new_xyz = self.group(xyz)
        else:
            sample_idx, new_xyz = None, xyz
        rk, W, Wp, bias = self._derived()
        y, stats = ops.inter_so3conv(xyz, new_xyz, inter_idx, x.feats_cl, rk, W, Wp, bias, self.sigma, order=self.order(new_xyz), want_stats=True, Wp32=self._wp32(), Wq=self._wq(),
                                     Wqn=None if ops.inter_planes_form(self.dim_in) == 32 else self._wqn(), Wq32=self._wq32(), kq=self._kq(), Wqh=self._wqh(), feats_planes=getattr(x, "feats_planes", None))
        cloud = SphericalPointCloud(new_xyz, None, self.anchors, feats_cl=y)
        cloud.in_stats = stats          # InstanceNorm (mean, rstd) of the output, a by-product of the conv's epilogue
        return inter_idx, None, sample_idx, cloud


class IntraSO3Conv(nn.Module):
    """modules.py:131-153."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        intra_idx = K.get_intra_idx()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.kernel_size = intra_idx.shape[1]
        self.basic_conv = BasicSO3Conv(dim_in, dim_out, self.kernel_size)
        self.register_buffer("anchors", torch.from_numpy(K.get_anchors()))
        self.register_buffer("intra_idx", torch.from_numpy(intra_idx).long())
        self._d = _Derived()
        self._wq = None
        self._wqh = None

    def _derived(self):
        W, bias = self.basic_conv.W, self.basic_conv.bias

        def build():
            ks, c = self.kernel_size, self.dim_in
            # kernel K order is tap-major: W2[o, tap*c + ch] = W[o, ch*12 + tap]
            W2 = W.detach().view(self.dim_out, c, ks).permute(0, 2, 1).reshape(self.dim_out, ks * c).contiguous()
            Wp32 = ops.permute_weight_frag32(W2) if (c == self.dim_out and c in (32, 64)) else None       # the 32x32x2 kernel's fragment order
            self._wq = ops.intra_weight_split(W2) if (c == self.dim_out and c in (32, 64)) else None       # the weight-stationary split kernel's
            self._wqh = ops.intra_weight_split_f16(W2) if (c == self.dim_out and c in (32, 64)) else None  # ... and its two-plane fp16 form's
            return ops.permute_weight_frag(W2), bias.detach().reshape(-1).contiguous(), self.intra_idx.to(torch.int32).contiguous(), Wp32

        return self._d.get((W, bias, self.intra_idx), build)

    def forward(self, x, mean=None, rstd=None, want_stats=False):
        Wp, bias, idx32, Wp32 = self._derived()
        if want_stats:
            y, stats = ops.intra_so3conv(x.feats_cl, idx32, Wp, bias, self.dim_out, mean, rstd, want_stats=True, Wp32=Wp32, Wq=self._wq, Wqh=self._wqh)
            cloud = SphericalPointCloud(x.xyz, None, self.anchors, feats_cl=y)
            cloud.in_stats = stats      # InstanceNorm (mean, rstd) of the output, a by-product of the conv's epilogue
            return cloud
        y = ops.intra_so3conv(x.feats_cl, idx32, Wp, bias, self.dim_out, mean, rstd, Wp32=Wp32, Wq=self._wq, Wqh=self._wqh)
        return SphericalPointCloud(x.xyz, None, self.anchors, feats_cl=y)


def get_anchors(k=60):
    return K.get_anchors(k)


def get_intra_idx():
    return K.get_intra_idx()
