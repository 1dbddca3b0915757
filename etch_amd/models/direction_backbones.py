"""MI355X counterpart of /root/reference/src/models/direction_backbones.py: same module tree / state-dict
keys (BatchLinear, BatchMLP, DotProdAttention, MultiHeadAttention, StackedMHSA); dense layers run on the
fp32 matrix cores (etch_linear), the 60x60 attention core is a dedicated kernel."""
import torch
from torch import nn

from .. import ops
from ..vgtk_so3conv import _Derived


class BatchLinear(nn.Linear):
    """direction_backbones.py:6-35."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features=in_features, out_features=out_features, bias=bias)
        nn.init.xavier_normal_(self.weight, gain=1)
        if bias:
            nn.init.constant_(self.bias, 0.0)

    def forward(self, x):
        nf, ni = x.shape[0], x.shape[1]
        y = ops.linear(x.reshape(nf * ni, self.in_features), self.weight.detach(), bias=None if self.bias is None else self.bias.detach())
        return y.view(nf, ni, self.out_features)


class BatchMLP(nn.Module):
    """direction_backbones.py:38-75."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.net = nn.Sequential(nn.Linear(in_features, out_features), nn.ReLU(), nn.Linear(out_features, out_features))

    def forward(self, x):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        h = ops.linear(x2, self.net[0].weight.detach(), bias=self.net[0].bias.detach(), act="relu")
        y = ops.linear(h, self.net[2].weight.detach(), bias=self.net[2].bias.detach())
        return y.view(*shp[:-1], self.out_features)


class DotProdAttention(nn.Module):
    """direction_backbones.py:78-129 (parameter-free here: linear_transform=False on the path)."""

    def __init__(self, embedding_dim, values_dim, linear_transform=False):
        super().__init__()
        assert not linear_transform
        self.embedding_dim, self.values_dim = embedding_dim, values_dim


class MultiHeadAttention(nn.Module):
    """direction_backbones.py:132-194.  forward(keys, queries, values) with keys is queries is values (self-attention over the 60
    anchor tokens, 8 heads of embedding_dim / 8).  The ETCH release (embedding 64 -> heads of 8) runs one fused kernel per layer
    (etch_mhsa_layer); other widths (encoder depths 1 / 3 / 4: 32 / 128 / 256, models_pointcloud.py:34-48) and value_dim != embedding_dim
    run the QKV GEMM + attention kernel + head_combine GEMM chain."""

    SUPPORTED_DIMS = (32, 64, 128, 256)

    def __init__(self, embedding_dim, value_dim, num_heads):
        super().__init__()
        if num_heads != 8 or embedding_dim not in self.SUPPORTED_DIMS:
            raise NotImplementedError(f"MultiHeadAttention: 8 heads over embedding_dim in {self.SUPPORTED_DIMS} are built (got {num_heads} heads, "
                                      f"{embedding_dim} dims)")
        self.embedding_dim, self.num_heads, self.value_dim = embedding_dim, num_heads, value_dim
        self.head_size = embedding_dim // num_heads
        self.key_transform = BatchLinear(embedding_dim, embedding_dim, bias=False)
        self.query_transform = BatchLinear(embedding_dim, embedding_dim, bias=False)
        self.value_transform = BatchLinear(embedding_dim, embedding_dim, bias=False)
        self.attention = DotProdAttention(embedding_dim=embedding_dim, values_dim=embedding_dim, linear_transform=False)
        self.head_combine = BatchLinear(embedding_dim, value_dim)
        self._d = _Derived()

    def _wqkv(self):
        ws = (self.query_transform.weight, self.key_transform.weight, self.value_transform.weight)
        return self._d.get(ws, lambda: torch.cat([w.detach() for w in ws], 0).contiguous())

    def heads(self, x):
        """x [T*60, E] -> the concatenated head outputs [T*60, E] (before head_combine)."""
        E = self.embedding_dim
        T = x.shape[0] // 60
        if E == 64 and x.is_contiguous():
            return ops.mhsa_layer(x, self.query_transform.weight.detach(), self.key_transform.weight.detach(), self.value_transform.weight.detach(),
                                  mode=2)
        qkv = ops.linear(x, self._wqkv())                               # [T*60, 3E] = q | k | v
        return ops.mhsa_attention(qkv, T, 0, E, 2 * E, embedding_dim=E)

    def forward(self, keys, queries, values, residual=False):
        assert keys is queries and keys is values and keys.shape[1] == 60
        T = keys.shape[0]
        x = keys.reshape(T * 60, self.embedding_dim)
        if self.embedding_dim == 64 and self.value_dim == 64 and x.is_contiguous():
            # the whole layer in one kernel: q|k|v and the attention output never leave the chip
            y = ops.mhsa_layer(x, self.query_transform.weight.detach(), self.key_transform.weight.detach(), self.value_transform.weight.detach(),
                               self.head_combine.weight.detach(), self.head_combine.bias.detach(), mode=0 if residual else 1)
            return y.view(T, 60, self.value_dim)
        att = self.heads(x)
        y = ops.linear(att, self.head_combine.weight.detach(), bias=self.head_combine.bias.detach(),
                       res=x if residual else None, res_mode=2 if residual else 0)
        return y.view(T, 60, self.value_dim)


class StackedMHSA(nn.Module):
    """direction_backbones.py:197-223."""

    def __init__(self, embedding_dim, value_dim, num_heads, num_layers):
        super().__init__()
        layers = [MultiHeadAttention(embedding_dim, embedding_dim, num_heads) for _ in range(num_layers - 1)]
        layers.append(MultiHeadAttention(embedding_dim, value_dim, num_heads))
        self.self_attention_layers = nn.ModuleList(layers)
        self.num_layers = num_layers

    def forward(self, point_feats):
        for n, layer in enumerate(self.self_attention_layers):
            # residual (point_feats + new) fused into the head_combine epilogue for all but the last layer
            point_feats = layer(point_feats, point_feats, point_feats, residual=(n != self.num_layers - 1))
        return point_feats
