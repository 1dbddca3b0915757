"""MI355X counterpart of /root/reference/src/models/pointops.py (the functions on the inference path).

Same names and argument order.  Two host-side economies that do not change any result:
  * offsets are read back to the host once per tensor (the reference does `.item()` per scan), and
  * identical kNN queries are computed once per forward (the reference recomputes the same query twice per
    PointTransformerLayer, pointtransformer_seg.py:28-29, and again in every block of a level)."""
import torch

from .. import ops

_knn_cache = None  # dict while a Point-Transformer forward is active


def host_offsets(o):
    """Host copy of an offset tensor, memoised ON the tensor object (never by address: freed offsets get re-used)."""
    v = getattr(o, "_etch_host", None)
    if v is None or getattr(o, "_etch_host_version", -1) != o._version:
        v = [int(t) for t in o.tolist()]
        set_host_offsets(o, v)
    return v


def set_host_offsets(o, values):
    o._etch_host = list(values)
    o._etch_host_version = o._version
    return o


_offset_tensors = {}   # (values, device) -> device int32 tensor; offsets are a pure function of (B, N, strides)


def offsets_tensor(values, device):
    """Device int32 offsets carrying their host copy, built ONCE per distinct value list: uploading a small host list is a
    synchronous pageable copy that would stall the host behind every kernel already queued (one stall per level per forward)."""
    device = torch.device(device)
    key = (tuple(int(v) for v in values), device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _offset_tensors.get(key)
    if t is None:
        t = set_host_offsets(torch.tensor(list(key[0]), dtype=torch.int32, device=device), key[0])
        torch.cuda.current_stream(device).synchronize()      # first use only: visible to every stream from here on
        if len(_offset_tensors) > 256:
            _offset_tensors.clear()
        _offset_tensors[key] = t
    return t


class knn_scope:
    """Enables FPS / kNN / sub-sampling de-duplication for the duration of one forward.  Scopes nest: the model opens
    one around BOTH Point-Transformer nets, which see the same points and offsets and therefore the same indices."""

    def __enter__(self):
        global _knn_cache
        self._prev = _knn_cache
        if _knn_cache is None:
            _knn_cache = {}
        return self

    def __exit__(self, *a):
        global _knn_cache
        _knn_cache = self._prev


def _memo(key, keep, fn):
    if _knn_cache is None:
        return fn()
    hit = _knn_cache.get(key)
    if hit is None:
        hit = (fn(), keep)          # `keep` pins the key tensors so their addresses stay unique inside the scope
        _knn_cache[key] = hit
    return hit[0]


def make_offsets(values, device, like=None):
    """Device int32 offset tensor carrying its host copy: one shared tensor object per distinct offset list (both nets and the
    index prefetch see the same object per level; built once, see offsets_tensor)."""
    return offsets_tensor(values, device)


def furthestsampling(xyz, offset, new_offset):
    """pointops.py:10-28."""
    oh, noh = host_offsets(offset), host_offsets(new_offset)
    return _memo(("fps", xyz.data_ptr(), tuple(oh), tuple(noh)), (xyz,), lambda: ops.furthestsampling(xyz, offset, new_offset, oh, noh))


def gather_rows(x, idx):
    """n_p = p[idx.long(), :] (pointtransformer_seg.py:60), memoised so both nets share the sub-sampled point tensors."""
    return _memo(("rows", x.data_ptr(), idx.data_ptr()), (x, idx), lambda: ops.gather_rows(x, idx))


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """pointops.py:32-45: -> idx (m,nsample) int32, dist (m,nsample) = sqrt(d2)."""
    if new_xyz is None:
        new_xyz = xyz
    oh, noh = host_offsets(offset), host_offsets(new_offset)
    return _memo(("knn", nsample, xyz.data_ptr(), new_xyz.data_ptr(), tuple(oh), tuple(noh)), (xyz, new_xyz),
                 lambda: ops.knnquery(nsample, xyz, new_xyz, offset, new_offset, noh))


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True):
    """pointops.py:79-100: -> (m, nsample, 3+c) or (m, nsample, c)."""
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        idx = knnquery(nsample, xyz, new_xyz, offset, new_offset)[0]
    m, c = new_xyz.shape[0], feat.shape[1]
    g = ops.pt_group(xyz, new_xyz, feat, idx).view(m, nsample, 3 + c)
    return g if use_xyz else g[:, :, 3:]


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3, add_to=None):
    """pointops.py:164-178 (weights from NON-squared distances).  `add_to` fuses the `linear1(x1) + ...` of TransitionUp."""
    assert k == 3
    r = knnquery(k, xyz, new_xyz, offset, new_offset)
    idx, dist = r[0], r[1]
    if add_to is None:
        add_to = torch.zeros((new_xyz.shape[0], feat.shape[1]), dtype=torch.float32, device=feat.device)
    return ops.pt_interp_add(add_to, feat, idx, dist)
