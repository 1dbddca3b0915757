"""GPU parity of the index kernels against the oracle: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu


def scan(seed, n, sigma=(0.14, 0.31, 0.085)):
    return (np.random.default_rng(seed).standard_normal((n, 3)) * np.array(sigma)).astype(np.float32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n,m,sigma", [(5000, 2500, (0.14, 0.31, 0.085)), (5000, 2500, (0.20, 0.45, 0.12)), (1024, 512, (0.14, 0.31, 0.085)),
                                       (160, 80, (0.14, 0.31, 0.085)), (37, 20, (0.5, 0.5, 0.5)), (9000, 300, (0.14, 0.31, 0.085)),
                                       (20000, 200, (0.14, 0.31, 0.085)), (12000, 700, (0.14, 0.31, 0.085)), (20000, 10000, (0.14, 0.31, 0.085)),
                                       (24000, 150, (0.14, 0.31, 0.085))])
@pytest.mark.parametrize("split", [None, 2, 4, 8])        # None: the dispatch's own choice; G: every scan forced over G workgroups
def test_fps_vgtk(n, m, sigma, split):
    from etch_amd import ops
    if split is not None and n > 8 * 1024 * split:
        pytest.skip("more than 8 chunks of 1024 points per workgroup: not a split shape")
    x = np.stack([scan(1000 + b, n, sigma).T for b in range(3)])
    idx = ops.furthest_point_sampling(dev(x), m, split=split)
    got = idx.cpu().numpy()
    assert np.array_equal(got, O.furthest_point_sampling(x, m))
    assert not ops.fps_split_failed(idx)


@pytest.mark.parametrize("split", [None, 2, 4, 8])
def test_fps_vgtk_ties_and_origin_skip(split):
    from etch_amd import ops
    rng = np.random.default_rng(3)
    for n in (37, 64, 200, 1500, 2600, 4100, 9000, 17000):
        if split is not None and n > 8 * 1024 * split:
            continue
        pts = rng.integers(-3, 4, (2, n, 3)).astype(np.float32) * 0.25 + 0.125
        pts[:, 5] = 0.0      # origin point: never a candidate
        pts[:, 7] = 0.01
        x = np.ascontiguousarray(pts.transpose(0, 2, 1))
        got = ops.furthest_point_sampling(dev(x), n // 2, split=split).cpu().numpy()
        assert np.array_equal(got, O.furthest_point_sampling(x, n // 2)), n


@pytest.mark.parametrize("n,m,r,ns", [(5000, 2500, 0.08, 64), (2500, 2500, 0.11313708498984763, 32), (2500, 1250, 0.16, 64),
                                      (1250, 1250, 0.16, 32), (300, 100, 0.05, 16), (70, 70, 10.0, 8), (5000, 64, 0.4, 100)])
def test_ball_query(n, m, r, ns):
    from etch_amd import ops
    for sigma in ((0.14, 0.31, 0.085), (0.20, 0.45, 0.12)):
        x = np.stack([scan(2000 + b, n, sigma).T for b in range(2)])
        q = np.ascontiguousarray(x[:, :, :m])
        got = ops.ball_query(dev(q), dev(x), r, ns).cpu().numpy()
        assert np.array_equal(got, O.ball_query(q, x, r, ns))


def test_ball_query_known_answers():
    from etch_amd import ops
    sup = np.array([[1, 0, 0], [0, 0, 0], [0.01, 0, 0], [2, 0, 0], [0.02, 0, 0]], np.float32).T[None]
    q = np.array([[0, 0, 0], [9, 9, 9]], np.float32).T[None]
    f = lambda ns: ops.ball_query(dev(q), dev(sup), 0.1, ns).cpu().numpy().tolist()
    assert f(8) == [[[1, 2, 4, 1, 2, 4, 1, 2], [0] * 8]]
    assert f(4) == [[[1, 2, 4, 0], [0] * 4]]
    assert f(3) == [[[1, 2, 4], [0] * 3]]


def test_gather_points():
    from etch_amd import ops
    rng = np.random.default_rng(0)
    pts = rng.standard_normal((3, 3, 5001)).astype(np.float32)
    idx = rng.integers(0, 5001, (3, 160000)).astype(np.int32)
    got = ops.gather_points_forward(dev(pts), dev(idx)).cpu().numpy()
    assert np.array_equal(got, O.gather_points_forward(pts, idx))


@pytest.mark.parametrize("k,segs,qsegs", [(8, [5000, 5000], None), (16, [1250, 1250, 1250], [312, 312, 312]), (16, [312, 300], None),
                                          (16, [19, 19], None), (3, [1250, 1250], [5000, 5000]), (3, [19, 19], [78, 78]), (16, [78], [19]),
                                          (16, [5, 40], None)])
def test_knn(k, segs, qsegs):
    from etch_amd import ops
    p = np.concatenate([scan(3000 + i, n) for i, n in enumerate(segs)])
    o = np.cumsum(segs).astype(np.int32)
    if qsegs is None:
        q, qo = p, o
    else:
        q = np.concatenate([scan(4000 + i, n) for i, n in enumerate(qsegs)])
        qo = np.cumsum(qsegs).astype(np.int32)
    ri, rd2 = O.knnquery(k, p, q, o, qo)
    for wave_kernel in (True, False):
        idx, dist = ops.knnquery(k, dev(p), dev(q), dev(o), dev(qo), wave_kernel=wave_kernel)
        assert np.array_equal(idx.cpu().numpy(), ri), wave_kernel
        assert np.array_equal(dist.cpu().numpy(), np.sqrt(rd2)), wave_kernel


def test_knn_exact_ties_follow_heap_order():
    from etch_amd import ops
    rng = np.random.default_rng(5)
    p = (rng.integers(-2, 3, (400, 3)).astype(np.float32)) * 0.5   # lattice: many exactly tied distances + duplicates
    o = np.array([150, 400], np.int32)
    for k in (3, 8, 16):
        ri, rd2 = O.knnquery(k, p, p, o, o)
        for wave_kernel in (True, False):
            idx, dist = ops.knnquery(k, dev(p), dev(p), dev(o), dev(o), sqrt=False, wave_kernel=wave_kernel)
            assert np.array_equal(idx.cpu().numpy(), ri), (k, wave_kernel)
            assert np.array_equal(dist.cpu().numpy(), rd2), (k, wave_kernel)


@pytest.mark.parametrize("segs,stride", [([5000, 5000, 5000], 4), ([1250, 1250], 4), ([312, 312], 4), ([78, 78, 78], 4), ([19, 19], 4),
                                         ([150, 90], 4), ([4999, 130, 2047], 3), ([20000, 20000], 4), ([11000, 300, 20480], 4)])
@pytest.mark.parametrize("split", [None, 2, 4, 8])
def test_fps_pointops(segs, stride, split):
    from etch_amd import ops
    if split is not None and max(segs) > 8 * 1024 * split:
        pytest.skip("more than 8 chunks of 1024 points per workgroup: not a split shape")
    p = np.concatenate([scan(5000 + i, n) for i, n in enumerate(segs)])
    o = np.cumsum(segs).astype(np.int32)
    no = np.cumsum([n // stride for n in segs]).astype(np.int32)
    got = ops.furthestsampling(dev(p), dev(o), dev(no), split=split).cpu().numpy()
    assert np.array_equal(got, O.furthestsampling(p, o, no))


def test_fps_split_dispatch_and_failure_is_loud():
    """The dispatch never splits by itself (measured slower at every size the path runs, profiles/r04_fps_split.txt); forced, a workgroup that
    never arrives makes every workgroup of its scan give up within the spin bound: the fail word is set and the scan's remaining indices are
    INT_MIN (a later gather with them faults instead of silently sampling garbage)."""
    import ctypes

    from etch_amd import _lib, ops
    assert ops.fps_split_default(32, 5000) == 1 and ops.fps_split_default(64, 20000) == 1 and ops.fps_split_default(1, 20000) == 1
    x = np.stack([scan(1000 + b, 12000).T for b in range(2)])
    ref = O.furthest_point_sampling(x, 600)
    lib = _lib.lib()
    try:
        _lib.check(lib.etch_fps_split_debug(ctypes.c_uint(2000), 1), "etch_fps_split_debug")      # workgroup 1 of every scan returns at once
        idx = ops.furthest_point_sampling(dev(x), 600, split=4)
        assert ops.fps_split_failed(idx)
        got = idx.cpu().numpy()
        assert (got[:, 0] == 0).all() and (got[:, 1:] == np.iinfo(np.int32).min).all()
    finally:
        _lib.check(lib.etch_fps_split_debug(ctypes.c_uint(0), -1), "etch_fps_split_debug")
    idx = ops.furthest_point_sampling(dev(x), 600, split=4)
    assert not ops.fps_split_failed(idx) and np.array_equal(idx.cpu().numpy(), ref)


def test_pybind_module_mirrors():
    """Same call conventions as the reference's pybind modules (caller-allocated outputs for pointops)."""
    from etch_amd import epn_gathering, epn_grouping, pointops_cuda
    x = np.stack([scan(1, 600).T])
    xt = dev(x)
    i = epn_grouping.furthest_point_sampling(xt, 300)
    q = epn_gathering.gather_points_forward(xt, i)
    bq = epn_grouping.ball_query(q, xt, 0.1, 16)
    assert np.array_equal(bq.cpu().numpy(), O.ball_query(q.cpu().numpy(), x, 0.1, 16))
    p = dev(scan(2, 500))
    o = torch.tensor([200, 500], dtype=torch.int32).cuda()
    idx = torch.zeros(500, 8, dtype=torch.int32).cuda()
    d2 = torch.zeros(500, 8).cuda()
    pointops_cuda.knnquery_cuda(500, 8, p, p, o, o, idx, d2)
    ri, rd = O.knnquery(8, p.cpu().numpy(), p.cpu().numpy(), o.cpu().numpy(), o.cpu().numpy())
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(d2.cpu().numpy(), rd)
    no = torch.tensor([50, 125], dtype=torch.int32).cuda()
    out = torch.zeros(125, dtype=torch.int32).cuda()
    pointops_cuda.furthestsampling_cuda(2, 300, p, o, no, torch.empty(500).cuda(), out)
    assert np.array_equal(out.cpu().numpy(), O.furthestsampling(p.cpu().numpy(), o.cpu().numpy(), no.cpu().numpy()))


@pytest.mark.parametrize("b,c,n,m,dup", [(3, 3, 5000, 2500, False), (2, 3, 700, 900, True), (1, 17, 129, 64, True), (2, 3, 50, 0, False)])
def test_gather_points_backward_bit_exact(b, c, n, m, dup):
    """epn_gathering.gather_points_backward vs the C oracle (ascending-index summation order on both sides) and vs torch autograd
    of the forward gather."""
    from etch_amd import epn_gathering
    from oracle import ops as O
    rng = np.random.default_rng(b * 100 + c + n)
    if dup:
        idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    else:
        idx = np.stack([rng.permutation(n)[:m] for _ in range(b)]).astype(np.int32).reshape(b, m)
    g = rng.standard_normal((b, c, m)).astype(np.float32)
    got = epn_gathering.gather_points_backward(torch.from_numpy(g).cuda(), torch.from_numpy(idx).cuda(), n)
    want = O.gather_points_backward(g, idx, n)
    assert got.shape == (b, c, n)
    assert np.array_equal(got.cpu().numpy(), want)
    if m:
        x = torch.zeros(b, c, n, dtype=torch.float64, requires_grad=True)
        torch.gather(x, 2, torch.from_numpy(idx).long()[:, None, :].expand(b, c, m)).backward(torch.from_numpy(g).double())
        assert np.abs(got.cpu().numpy() - x.grad.numpy()).max() < 1e-5


def test_knn_eviction_tie_straddling_the_kth_rank():
    """Two DIFFERENT support points exactly tied at the current k-th distance when a closer candidate arrives: the reference's reheap
    evicts whichever of them sits at the heap root, the register list its last entry.  If only one of the two survives to the end no
    tie is visible in the final list any more -- the kernel has to notice the ambiguous eviction when it happens (and redo the query
    with the literal heap).  Constructed cases + a random lattice sweep where such evictions are frequent."""
    from etch_amd import ops
    # query at the origin; k = 3.  Support in index order: a (d=4), b and c (both d=9, different points), then closer points arrive.
    base = np.array([[2, 0, 0], [3, 0, 0], [0, 3, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.5], [0, 0, 0.5]], np.float32)
    for perm in ([0, 1, 2, 3, 4, 5, 6], [1, 2, 0, 3, 4, 5, 6], [2, 1, 0, 6, 5, 4, 3], [1, 0, 2, 5, 3, 6, 4]):
        p = base[perm]
        q = np.zeros((1, 3), np.float32)
        o, qo = np.array([len(p)], np.int32), np.array([1], np.int32)
        for k in (2, 3, 4, 5):
            ri, rd2 = O.knnquery(k, p, q, o, qo)
            idx, dist = ops.knnquery(k, dev(p), dev(q), dev(o), dev(qo), sqrt=False)
            assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(dist.cpu().numpy(), rd2), (perm, k)
    rng = np.random.default_rng(9)
    for trial in range(6):
        p = rng.integers(-3, 4, (700, 3)).astype(np.float32) * 0.25          # coarse lattice: ties everywhere, incl. at the k-th rank
        p = p[rng.permutation(len(p))]
        o = np.array([300, 700], np.int32)
        for k in (2, 5, 8, 16):
            ri, rd2 = O.knnquery(k, p, p, o, o)
            idx, dist = ops.knnquery(k, dev(p), dev(p), dev(o), dev(o), sqrt=False)
            assert np.array_equal(idx.cpu().numpy(), ri), (trial, k)
            assert np.array_equal(dist.cpu().numpy(), rd2), (trial, k)


@pytest.mark.parametrize("k", [17, 32, 64, 100])
def test_knn_large_nsample_up_to_the_reference_limit(k):
    """nsample up to 100, the capacity of the reference kernel's heap arrays (knnquery_cuda_kernel.cu:86-87): the literal LDS-heap scan."""
    from etch_amd import ops
    segs = [400, 150]
    p = np.concatenate([scan(3100 + i, n) for i, n in enumerate(segs)])
    o = np.cumsum(segs).astype(np.int32)
    q = np.concatenate([scan(4100 + i, n) for i, n in enumerate([37, 21])])
    qo = np.array([37, 58], np.int32)
    ri, rd2 = O.knnquery(k, p, q, o, qo)
    idx, dist = ops.knnquery(k, dev(p), dev(q), dev(o), dev(qo), sqrt=False)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(dist.cpu().numpy(), rd2)


def test_reference_launcher_symbols_without_host_offsets():
    """The reference's OWN launcher symbols (knnquery_cuda_kernel.h:14, sampling_cuda_kernel.h:14, + pointer-level forms of the vgtk
    launches) called through ctypes with device pointers only: the offsets are built on the device and never read back by the host;
    results must equal the oracle's and etch_knnquery's (which is handed the segment count and sizes by the host)."""
    import ctypes

    from etch_amd import _lib, ops
    L = _lib.lib()
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    segs, qsegs = [700, 1300, 64, 2000], [300, 41, 64, 500]
    p = np.concatenate([scan(5100 + i, n) for i, n in enumerate(segs)])
    q = np.concatenate([scan(6100 + i, n) for i, n in enumerate(qsegs)])
    dp, dq = dev(p), dev(q)
    o = torch.cumsum(torch.tensor(segs, dtype=torch.int32, device="cuda"), 0).int()          # device-resident offsets
    qo = torch.cumsum(torch.tensor(qsegs, dtype=torch.int32, device="cuda"), 0).int()
    m = int(sum(qsegs))
    for k in (1, 3, 8, 16, 40):
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda")
        d2 = torch.empty((m, k), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        L.knnquery_cuda_launcher(ci(m), ci(k), vp(dp.data_ptr()), vp(dq.data_ptr()), vp(o.data_ptr()), vp(qo.data_ptr()), vp(idx.data_ptr()),
                                 vp(d2.data_ptr()))
        torch.cuda.synchronize()
        ri, rd2 = O.knnquery(k, p, q, np.cumsum(segs).astype(np.int32), np.cumsum(qsegs).astype(np.int32))
        assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(d2.cpu().numpy(), rd2), k
        i2, dd = ops.knnquery(k, dp, dq, o, qo, sqrt=False)
        assert torch.equal(i2, idx) and torch.equal(dd, d2)
        # the pybind-module mirror takes the same route (no .tolist() of the offsets)
        from etch_amd import pointops_cuda
        idx3, d3 = torch.empty_like(idx), torch.empty_like(d2)
        pointops_cuda.knnquery_cuda(m, k, dp, dq, o, qo, idx3, d3)
        assert torch.equal(idx3, idx) and torch.equal(d3, d2)
    # furthestsampling_cuda_launcher(b, n_max, xyz, offset, new_offset, tmp, idx)
    no = torch.cumsum(torch.tensor([s // 4 for s in segs], dtype=torch.int32, device="cuda"), 0).int()
    mt = sum(s // 4 for s in segs)
    fidx = torch.zeros((mt,), dtype=torch.int32, device="cuda")
    tmp = torch.full((len(p),), 1e10, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    L.furthestsampling_cuda_launcher(ci(len(segs)), ci(max(segs)), vp(dp.data_ptr()), vp(o.data_ptr()), vp(no.data_ptr()), vp(tmp.data_ptr()),
                                     vp(fidx.data_ptr()))
    torch.cuda.synchronize()
    want = O.furthestsampling(p, np.cumsum(segs).astype(np.int32), np.cumsum([s // 4 for s in segs]).astype(np.int32))
    assert np.array_equal(fidx.cpu().numpy(), want)
    # vgtk launches, pointer level
    x = np.stack([scan(7000 + b, 900).T for b in range(2)])
    dx = dev(x)
    nq = 450
    fi = torch.zeros((2, nq), dtype=torch.int32, device="cuda")
    temp = torch.full((2, 900), 1e10, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    L.furthest_point_sampling_cuda_launcher(ci(2), ci(900), ci(nq), vp(dx.data_ptr()), vp(temp.data_ptr()), vp(fi.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(fi.cpu().numpy(), O.furthest_point_sampling(x, nq))
    new = torch.empty((2, 3, nq), dtype=torch.float32, device="cuda")
    L.gather_points_forward_cuda_launcher(ci(2), ci(3), ci(900), ci(nq), vp(dx.data_ptr()), vp(fi.data_ptr()), vp(new.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(new.cpu().numpy(), np.take_along_axis(x, fi.cpu().numpy()[:, None, :].astype(np.int64), 2))
    bi = torch.empty((2, nq, 32), dtype=torch.int32, device="cuda")
    L.ball_query_cuda_launcher(ci(2), ci(900), ci(nq), cf(0.12), ci(32), vp(new.data_ptr()), vp(dx.data_ptr()), vp(bi.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(bi.cpu().numpy(), O.ball_query(new.cpu().numpy(), x, 0.12, 32))
    g = torch.randn(2, 3, nq, device="cuda")
    gp = torch.empty((2, 3, 900), dtype=torch.float32, device="cuda")
    L.gather_points_backward_cuda_launcher(ci(2), ci(3), ci(900), ci(nq), vp(g.data_ptr()), vp(fi.data_ptr()), vp(gp.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(gp.cpu().numpy(), O.gather_points_backward(g.cpu().numpy(), fi.cpu().numpy(), 900))
