"""Drop-in for the reference's native module svecalign/vecalign/dp_core.pyx: the same five
functions, same argument order and meaning, same error behaviour, evaluated by HIP kernels through
the C ABI (include/svx.h).  Arguments are numpy arrays (or torch CUDA tensors, which skip the
host<->device copies); results are numpy arrays like the reference's.

  make_dense_costs   dp_core.pyx:36-77    -> svx_dense_costs
  dense_dp           dp_core.pyx:79-141   -> svx_dense_dp
  score_path         dp_core.pyx:143-161  -> svx_score_path
  make_sparse_costs  dp_core.pyx:165-267  -> svx_sparse_costs
  sparse_dp          dp_core.pyx:269-404  -> svx_sparse_dp
"""
import ctypes

import numpy as np

from .. import _lib

_CNAME = {np.dtype(np.float32): "float", np.dtype(np.float64): "double", np.dtype(np.int32): "int",
          np.dtype(np.int64): "long", np.dtype(np.float16): "npy_half"}


def _typed(a, dtype, ndim, what):
    """The Cython buffer checks of the reference: exact dtype and ndim, else ValueError."""
    is_t = hasattr(a, "data_ptr")
    if not is_t:
        a = np.asarray(a) if not isinstance(a, np.ndarray) else a
        if a.dtype != np.dtype(dtype):
            raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s'" %
                             (_CNAME[np.dtype(dtype)], _CNAME.get(a.dtype, str(a.dtype))))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)" % (ndim, a.ndim))
    return a


def _dev(ctx, a, tdtype=None):
    t = ctx.torch
    if hasattr(a, "data_ptr"):
        x = a.to(ctx.tdev)
        if tdtype is not None and x.dtype != tdtype:
            x = x.to(tdtype)
        return x.contiguous()
    return t.from_numpy(np.ascontiguousarray(a)).to(ctx.tdev)


def _p(x):
    return ctypes.c_void_p(x.data_ptr())


def _types(alignment_types):
    for x, y in alignment_types:  # dp_core.pyx:24-34
        assert (x > 0)
        assert (y > 0)
    flat = [int(v) for xy in alignment_types for v in xy]
    arr = (ctypes.c_int32 * max(1, len(flat)))(*flat)
    return arr, len(flat) // 2


def make_dense_costs(vecs0, vecs1, norm0, norm1, offset0=0, offset1=0):
    ctx = _lib.context()
    vecs0 = _typed(vecs0, np.float32, 3, "vecs0"); vecs1 = _typed(vecs1, np.float32, 3, "vecs1")
    norm0 = _typed(norm0, np.float32, 2, "norm0"); norm1 = _typed(norm1, np.float32, 2, "norm1")
    assert vecs0.shape[0] > offset0
    assert vecs1.shape[0] > offset1
    assert norm0.shape[0] > offset0
    assert norm1.shape[0] > offset1
    k0, s0, d = vecs0.shape
    k1, s1, d1 = vecs1.shape
    assert norm0.shape[1] == s0
    assert norm1.shape[1] == s1
    assert d1 == d
    t = ctx.torch
    v0, v1, n0, n1 = _dev(ctx, vecs0), _dev(ctx, vecs1), _dev(ctx, norm0), _dev(ctx, norm1)
    costs = t.empty((s0, s1), dtype=t.float32, device=ctx.tdev)
    ctx.check(ctx.lib.svx_dense_costs(ctx.h, _p(v0), k0, s0, _p(v1), k1, s1, d, _p(n0), _p(n1), offset0, offset1, _p(costs)))
    return costs.cpu().numpy()


def dense_dp(alignment_cost, pen):
    ctx = _lib.context()
    alignment_cost = _typed(alignment_cost, np.float32, 2, "alignment_cost")
    s0, s1 = alignment_cost.shape
    t = ctx.torch
    c = _dev(ctx, alignment_cost)
    csum = t.empty((s0 + 1, s1 + 1), dtype=t.float64, device=ctx.tdev)
    bp = t.empty((s0 + 1, s1 + 1), dtype=t.int32, device=ctx.tdev)
    ctx.check(ctx.lib.svx_dense_dp(ctx.h, _p(c), s0, s1, float(np.float32(pen)), _p(csum), _p(bp)))
    return csum.cpu().numpy(), bp.cpu().numpy()


def score_path(xx, yy, norm1, norm2, vecs1, vecs2, out):
    ctx = _lib.context()
    xx = _typed(xx, np.int32, 1, "xx"); yy = _typed(yy, np.int32, 1, "yy")
    norm1 = _typed(norm1, np.float32, 1, "norm1"); norm2 = _typed(norm2, np.float32, 1, "norm2")
    vecs1 = _typed(vecs1, np.float32, 2, "vecs1"); vecs2 = _typed(vecs2, np.float32, 2, "vecs2")
    out = _typed(out, np.float32, 1, "out")
    n = xx.shape[0]
    if n == 0:
        return
    xmax, ymax = int(np.max(xx)), int(np.max(yy))
    if int(np.min(xx)) < -vecs1.shape[0] or xmax >= vecs1.shape[0] or int(np.min(yy)) < -vecs2.shape[0] or ymax >= vecs2.shape[0]:
        raise IndexError("Out of bounds on buffer access (axis 0)")
    t = ctx.torch
    o = t.empty(n, dtype=t.float32, device=ctx.tdev)
    dx, dy = _dev(ctx, np.mod(xx, vecs1.shape[0]).astype(np.int32)), _dev(ctx, np.mod(yy, vecs2.shape[0]).astype(np.int32))
    n1, n2, v1, v2 = _dev(ctx, norm1), _dev(ctx, norm2), _dev(ctx, vecs1), _dev(ctx, vecs2)  # keep alive across the call
    ctx.check(ctx.lib.svx_score_path(ctx.h, _p(dx), _p(dy), n, _p(n1), _p(n2), _p(v1), vecs1.shape[0], _p(v2),
                                     vecs2.shape[0], vecs1.shape[1], _p(o)))
    out[:] = o.cpu().numpy()


def make_sparse_costs(vecs0, vecs1, norms0, norms1, x_y_path, alignment_types, width_over2):
    ctx = _lib.context()
    vecs0 = _typed(vecs0, np.float32, 3, "vecs0"); vecs1 = _typed(vecs1, np.float32, 3, "vecs1")
    norms0 = _typed(norms0, np.float32, 2, "norms0"); norms1 = _typed(norms1, np.float32, 2, "norms1")
    path = np.array(x_y_path).astype(np.int32).reshape(-1, 2)
    assert (vecs0.shape[0] == norms0.shape[0])
    assert (vecs1.shape[0] == norms1.shape[0])
    assert (vecs0.shape[1] == norms0.shape[1])
    assert (vecs1.shape[1] == norms1.shape[1])
    assert (vecs0.shape[2] == vecs1.shape[2])
    types, T = _types(alignment_types)
    A, B = path.shape[0], 2 * int(width_over2)
    t = ctx.torch
    costs = t.empty((T, A, B), dtype=t.float32, device=ctx.tdev)
    boff = t.empty(A, dtype=t.int32, device=ctx.tdev)
    dpath = _dev(ctx, path)
    v0, v1, n0, n1 = _dev(ctx, vecs0), _dev(ctx, vecs1), _dev(ctx, norms0), _dev(ctx, norms1)  # keep alive across the call
    ctx.check(ctx.lib.svx_sparse_costs(ctx.h, _p(v0), vecs0.shape[0], vecs0.shape[1], _p(v1), vecs1.shape[0], vecs1.shape[1],
                                       vecs0.shape[2], _p(n0), _p(n1), _p(dpath), A, types, T, int(width_over2),
                                       _p(costs), _p(boff)))
    return costs.cpu().numpy(), boff.cpu().numpy()


def sparse_dp(a_b_costs, b_offset_in, alignment_types, del_penalty, x_in_size, y_in_size):
    ctx = _lib.context()
    a_b_costs = _typed(a_b_costs, np.float32, 3, "a_b_costs")
    b_offset_in = _typed(b_offset_in, np.int32, 1, "b_offset_in")
    types, T = _types(alignment_types)
    Tc, A, B = a_b_costs.shape
    assert Tc == T
    t = ctx.torch
    csum = t.empty((A + 2, B), dtype=t.float64, device=ctx.tdev)
    xp = t.empty((A + 2, B), dtype=t.int32, device=ctx.tdev)
    yp = t.empty((A + 2, B), dtype=t.int32, device=ctx.tdev)
    bout = t.empty(A + 2, dtype=t.int32, device=ctx.tdev)
    c = _dev(ctx, a_b_costs) if a_b_costs.size else t.empty(1, dtype=t.float32, device=ctx.tdev)
    bin_ = _dev(ctx, b_offset_in)
    ctx.check(ctx.lib.svx_sparse_dp(ctx.h, _p(c), _p(bin_), A, B, types, T, float(del_penalty),
                                    int(x_in_size), int(y_in_size), _p(csum), _p(xp), _p(yp), _p(bout)))
    return csum.cpu().numpy(), xp.cpu().numpy(), yp.cpu().numpy(), bout.cpu().numpy()
