"""Corridor generation: batched firi::firi (gcopter/firi.hpp:268-416), the convexCover loop around it
(gcopter/sfc_gen.hpp:116-186), sfc_gen::shortCut (sfc_gen.hpp:188-226) and the polytope tests it rests on
(geo_utils::findInterior / overlap, gcopter/geo_utils.hpp:43-85).  Polytopes are in GCOPTER's raw form, rows h with h.[x;1] <= 0;
`to_planner_form` is the normalise-and-negate step LearningPlanner applies before the QP
(learning_planner.hpp:293-299)."""
import ctypes

import numpy as np

from .context import default_context


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def firi_params(**over):
    from ._lib import FiriParams, load
    p = FiriParams()
    load().anet_firi_default_params(ctypes.byref(p))
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def firi(bd, pc, a, b, n_points=None, max_rows=64, params=None, ctx=None, iterations=None):
    """Batched firi::firi.  bd (B,Mb,4); pc (B,Np,3) with n_points (B,) valid points each (default: all);
    a, b (B,3).  iterations: optional (B,) int32 pass count per corridor (anet_firi_var; default: params.iterations for
    all).  Returns dict(hpoly (B,max_rows,4) zero-padded, n_rows (B,), ok (B,), ellipsoid (B,15))."""
    ctx = ctx or default_context()
    bd = np.ascontiguousarray(bd, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    B, Mb, _ = bd.shape
    pc = np.ascontiguousarray(pc, dtype=np.float64).reshape(B, -1, 3)
    Np = pc.shape[1]
    npts = np.full(B, Np, dtype=np.int32) if n_points is None else np.ascontiguousarray(n_points, dtype=np.int32)
    if a.shape != (B, 3) or b.shape != (B, 3) or npts.shape != (B,) or (npts > Np).any() or (npts < 0).any():
        raise ValueError("shape mismatch")
    hp = np.zeros((B, max_rows, 4)); nh = np.zeros(B, dtype=np.int32); ok = np.zeros(B, dtype=np.int32)
    ell = np.zeros((B, 15))
    pp = ctypes.cast(ctypes.pointer(params), ctypes.c_void_p) if params is not None else None
    its = None
    if iterations is not None:
        its = np.ascontiguousarray(iterations, dtype=np.int32)
        if its.shape != (B,) or (its < 1).any():
            raise ValueError("iterations: (B,) pass counts >= 1")
    ctx.check(ctx.lib.anet_firi_var(ctx.handle, B, Mb, Np, int(max_rows), _p(bd), _p(pc) if Np else None,
                                    _p(npts) if Np else None, _p(a), _p(b), _p(its) if its is not None else None, pp,
                                    _p(hp), _p(nh), _p(ok), _p(ell)))
    return dict(hpoly=hp, n_rows=nh, ok=ok, ellipsoid=ell)


def firi_dev(bd, pc, n_points, a, b, max_rows=64, params=None, stream=None, ctx=None):
    """anet_firi_dev: the same with torch CUDA tensors in and out (float64 bd (B,Mb,4), pc (B,Np,3), a, b (B,3);
    int32 n_points (B,)).  Nothing leaves the device; asynchronous on `stream` (default: torch's current stream).
    Returns dict(hpoly (B,max_rows,4), n_rows, ok, ellipsoid) of device tensors."""
    import torch
    ctx = ctx or default_context()
    B, Mb, _ = bd.shape
    Np = pc.shape[1]
    for t in (bd, pc, a, b):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError("float64 contiguous CUDA tensors expected")
    if not (n_points.is_cuda and n_points.dtype == torch.int32 and n_points.is_contiguous() and n_points.shape == (B,)):
        raise ValueError("n_points: int32 contiguous CUDA tensor of shape (B,)")
    dev = bd.device
    hp = torch.empty((B, max_rows, 4), device=dev, dtype=torch.float64)
    nh = torch.empty(B, device=dev, dtype=torch.int32); ok = torch.empty(B, device=dev, dtype=torch.int32)
    ell = torch.empty((B, 15), device=dev, dtype=torch.float64)
    work = torch.empty(int(ctx.lib.anet_firi_workspace(B, Np, int(max_rows))), device=dev, dtype=torch.float64)
    st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
    pp = ctypes.cast(ctypes.pointer(params), ctypes.c_void_p) if params is not None else None
    ctx.check(ctx.lib.anet_firi_dev(ctx.handle, B, Mb, Np, int(max_rows), ctypes.c_void_p(bd.data_ptr()),
                                    ctypes.c_void_p(pc.data_ptr()) if Np else None,
                                    ctypes.c_void_p(n_points.data_ptr()) if Np else None, ctypes.c_void_p(a.data_ptr()),
                                    ctypes.c_void_p(b.data_ptr()), pp, ctypes.c_void_p(work.data_ptr()),
                                    ctypes.c_void_p(hp.data_ptr()), ctypes.c_void_p(nh.data_ptr()),
                                    ctypes.c_void_p(ok.data_ptr()), ctypes.c_void_p(ell.data_ptr()), ctypes.c_void_p(st)))
    return dict(hpoly=hp, n_rows=nh, ok=ok, ellipsoid=ell, _work=work)


def to_planner_form(hpoly, n_rows):
    """learning_planner.hpp:293-299: rows divided by the norm of their normal, offset negated: a.x <= b."""
    out = np.zeros_like(hpoly)
    for i, k in enumerate(n_rows):
        nrm = np.linalg.norm(hpoly[i, :k, :3], axis=1, keepdims=True)
        out[i, :k, :3] = hpoly[i, :k, :3] / nrm
        out[i, :k, 3] = -hpoly[i, :k, 3] / nrm[:, 0]
    return out


def pack_model_inputs(ini_pva, fin_pva, hpolys, max_rows=50, max_seg=5):
    """The tensors LearningPlanner::callModel hands to the time-allocation network (learning_planner.hpp:147-170), which are
    also what MinTrajOpt.update / OsqpLayer take (min_traj_opt.py:68-90): state (9,2) float32 -- rows px,vx,ax,py,..,
    column 0 start, column 1 end -- and the corridor (max_rows,4,max_seg) float32, polytope i in [:, :, i] in PLANNER form
    (`to_planner_form`: unit normals, a.x <= b), zero rows / zero polytopes as padding.  Raises when the corridor is
    longer than the model takes (the planner gives that try up, learning_planner.hpp:286-290) or a polytope has more
    rows than the tensor."""
    ini = np.asarray(ini_pva, dtype=np.float64).reshape(3, 3); fin = np.asarray(fin_pva, dtype=np.float64).reshape(3, 3)
    if len(hpolys) > max_seg:
        raise ValueError(f"corridor of {len(hpolys)} polytopes, the model takes {max_seg}")
    state = np.stack([ini.reshape(9), fin.reshape(9)], axis=1).astype(np.float32)     # row = axis, cols p,v,a -> px,vx,ax,py,..
    out = np.zeros((max_rows, 4, max_seg), dtype=np.float32)
    for i, h in enumerate(hpolys):
        h = np.asarray(h, dtype=np.float64)
        if h.shape[0] > max_rows:
            raise ValueError(f"polytope {i} has {h.shape[0]} rows, the model takes {max_rows}")
        out[:h.shape[0], :, i] = h
    return state, out


def convex_cover(path, points, low_corner, high_corner, progress, rng_range, eps=1.0e-6, max_rows=64, ctx=None):
    """sfc_gen::convexCover (sfc_gen.hpp:116-186): walk the path in steps of at most `progress`, one FIRI
    polytope per step inside the box [segment -/+ range] clipped to the map, plus a gap polytope where
    consecutive ones barely overlap at the shared point.  The segments are independent, so all FIRI calls
    of a path run as ONE batch (then one more batch for the gap polytopes).  Returns a list of (n_i,4)
    arrays in raw form."""
    path = [np.asarray(p, dtype=np.float64) for p in path]
    points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    lo_c = np.asarray(low_corner, dtype=np.float64); hi_c = np.asarray(high_corner, dtype=np.float64)
    segs = []
    bq = path[0]
    i = 1
    while i < len(path):
        aq = bq
        if np.linalg.norm(aq - path[i]) > progress:
            d = path[i] - aq
            bq = d / np.linalg.norm(d) * progress + aq
        else:
            bq = path[i]
            i += 1
        segs.append((aq, bq))
    if not segs:
        return []
    B = len(segs)
    bd = np.zeros((B, 6, 4)); sel = []
    for k, (aq, bq) in enumerate(segs):
        hi = np.minimum(np.maximum(aq, bq) + rng_range, hi_c)
        lo = np.maximum(np.minimum(aq, bq) - rng_range, lo_c)
        for ax in range(3):
            bd[k, 2 * ax, ax] = 1.0; bd[k, 2 * ax, 3] = -hi[ax]
            bd[k, 2 * ax + 1, ax] = -1.0; bd[k, 2 * ax + 1, 3] = lo[ax]
        inside = ((points @ bd[k, :, :3].T + bd[k, :, 3]).max(axis=1) < 0.0) if len(points) else np.zeros(0, dtype=bool)
        sel.append(points[inside])
    Np = max(1, max(len(s) for s in sel))
    pc = np.zeros((B, Np, 3)); npts = np.zeros(B, dtype=np.int32)
    for k, s in enumerate(sel):
        pc[k, :len(s)] = s; npts[k] = len(s)
    A = np.array([s[0] for s in segs]); Bv = np.array([s[1] for s in segs])
    # One batch: the B segments with the default pass count and, speculatively, the B - 1 gap polytopes of
    # sfc_gen.hpp:171-179 (firi::firi(bd, pc, a, a, gap, 1): ONE pass, so only a planes kernel each) -- whether a gap
    # polytope is needed depends on the segments' results, but computing all of them costs less than a second call.
    jn = list(range(1, B))
    allr = firi(np.concatenate([bd, bd[jn]]), np.concatenate([pc, pc[jn]]), np.concatenate([A, A[jn]]),
                np.concatenate([Bv, A[jn]]), n_points=np.concatenate([npts, npts[jn]]), max_rows=max_rows,
                iterations=np.r_[np.full(B, firi_params().iterations, dtype=np.int32), np.ones(B - 1, dtype=np.int32)], ctx=ctx)
    polys = [allr["hpoly"][k, :allr["n_rows"][k]] for k in range(B)]
    gaps = {}
    for k in range(1, B):
        ah = np.r_[segs[k][0], 1.0]
        if 3 <= int((polys[k] @ ah > -eps).sum()) + int((polys[k - 1] @ ah > -eps).sum()):
            q = B + k - 1
            gaps[k] = allr["hpoly"][q, :allr["n_rows"][q]]
    out = []
    for k in range(B):
        if k in gaps:
            out.append(gaps[k])
        out.append(polys[k])
    return out


def polytope_depth(hpolys, normalise=True, ctx=None):
    """anet_polytope_depth on a list of (n_i,4) raw-form polytopes (or a zero-padded (B,H,4) array):
    depth (B,) = max t s.t. n.x + t <= -h3, and the point (B,3) attaining it.  -inf: empty polytope."""
    ctx = ctx or default_context()
    if isinstance(hpolys, np.ndarray) and hpolys.ndim == 3:
        hp = np.ascontiguousarray(hpolys, dtype=np.float64)
    else:
        H = max(1, max((len(h) for h in hpolys), default=1))
        hp = np.zeros((len(hpolys), H, 4))
        for i, h in enumerate(hpolys):
            hp[i, :len(h)] = h
    B, H, _ = hp.shape
    depth = np.zeros(B); point = np.zeros((B, 3))
    ctx.check(ctx.lib.anet_polytope_depth(ctx.handle, B, H, _p(hp), 1 if normalise else 0, _p(depth), _p(point)))
    return depth, point


def find_interior(hpoly, ctx=None):
    """geo_utils::findInterior (geo_utils.hpp:43-62): (found, interior point) of one raw-form polytope."""
    d, x = polytope_depth([np.asarray(hpoly, dtype=np.float64)], normalise=True, ctx=ctx)
    return bool(d[0] > 0.0 and np.isfinite(d[0])), x[0]


def overlap(hpoly0, hpoly1, eps=1.0e-6, ctx=None):
    """geo_utils::overlap (geo_utils.hpp:64-85): do the two polytopes share a ball of 'radius' eps (rows not normalised)."""
    d, _ = polytope_depth([np.vstack([hpoly0, hpoly1])], normalise=False, ctx=ctx)
    return bool(d[0] > eps and np.isfinite(d[0]))


def overlap_pt(hpoly0, hpoly1, eps=1.0e-6, ctx=None):
    """geo_utils::overlapPt (geo_utils.hpp:88-111): the overlap test and the deepest common point (a waypoint candidate
    between two consecutive corridor polytopes)."""
    d, x = polytope_depth([np.vstack([hpoly0, hpoly1])], normalise=False, ctx=ctx)
    return bool(d[0] > eps and np.isfinite(d[0])), x[0]


def short_cut(hpolys, eps=0.1, ctx=None):
    """sfc_gen::shortCut (sfc_gen.hpp:188-226): walk the corridor from its last polytope, each time jumping to the
    EARLIEST polytope that still overlaps the current one (consecutive ones always count as overlapping).  The
    overlap tests of all pairs (i, j < i-1) are independent: one batched call, then the walk on the host.
    Returns the shortened list."""
    h = [np.asarray(x, dtype=np.float64) for x in hpolys]
    if len(h) == 1:
        h = [h[0], h[0]]
    M = len(h)
    pairs = [(i, j) for i in range(M) for j in range(i - 1)]
    ov = {}
    if pairs:
        d, _ = polytope_depth([np.vstack([h[i], h[j]]) for i, j in pairs], normalise=False, ctx=ctx)
        ov = {pq: bool(dd > eps and np.isfinite(dd)) for pq, dd in zip(pairs, d)}
    idx = [M - 1]
    i = M - 1
    while i > 0:
        j = next(j for j in range(i) if j == i - 1 or ov[(i, j)])
        idx.insert(0, j)
        i = j
    return [h[k] for k in idx]
