"""Typed wrappers for the dense surrogate kernels (csrc/nn.hip). All tensors fp32, CUDA/HIP, 2-D,
unit inner stride; row strides are passed through so column slices of wider buffers work."""
from __future__ import annotations

import ctypes

import torch

from . import _lib

ACT = {None: 0, "none": 0, "tanh": 1}
AGGR = {"sum": 0, "add": 0, "mean": 1, "max": 2, "mul": 3}


def _mat(t: torch.Tensor, name: str):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise _lib.NbdError(f"{name}: need a 2-D fp32 CUDA/HIP tensor with unit inner stride, got "
                            f"{t.dtype} {tuple(t.shape)} strides {t.stride()} on {t.device}")
    # a single row has no meaningful row stride (torch reports whatever the view it came from had)
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1], 1)


def _vec(t, n, name):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 1 or t.numel() != n or not t.is_contiguous():
        raise _lib.NbdError(f"{name}: need a contiguous fp32 CUDA vector of {n}")
    return t.data_ptr()


def linear(x, w, bias=None, act=None, out=None, rowscale=None, bias_rowscale=None):
    """out = act(rowscale * x @ w.T + bias_rowscale * bias); w is (out_features, in_features)."""
    n, k = x.shape
    m = w.shape[0]
    if w.shape[1] != k:
        raise _lib.NbdError(f"linear: x is {tuple(x.shape)}, w is {tuple(w.shape)}")
    ldx, ldw = _mat(x, "x"), _mat(w, "w")
    if out is None:
        out = torch.empty((n, m), dtype=torch.float32, device=x.device)
    if tuple(out.shape) != (n, m):
        raise _lib.NbdError(f"linear: out is {tuple(out.shape)}, expected {(n, m)}")
    ldy = _mat(out, "out")
    need = _lib.lib().nbd_linear_workspace_bytes(n, m, k)
    ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().nbd_linear_f32(x.data_ptr(), ldx, w.data_ptr(), ldw, _vec(bias, m, "bias"),
                                             _vec(rowscale, n, "rowscale"), _vec(bias_rowscale, n, "bias_rowscale"),
                                             ACT[act], out.data_ptr(), ldy, n, m, k, _lib.ptr(ws), need,
                                             _lib.current_stream(x.device)), "nbd_linear_f32")
    return out


def edgeconv_aggregate(pq, h, rowptr, src, fixed_k, aggr, out=None):
    n = pq.shape[0]
    ld = _mat(pq, "pq")
    if pq.shape[1] != 2 * h:
        raise _lib.NbdError(f"pq must be (n, 2h) = (n, {2 * h}), got {tuple(pq.shape)}")
    if out is None:
        out = torch.empty((n, h), dtype=torch.float32, device=pq.device)
    ldo = _mat(out, "out")
    if src is not None and (src.dtype != torch.int64 or not src.is_contiguous()):
        raise _lib.NbdError("src must be contiguous int64")
    if rowptr is not None and (rowptr.dtype != torch.int32 or rowptr.numel() != n + 1):
        raise _lib.NbdError("rowptr must be int32 [n+1]")
    n_edges = 0 if src is None else src.numel()
    if rowptr is None and n * fixed_k != n_edges:
        raise _lib.NbdError(f"fixed_k={fixed_k} x n={n} != {n_edges} edges")
    with _lib.on_device(pq.device):
        _lib.check(_lib.lib().nbd_edgeconv_aggregate_f32(pq.data_ptr(), ld, h, _lib.ptr(rowptr), _lib.ptr(src),
                                                         fixed_k, n, AGGR[aggr], out.data_ptr(), ldo,
                                                         _lib.current_stream(pq.device)), "nbd_edgeconv_aggregate_f32")
    return out


def edge_messages(pq, h, src, tgt):
    """(E, h) rows tanh(P_tgt + Q_src) for edges already grouped by target."""
    e = src.numel()
    ld = _mat(pq, "pq")
    m = torch.empty((e, h), dtype=torch.float32, device=pq.device)
    for t in (src, tgt):
        if t.dtype != torch.int64 or not t.is_contiguous() or t.numel() != e:
            raise _lib.NbdError("src/tgt must be contiguous int64 of equal length")
    with _lib.on_device(pq.device):
        _lib.check(_lib.lib().nbd_edge_messages_f32(pq.data_ptr(), ld, h, src.data_ptr(), tgt.data_ptr(), e,
                                                    m.data_ptr(), h, _lib.current_stream(pq.device)),
                   "nbd_edge_messages_f32")
    return m


def segment_reduce(m, rowptr, n, mode, out=None):
    h = m.shape[1]
    if out is None:
        out = torch.empty((n, h), dtype=torch.float32, device=m.device)
    with _lib.on_device(m.device):
        _lib.check(_lib.lib().nbd_segment_reduce_f32(m.data_ptr(), _mat(m, "m") if m.shape[0] else h, h,
                                                     rowptr.data_ptr(), n, AGGR[mode], out.data_ptr(), _mat(out, "out"),
                                                     _lib.current_stream(m.device)), "nbd_segment_reduce_f32")
    return out


def layernorm(x, gamma, beta, eps, out=None):
    n, c = x.shape
    ldx = _mat(x, "x")
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    ldy = _mat(out, "out")
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().nbd_layernorm_f32(x.data_ptr(), ldx, c, _vec(gamma, c, "gamma"), _vec(beta, c, "beta"),
                                                float(eps), out.data_ptr(), ldy, n,
                                                _lib.current_stream(x.device)), "nbd_layernorm_f32")
    return out


def reachable_cells(d: int, radius: float, margin: float = 1e-3):
    """Filter grid points (cell = (z*D + y)*D + x) a ContinuousConv sample can touch at all: ball_to_cube maps
    every edge inside |mapped| <= tanh(R) (contconv.py:30-33; the window is zero beyond R, :85-87), i.e. inside
    a ball of radius tanh(R) (D-1)/2 around the grid centre, and a grid point is touched only by samples in
    the unit cubes adjacent to it. Returns (cells int64 [K], cell_map int32 [D^3] with -1 for unreachable)."""
    import math
    c = (d - 1) / 2.0
    rad = math.tanh(radius) * c + margin
    ax = torch.arange(d, dtype=torch.float64)
    gap = torch.clamp((ax - c).abs() - 1.0, min=0.0)                 # distance from the centre to [p-1, p+1]
    g2 = gap ** 2
    dist2 = g2[:, None, None] + g2[None, :, None] + g2[None, None, :]
    keep = (dist2 <= rad * rad).reshape(-1)
    cells = torch.nonzero(keep).reshape(-1)
    cell_map = torch.full((d * d * d,), -1, dtype=torch.int32)
    cell_map[cells] = torch.arange(cells.numel(), dtype=torch.int32)
    return cells, cell_map


def ln_mlp_head_plan(c: int, head, gamma=None, beta=None):
    """What nbd_ln_mlp_head_f32 needs from a decoder chain [(W, b, act)] (gnn.head_chain) over c LayerNorm channels, or
    None when the shapes are outside the fused kernel (it then runs as layernorm + linears): the hidden layers'
    weights transposed ([in][out]) -- with hidden layers the LayerNorm's gamma / beta folded into the first Linear
    (W1 diag(gamma), b1 + W1 beta) --, the ctypes arrays, the tensors to keep alive."""
    n = len(head)
    if not 1 <= n <= 3 or any(act != ("tanh" if i < n - 1 else None) for i, (_, _, act) in enumerate(head)):
        return None
    dims = [c] + [w.shape[0] for w, _, _ in head]
    if any(w.shape[1] != d for (w, _, _), d in zip(head, dims[:-1])):
        return None
    darr = (ctypes.c_int * (n + 1))(*dims)
    if _lib.lib().nbd_ln_mlp_head_lds_bytes(c, n, darr) == 0:
        return None
    ws = [w for w, _, _ in head]
    bs = [b for _, b, _ in head]
    folded = n >= 2
    if folded:
        w1, b1 = ws[0], bs[0]
        if beta is not None:
            shift = w1 @ beta
            b1 = shift if b1 is None else b1 + shift
        if gamma is not None:
            w1 = w1 * gamma.unsqueeze(0)
        ws, bs = [w1] + ws[1:], [b1] + bs[1:]
    ws = [w.t().contiguous() if i < n - 1 else w.contiguous() for i, w in enumerate(ws)]
    bs = [None if b is None else b.contiguous() for b in bs]
    return {"n": n, "dims": darr, "out_dim": dims[-1], "keep": (ws, bs), "folded": folded,
            "w": (ctypes.c_void_p * n)(*[t.data_ptr() for t in ws]),
            "b": (ctypes.c_void_p * n)(*[_lib.ptr(t) for t in bs])}


def ln_mlp_head(x, gamma, beta, eps, plan, out=None, kick_vel=None, kick_c=0.0):
    """out (n, out_dim) = MLP(LayerNorm(x)) in one launch (nbd_ln_mlp_head_f32); kick_vel += kick_c * out when given.
    gamma / beta must be the ones the plan was built with (a plan with hidden layers carries them folded)."""
    n, c = x.shape
    ldx = _mat(x, "x")
    od = plan["out_dim"]
    if out is None or tuple(out.shape) != (n, od) or out.dtype != torch.float32 or out.stride(1) != 1 or out.device != x.device:
        out = torch.empty((n, od), dtype=torch.float32, device=x.device)
    if kick_vel is not None and (tuple(kick_vel.shape) != (n, od) or kick_vel.dtype != torch.float32 or not kick_vel.is_contiguous()):
        raise _lib.NbdError("ln_mlp_head: kick_vel must be a contiguous fp32 (n, out_dim) tensor")
    g_ptr = None if plan["folded"] else _vec(gamma, c, "gamma")
    b_ptr = None if plan["folded"] else _vec(beta, c, "beta")
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().nbd_ln_mlp_head_f32(x.data_ptr(), ldx, c, g_ptr, b_ptr,
                                                  float(eps), plan["n"], plan["w"], plan["b"], plan["dims"], out.data_ptr(),
                                                  out.stride(0), _lib.ptr(kick_vel), float(kick_c), n,
                                                  _lib.current_stream(x.device)), "nbd_ln_mlp_head_f32")
    return out


def contconv_bin(pos, feat, rowptr, centres, d, radius_sq, out=None, node_begin=0, count=None, cell_map=None,
                 cells_out=None):
    """A (count, cells_out * I): feature-side trilinear binning of ContinuousConv (contconv.py:80-93) for the
    nodes [node_begin, node_begin + count); cell_map (int32 [d^3] on the device) drops unreachable cells."""
    n, i_ch = feat.shape
    count = n - node_begin if count is None else count
    ldf = _mat(feat, "feat")
    if pos.shape != (n, 3) or pos.dtype != torch.float32 or not pos.is_contiguous():
        raise _lib.NbdError("pos must be contiguous fp32 (n,3)")
    if rowptr.dtype != torch.int32 or rowptr.numel() != n + 1 or centres.dtype != torch.int32:
        raise _lib.NbdError("rowptr int32 [n+1] / centres int32 required")
    if node_begin < 0 or count < 0 or node_begin + count > n:
        raise _lib.NbdError(f"node range [{node_begin}, {node_begin + count}) outside [0, {n})")
    if cell_map is None:
        cells_out = d * d * d
    elif cell_map.dtype != torch.int32 or cell_map.numel() != d * d * d or not cell_map.is_cuda or cells_out is None:
        raise _lib.NbdError("cell_map must be an int32 CUDA tensor of d^3 entries, with cells_out")
    kc = cells_out * i_ch
    if out is None:
        out = torch.empty((count, kc), dtype=torch.float32, device=feat.device)
    if out.dim() != 2 or out.shape[0] < count or out.shape[1] != kc or not out.is_contiguous():
        raise _lib.NbdError(f"A must be contiguous (>= {count}, {kc})")
    with _lib.on_device(feat.device):
        _lib.check(_lib.lib().nbd_contconv_bin_f32(pos.data_ptr(), feat.data_ptr(), ldf, i_ch, rowptr.data_ptr(),
                                                   centres.data_ptr(), node_begin, count, d, float(radius_sq),
                                                   _lib.ptr(cell_map), cells_out, out.data_ptr(),
                                                   _lib.current_stream(feat.device)),
                   "nbd_contconv_bin_f32")
    return out


def ball_to_cube(r: torch.Tensor) -> torch.Tensor:
    """contconv.py:30-33 on the GPU (nbd_ball_to_cube_f32): r (..., 3) -> r / (|r| + 1e-8) * tanh |r|."""
    if not r.is_cuda:
        raise _lib.NbdError("ball_to_cube: tensors must live on the GPU (no CPU path)")
    shape = r.shape
    if shape[-1] != 3:
        raise _lib.NbdError("ball_to_cube: r must be (..., 3)")
    rc = r.detach().to(torch.float32).reshape(-1, 3).contiguous()
    out = torch.empty_like(rc)
    with _lib.on_device(rc.device):
        _lib.check(_lib.lib().nbd_ball_to_cube_f32(rc.data_ptr(), rc.shape[0], out.data_ptr(), _lib.current_stream(rc.device)),
                   "nbd_ball_to_cube_f32")
    return out.reshape(shape)


def trilinear_interpolate(filters: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """contconv.py:53-78 on the GPU (nbd_trilinear_interpolate_f32): filters (D, D, D, I, O), coords (N, 3) in [0, D - 1]
    -> (N, I, O), the reference's grid_sample call (align_corners, zero padding, component 0 along the last grid axis)."""
    if not (filters.is_cuda and coords.is_cuda):
        raise _lib.NbdError("trilinear_interpolate: tensors must live on the GPU (no CPU path)")
    d, i_ch, o_ch = filters.shape[0], filters.shape[3], filters.shape[4]
    if coords.dim() != 2 or coords.shape[1] != 3:
        raise _lib.NbdError("trilinear_interpolate: coords must be (N, 3)")
    f = filters.detach().to(torch.float32).contiguous()
    c = coords.detach().to(torch.float32).contiguous()
    out = torch.empty((c.shape[0], i_ch, o_ch), dtype=torch.float32, device=f.device)
    with _lib.on_device(f.device):
        _lib.check(_lib.lib().nbd_trilinear_interpolate_f32(f.data_ptr(), d, i_ch, o_ch, c.data_ptr(), c.shape[0], out.data_ptr(),
                                                            _lib.current_stream(f.device)), "nbd_trilinear_interpolate_f32")
    return out


def contconv_fused_supported(i_ch: int, o_ch: int, n_cells: int) -> bool:
    return bool(_lib.lib().nbd_contconv_fused_supported(int(i_ch), int(o_ch), int(n_cells)))


def contconv_shuffle_filters(filters: torch.Tensor, cells: torch.Tensor) -> torch.Tensor:
    """filters (D,D,D,I,O) -> the operand nbd_contconv_fused_f32 reads (see include/nbd.h): the kept cells' I x O matrices
    in MFMA fragment order, every element split into three bf16 terms (hi, mid, lo: together its 24 significant bits). A
    layout transform of the weights: done once per weight update, by one launch (nbd_contconv_shuffle_filters_f32)."""
    return contconv_shuffle_filters_kernel(filters, cells)


def contconv_shuffle_filters_kernel(filters: torch.Tensor, cells: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """nbd_contconv_shuffle_filters_f32: what the training step calls four times (two layers, forward and the transposed
    operand of the feature gradient); `cells` int64 on the device."""
    d, i_ch, o_ch = filters.shape[0], filters.shape[3], filters.shape[4]
    f = filters.detach()
    f = f if (f.is_contiguous() and f.dtype == torch.float32) else f.contiguous().float()
    k = int(cells.numel())
    L = _lib.lib()
    ii, oo = (o_ch, i_ch) if transposed else (i_ch, o_ch)
    out = torch.empty(L.nbd_contconv_filter_floats(ii, oo, k), dtype=torch.float32, device=f.device)
    with _lib.on_device(f.device):
        _lib.check(L.nbd_contconv_shuffle_filters_f32(f.data_ptr(), cells.data_ptr(), k, i_ch, o_ch, int(transposed),
                                                      out.data_ptr(), _lib.current_stream(f.device)),
                   "nbd_contconv_shuffle_filters_f32")
    return out


def contconv_filter_grad_full(feat, g, rowptr, pair_buf, edge_capacity: int, n_cells: int, cell_map, d: int):
    """(d, d, d, I, O): the filter gradient over the full grid, zeros at the cells no sample can reach."""
    n, i_ch, o_ch = rowptr.numel() - 1, feat.shape[1], g.shape[1]
    ldf, ldg = _mat(feat, "feat"), _mat(g, "g")
    L = _lib.lib()
    out = torch.empty((d, d, d, i_ch, o_ch), dtype=torch.float32, device=feat.device)
    need = n_cells * i_ch * o_ch * 4 + L.nbd_contconv_filter_grad_workspace_bytes(n, int(n_cells), i_ch, o_ch)
    ws = _ws(need, feat.device)
    with _lib.on_device(feat.device):
        _lib.check(L.nbd_contconv_filter_grad_full_f32(feat.data_ptr(), ldf, i_ch, g.data_ptr(), ldg, o_ch, rowptr.data_ptr(), n,
                                                       int(edge_capacity), pair_buf.data_ptr(), int(n_cells),
                                                       _lib.ptr(cell_map), d * d * d, out.data_ptr(), _lib.ptr(ws), need,
                                                       _lib.current_stream(feat.device)), "nbd_contconv_filter_grad_full_f32")
    return out


def contconv_pairs(pos, rowptr, centres, edge_capacity: int, d: int, radius_sq: float, cell_map, n_cells: int):
    """Pair lists of one (graph, filter resolution): opaque device buffer for contconv_fused (see include/nbd.h)."""
    n = pos.shape[0]
    if pos.shape != (n, 3) or pos.dtype != torch.float32 or not pos.is_contiguous():
        raise _lib.NbdError("pos must be contiguous fp32 (n,3)")
    if rowptr.dtype != torch.int32 or rowptr.numel() != n + 1 or centres.dtype != torch.int32:
        raise _lib.NbdError("rowptr int32 [n+1] / centres int32 required")
    if centres.numel() > edge_capacity:
        edge_capacity = centres.numel()
    if cell_map is not None and (cell_map.dtype != torch.int32 or cell_map.numel() != d * d * d or not cell_map.is_cuda):
        raise _lib.NbdError("cell_map must be an int32 CUDA tensor of d^3 entries")
    L = _lib.lib()
    nbytes = L.nbd_contconv_pairs_bytes(n, int(edge_capacity), int(n_cells))
    buf = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=pos.device)
    with _lib.on_device(pos.device):
        _lib.check(L.nbd_contconv_pairs_f32(pos.data_ptr(), rowptr.data_ptr(), centres.data_ptr(), n, int(edge_capacity),
                                            int(d), float(radius_sq), _lib.ptr(cell_map), int(n_cells), buf.data_ptr(),
                                            buf.numel(), _lib.current_stream(pos.device)), "nbd_contconv_pairs_f32")
    return buf, int(edge_capacity)


def contconv_pairs_batch(pos, rowptr, centres, edge_capacity: int, radius_sq: float, jobs):
    """Pair lists of several filter resolutions of one graph in ONE launch. jobs: list of (d, cell_map, n_cells);
    returns a list of (buffer, edge_capacity) in the same order."""
    import ctypes
    n = pos.shape[0]
    if pos.shape != (n, 3) or pos.dtype != torch.float32 or not pos.is_contiguous():
        raise _lib.NbdError("pos must be contiguous fp32 (n,3)")
    if rowptr.dtype != torch.int32 or rowptr.numel() != n + 1 or centres.dtype != torch.int32:
        raise _lib.NbdError("rowptr int32 [n+1] / centres int32 required")
    edge_capacity = max(int(edge_capacity), centres.numel())
    L = _lib.lib()
    k = len(jobs)
    bufs = []
    for d, cmap, nc in jobs:
        if cmap is not None and (cmap.dtype != torch.int32 or cmap.numel() != d * d * d or not cmap.is_cuda):
            raise _lib.NbdError("cell_map must be an int32 CUDA tensor of d^3 entries")
        bufs.append(torch.empty(max(L.nbd_contconv_pairs_bytes(n, edge_capacity, int(nc)), 16), dtype=torch.uint8,
                                device=pos.device))
    ds = (ctypes.c_int * k)(*[int(j[0]) for j in jobs])
    maps = (ctypes.c_void_p * k)(*[_lib.ptr(j[1]) for j in jobs])
    ncs = (ctypes.c_int * k)(*[int(j[2]) for j in jobs])
    ptrs = (ctypes.c_void_p * k)(*[b.data_ptr() for b in bufs])
    sizes = (ctypes.c_size_t * k)(*[b.numel() for b in bufs])
    with _lib.on_device(pos.device):
        _lib.check(L.nbd_contconv_pairs_batch_f32(pos.data_ptr(), rowptr.data_ptr(), centres.data_ptr(), n, edge_capacity,
                                                  float(radius_sq), k, ds, maps, ncs, ptrs, sizes,
                                                  _lib.current_stream(pos.device)), "nbd_contconv_pairs_batch_f32")
    return [(b, edge_capacity) for b in bufs]


def contconv_pairs_jobs(pos, jobs):
    """nbd_contconv_pairs_jobs_f32: up to four pair-list jobs over the same nodes in ONE launch. jobs: dicts with
    rowptr, centres, deg (None = CSR), edge_capacity, d, cell_map, n_cells, adjoint. Returns [(buffer, edge_capacity)]."""
    n = pos.shape[0]
    if pos.shape != (n, 3) or pos.dtype != torch.float32 or not pos.is_contiguous():
        raise _lib.NbdError("pos must be contiguous fp32 (n,3)")
    if not 1 <= len(jobs) <= _lib.CC_MAX_RES:
        raise _lib.NbdError(f"contconv_pairs_jobs: 1 .. {_lib.CC_MAX_RES} jobs per launch")
    L = _lib.lib()
    arr = (_lib.CcPairsJob * len(jobs))()
    out = []
    r2 = None
    for a, j in zip(arr, jobs):
        rowptr, centres, deg, cmap = j["rowptr"], j["centres"], j.get("deg"), j.get("cell_map")
        if rowptr.dtype != torch.int32 or rowptr.numel() != n + 1 or centres.dtype != torch.int32 or \
                (deg is not None and (deg.dtype != torch.int32 or deg.numel() != n)):
            raise _lib.NbdError("rowptr int32 [n+1] / centres int32 / deg int32 [n] required")
        d, nc = int(j["d"]), int(j["n_cells"])
        if cmap is not None and (cmap.dtype != torch.int32 or cmap.numel() != d * d * d or not cmap.is_cuda):
            raise _lib.NbdError("cell_map must be an int32 CUDA tensor of d^3 entries")
        cap_e = max(int(j["edge_capacity"]), centres.numel())
        buf = torch.empty(max(L.nbd_contconv_pairs_bytes(n, cap_e, nc), 16), dtype=torch.uint8, device=pos.device)
        a.rowptr, a.centres, a.deg, a.edge_capacity = rowptr.data_ptr(), centres.data_ptr(), _lib.ptr(deg), cap_e
        a.filter_resolution, a.cell_map, a.n_cells, a.adjoint = d, _lib.ptr(cmap), nc, int(bool(j.get("adjoint")))
        a.pair_lists, a.pair_lists_bytes = buf.data_ptr(), buf.numel()
        r2 = float(j["radius_sq"]) if r2 is None else r2
        out.append((buf, cap_e))
    with _lib.on_device(pos.device):
        _lib.check(L.nbd_contconv_pairs_jobs_f32(pos.data_ptr(), n, r2, len(jobs), arr, _lib.current_stream(pos.device)),
                   "nbd_contconv_pairs_jobs_f32")
    return out


def contconv_filter_grad(feat, g, rowptr, pair_buf, edge_capacity: int, n_cells: int):
    """(n_cells, I, O) = sum over the touched (node, cell) blocks of A[node][cell]^T g[node] (nbd_contconv_filter_grad_f32):
    ContinuousConv's filter gradient over the forward pair lists, cells in compact (kept) order."""
    n, i_ch, o_ch = rowptr.numel() - 1, feat.shape[1], g.shape[1]
    ldf, ldg = _mat(feat, "feat"), _mat(g, "g")
    L = _lib.lib()
    out = torch.empty((n_cells, i_ch, o_ch), dtype=torch.float32, device=feat.device)
    need = L.nbd_contconv_filter_grad_workspace_bytes(n, int(n_cells), i_ch, o_ch)
    ws = _ws(need, feat.device)
    with _lib.on_device(feat.device):
        _lib.check(L.nbd_contconv_filter_grad_f32(feat.data_ptr(), ldf, i_ch, g.data_ptr(), ldg, o_ch, rowptr.data_ptr(), n,
                                                  int(edge_capacity), pair_buf.data_ptr(), int(n_cells), out.data_ptr(),
                                                  _lib.ptr(ws), need, _lib.current_stream(feat.device)),
                   "nbd_contconv_filter_grad_f32")
    return out


def contconv_pairs_inv_degree(pair_buf, n: int, edge_capacity: int, n_cells: int) -> torch.Tensor:
    """float32 [n] view into the pair-list buffer: 1 / max(in-degree, 1), the mean aggregation's row scale, which the
    pair kernel writes as a by-product (no degree_scale launch)."""
    L = _lib.lib()
    off = (ctypes.c_size_t * 9)()
    _lib.check(L.nbd_contconv_pairs_layout(int(n), int(edge_capacity), int(n_cells), off), "nbd_contconv_pairs_layout")
    return pair_buf[off[8]:off[8] + 4 * n].view(torch.float32)


def contconv_pairs_stats(pair_buf, n: int, edge_capacity: int, n_cells: int) -> dict:
    """{"steps": 16-row MFMA steps the fused kernel will run over this graph, "cost": the weight its workgroup split
    balances (sum over steps of max(64, pairs): CC_COST_MIN)} -- one host read-back: for reports (bench.py's executed-flop figure),
    not for the rollout loop."""
    L = _lib.lib()
    off = (ctypes.c_size_t * 9)()
    _lib.check(L.nbd_contconv_pairs_layout(int(n), int(edge_capacity), int(n_cells), off), "nbd_contconv_pairs_layout")
    tiles = (n + 127) // 128
    steps = pair_buf[off[5]:off[5] + 4 * tiles].view(torch.int32).sum()
    cost = pair_buf[off[6]:off[6] + 4 * tiles].view(torch.int32).to(torch.int64).sum()
    return {"steps": int(steps.item()), "cost": int(cost.item())}


def contconv_fused(feat, rowptr, pair_buf, edge_capacity: int, filt_shuffled, n_cells: int, o_ch: int, rowscale=None,
                   act=None, out=None):
    """out (n, o_ch) = act(rowscale * sum_cells A[n][cell] . F[cell]): the block-sparse fused ContinuousConv."""
    n, i_ch = rowptr.numel() - 1, feat.shape[1]          # rows = aggregation targets; feat rows = source nodes
    ldf = _mat(feat, "feat")
    L = _lib.lib()
    if filt_shuffled.dtype != torch.float32 or not filt_shuffled.is_contiguous() or \
            filt_shuffled.numel() != L.nbd_contconv_filter_floats(i_ch, o_ch, n_cells):
        raise _lib.NbdError("filt_shuffled: wrong size for (in, out, cells) -- use contconv_shuffle_filters")
    if out is None:
        out = torch.empty((n, o_ch), dtype=torch.float32, device=feat.device)
    ldo = _mat(out, "out")
    if out.shape != (n, o_ch):
        raise _lib.NbdError(f"out must be ({n}, {o_ch})")
    need = L.nbd_contconv_fused_workspace_bytes(n, n_cells, o_ch)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=feat.device)
    with _lib.on_device(feat.device):
        _lib.check(L.nbd_contconv_fused_f32(feat.data_ptr(), ldf, i_ch, rowptr.data_ptr(), n, int(edge_capacity),
                                            pair_buf.data_ptr(), filt_shuffled.data_ptr(), int(n_cells), int(o_ch),
                                            _lib.ptr(rowscale), 1 if act == "tanh" else 0, out.data_ptr(), ldo,
                                            ws.data_ptr(), ws.numel(), _lib.current_stream(feat.device)),
                   "nbd_contconv_fused_f32")
    return out


def degree_scale(rowptr, n, mode, device):
    out = torch.empty(n, dtype=torch.float32, device=device)
    with _lib.on_device(device):
        _lib.check(_lib.lib().nbd_degree_scale_f32(rowptr.data_ptr(), n, mode, out.data_ptr(),
                                                   _lib.current_stream(device)), "nbd_degree_scale_f32")
    return out


EPILOGUE = {"write_x": 0, "next_pq": 1, "final_head": 2, "final_ln": 3, "next_pq_folded": 4}


def gnn_layer_args(*, n, h, aggr, rowptr, src, fixed_k, w2t, b2, epilogue, out, pq=None, x=None, f=0, wpq=None, bpq=None,
                   w_ep=None, b_ep=None, ep_out=0, enc=None, e=0, ln_g=None, ln_b=None, ln_eps=1e-5, kick_vel=None, kick_c=0.0):
    """The `nbd_gnn_layer_args` struct of one fused EdgeConv layer, or None when the shape is outside what the fused
    kernel supports (the caller then takes the general multi-kernel path)."""
    a = _lib.GnnLayerArgs()
    dev = out.device
    kp = 64 * ((h + 63) // 64)
    lds_floats = kp * ep_out if epilogue == "next_pq_folded" else kp * h + kp * (ep_out if epilogue == "next_pq" else 0)
    if h > 128 or (pq is None and f > 8) or lds_floats * 4 > 64 * 1024:
        return None
    if epilogue == "final_head" and ep_out > 8:
        return None
    if epilogue in ("final_head", "final_ln") and e > 256:
        return None
    for t in (w2t, b2, wpq, bpq, w_ep, b_ep, ln_g, ln_b):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda):
            raise _lib.NbdError("gnn_layer: weights must be contiguous fp32 CUDA tensors")
    a.rowptr, a.src, a.fixed_k, a.n = _lib.ptr(rowptr), _lib.ptr(src), fixed_k, n
    a.pq, a.ldpq = _lib.ptr(pq), (_mat(pq, "pq") if pq is not None else 0)
    a.x, a.ldx, a.f = _lib.ptr(x), (_mat(x, "x") if x is not None else 0), f
    a.wpq, a.bpq = _lib.ptr(wpq), _lib.ptr(bpq)
    a.h, a.aggr, a.w2t, a.b2 = h, AGGR[aggr], _lib.ptr(w2t), b2.data_ptr()
    a.epilogue, a.w_ep, a.b_ep, a.ep_out = EPILOGUE[epilogue], _lib.ptr(w_ep), _lib.ptr(b_ep), ep_out
    a.enc, a.ldenc, a.e = _lib.ptr(enc), (_mat(enc, "enc") if enc is not None else 0), e
    a.ln_g, a.ln_b, a.ln_eps = _lib.ptr(ln_g), _lib.ptr(ln_b), float(ln_eps)
    a.out, a.ldout = out.data_ptr(), _mat(out, "out")
    if kick_vel is not None:
        if epilogue != "final_head" or kick_vel.shape != (n, ep_out) or kick_vel.dtype != torch.float32 or \
                not kick_vel.is_contiguous() or kick_vel.device != dev:
            raise _lib.NbdError("gnn_layer: kick_vel must be a contiguous fp32 (n, ep_out) tensor on the layer's device")
        a.kick_vel, a.kick_c = kick_vel.data_ptr(), float(kick_c)
    return a


def gnn_layer(**kw):
    """One fused EdgeConv layer (nbd_gnn_layer_f32). Returns False (nothing launched) when the shape is
    outside what the fused kernel supports, so the caller can take the general multi-kernel path.
    kick_vel (n, ep_out), final_head only: vel += kick_c * out in the epilogue (Trainer.step's second half-kick)."""
    a = gnn_layer_args(**kw)
    if a is None:
        return False
    dev = kw["out"].device
    with _lib.on_device(dev):
        _lib.check(_lib.lib().nbd_gnn_layer_f32(ctypes.byref(a), _lib.current_stream(dev)), "nbd_gnn_layer_f32")
    return True


# ---------------------------------------------------------------------------- backward (csrc/train.hip)
def _ws(nbytes, device):
    return torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None


def act_bwd(dy, y, act, rowscale=None):
    """rowscale * dy * act'(y) with y the forward output (tanh: 1 - y^2)."""
    n, c = dy.shape
    g = torch.empty((n, c), dtype=torch.float32, device=dy.device)
    with _lib.on_device(dy.device):
        _lib.check(_lib.lib().nbd_act_bwd_f32(dy.data_ptr(), _mat(dy, "dy"), y.data_ptr() if y is not None else None,
                                              _mat(y, "y") if y is not None else 0, ACT[act],
                                              _vec(rowscale, n, "rowscale"), g.data_ptr(), c, n, c,
                                              _lib.current_stream(dy.device)), "nbd_act_bwd_f32")
    return g


def colsum(x, rowweight=None):
    n, c = x.shape
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    need = _lib.lib().nbd_colsum_workspace_bytes(n, c)
    ws = _ws(need, x.device)
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().nbd_colsum_f32(x.data_ptr(), _mat(x, "x") if n else c, _vec(rowweight, n, "rowweight"),
                                             n, c, out.data_ptr(), _lib.ptr(ws), need,
                                             _lib.current_stream(x.device)), "nbd_colsum_f32")
    return out


def linear_wgrad(g, x):
    """(m, k) = g^T x for g (n, m), x (n, k): torch.nn.Linear's weight gradient."""
    n, m = g.shape
    k = x.shape[1]
    if x.shape[0] != n:
        raise _lib.NbdError(f"linear_wgrad: g is {tuple(g.shape)}, x is {tuple(x.shape)}")
    dw = torch.empty((m, k), dtype=torch.float32, device=g.device)
    need = _lib.lib().nbd_linear_wgrad_workspace_bytes(n, m, k)
    ws = _ws(need, g.device)
    with _lib.on_device(g.device):
        _lib.check(_lib.lib().nbd_linear_wgrad_f32(g.data_ptr(), _mat(g, "g") if n else m, x.data_ptr(),
                                                   _mat(x, "x") if n else k, n, m, k, dw.data_ptr(), k, _lib.ptr(ws),
                                                   need, _lib.current_stream(g.device)), "nbd_linear_wgrad_f32")
    return dw


def linear_wgrad_bias(g, x, rowweight=None):
    """(dW (m, k), db (m,)) = (g^T x, sum_n w_n g_n) in the same launches (nbd_linear_wgrad_bias_f32)."""
    n, m = g.shape
    k = x.shape[1]
    if x.shape[0] != n:
        raise _lib.NbdError(f"linear_wgrad_bias: g is {tuple(g.shape)}, x is {tuple(x.shape)}")
    dw = torch.empty((m, k), dtype=torch.float32, device=g.device)
    db = torch.empty(m, dtype=torch.float32, device=g.device)
    need = _lib.lib().nbd_linear_wgrad_bias_workspace_bytes(n, m, k)
    ws = _ws(need, g.device)
    with _lib.on_device(g.device):
        _lib.check(_lib.lib().nbd_linear_wgrad_bias_f32(g.data_ptr(), _mat(g, "g") if n else m, x.data_ptr(),
                                                        _mat(x, "x") if n else k, _vec(rowweight, n, "rowweight"), n, m, k,
                                                        dw.data_ptr(), k, db.data_ptr(), _lib.ptr(ws), need,
                                                        _lib.current_stream(g.device)), "nbd_linear_wgrad_bias_f32")
    return dw, db


def edgeconv_aggregate_bwd(pq, h, ds, rowptr, src, fixed_k, rowptr_t, tgt_t, aggr):
    n = pq.shape[0]
    dpq = torch.empty((n, 2 * h), dtype=torch.float32, device=pq.device)
    if rowptr_t.dtype != torch.int32 or rowptr_t.numel() != n + 1 or tgt_t.dtype != torch.int32:
        raise _lib.NbdError("rowptr_t int32 [n+1] / tgt_t int32 required")
    with _lib.on_device(pq.device):
        _lib.check(_lib.lib().nbd_edgeconv_aggregate_bwd_f32(
            pq.data_ptr(), _mat(pq, "pq"), h, ds.data_ptr(), _mat(ds, "ds"), _lib.ptr(rowptr), _lib.ptr(src), fixed_k,
            rowptr_t.data_ptr(), _lib.ptr(tgt_t), n, AGGR[aggr], dpq.data_ptr(), 2 * h,
            _lib.current_stream(pq.device)), "nbd_edgeconv_aggregate_bwd_f32")
    return dpq


def layernorm_bwd(x, gamma, eps, dy):
    n, c = x.shape
    dx = torch.empty((n, c), dtype=torch.float32, device=x.device)
    dg = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    need = _lib.lib().nbd_layernorm_bwd_workspace_bytes(n, c)
    ws = _ws(need, x.device)
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().nbd_layernorm_bwd_f32(x.data_ptr(), _mat(x, "x") if n else c, c, _vec(gamma, c, "gamma"),
                                                    float(eps), dy.data_ptr(), _mat(dy, "dy") if n else c,
                                                    dx.data_ptr(), c, dg.data_ptr(), db.data_ptr(), n, _lib.ptr(ws),
                                                    need, _lib.current_stream(x.device)), "nbd_layernorm_bwd_f32")
    return dx, dg, db


def segment_max_bwd(m, x, rowptr, dx):
    e, h = m.shape
    n = x.shape[0]
    dm = torch.empty((e, h), dtype=torch.float32, device=m.device)
    with _lib.on_device(m.device):
        _lib.check(_lib.lib().nbd_segment_max_bwd_f32(m.data_ptr(), _mat(m, "m") if e else h, h, x.data_ptr(),
                                                      _mat(x, "x"), rowptr.data_ptr(), n, dx.data_ptr(), _mat(dx, "dx"),
                                                      dm.data_ptr(), h, _lib.current_stream(m.device)),
                   "nbd_segment_max_bwd_f32")
    return dm


def segment_mul_bwd(m, rowptr, n, dx):
    """Gradient of the per-target product (segment_reduce mode "mul") with respect to its rows."""
    e, h = m.shape
    dm = torch.empty((e, h), dtype=torch.float32, device=m.device)
    with _lib.on_device(m.device):
        _lib.check(_lib.lib().nbd_segment_mul_bwd_f32(m.data_ptr(), _mat(m, "m") if e else h, h, rowptr.data_ptr(), n,
                                                      dx.data_ptr(), _mat(dx, "dx"), dm.data_ptr(), h,
                                                      _lib.current_stream(m.device)), "nbd_segment_mul_bwd_f32")
    return dm


def batchnorm_train_fwd(x, gamma, beta, eps, act):
    """y = act(BatchNorm1d_train(x)); returns (y, mean, biased var, rstd)."""
    n, c = x.shape
    if n < 2:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
    dev = x.device
    y = torch.empty((n, c), dtype=torch.float32, device=dev)
    mean, var, rstd = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(3))
    need = _lib.lib().nbd_batchnorm_train_workspace_bytes(n, c)
    ws = _ws(need, dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().nbd_batchnorm_train_fwd_f32(x.data_ptr(), _mat(x, "x"), n, c, _vec(gamma, c, "gamma"),
                                                          _vec(beta, c, "beta"), float(eps), ACT[act], y.data_ptr(), c,
                                                          mean.data_ptr(), var.data_ptr(), rstd.data_ptr(),
                                                          _lib.ptr(ws), need, _lib.current_stream(dev)),
                   "nbd_batchnorm_train_fwd_f32")
    return y, mean, var, rstd


def batchnorm_train_bwd(x, gamma, mean, rstd, act, y, dy):
    n, c = x.shape
    dev = x.device
    dx = torch.empty((n, c), dtype=torch.float32, device=dev)
    dg, db = torch.empty(c, dtype=torch.float32, device=dev), torch.empty(c, dtype=torch.float32, device=dev)
    need = _lib.lib().nbd_batchnorm_train_workspace_bytes(n, c)
    ws = _ws(need, dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().nbd_batchnorm_train_bwd_f32(x.data_ptr(), _mat(x, "x"), n, c, _vec(gamma, c, "gamma"),
                                                          mean.data_ptr(), rstd.data_ptr(), ACT[act], y.data_ptr(),
                                                          _mat(y, "y"), dy.data_ptr(), _mat(dy, "dy"), dx.data_ptr(), c,
                                                          dg.data_ptr(), db.data_ptr(), _lib.ptr(ws), need,
                                                          _lib.current_stream(dev)), "nbd_batchnorm_train_bwd_f32")
    return dx, dg, db


def contconv_bin_bwd(pos, da, i_ch, d, radius_sq, rowptr_s=None, tgt_s=None, deg=None, cap=0, cell_map=None,
                     cells_out=None):
    """dfeat (n, i_ch): adjoint of contconv_bin w.r.t. the features; lists are per SOURCE (see nbd.h)."""
    n = pos.shape[0]
    cells_out = d * d * d if cell_map is None else cells_out
    if da.shape != (n, cells_out * i_ch) or not da.is_contiguous():
        raise _lib.NbdError(f"dA must be contiguous ({n}, {cells_out * i_ch})")
    for t in (rowptr_s, tgt_s, deg, cell_map):
        if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
            raise _lib.NbdError("contconv_bin_bwd: index lists must be contiguous int32")
    dfeat = torch.empty((n, i_ch), dtype=torch.float32, device=pos.device)
    with _lib.on_device(pos.device):
        _lib.check(_lib.lib().nbd_contconv_bin_bwd_f32(pos.data_ptr(), da.data_ptr(), i_ch, _lib.ptr(rowptr_s),
                                                       tgt_s.data_ptr(), _lib.ptr(deg), cap, n, d, float(radius_sq),
                                                       _lib.ptr(cell_map), cells_out, dfeat.data_ptr(), i_ch,
                                                       _lib.current_stream(pos.device)),
                   "nbd_contconv_bin_bwd_f32")
    return dfeat
