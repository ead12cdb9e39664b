"""Neighbour-graph construction on the GPU (csrc/graph.hip) behind PyG-shaped functions.

knn_graph / radius_graph keep the signatures the reference imports from torch_geometric
(gnn.py:5, contconv.py:5) for the arguments it uses; `radius_lists` is the sync-free padded form
the ContinuousConv pipeline consumes."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .direct import _chk


def mark(t: torch.Tensor, attr: str) -> None:
    """Record a structural fact about tensor `t` (sorted / grouped) together with its version counter, so that an
    in-place edit of the tensor silently retires the fact."""
    setattr(t, attr, t._version)


def marked(t, attr: str) -> bool:
    return getattr(t, attr, None) == getattr(t, "_version", -1)


def _stream(dev):
    return _lib.current_stream(dev)


def _segments(batch: torch.Tensor | None, n: int, device):
    """Per-node [lo, hi) of its batch segment (batch must be sorted, as PyG requires). No host sync when
    the vector is known to be sorted (built by nbd.data.collate, or seen before): the bounds are two
    binary searches of the vector in itself, remembered on the tensor object."""
    if batch is None:
        return None, None
    if batch.numel() != n:
        raise _lib.NbdError(f"batch has {batch.numel()} entries for {n} nodes")
    memo = getattr(batch, "_nbd_segments", None)
    if memo is not None and memo[0] == batch._version and memo[1].device == torch.device(device):
        return memo[1], memo[2]
    b = batch.to(device=device, dtype=torch.int64).contiguous()
    if not marked(batch, "_nbd_sorted") and n > 1 and bool((b[1:] < b[:-1]).any()):   # one sync, first sight only
        raise _lib.NbdError("batch vector must be sorted (PyG convention)")
    lo = torch.searchsorted(b, b, right=False).to(torch.int32).contiguous()
    hi = torch.searchsorted(b, b, right=True).to(torch.int32).contiguous()
    try:
        batch._nbd_segments = (batch._version, lo, hi)
    except (AttributeError, RuntimeError):
        pass
    return lo, hi


def knn_layout(batch: torch.Tensor, n: int, k: int, loop: bool, device):
    """(first edge of every centre int64 [n], edge count) of knn_graph(batch=...): they depend only on (segment sizes, k,
    loop), so they are remembered on the batch vector -- a rollout that advances several scenes together pays the one host
    read-back of the edge count once, not per step (and none at all inside a hipGraph capture)."""
    lo, hi = _segments(batch, n, device)
    key = (batch._version, int(k), bool(loop), str(device))
    memo = getattr(batch, "_nbd_knn_layout", None)
    if memo is not None and key in memo:
        return memo[key]
    per = torch.clamp((hi - lo).to(torch.int64) - (0 if loop else 1), min=0, max=k)
    off = (torch.cumsum(per, 0) - per).contiguous()
    val = (off, int(per.sum().item()))
    try:
        if memo is None:
            batch._nbd_knn_layout = memo = {}
        memo[key] = val
    except (AttributeError, RuntimeError):
        pass
    return val


def knn_graph(x: torch.Tensor, k: int, batch: torch.Tensor | None = None, loop: bool = False,
              hint: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """int64 edge_index [2, E]: per centre i its k nearest j (ascending (d2, j)); row 0 = j, row 1 = i.
    hint: an earlier edge_index of the same shape for a similar configuration (the previous rollout step): its
    neighbours' current distances bound the search and save one of the two candidate scans; the result does not
    depend on it. out: write into this (2, E) tensor (may be the hint itself)."""
    n = x.shape[0]
    if x.dim() != 2 or x.shape[1] != 3:
        raise _lib.NbdError(f"knn_graph: this build searches 3-D positions (n,3) as the reference does "
                            f"(gnn.py:13), got {tuple(x.shape)}")
    if not loop and hint is None and out is None and not torch.cuda.is_current_stream_capturing():
        # torch_cluster 1.6.3: knn(x, x, k + 1) with self as a candidate, then row == col dropped. One boolean
        # compaction (a host sync), so only outside the rollout's fast path (hint / out given, or inside a graph
        # capture), which masks the diagonal in the kernel instead: the two rules differ only for a centre with
        # >= k + 1 LOWER-indexed bodies at distance exactly 0, which then keeps k + 1 neighbours here.
        ei1 = knn_graph(x, k + 1, batch, loop=True)
        ei = ei1[:, ei1[0] != ei1[1]].contiguous()
        mark(ei, "_nbd_grouped")
        return ei
    pos = x.contiguous()
    _chk(pos, (n, 3), "x")
    dev = pos.device
    lo, hi = _segments(batch, n, dev)
    if lo is None:
        kk = max(min(k, n - (0 if loop else 1)), 0)
        e, off = n * kk, None
    else:
        off, e = knn_layout(batch, n, k, loop, dev)
    if out is not None:
        if out.shape != (2, e) or out.dtype != torch.int64 or not out.is_contiguous() or out.device != dev:
            raise _lib.NbdError(f"knn_graph: out must be a contiguous int64 (2, {e}) tensor on {dev}")
        ei = out
    else:
        ei = torch.empty((2, e), dtype=torch.int64, device=dev)
    use_hint = (hint is not None and lo is None and e > 0 and hint.shape == (2, e) and hint.dtype == torch.int64
                and hint.is_contiguous() and hint.device == dev)
    if e:
        with _lib.on_device(dev):
            if use_hint:
                _lib.check(_lib.lib().nbd_knn_graph_hint_f32(pos.data_ptr(), n, k, int(loop), None, None, None, e,
                                                             ei.data_ptr(), hint.data_ptr(), _stream(dev)),
                           "nbd_knn_graph_hint_f32")
            else:
                _lib.check(_lib.lib().nbd_knn_graph_f32(pos.data_ptr(), n, k, int(loop), _lib.ptr(lo), _lib.ptr(hi),
                                                        _lib.ptr(off), e, ei.data_ptr(), _stream(dev)),
                           "nbd_knn_graph_f32")
    mark(ei, "_nbd_grouped")      # edges come out grouped by centre (row 1 ascending): csr_by_target need not check
    return ei


@dataclass
class RadiusLists:
    """radius_graph in padded form + its transpose (what ContinuousConv aggregates over)."""
    n: int
    cap: int
    nbr: torch.Tensor       # (n, cap) int32, first deg[i] entries valid: neighbours j of centre i, ascending
    deg: torch.Tensor       # (n,) int32
    last: torch.Tensor      # (n,) int32 largest listed j
    rowptr: torch.Tensor    # (n+1,) int32 CSR by neighbour j (= edge_index[0], the aggregation target)
    centres: torch.Tensor   # (>= E,) int32 centres c listing j, ascending per row (first rowptr[n] valid)


def radius_r2(r: float) -> float:
    """torch_cluster 1.6.3 hands its kernel r * r computed in double and cast to fp32 (0.7 -> 0.49000001; the fp32
    product of the fp32 radius would be 0.48999998)."""
    return float(np.float32(float(r) * float(r)))


class RadiusCache:
    """State of nbd_radius_cached_search_f32 for ONE sequence of similar configurations (a rollout): candidate lists
    of the first `wide_cap` indices within r + skin of every centre and the positions they were built at. The search
    result does not depend on it (it is exact either way); it only decides how often the O(n^2) scan runs."""
    SKIN = float(__import__("os").environ.get("NBD_RADIUS_SKIN", "0.15"))          # in units of r
    MARGIN = 0.45        # rebuild when a body has moved MARGIN * skin (0.5 would be the exact bound)

    def __init__(self, wide_cap: int = 192):      # measured on the ContinuousConv rollout step (N = 16 384): 128 / 192 / 256 -> 1.021 / 1.006 / 1.096 ms
        self.wide_cap, self.key, self.state, self.ws = int(wide_cap), None, None, None

    def buffers(self, n, r, cap, dev):
        L = _lib.lib()
        wide = max(self.wide_cap, 4 * cap)
        key = (n, float(r), cap, wide, str(dev))
        if key != self.key:
            self.state = torch.zeros(L.nbd_radius_cached_state_bytes(n, wide), dtype=torch.uint8, device=dev)
            self.ws = torch.empty(max(L.nbd_radius_cached_workspace_bytes(n, wide), 1), dtype=torch.uint8, device=dev)
            self.key = key
        return wide, self.state, self.ws

    def rebuilds(self) -> int:
        """How many times the O(n^2) candidate-list build has run in this state (one host read-back: for reports,
        not for the rollout loop). The first search of a sequence counts as one."""
        if self.state is None:
            return 0
        return int(self.state[:16].view(torch.int32)[3].item())


def radius_lists(pos: torch.Tensor, r: float, batch=None, loop: bool = False, max_num_neighbors: int = 32,
                 transpose: bool = True, scan_transpose: bool = False, cache: RadiusCache | None = None) -> RadiusLists:
    n = pos.shape[0]
    _chk(pos, (n, 3), "pos")
    dev = pos.device
    # torch_cluster 1.6.3: radius(x, x, r, ..., max_num_neighbors if loop else max_num_neighbors + 1) with self as a
    # candidate, THEN row == col is dropped -- so without self loops a centre with >= 33 lower-indexed hits keeps 33
    drop_self = not loop
    cap = int(max_num_neighbors) + (1 if drop_self else 0)
    search_loop = 1
    lo, hi = _segments(batch, n, dev)
    nbr = torch.empty((n, max(cap, 1)), dtype=torch.int32, device=dev)
    deg = torch.empty(n, dtype=torch.int32, device=dev)
    last = torch.empty(n, dtype=torch.int32, device=dev)
    r2 = radius_r2(r)
    L, st = _lib.lib(), _stream(dev)
    rowptr = centres = None
    # streaming search (lane = centre) when its slice lists are small; the one-wave-per-centre search for
    # very large caps (its early exit at `cap` hits does not need them)
    need = L.nbd_radius_search_workspace_bytes(n, cap)
    stream_ok = 0 < need <= (1 << 29)

    use_cache = cache is not None and batch is None and cap > 0 and n > 0 and stream_ok
    if use_cache:       # the rebuild's slice lists grow with the wide cap: keep them under 1 GiB or search plainly
        use_cache = L.nbd_radius_cached_workspace_bytes(n, max(cache.wide_cap, 4 * cap)) <= (1 << 30)

    def search(indeg_ptr):
        if use_cache:
            wide, state, cws = cache.buffers(n, r, cap, dev)
            skin = np.float32(RadiusCache.SKIN) * np.float32(r)
            rw = np.float32(r) + skin
            moved = np.float32(RadiusCache.MARGIN) * skin
            _lib.check(L.nbd_radius_cached_search_f32(pos.data_ptr(), n, r2, float(np.float32(rw * rw)),
                                                      float(np.float32(moved * moved)), search_loop, cap, wide,
                                                      state.data_ptr(), state.numel(), nbr.data_ptr(), deg.data_ptr(),
                                                      last.data_ptr(), indeg_ptr, cws.data_ptr(), cws.numel(), st),
                       "nbd_radius_cached_search_f32")
        elif stream_ok:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)      # only this branch needs it (up to 512 MiB)
            _lib.check(L.nbd_radius_search_ws_f32(pos.data_ptr(), n, r2, search_loop, cap, _lib.ptr(lo), _lib.ptr(hi),
                                                  nbr.data_ptr(), deg.data_ptr(), last.data_ptr(), indeg_ptr,
                                                  ws.data_ptr(), need, st), "nbd_radius_search_ws_f32")
        else:
            _lib.check(L.nbd_radius_search_f32(pos.data_ptr(), n, r2, search_loop, cap, _lib.ptr(lo), _lib.ptr(hi),
                                               nbr.data_ptr(), deg.data_ptr(), last.data_ptr(), indeg_ptr, st),
                       "nbd_radius_search_f32")
        if drop_self and n > 0 and cap > 0:
            _lib.check(L.nbd_radius_drop_self_i32(nbr.data_ptr(), deg.data_ptr(), last.data_ptr(), indeg_ptr, n, cap, st),
                       "nbd_radius_drop_self_i32")
    with _lib.on_device(dev):
        if not transpose:
            search(None)
        else:
            indeg = torch.empty(n, dtype=torch.int32, device=dev)
            rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
            centres = torch.empty(max(n * cap, 1), dtype=torch.int32, device=dev)   # E <= n*cap: no sync needed
            search(indeg.data_ptr())
            _lib.check(L.nbd_exclusive_scan_i32(indeg.data_ptr(), n, rowptr.data_ptr(), st), "exclusive_scan")
            if use_cache and not scan_transpose:   # rows straight from the candidate lists, already in order
                wide, state, _ = cache.buffers(n, r, cap, dev)
                _lib.check(L.nbd_radius_cached_transpose_f32(pos.data_ptr(), n, r2, int(loop), wide, state.data_ptr(),
                                                             state.numel(), last.data_ptr(), rowptr.data_ptr(),
                                                             centres.data_ptr(), st), "nbd_radius_cached_transpose_f32")
            elif scan_transpose:    # O(N^2) scanning transpose (kept as the independent cross-check)
                _lib.check(L.nbd_radius_transpose_fill_f32(pos.data_ptr(), n, r2, int(loop), _lib.ptr(lo), _lib.ptr(hi),
                                                           last.data_ptr(), rowptr.data_ptr(), centres.data_ptr(), st),
                           "radius_transpose_fill")
            else:                   # O(E): scatter + per-row sort
                scratch = torch.empty_like(centres)
                _lib.check(L.nbd_radius_transpose_lists(nbr.data_ptr(), deg.data_ptr(), n, cap, rowptr.data_ptr(),
                                                        indeg.data_ptr(), scratch.data_ptr(), centres.data_ptr(), st),
                           "nbd_radius_transpose_lists")
    return RadiusLists(n, cap, nbr, deg, last, rowptr, centres)


def radius_graph(x: torch.Tensor, r: float, batch=None, loop: bool = False, max_num_neighbors: int = 32) -> torch.Tensor:
    """int64 edge_index [2, E] as torch_geometric.nn.radius_graph returns it (one host sync for E)."""
    lists = radius_lists(x, r, batch, loop, max_num_neighbors, transpose=False)
    n, dev = lists.n, x.device
    ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    L, st = _lib.lib(), _stream(dev)
    with _lib.on_device(dev):
        _lib.check(L.nbd_exclusive_scan_i32(lists.deg.data_ptr(), n, ptr.data_ptr(), st), "exclusive_scan")
        e = int(ptr[n].item()) if n else 0
        ei = torch.empty((2, e), dtype=torch.int64, device=dev)
        if e:
            _lib.check(L.nbd_ell_to_edge_index(lists.nbr.data_ptr(), lists.deg.data_ptr(), ptr.data_ptr(), n, lists.cap,
                                               e, ei.data_ptr(), st), "nbd_ell_to_edge_index")
    return ei


def csr_by_target(edge_index: torch.Tensor, n: int, return_tgt: bool = False):
    """(rowptr int32 [n+1], src int64 [E]) for edges grouped by edge_index[1]; already-grouped input
    (what knn_graph / PyG produce) is used as is, anything else is stably sorted first. With
    return_tgt also the target of every edge in that order."""
    tgt = edge_index[1]
    src = edge_index[0]
    if (marked(edge_index, "_nbd_grouped") and edge_index.is_cuda and edge_index.dtype == torch.int64
            and edge_index.stride(1) == 1):
        # known to be grouped by ascending target (knn_graph's output, collated batches of it): one launch
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=edge_index.device)
        with _lib.on_device(edge_index.device):
            _lib.check(_lib.lib().nbd_rowptr_sorted_i64(tgt.data_ptr(), tgt.numel(), n, rowptr.data_ptr(),
                                                        _stream(edge_index.device)), "nbd_rowptr_sorted_i64")
        return (rowptr, src.contiguous(), tgt.contiguous()) if return_tgt else (rowptr, src.contiguous())
    if not marked(edge_index, "_nbd_grouped") and tgt.numel() > 1 and bool((tgt[1:] < tgt[:-1]).any()):
        order = torch.sort(tgt, stable=True).indices
        tgt, src = tgt[order], src[order]
    counts = torch.bincount(tgt, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device=edge_index.device)
    rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    if return_tgt:
        return rowptr, src.contiguous(), tgt.contiguous()
    return rowptr, src.contiguous()


def csr_by_key(key: torch.Tensor, val: torch.Tensor, n: int, validate: bool = True):
    """(rowptr int32 [n+1], vals int32 [E]): the edge list grouped by `key`, each group's `val`s ascending.
    csr_by_key(edge_index[0], edge_index[1], n) is the adjacency transposed (targets of each source)."""
    if key.dtype != torch.int64 or val.dtype != torch.int64 or key.shape != val.shape or key.dim() != 1:
        raise _lib.NbdError("csr_by_key: key/val must be int64 vectors of equal length")
    if not key.is_cuda:
        raise _lib.NbdError("csr_by_key: tensors must live on the GPU (no CPU path)")
    key, val = key.contiguous(), val.contiguous()
    e, dev = key.numel(), key.device
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    cursor = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    scratch = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    out = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().nbd_csr_by_key_i64(key.data_ptr(), val.data_ptr(), e, n, rowptr.data_ptr(),
                                                 cursor.data_ptr(), scratch.data_ptr(), out.data_ptr(), bad.data_ptr(),
                                                 _lib.current_stream(dev)), "nbd_csr_by_key_i64")
    if validate and int(bad.item()):      # one host sync; out-of-range keys are skipped by the kernels either way
        raise _lib.NbdError(f"csr_by_key: an index lies outside [0, {n})")
    return rowptr, out[:e]
