"""torch.autograd glue for the HIP layers: each Function's forward AND backward are C-ABI kernels
(csrc/nn.hip, csrc/train.hip); torch only records the graph and owns the tensors. This is what makes
`loss.backward()` of the reference's training step (gnn.py:163-191, contconv.py:242-247) run on the
MI355X kernels while `optimizer.step()` keeps working on ordinary `.grad` fields.

All backward sums have a fixed order (gathers over sorted adjacency, slab partials reduced by a second
kernel), so gradients are bit-identical run to run.
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import graphops, nnops


class EdgeLists:
    """The edges of one batch in the two groupings the kernels gather over: by target (forward and dP)
    and by source (dQ). The by-source lists are built on first use and cached."""

    def __init__(self, n, rowptr, src, fixed_k, tgt=None):
        self.n, self.rowptr, self.src, self.fixed_k = n, rowptr, src, fixed_k
        self._by_source = None
        self._tgt = tgt             # int64 [E] targets in by-target order, when the caller has them (no sync)

    def targets(self):
        """int64 [E] target index of every edge, in by-target order."""
        if self._tgt is None:
            dev = self.src.device
            if self.rowptr is None:
                self._tgt = torch.arange(self.n, device=dev, dtype=torch.int64).repeat_interleave(max(self.fixed_k, 0))
            else:
                deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
                self._tgt = torch.repeat_interleave(torch.arange(self.n, device=dev, dtype=torch.int64), deg)
        return self._tgt

    def by_source(self):
        if self._by_source is None:
            self._by_source = graphops.csr_by_key(self.src, self.targets(), self.n, validate=False)
        return self._by_source

    def csr_rowptr(self):
        """int32 [n+1] by-target row pointer (materialised for regular-k graphs)."""
        if self.rowptr is not None:
            return self.rowptr
        k = max(self.fixed_k, 0)
        return (torch.arange(self.n + 1, device=self.src.device, dtype=torch.int64) * k).to(torch.int32)


class LinearFn(Function):
    """y = act(x w^T + bias_rowscale * b) (nbd_linear_f32); backward: nbd_act_bwd_f32, nbd_linear_f32 with
    w^T (dx), nbd_linear_wgrad_f32 (dw), nbd_colsum_f32 (db)."""

    @staticmethod
    def forward(ctx, x, w, b, act, bias_rowscale):
        w = w.contiguous()
        y = nnops.linear(x, w, b.contiguous() if b is not None else None, act=act, bias_rowscale=bias_rowscale)
        ctx.save_for_backward(x, w, y if act == "tanh" else None, bias_rowscale)
        ctx.act, ctx.has_bias = act, b is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y, brs = ctx.saved_tensors
        dy = dy if dy.stride(1) == 1 else dy.contiguous()
        g = nnops.act_bwd(dy, y, "tanh") if ctx.act == "tanh" else dy
        dx = nnops.linear(g, w.t().contiguous()) if ctx.needs_input_grad[0] else None
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if want_w and want_b:       # the bias gradient as one more column of the weight gradient's product
            dw, db = nnops.linear_wgrad_bias(g, x, rowweight=brs)
        else:
            dw = nnops.linear_wgrad(g, x) if want_w else None
            db = nnops.colsum(g, rowweight=brs) if want_b else None
        return dx, dw, db, None, None


def linear(x, w, b=None, act=None, bias_rowscale=None):
    return LinearFn.apply(x, w, b, act, bias_rowscale)


class EdgeAggregateFn(Function):
    """S_i = aggr_j tanh(P_i + Q_j), aggr in {sum, mean} (nbd_edgeconv_aggregate_f32 and its backward)."""

    @staticmethod
    def forward(ctx, pq, lists, h, aggr):
        s = nnops.edgeconv_aggregate(pq, h, lists.rowptr, lists.src, lists.fixed_k, aggr)
        ctx.save_for_backward(pq)
        ctx.lists, ctx.h, ctx.aggr = lists, h, aggr
        return s

    @staticmethod
    @once_differentiable
    def backward(ctx, ds):
        (pq,) = ctx.saved_tensors
        lists = ctx.lists
        rowptr_t, tgt_t = lists.by_source()
        ds = ds if ds.stride(1) == 1 else ds.contiguous()
        dpq = nnops.edgeconv_aggregate_bwd(pq, ctx.h, ds, lists.rowptr, lists.src, lists.fixed_k, rowptr_t, tgt_t,
                                           ctx.aggr)
        return dpq, None, None, None


class EdgeMessagesFn(Function):
    """m_e = tanh(P_tgt(e) + Q_src(e)) per edge (aggr = max path, nbd_edge_messages_f32)."""

    @staticmethod
    def forward(ctx, pq, lists, h):
        m = nnops.edge_messages(pq, h, lists.src, lists.targets())
        ctx.save_for_backward(m)
        ctx.lists, ctx.h = lists, h
        return m

    @staticmethod
    @once_differentiable
    def backward(ctx, dm):
        (m,) = ctx.saved_tensors
        lists, h = ctx.lists, ctx.h
        dpre = nnops.act_bwd(dm if dm.stride(1) == 1 else dm.contiguous(), m, "tanh")          # (E, h)
        n = lists.n
        dpq = torch.empty((n, 2 * h), dtype=torch.float32, device=m.device)
        nnops.segment_reduce(dpre, lists.csr_rowptr(), n, "sum", out=dpq[:, :h])
        e = lists.src.numel()
        rowptr_e, edge_ids = graphops.csr_by_key(lists.src, torch.arange(e, device=m.device, dtype=torch.int64), n,
                                                validate=False)
        nnops.segment_reduce(dpre.index_select(0, edge_ids.to(torch.int64)), rowptr_e, n, "sum", out=dpq[:, h:])
        return dpq, None, None


class SegmentMaxFn(Function):
    """x_i = max over the rows of target i (empty -> 0); the gradient goes to the first row attaining it."""

    @staticmethod
    def forward(ctx, m, rowptr, n):
        x = nnops.segment_reduce(m, rowptr, n, "max")
        ctx.save_for_backward(m, x, rowptr)
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, dx):
        m, x, rowptr = ctx.saved_tensors
        return nnops.segment_max_bwd(m, x, rowptr, dx if dx.stride(1) == 1 else dx.contiguous()), None, None


class SegmentMulFn(Function):
    """x_i = product over the rows of target i (empty -> 1: torch_scatter's scatter_mul); each row's gradient is the
    product of the others times the target's."""

    @staticmethod
    def forward(ctx, m, rowptr, n):
        x = nnops.segment_reduce(m, rowptr, n, "mul")
        ctx.save_for_backward(m, rowptr)
        ctx.n = n
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, dx):
        m, rowptr = ctx.saved_tensors
        return nnops.segment_mul_bwd(m, rowptr, ctx.n, dx if dx.stride(1) == 1 else dx.contiguous()), None, None


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y = nnops.layernorm(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = nnops.layernorm_bwd(x, gamma, ctx.eps, dy if dy.stride(1) == 1 else dy.contiguous())
        return dx, dg, db, None


class BatchNormActFn(Function):
    """act(BatchNorm1d(x)) with BATCH statistics (training mode); returns (y, batch mean, biased batch var)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act):
        y, mean, var, rstd = nnops.batchnorm_train_fwd(x, gamma, beta, eps, act)
        ctx.save_for_backward(x, gamma, mean, rstd, y)
        ctx.act = act
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dmean, _dvar):
        x, gamma, mean, rstd, y = ctx.saved_tensors
        dx, dg, db = nnops.batchnorm_train_bwd(x, gamma, mean, rstd, ctx.act, y, dy if dy.stride(1) == 1 else dy.contiguous())
        return dx, dg, db, None, None


def batchnorm_act(x, bn: torch.nn.BatchNorm1d, act):
    """Training-mode BatchNorm1d + activation, updating bn's running statistics as torch does
    (momentum blend, unbiased variance, num_batches_tracked)."""
    y, mean, var = BatchNormActFn.apply(x, bn.weight, bn.bias, bn.eps, act)
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            n = x.shape[0]
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var, alpha=mom * n / (n - 1))
    return y


class ContConvFn(Function):
    """ContinuousConv layer (contconv.py:80-98): out = act(scale_n * bin(pos, feat)[n] . filters).
    Backward: dfilters = bin^T (scale g) on the fp32-MFMA wgrad kernel (the binned matrix is recomputed,
    not kept), dfeat = adjoint binning of (scale g) filters^T gathered per source."""

    @staticmethod
    def forward(ctx, feat, filters, pos, fwd_lists, bwd_lists, d, r2, scale, act, cells):
        i_ch, o_ch = filters.shape[3], filters.shape[4]
        rowptr, centres = fwd_lists
        idx, cmap, k = cells
        feat = feat if feat.stride(1) == 1 else feat.contiguous()
        a = nnops.contconv_bin(pos, feat, rowptr, centres, d, r2, cell_map=cmap, cells_out=k)
        w = filters.reshape(d * d * d, i_ch, o_ch).index_select(0, idx).reshape(k * i_ch, o_ch)
        out = nnops.linear(a, w.t().contiguous(), None, act=act, rowscale=scale)
        ctx.save_for_backward(feat, filters, pos, out, scale)
        ctx.fwd_lists, ctx.bwd_lists, ctx.d, ctx.r2, ctx.act, ctx.cells = fwd_lists, bwd_lists, d, r2, act, cells
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        feat, filters, pos, out, scale = ctx.saved_tensors
        d, r2 = ctx.d, ctx.r2
        idx, cmap, k = ctx.cells
        i_ch, o_ch = filters.shape[3], filters.shape[4]
        dout = dout if dout.stride(1) == 1 else dout.contiguous()
        gs = nnops.act_bwd(dout, out if ctx.act == "tanh" else None, ctx.act, rowscale=scale)
        dfilters = dfeat = None
        if ctx.needs_input_grad[1]:
            a = nnops.contconv_bin(pos, feat, ctx.fwd_lists[0], ctx.fwd_lists[1], d, r2, cell_map=cmap, cells_out=k)
            dw = nnops.linear_wgrad(a, gs).reshape(k, i_ch, o_ch)
            del a
            # unreachable grid points never enter a product: their gradient is exactly zero
            dfilters = torch.zeros((d * d * d, i_ch, o_ch), dtype=torch.float32, device=dw.device)
            dfilters.index_copy_(0, idx, dw)
            dfilters = dfilters.reshape(filters.shape)
        if ctx.needs_input_grad[0]:
            w = filters.reshape(d * d * d, i_ch, o_ch).index_select(0, idx).reshape(k * i_ch, o_ch).contiguous()
            da = nnops.linear(gs, w)
            dfeat = nnops.contconv_bin_bwd(pos, da, i_ch, d, r2, cell_map=cmap, cells_out=k, **ctx.bwd_lists)
        return dfeat, dfilters, None, None, None, None, None, None, None, None


_VIRTUAL_ROWPTR = {}


def ell_rowptr(n, cap, device):
    """int32 [n + 1] = i * cap: the row starts of a padded (ELL) list array, built once per (n, cap, device)."""
    key = (int(n), int(cap), str(device))
    rp = _VIRTUAL_ROWPTR.get(key)
    if rp is None:
        if len(_VIRTUAL_ROWPTR) > 64:
            _VIRTUAL_ROWPTR.clear()
        rp = (torch.arange(n + 1, device=device, dtype=torch.int64) * cap).to(torch.int32)
        _VIRTUAL_ROWPTR[key] = rp
    return rp


class ConvGraph:
    """The edges of one ContinuousConv forward pass in the two groupings the fused kernels walk -- by aggregation
    target (forward, filter gradient) and by feature source (feature gradient) -- and the pair lists built from them,
    per filter resolution, on first use or several per launch (prebuild)."""

    def __init__(self, pos, r2, fwd, adj):
        self.pos, self.r2, self.n = pos, float(r2), pos.shape[0]
        self.fwd, self.adj = fwd, adj        # (rowptr, listed nodes, deg or None, edge capacity)
        self._pairs = {}

    @classmethod
    def from_lists(cls, pos, r2, lists):
        """From graphops.radius_lists: its transposed CSR is the forward grouping, its own padded per-centre lists
        (rows of `cap` entries, `deg` valid) the adjoint one -- no conversion."""
        adj = None
        if lists.deg is not None and lists.nbr is not None:
            n, cap = lists.n, lists.nbr.shape[1]
            adj = (ell_rowptr(n, cap, pos.device), lists.nbr.reshape(-1), lists.deg, n * cap)
        return cls(pos, r2, (lists.rowptr, lists.centres, None, lists.centres.numel()), adj)

    @classmethod
    def from_edge_index(cls, pos, r2, rowptr, centres, edge_index):
        rp, tg = graphops.csr_by_key(edge_index[1], edge_index[0], pos.shape[0])
        if tg.numel() == 0:
            tg = torch.zeros(1, dtype=torch.int32, device=pos.device)
        return cls(pos, r2, (rowptr, centres, None, centres.numel()), (rp, tg, None, tg.numel()))

    def _job(self, d, cmap, n_cells, adjoint):
        rowptr, centres, deg, cap_e = self.adj if adjoint else self.fwd
        return dict(rowptr=rowptr, centres=centres, deg=deg, edge_capacity=cap_e, d=d, cell_map=cmap, n_cells=n_cells,
                    adjoint=adjoint, radius_sq=self.r2)

    def prebuild(self, wants):
        """wants: [(d, cell_map, n_cells, adjoint)] -- built four jobs per launch, skipping what exists."""
        todo = []
        for d, cmap, nc, adjoint in wants:
            key = (int(d), int(nc), bool(adjoint))
            if key not in self._pairs and key not in [t[0] for t in todo] and self.n > 0:
                todo.append((key, self._job(d, cmap, nc, adjoint)))
        for lo in range(0, len(todo), 4):
            got = nnops.contconv_pairs_jobs(self.pos, [j for _, j in todo[lo:lo + 4]])
            for (key, _), g in zip(todo[lo:lo + 4], got):
                self._pairs[key] = g

    def pairs(self, d, cmap, n_cells, adjoint=False):
        key = (int(d), int(n_cells), bool(adjoint))
        if key not in self._pairs:
            self.prebuild([(d, cmap, n_cells, adjoint)])
        return self._pairs[key]

    def rows(self, adjoint=False):
        return (self.adj if adjoint else self.fwd)[0]


class ContConvFusedFn(Function):
    """ContinuousConv layer (contconv.py:80-98) on the block-sparse fused kernels, forward AND backward: the binned
    matrix A is formed in neither.
      forward   out = act(scale_n * sum_cells A[n][cell] . F[cell])              nbd_contconv_fused_f32, forward lists
      dfilters  [cell] = sum_n A[n][cell]^T g[n],  g = scale * act'(out) * dout   nbd_contconv_filter_grad_f32
      dfeat     = the forward kernel over the ADJOINT lists (rows = feature sources) with g as the features and every
                  cell's filter transposed: dfeat[c] = sum_cells B[c][cell] . F[cell]^T, B = the adjoint blocks' sums of g rows."""

    @staticmethod
    def forward(ctx, feat, filters, graph, d, scale, act, cells):
        idx, cmap, k = cells
        o_ch = filters.shape[4]
        feat = feat if (feat.stride(1) == 1 and feat.stride(0) % 2 == 0 and feat.data_ptr() % 8 == 0) else feat.contiguous()
        pb, cap_e = graph.pairs(d, cmap, k)
        wf = nnops.contconv_shuffle_filters_kernel(filters, idx)
        out = nnops.contconv_fused(feat, graph.rows(), pb, cap_e, wf, k, o_ch, rowscale=scale, act=act)
        ctx.save_for_backward(feat, filters, out, scale)
        ctx.graph, ctx.d, ctx.act, ctx.cells = graph, d, act, cells
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        feat, filters, out, scale = ctx.saved_tensors
        graph, d = ctx.graph, ctx.d
        idx, cmap, k = ctx.cells
        i_ch, o_ch = filters.shape[3], filters.shape[4]
        dout = dout if dout.stride(1) == 1 else dout.contiguous()
        gs = nnops.act_bwd(dout, out if ctx.act == "tanh" else None, ctx.act, rowscale=scale)
        dfilters = dfeat = None
        if ctx.needs_input_grad[1]:
            pb, cap_e = graph.pairs(d, cmap, k)
            # over the full grid in one call: unreachable grid points never enter a product, their gradient is exactly zero
            dfilters = nnops.contconv_filter_grad_full(feat, gs, graph.rows(), pb, cap_e, k, cmap, d)
        if ctx.needs_input_grad[0]:
            pa, cap_a = graph.pairs(d, cmap, k, adjoint=True)
            wt = nnops.contconv_shuffle_filters_kernel(filters, idx, transposed=True)
            dfeat = nnops.contconv_fused(gs, graph.rows(adjoint=True), pa, cap_a, wt, k, i_ch)
        return dfeat, dfilters, None, None, None, None, None


def _flat_like(tensors):
    """One allocation for all parameter gradients of a model pass, handed out as views shaped like `tensors` (a dozen
    separate allocations are a dozen trips through the caching allocator per training step; each view starts 16-byte
    aligned)."""
    sizes = [(t.numel() + 3) // 4 * 4 for t in tensors]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=tensors[0].device)
    out, at = [], 0
    for t, sz in zip(tensors, sizes):
        out.append(flat[at:at + t.numel()].view(t.shape))
        at += sz
    return out


def _versions(tensors):
    return tuple(t._version for t in tensors)


def _check_once_and_unchanged(ctx, tensors, what):
    """The model-level nodes keep raw pointers to parameters, activations and ONE workspace that the backward pass overwrites:
    a second backward (retain_graph=True) would read what the first one destroyed, and a parameter edited in place between
    forward and backward would be differentiated at its new value. Both are refused loudly (save_for_backward's version check
    does the same for ordinary nodes)."""
    if getattr(ctx, "_nbd_backward_done", False):
        raise RuntimeError(f"{what}: backward through this node a second time -- its workspace was consumed by the first pass "
                           f"(run the forward again; retain_graph is not supported on the one-call training path)")
    if _versions(tensors) != ctx._nbd_versions:
        raise RuntimeError(f"{what}: a parameter or the input was modified in place between forward and backward")
    ctx._nbd_backward_done = True



class GnnModelFn(Function):
    """GraphModel.forward (gnn.py:130-148) as ONE autograd node: nbd_gnn_train_forward_f32 enqueues the whole forward and
    keeps its activations in a workspace, nbd_gnn_train_backward_f32 the whole adjoint, writing every parameter gradient
    (csrc/train_model.hip). The per-layer Functions above remain for what this does not cover (aggr = "max", dropout)."""

    @staticmethod
    def forward(ctx, x_in, lists, spec, *params):
        import ctypes
        from . import _lib
        n, dev = x_in.shape[0], x_in.device
        p = [t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float() for t in params]
        a = _lib.GnnTrainArgs()
        a.n, a.rowptr, a.src, a.fixed_k = n, _lib.ptr(lists.rowptr), _lib.ptr(lists.src), int(lists.fixed_k)
        a.aggr = 1 if spec["aggr"] == "mean" else 0
        a.x, a.ldx, a.f = x_in.data_ptr(), x_in.stride(0), x_in.shape[1]
        k = 0
        a.n_enc = len(spec["enc_dims"]) - 1 if spec["enc_dims"] else 0
        for i in range(a.n_enc):
            a.enc_w[i], a.enc_b[i] = p[k].data_ptr(), p[k + 1].data_ptr()
            k += 2
        for i, d in enumerate(spec["enc_dims"] or []):
            a.enc_dim[i] = d
        a.n_layers, a.h = spec["n_layers"], spec["h"]
        for l in range(a.n_layers):
            a.w1[l], a.b1[l], a.w2[l], a.b2[l] = (p[k + j].data_ptr() for j in range(4))
            k += 4
        a.ln_g, a.ln_b, a.ln_eps = p[k].data_ptr(), p[k + 1].data_ptr(), float(spec["ln_eps"])
        k += 2
        a.n_head = len(spec["head_dims"]) - 1
        for i in range(a.n_head):
            a.head_w[i], a.head_b[i] = p[k].data_ptr(), p[k + 1].data_ptr()
            k += 2
        for i, d in enumerate(spec["head_dims"]):
            a.head_dim[i] = d
        out = torch.empty((n, spec["head_dims"][-1]), dtype=torch.float32, device=dev)
        a.out, a.ldout = out.data_ptr(), out.stride(0)
        L = _lib.lib()
        need = L.nbd_gnn_train_workspace_bytes(ctypes.byref(a))
        if need == 0:
            raise _lib.NbdError("nbd_gnn_train_workspace_bytes: configuration rejected")
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
        with _lib.on_device(dev):
            _lib.check(L.nbd_gnn_train_forward_f32(ctypes.byref(a), _lib.current_stream(dev)), "nbd_gnn_train_forward_f32")
        ctx.args, ctx.keep, ctx.lists = a, (x_in, ws, p, out.data_ptr()), lists      # (the output itself is not read back)
        ctx._nbd_versions = _versions([x_in, *params])
        ctx._nbd_params = [x_in, *params]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        import ctypes
        from . import _lib
        a, (x_in, ws, p, _), lists = ctx.args, ctx.keep, ctx.lists
        _check_once_and_unchanged(ctx, ctx._nbd_params, "GnnModelFn")
        dev = x_in.device
        dout = dout if (dout.stride(1) == 1 and dout.dtype == torch.float32) else dout.contiguous().float()
        rowptr_t, tgt_t = lists.by_source()
        a.rowptr_t, a.tgt_t = rowptr_t.data_ptr(), tgt_t.data_ptr()
        grads = _flat_like(p)
        g = _lib.GnnTrainGrads()
        k = 0
        for i in range(a.n_enc):
            g.enc_w[i], g.enc_b[i] = grads[k].data_ptr(), grads[k + 1].data_ptr()
            k += 2
        for l in range(a.n_layers):
            g.w1[l], g.b1[l], g.w2[l], g.b2[l] = (grads[k + j].data_ptr() for j in range(4))
            k += 4
        g.ln_g, g.ln_b = grads[k].data_ptr(), grads[k + 1].data_ptr()
        k += 2
        for i in range(a.n_head):
            g.head_w[i], g.head_b[i] = grads[k].data_ptr(), grads[k + 1].data_ptr()
            k += 2
        with _lib.on_device(dev):
            _lib.check(_lib.lib().nbd_gnn_train_backward_f32(ctypes.byref(a), dout.data_ptr(), dout.stride(0), ctypes.byref(g),
                                                             _lib.current_stream(dev)), "nbd_gnn_train_backward_f32")
        return (None, None, None, *grads)


class ContConvModelFn(Function):
    """ContinuousConvModel.forward (contconv.py:218-234) as ONE autograd node over nbd_cc_train_forward_f32 /
    nbd_cc_train_backward_f32 (csrc/train_model.hip); the pair lists come from the caller's ConvGraph (one launch for all
    layers and both groupings). spec: dict built by ContinuousConvModel._one_call_train."""

    @staticmethod
    def forward(ctx, x, graph, spec, *params):
        import ctypes
        from . import _lib
        n, dev = x.shape[0], x.device
        p = [t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float() for t in params]
        a = _lib.CcTrainArgs()
        a.n, a.x, a.ldx, a.in_ch = n, x.data_ptr(), x.stride(0), x.shape[1]
        k = 0
        enc_dims = spec["enc_dims"]
        a.n_enc = len(enc_dims) - 1 if enc_dims else 0
        for i in range(a.n_enc):
            a.enc_w[i], a.enc_b[i] = p[k].data_ptr(), p[k + 1].data_ptr()
            k += 2
        for i, d in enumerate(enc_dims or []):
            a.enc_dim[i] = d
        a.enc_bn = 1 if spec["bns"] else 0
        for i, bn in enumerate(spec["bns"] or []):
            a.bn_g[i], a.bn_b[i], a.bn_eps[i] = p[k].data_ptr(), p[k + 1].data_ptr(), float(bn.eps)
            k += 2
            if bn.track_running_stats and bn.running_mean is not None:
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                a.bn_rmean[i], a.bn_rvar[i], a.bn_momentum[i] = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(mom)
        layers = spec["layers"]                  # [(d, idx int64, cmap int32, n_cells)]
        a.n_layers, a.cdim = len(layers), spec["cdim"]
        keep = []
        for l, (d, idx, cmap, nc) in enumerate(layers):
            a.filt[l] = p[k].data_ptr()
            k += 1
            a.kept[l], a.cell_map[l], a.n_cells[l], a.cells_total[l] = idx.data_ptr(), cmap.data_ptr(), nc, d * d * d
            pf, cap_f = graph.pairs(d, cmap, nc)
            a.pairs_fwd[l] = pf.data_ptr()
            keep.append(pf)
            if l > 0 or a.n_enc > 0:
                pa, cap_a = graph.pairs(d, cmap, nc, adjoint=True)
                a.pairs_adj[l] = pa.data_ptr()
                a.cap_adj = cap_a
                keep.append(pa)
            a.cap_fwd = cap_f
        a.rowptr_fwd = graph.rows().data_ptr()
        if graph.adj is not None:
            a.rowptr_adj = graph.rows(adjoint=True).data_ptr()
        scale = None
        if spec["mean"]:
            d0, _, cmap0, nc0 = layers[0]
            pb, cap_e = graph.pairs(d0, cmap0, nc0)
            scale = nnops.contconv_pairs_inv_degree(pb, n, cap_e, nc0)
            a.scale = scale.data_ptr()
        a.ln_g, a.ln_b, a.ln_eps = p[k].data_ptr(), p[k + 1].data_ptr(), float(spec["ln_eps"])
        k += 2
        a.n_head = len(spec["head_dims"]) - 1
        for i in range(a.n_head):
            a.head_w[i], a.head_b[i] = p[k].data_ptr(), p[k + 1].data_ptr()
            k += 2
        for i, d in enumerate(spec["head_dims"]):
            a.head_dim[i] = d
        out = torch.empty((n, spec["head_dims"][-1]), dtype=torch.float32, device=dev)
        a.out, a.ldout = out.data_ptr(), out.stride(0)
        L = _lib.lib()
        need = L.nbd_cc_train_workspace_bytes(ctypes.byref(a))
        if need == 0:
            raise _lib.NbdError("nbd_cc_train_workspace_bytes: configuration rejected")
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
        with _lib.on_device(dev):
            _lib.check(L.nbd_cc_train_forward_f32(ctypes.byref(a), _lib.current_stream(dev)), "nbd_cc_train_forward_f32")
        ctx.args, ctx.keep, ctx.spec = a, (x, ws, p, graph, keep, scale), spec
        ctx._nbd_versions = _versions([x, *params])
        ctx._nbd_params = [x, *params]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        import ctypes
        from . import _lib
        a, (x, ws, p, graph, keep, scale), spec = ctx.args, ctx.keep, ctx.spec
        _check_once_and_unchanged(ctx, ctx._nbd_params, "ContConvModelFn")
        dev = x.device
        dout = dout if (dout.stride(1) == 1 and dout.dtype == torch.float32) else dout.contiguous().float()
        grads = _flat_like(p)
        g = _lib.CcTrainGrads()
        k = 0
        for i in range(a.n_enc):
            g.enc_w[i], g.enc_b[i] = grads[k].data_ptr(), grads[k + 1].data_ptr()
            k += 2
        for i in range(len(spec["bns"] or [])):
            g.bn_g[i], g.bn_b[i] = grads[k].data_ptr(), grads[k + 1].data_ptr()
            k += 2
        for l in range(a.n_layers):
            g.filt[l] = grads[k].data_ptr()
            k += 1
        g.ln_g, g.ln_b = grads[k].data_ptr(), grads[k + 1].data_ptr()
        k += 2
        for i in range(a.n_head):
            g.head_w[i], g.head_b[i] = grads[k].data_ptr(), grads[k + 1].data_ptr()
            k += 2
        with _lib.on_device(dev):
            _lib.check(_lib.lib().nbd_cc_train_backward_f32(ctypes.byref(a), dout.data_ptr(), dout.stride(0), ctypes.byref(g),
                                                            _lib.current_stream(dev)), "nbd_cc_train_backward_f32")
        return (None, None, None, *grads)
