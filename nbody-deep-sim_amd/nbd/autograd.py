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

    def __init__(self, n, rowptr, src, fixed_k, edge_index=None):
        self.n, self.rowptr, self.src, self.fixed_k = n, rowptr, src, fixed_k
        self._edge_index = edge_index
        self._by_source = None
        self._tgt = None

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
            self._by_source = graphops.csr_by_key(self.src, self.targets(), self.n)
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
        dw = nnops.linear_wgrad(g, x) if ctx.needs_input_grad[1] else None
        db = nnops.colsum(g, rowweight=brs) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
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
        rowptr_e, edge_ids = graphops.csr_by_key(lists.src, torch.arange(e, device=m.device, dtype=torch.int64), n)
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
