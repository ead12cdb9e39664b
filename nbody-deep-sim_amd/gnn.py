"""MI355X drop-in for the reference's gnn.py: `transform_to_graph` and `GraphModel` with the same
constructor arguments, attributes (`neighbors`, ...), parameter names (`gnns.{l}.nn.{0,2}.*`,
`node_encoder.lins.*`, `layer_norm.*`, `output.*` -- reference state_dicts load unchanged) and the
inference methods `forward(data)`, `predict(pos, feat)`, `predict_graph`, `eval_graph_batch`.

The forward pass runs in hand-written HIP kernels (csrc/graph.hip, csrc/nn.hip):
  knn_graph            -> nbd_knn_graph_f32                         (gnn.py:13)
  EdgeConv layer l     -> nbd_linear_f32 [x -> P|Q]  +  nbd_edgeconv_aggregate_f32  +  nbd_linear_f32 [W2]
  LayerNorm + head     -> nbd_layernorm_f32 + nbd_linear_f32        (gnn.py:146-148)
EdgeConv is evaluated in an algebraically factored form (exact in real arithmetic, differs from the
per-edge order by fp32 rounding only):
  nn([x_i || x_j - x_i]) = W2 tanh((W1a - W1b) x_i + b1 + W1b x_j) + b2 = W2 tanh(P_i + Q_j) + b2
  sum/mean over j commute with the affine W2, so W2 is applied once per node, not once per edge.
Training (gnn.py:150-191): when gradients are enabled, forward() runs the same kernels through
torch.autograd Functions (nbd/autograd.py) whose backward passes are HIP kernels too (csrc/train.hip);
`compute_loss`, `train_batch` and `train_graph_batch` then behave as in the reference, and any
torch.optim optimiser steps on the `.grad` fields.
"""
from __future__ import annotations

import ctypes
import os
import time

import torch
from torch.nn import LayerNorm, Linear, ModuleList, Sequential, Tanh

from nbd import autograd as ag
from nbd import graphops, nnops
from nbd._lib import NbdError, NbdUnsupported
from nbd.data import Data


def transform_to_graph(positions, features, y, batch=None, neighbors=50, device="cuda"):
    """gnn.py:11-22: kNN graph (loop=False) + Data(x=[positions|features], edge_index, y, batch)."""
    graph = graphops.knn_graph(positions, k=neighbors, batch=batch, loop=False)
    return Data(x=torch.cat((positions, features), dim=-1), edge_index=graph, edge_attr=None, y=y, batch=batch)


class MLP(torch.nn.Module):
    """Parameter container with torch_geometric.nn.MLP's layout (lins.{i}, norms.{i}.module) for the
    way the reference builds it: act='tanh', plain last layer, BatchNorm unless norm=None."""

    def __init__(self, channels, norm="batch_norm", dropout=0.0):
        super().__init__()
        self.channels = list(channels)
        self.lins = ModuleList(Linear(a, b) for a, b in zip(channels[:-1], channels[1:]))
        self.norms = ModuleList()
        for c in channels[1:-1]:
            if norm is None:
                self.norms.append(torch.nn.Identity())
            else:
                holder = torch.nn.Module()
                holder.module = torch.nn.BatchNorm1d(c)
                self.norms.append(holder)
        self.has_norm = norm is not None
        self.dropout = dropout

    def folded(self):
        """[(W, b, act)] with the eval-mode BatchNorm folded into the preceding Linear."""
        out = []
        last = len(self.lins) - 1
        for i, lin in enumerate(self.lins):
            w, b = lin.weight.detach(), lin.bias.detach()
            if i < last and self.has_norm:
                bn = self.norms[i].module
                s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
                w = w * s.unsqueeze(1)
                b = (b - bn.running_mean) * s + bn.bias.detach()
            out.append((w.contiguous(), b.contiguous(), "tanh" if i < last else None))
        return out


class EdgeConv(torch.nn.Module):
    """Parameter holder named like torch_geometric.nn.EdgeConv (`nn`, `aggr`)."""

    def __init__(self, nn, aggr):
        super().__init__()
        self.nn, self.aggr = nn, aggr


def run_chain(x, chain, out_last=None):
    """Apply [(W, b, act)] with nbd_linear_f32; the last layer may write into `out_last` (a slice)."""
    for li, (w, b, act) in enumerate(chain):
        x = nnops.linear(x, w, b, act=act, out=out_last if li == len(chain) - 1 else None)
    return x


def head_chain(output):
    """torch Sequential(Linear, Tanh, Linear, ...) or a single Linear -> [(W, b, act)]."""
    if isinstance(output, Linear):
        return [(output.weight.detach().contiguous(), output.bias.detach().contiguous(), None)]
    lins = [m for m in output if isinstance(m, Linear)]
    return [(l.weight.detach().contiguous(), l.bias.detach().contiguous(), "tanh" if i < len(lins) - 1 else None)
            for i, l in enumerate(lins)]


class _WeightCache:
    """Derived (folded / re-laid-out) weights, rebuilt when any parameter or buffer changes."""

    def __init__(self, module):
        self.module, self.key, self.value = module, None, None

    def get(self, builder):
        # every parameter and buffer of the module tree, by a plain walk over the modules' own dicts: parameters() /
        # buffers() build names and de-duplicate through generators -- 40 % of an eager predict()'s host time, measured
        # (tools/profile_predict_host.py) -- and the key only needs the tensors
        tensors, stack = [], [self.module]
        while stack:
            mod = stack.pop()
            tensors.extend(t for t in mod._parameters.values() if t is not None)
            tensors.extend(t for t in mod._buffers.values() if t is not None)
            stack.extend(c for c in mod._modules.values() if c is not None)
        key = tuple((t.data_ptr(), t._version, t.device) for t in tensors)
        if key != self.key:
            with torch.no_grad():
                self.value = builder()
            self.key = key
        return self.value


def ensure_eval(module):
    """module.eval() only if some module of the tree is in training mode: Module.eval() re-assigns `training` on every
    submodule through __setattr__ on every call (a third of an eager predict()'s host time)."""
    stack = [module]
    while stack:
        mod = stack.pop()
        if mod.training:
            module.eval()
            return
        stack.extend(c for c in mod._modules.values() if c is not None)


class GraphModel(torch.nn.Module):
    def __init__(self, input_dim=1, output_hiddens=None, output_dim=3, node_encoder_dims=None, gnn_dim=128,
                 encoder_dropout=0.0, message_passing_steps=4, aggr="sum", device="cpu", neighbors=50,
                 scale_factor=1):
        super().__init__()
        self.device = device
        self.neighbors = neighbors
        self.node_encoder_dims = node_encoder_dims
        self.message_passing_steps = message_passing_steps
        self.aggr = aggr
        self.output_hiddens = output_hiddens
        self.output_dim = output_dim
        self.input_dim = input_dim
        self.gnn_dim = gnn_dim
        self.encoder_dropout = encoder_dropout
        self.scale_factor = scale_factor
        if aggr not in ("sum", "add", "mean", "max"):
            raise NotImplementedError(f"aggr={aggr!r}: supported aggregations are sum/add/mean (second Linear "
                                      "hoisted out of the edge sum) and max (per-edge messages materialised)")
        if node_encoder_dims:                                                     # gnn.py:56-65
            self.node_encoder = MLP([input_dim] + list(node_encoder_dims) + [gnn_dim], norm=None,
                                    dropout=encoder_dropout)
        else:
            self.node_encoder = torch.nn.Identity()
        self.gnns = ModuleList()                                                  # gnn.py:71-95
        for i in range(message_passing_steps):
            fin = input_dim if (i == 0 and node_encoder_dims is None) else gnn_dim
            self.gnns.append(EdgeConv(nn=Sequential(Linear(fin * 2, gnn_dim), Tanh(), Linear(gnn_dim, gnn_dim)),
                                      aggr=aggr))
        out_dim = gnn_dim + input_dim if node_encoder_dims is None else gnn_dim * 2   # gnn.py:97-100
        self.layer_norm = LayerNorm(out_dim)
        if output_hiddens:                                                        # gnn.py:105-114
            layers, dims = [], [out_dim] + list(output_hiddens) + [output_dim]
            for i in range(len(dims) - 1):
                layers.append(Linear(dims[i], dims[i + 1]))
                if i < len(dims) - 2:
                    layers.append(Tanh())
            self.output = Sequential(*layers)
        else:
            self.output = Linear(out_dim, output_dim)
        self._cache = _WeightCache(self)
        self.use_fused = True          # one launch per EdgeConv layer when the shapes allow (csrc/gnn_fused.hip)
        self._fused_out = None
        self._out_hint = None
        self._brs_const = None
        self._knn_buf = None
        self._one_call = None          # cached nbd_gnn_forward_args of predict() (see _one_call_plan)
        self.last_path = None          # which path the last inference call took: "one_call+tables" / "one_call" (+ "+pre_advance"),
                                       # "fused" (per-kernel fused layers), "general" (linear + aggregate kernels), "max"
        self.use_exp_tables = os.environ.get("NBD_GNN_EXP_TABLES", "1") != "0"   # 0: the one-call pass without tables
        self.use_one_call = True       # predict(): search + layers through ONE C-ABI call when the configuration allows
        self.to(device)

    def get_config(self):
        return {"input_dim": self.input_dim, "output_hiddens": self.output_hiddens, "output_dim": self.output_dim,
                "node_encoder_dims": self.node_encoder_dims, "gnn_dim": self.gnn_dim,
                "encoder_dropout": self.encoder_dropout, "message_passing_steps": self.message_passing_steps,
                "aggr": self.aggr, "device": self.device, "neighbors": self.neighbors}

    # ------------------------------------------------------------------ derived weights
    def _build_weights(self):
        layers = []
        for g in self.gnns:
            w1, b1 = g.nn[0].weight.detach(), g.nn[0].bias.detach()
            f = w1.shape[1] // 2
            w1a, w1b = w1[:, :f], w1[:, f:]
            wpq = torch.cat([w1a - w1b, w1b], dim=0).contiguous()                 # (2H, F): rows P then Q
            bpq = torch.cat([b1, torch.zeros_like(b1)]).contiguous()
            layers.append((wpq, bpq, g.nn[2].weight.detach().contiguous(), g.nn[2].bias.detach().contiguous()))
        enc = self.node_encoder.folded() if isinstance(self.node_encoder, MLP) else None
        # transposed copies ([in][out]) for the fused layer kernel's LDS mat-vecs
        fused = [(wpq.t().contiguous(), w2.t().contiguous()) for (wpq, _, w2, _) in layers]
        # layer l's second Linear folded into layer l+1's [P|Q] Linear (both act on the node's aggregate, nothing
        # non-linear in between): next_pq = (Wpq' W2) S + beta (Wpq' b2) + bpq'  -- one mat-vec per node, not two
        folded = []
        for li in range(len(layers) - 1):
            wpq_n, w2, b2 = layers[li + 1][0], layers[li][2], layers[li][3]
            m = nnops.linear(wpq_n, w2.t().contiguous())                          # (2H, H) = Wpq' W2
            c = nnops.linear(b2.unsqueeze(0).contiguous(), wpq_n).reshape(-1)     # (2H,)  = Wpq' b2
            folded.append((m.t().contiguous(), c.contiguous()))
        head = head_chain(self.output)
        return {"enc": enc, "layers": layers, "head": head, "fused_t": fused, "folded": folded,
                "head_plan": nnops.ln_mlp_head_plan(self.layer_norm.normalized_shape[0], head, self.layer_norm.weight.detach(),
                                                            self.layer_norm.bias.detach())}

    # ------------------------------------------------------------------ forward (gnn.py:130-148)
    def forward(self, data):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_autograd(data)
        if self.training and self.encoder_dropout > 0 and isinstance(self.node_encoder, MLP):
            return self._forward_autograd(data)          # dropout active: not the folded inference path
        x7 = data.x
        if not x7.is_cuda:
            raise NbdError("GraphModel.forward: data must live on the GPU (no CPU path)")
        x_in = torch.cat((x7[:, :3], x7[:, 6:]), dim=-1) if self.input_dim == 4 else x7
        return self._forward_inference(x_in.to(torch.float32), data.edge_index, getattr(data, "_regular_k", None))

    def _forward_inference(self, x_in, ei, reg, out=None):
        """Inference forward on the model input x_in (n, input_dim) = [pos | mass] or [pos | vel | mass]."""
        w = self._cache.get(self._build_weights)
        n = x_in.shape[0]
        h = self.gnn_dim
        dev = x_in.device
        enc_dim = self.input_dim if w["enc"] is None else h
        # edges grouped by target (edge_index[1]); regular kNN output needs no CSR
        e = ei.shape[1]
        fixed_k, rowptr, src = -1, None, ei[0].contiguous()
        if reg is not None and n * reg == e:
            fixed_k = reg
        else:
            rowptr, src = graphops.csr_by_target(ei, n)
        aggr = "mean" if self.aggr == "mean" else "sum"
        if self.use_fused and self.aggr != "max" and w["enc"] is None:
            # the fused layer kernels read the encoder output through (pointer, row stride): the model input
            # itself when there is no encoder -- no concatenation buffer, no copy
            x_c = x_in if x_in.stride(1) == 1 else x_in.contiguous()
            self._out_hint = out           # a caller-owned (n, output_dim) buffer the last fused layer may write into
            done = self._forward_fused(w, x_c, n, h, rowptr, src, fixed_k, aggr, None)
            self._out_hint = None
            if done:
                self.last_path = "fused"
                return self._fused_out
        cat_buf = torch.empty((n, enc_dim + h), dtype=torch.float32, device=dev)
        enc_view, gnn_view = cat_buf[:, :enc_dim], cat_buf[:, enc_dim:]
        if w["enc"] is None:
            enc_view.copy_(x_in)
        else:
            run_chain(x_in.contiguous(), w["enc"], out_last=enc_view)
        if self.aggr == "mean":
            brs_mode = 2
        else:
            brs_mode = 1
        if rowptr is None:
            val = float(fixed_k) if brs_mode == 1 else (1.0 if fixed_k > 0 else 0.0)
            key = (n, val, str(dev))
            if self._brs_const is None or self._brs_const[0] != key:       # a constant vector: built once
                self._brs_const = (key, torch.full((n,), val, dtype=torch.float32, device=dev))
            brs = self._brs_const[1]
        else:
            brs = nnops.degree_scale(rowptr, n, brs_mode, dev)
        if self.aggr == "max":
            self.last_path = "max"
            return self._forward_max(w, enc_view, gnn_view, cat_buf, n, h, rowptr, src, fixed_k, ei)
        if self.use_fused and self._forward_fused(w, enc_view, n, h, rowptr, src, fixed_k, aggr, cat_buf):
            self.last_path = "fused"
            return self._fused_out
        self.last_path = "general"
        x = enc_view
        for li, (wpq, bpq, w2, b2) in enumerate(w["layers"]):
            pq = nnops.linear(x, wpq, bpq)                                        # (n, 2H) = [P | Q]
            s = nnops.edgeconv_aggregate(pq, h, rowptr, src, fixed_k, aggr)
            last = li == len(w["layers"]) - 1
            x = nnops.linear(s, w2, b2, bias_rowscale=brs, out=gnn_view if last else None)
        if w["head_plan"] is not None:
            return nnops.ln_mlp_head(cat_buf, self.layer_norm.weight.detach(), self.layer_norm.bias.detach(),
                                     self.layer_norm.eps, w["head_plan"])
        ln = nnops.layernorm(cat_buf, self.layer_norm.weight.detach(), self.layer_norm.bias.detach(),
                             self.layer_norm.eps)
        return run_chain(ln, w["head"])

    def _forward_max(self, w, enc_view, gnn_view, cat_buf, n, h, rowptr, src, fixed_k, ei):
        """aggr="max": x_i' = max_j (W2 tanh(P_i + Q_j) + b2); nodes without edges get 0 (PyG fills empty
        max-aggregations with 0)."""
        dev = enc_view.device
        if rowptr is None:
            rowptr = torch.arange(0, (n + 1) * fixed_k, max(fixed_k, 1), dtype=torch.int32, device=dev)[:n + 1] \
                if fixed_k > 0 else torch.zeros(n + 1, dtype=torch.int32, device=dev)
            tgt = torch.arange(n, device=dev, dtype=torch.int64).repeat_interleave(max(fixed_k, 0))
        else:
            tgt = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64),
                                          (rowptr[1:] - rowptr[:-1]).to(torch.int64))
        x = enc_view
        for li, (wpq, bpq, w2, b2) in enumerate(w["layers"]):
            pq = nnops.linear(x, wpq, bpq)
            msg = nnops.linear(nnops.edge_messages(pq, h, src, tgt), w2, b2)
            last = li == len(w["layers"]) - 1
            x = nnops.segment_reduce(msg, rowptr, n, "max", out=gnn_view if last else None)
        ln = nnops.layernorm(cat_buf, self.layer_norm.weight.detach(), self.layer_norm.bias.detach(),
                             self.layer_norm.eps)
        return run_chain(ln, w["head"])

    def _forward_fused(self, w, enc, n, h, rowptr, src, fixed_k, aggr, cat_buf):
        """One nbd_gnn_layer_f32 launch per EdgeConv layer (aggregation + W2 + next [P|Q] or
        LayerNorm + head). Returns False, having launched nothing, if a shape is outside the fused
        kernel's limits; the general path above then runs."""
        layers, head = w["layers"], w["head"]
        e = enc.shape[1]
        dev = enc.device
        n_layers = len(layers)
        single_head = len(head) == 1 and head[0][0].shape[0] <= 8
        # feasibility first, so that a refusal never leaves a half-run forward behind
        if h > 128 or e > 256:
            return False
        for li in range(n_layers):
            kp = 64 * ((h + 63) // 64)
            lds = kp * 2 * h if li < n_layers - 1 else kp * h        # folded [P|Q] matrix, or W2^T on the last layer
            if lds * 4 > 64 * 1024:
                return False
        ln_g, ln_b = self.layer_norm.weight.detach(), self.layer_norm.bias.detach()
        pq = None
        if e > 8:                                           # first [P|Q] as a plain Linear
            pq = nnops.linear(enc, layers[0][0], layers[0][1])
        for li, (wpq, bpq, w2, b2) in enumerate(layers):
            kw = dict(n=n, h=h, aggr=aggr, rowptr=rowptr, src=src, fixed_k=fixed_k, w2t=w["fused_t"][li][1], b2=b2)
            if pq is not None:
                kw["pq"] = pq
            else:
                kw.update(x=enc, f=e, wpq=wpq, bpq=layers[li][1][:h].contiguous())
            if li < n_layers - 1:
                nxt = torch.empty((n, 2 * h), dtype=torch.float32, device=dev)
                m_t, c = w["folded"][li]
                kw_f = dict(kw, w2t=None, b2=c)
                ok = nnops.gnn_layer(epilogue="next_pq_folded", w_ep=m_t, b_ep=layers[li + 1][1], ep_out=2 * h,
                                     out=nxt, **kw_f)
                pq = nxt
            elif single_head:
                hint = getattr(self, "_out_hint", None)
                if (hint is not None and tuple(hint.shape) == (n, head[0][0].shape[0]) and hint.dtype == torch.float32
                        and hint.is_contiguous() and hint.device == dev):
                    out = hint
                else:
                    out = torch.empty((n, head[0][0].shape[0]), dtype=torch.float32, device=dev)
                kick = getattr(self, "_kick_hint", None)      # (vel, c): the caller's half-kick, done in the epilogue
                kw_k = dict(kick_vel=kick[0], kick_c=kick[1]) if kick is not None else {}
                ok = nnops.gnn_layer(epilogue="final_head", w_ep=head[0][0], b_ep=head[0][1],
                                     ep_out=head[0][0].shape[0], enc=enc, e=e, ln_g=ln_g, ln_b=ln_b,
                                     ln_eps=self.layer_norm.eps, out=out, **kw_k, **kw)
                self._fused_out = out
                self._kick_done = kick is not None
            else:
                z = torch.empty((n, e + h), dtype=torch.float32, device=dev)
                ok = nnops.gnn_layer(epilogue="final_ln", enc=enc, e=e, ln_g=ln_g, ln_b=ln_b,
                                     ln_eps=self.layer_norm.eps, out=z, **kw)
                if ok:
                    self._fused_out = run_chain(z, head)
            if not ok:
                raise NbdError("fused GNN layer refused a shape its feasibility check accepted")
        return True

    # ------------------------------------------------------------------ inference API
    # ------------------------------------------------------------------ predict() as ONE C-ABI call
    def _one_call_plan(self, w, n, kk, k, dev, ldx):
        """nbd_gnn_forward_args for (these weights, n, k): every pointer that does not change from call to call filled
        in once -- weights, the intermediate [P | Q] buffers, LayerNorm / head -- or None when the configuration is not
        the all-fused one (no encoder, input_dim <= 8, sum / mean, single-Linear head of <= 8 outputs)."""
        from nbd import _lib
        layers, head = w["layers"], w["head"]
        h, e, n_layers = self.gnn_dim, self.input_dim, len(layers)
        if (not self.use_fused or self.aggr == "max" or w["enc"] is not None or e > 8 or h > 128
                or n_layers > _lib.GNN_MAX_LAYERS or len(head) != 1 or head[0][0].shape[0] > 8):
            return None
        aggr = "mean" if self.aggr == "mean" else "sum"
        fa = _lib.GnnForwardArgs()
        keep = [w]
        dummy_x = torch.empty((1, ldx), dtype=torch.float32, device=dev)        # layout carrier; pointers patched per call
        dummy_out = torch.empty((1, head[0][0].shape[0]), dtype=torch.float32, device=dev)
        pq = None
        for li, (wpq, bpq, w2, b2) in enumerate(layers):
            kw = dict(n=n, h=h, aggr=aggr, rowptr=None, src=None, fixed_k=kk, w2t=w["fused_t"][li][1], b2=b2)
            if pq is not None:
                kw["pq"] = pq
            else:
                bp = layers[li][1][:h].contiguous()
                keep.append(bp)
                kw.update(x=dummy_x, f=e, wpq=wpq, bpq=bp)
            if li < n_layers - 1:
                nxt = torch.empty((n, 2 * h), dtype=torch.float32, device=dev)
                keep.append(nxt)
                m_t, c = w["folded"][li]
                a = nnops.gnn_layer_args(epilogue="next_pq_folded", w_ep=m_t, b_ep=layers[li + 1][1], ep_out=2 * h, out=nxt,
                                         **dict(kw, w2t=None, b2=c))
                pq = nxt
            else:
                a = nnops.gnn_layer_args(epilogue="final_head", w_ep=head[0][0], b_ep=head[0][1], ep_out=head[0][0].shape[0],
                                         enc=dummy_x, e=e, ln_g=self.layer_norm.weight.detach(),
                                         ln_b=self.layer_norm.bias.detach(), ln_eps=self.layer_norm.eps, out=dummy_out, **kw)
            if a is None:
                return None
            a.ldx = ldx if a.x else a.ldx
            fa.layers[li] = a
        fa.n, fa.k, fa.loop, fa.n_layers = n, k, 0, n_layers
        fa.layers[n_layers - 1].ldenc = ldx
        fa.layers[n_layers - 1].ldout = head[0][0].shape[0]
        # room for the layers' exponential tables (include/nbd.h: nbd_gnn_layer_args.epq) where the configuration has them
        need = _lib.lib().nbd_gnn_forward_workspace_bytes(ctypes.byref(fa)) if self.use_exp_tables else 0
        if need:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            keep.append(ws)
            fa.workspace, fa.workspace_bytes = ws.data_ptr(), need
        return {"fa": fa, "keep": keep, "out_dim": head[0][0].shape[0]}

    def _predict_one_call(self, x_in, pos, k, out=None, kick=None, advance=None):
        """kNN graph + all fused layers through nbd_gnn_forward_f32, or None when this call cannot go that way (first
        call of a sequence: no previous graph to reuse as buffer and hint; configuration not all-fused)."""
        from nbd import _lib
        n, dev = pos.shape[0], pos.device
        kk = max(min(k, n - 1), 0)
        buf = self._knn_buf
        if (not self.use_one_call or n == 0 or kk == 0 or buf is None or buf.shape != (2, n * kk) or buf.device != dev or x_in.dtype != torch.float32
                or x_in.stride(1) != 1 or pos.dtype != torch.float32 or not pos.is_contiguous()):
            return None
        w = self._cache.get(self._build_weights)
        ldx = x_in.stride(0)
        key = (id(w), n, kk, k, str(dev), ldx)
        plan = self._one_call
        if plan is None or plan["key"] != key:
            plan = self._one_call_plan(w, n, kk, k, dev, ldx)
            if plan is None:
                return None
            plan["key"] = key
            self._one_call = plan
        fa, od = plan["fa"], plan["out_dim"]
        if out is None or tuple(out.shape) != (n, od) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            out = torch.empty((n, od), dtype=torch.float32, device=dev)
        last = fa.layers[fa.n_layers - 1]
        fa.pos, fa.edge_index, fa.use_hint = pos.data_ptr(), buf.data_ptr(), 1
        fa.layers[0].x = x_in.data_ptr()
        last.enc, last.out = x_in.data_ptr(), out.data_ptr()
        if kick is not None:
            if kick[0].shape != (n, od) or kick[0].dtype != torch.float32 or not kick[0].is_contiguous():
                return None
            last.kick_vel, last.kick_c = kick[0].data_ptr(), float(kick[1])
        else:
            last.kick_vel, last.kick_c = None, 0.0
        if advance is not None:
            # (vel_half, pos_out, posm, dt): the leapfrog bookkeeping in the last layer's epilogue (include/nbd.h,
            # nbd_gnn_layer_args.adv_*); `pos` is then the pre-advanced position array the epilogue advances again
            vh, pos_out, posm, dt = advance
            # (a one-layer model's only launch gathers neighbour rows from x = posm, which this epilogue overwrites: the
            # C-ABI refuses that combination, NBD_E_UNSUPPORTED; the caller keeps its separate kick-drift launch)
            ok = (kick is not None and od == 3 and self.gnn_dim == 64 and self.input_dim <= 64 and fa.n_layers >= 2
                  and all(t.shape == (n, 3) and t.dtype == torch.float32 and t.is_contiguous() and t.device == dev for t in (vh, pos_out))
                  and posm.dtype == torch.float32 and posm.is_contiguous() and posm.shape[1] == 4 and posm.shape[0] >= n)
            if not ok:
                return None
            last.adv_vel_half, last.adv_pos, last.adv_posm = vh.data_ptr(), pos.data_ptr(), posm.data_ptr()
            last.adv_pos_out, last.adv_dt = pos_out.data_ptr(), float(dt)
        else:
            last.adv_vel_half = last.adv_pos = last.adv_posm = last.adv_pos_out = None
        with _lib.on_device(dev):
            _lib.check(_lib.lib().nbd_gnn_forward_f32(ctypes.byref(fa), _lib.current_stream(dev)), "nbd_gnn_forward_f32")
        self._kick_done = kick is not None
        self._advance_done = advance is not None
        self.last_path = ("one_call+tables" if fa.workspace_bytes else "one_call") + ("+pre_advance" if advance is not None else "")
        graphops.mark(buf, "_nbd_grouped")
        return out

    def predict(self, pos, feat, neighbors=None):
        """gnn.py:205-215. The reference never forwards `self.neighbors` here, so the graph uses
        transform_to_graph's default k = 50; `neighbors=` is this build's optional override."""
        ensure_eval(self)
        with torch.no_grad():
            k = 50 if neighbors is None else neighbors
            if not pos.is_cuda:
                raise NbdError("GraphModel.predict: tensors must live on the GPU (no CPU path)")
            # transform_to_graph + forward without materialising x = [pos | feat] first: the model input is
            # [pos | mass] (input_dim == 4, gnn.py:131-132) or [pos | feat]
            # the previous call's graph (same n, k) is both the hint and the output buffer of this search: in a
            # rollout consecutive configurations are close, and the search result does not depend on the hint
            kk = max(min(k, pos.shape[0] - 1), 0)
            x_in = torch.cat((pos, feat[:, 3:]), dim=-1) if self.input_dim == 4 else torch.cat((pos, feat), dim=-1)
            x_in = x_in.to(torch.float32)
            pred = self._predict_one_call(x_in, pos.contiguous(), k)       # search + layers in one C-ABI call
            if pred is not None:
                return pred
            buf = self._knn_buf
            if buf is not None and (buf.shape != (2, pos.shape[0] * kk) or buf.device != pos.device):
                buf = None
            ei = graphops.knn_graph(pos, k=k, batch=None, loop=False, hint=buf, out=buf)
            self._knn_buf = ei
            # First call of a sequence: the graph just built is the buffer and hint the one-call pass was missing -- take
            # that pass now (one more, hinted, search), so that this call and every later one run the SAME arithmetic
            # (exponential tables or not): predict() on the same input returns the same bits, call after call.
            pred = self._predict_one_call(x_in, pos.contiguous(), k)
            if pred is None:
                pred = self._forward_inference(x_in, ei, max(min(k, pos.shape[0] - 1), 0))
        return pred

    def predict_batched(self, pos, feat, batch, neighbors=None):
        """predict() for SEVERAL independent systems at once (Trainer.test_from_dir advances all scenes of a file
        together): `batch` (sorted int64, one entry per body) names every body's system; neighbours are searched inside
        a system only (knn_graph(batch=...), k = 50 as predict(), fewer in systems of <= k bodies), one set of launches for
        all of them. Same layer kernels as the first predict() call of a sequence (CSR form, exact tanh)."""
        ensure_eval(self)
        with torch.no_grad():
            k = 50 if neighbors is None else neighbors
            if not pos.is_cuda:
                raise NbdError("GraphModel.predict_batched: tensors must live on the GPU (no CPU path)")
            x_in = torch.cat((pos, feat[:, 3:]), dim=-1) if self.input_dim == 4 else torch.cat((pos, feat), dim=-1)
            x_in = x_in.to(torch.float32)
            # the search always writes into a buffer of the layout's size (remembered on `batch`): no per-call read-back of
            # the edge count, safe inside a hipGraph capture; self is masked in the kernel (the rollout's form of the rule)
            _, e = graphops.knn_layout(batch, pos.shape[0], k, False, pos.device)
            buf = getattr(self, "_knn_buf_batched", None)
            if buf is None or buf.shape != (2, e) or buf.device != pos.device:
                buf = torch.empty((2, e), dtype=torch.int64, device=pos.device)
            ei = graphops.knn_graph(pos.contiguous(), k=k, batch=batch, loop=False, out=buf)
            self._knn_buf_batched = ei
            return self._forward_inference(x_in, ei, None)

    supports_pre_advance = True          # _predict_posm(advance=...): see Trainer._capture_step

    def _predict_posm(self, posm, pos, k=50, out=None, kick=None, advance=None):
        """predict() for callers that already hold the packed rows {x, y, z, mass} the kick-drift kernel writes
        (Trainer's captured rollout step): with input_dim == 4 that IS the model input [pos | mass]
        (gnn.py:131-132), so nothing is concatenated. Same graph, same kernels, same values as predict().
        kick = (vel, c): apply vel += c * prediction in the last layer's epilogue when the fused path runs with a
        single-Linear head; `self._kick_done` tells the caller whether it did (otherwise the caller kicks)."""
        ensure_eval(self)
        self._kick_done = False
        self._advance_done = False
        self._kick_hint = kick
        try:
            return self._predict_posm_impl(posm, pos, k, out, advance)
        finally:
            self._kick_hint = None

    def _predict_posm_impl(self, posm, pos, k, out, advance=None):
        """advance = (vel_half, pos_out, dt): also do the step's leapfrog bookkeeping in the last layer's epilogue
        (see _predict_one_call); `self._advance_done` tells the caller whether it happened -- it never does on the
        general path below, and the caller must then not have relied on it (Trainer checks before capturing)."""
        with torch.no_grad():
            n = pos.shape[0]
            kk = max(min(k, n - 1), 0)
            adv = None if advance is None else (advance[0], advance[1], posm, advance[2])
            pred = self._predict_one_call(posm[:n], pos, k, out=out, kick=self._kick_hint, advance=adv)
            if pred is not None:
                return pred
            if advance is not None:
                raise NbdUnsupported("GraphModel._predict_posm: the pre-advancing step needs the one-call fused path")
            buf = self._knn_buf
            if buf is not None and (buf.shape != (2, n * kk) or buf.device != pos.device):
                buf = None
            ei = graphops.knn_graph(pos, k=k, batch=None, loop=False, hint=buf, out=buf)
            self._knn_buf = ei
            return self._forward_inference(posm[:n], ei, kk, out=out)

    def predict_graph(self, data):
        self.eval()
        with torch.no_grad():
            return self.forward(data)

    def eval_graph_batch(self, data):
        """(rmse, mse, seconds) as gnn.py:193-203; the time brackets forward() with a device sync."""
        self.eval()
        with torch.no_grad():
            torch.cuda.synchronize()
            start = time.time()
            acc_pred = self.forward(data)
            torch.cuda.synchronize()
            end = time.time()
            mse_loss = torch.nn.functional.mse_loss(acc_pred, data.y, reduction="mean")
            loss = torch.sqrt(mse_loss)
        return loss.item(), mse_loss.item(), end - start

    # ------------------------------------------------------------------ training (gnn.py:150-191)
    def _graph_lists(self, data, n):
        ei = data.edge_index
        reg = getattr(data, "_regular_k", None)
        if reg is not None and n * reg == ei.shape[1]:
            return ag.EdgeLists(n, None, ei[0].contiguous(), reg, tgt=ei[1].contiguous())
        cached = getattr(data, "_edge_lists", None)
        if cached is None or cached.n != n:
            rowptr, src, tgt = graphops.csr_by_target(ei, n, return_tgt=True)
            cached = ag.EdgeLists(n, rowptr, src, -1, tgt=tgt)
            try:
                data._edge_lists = cached      # the graph of a batch does not change between epochs
            except AttributeError:
                pass
        return cached

    use_one_call_train = True    # forward + backward of the whole model through ONE C-ABI call each (csrc/train_model.hip)

    def _one_call_train(self, x_in, lists):
        """The training forward as one autograd node (ag.GnnModelFn), or None when the configuration is outside what
        nbd_gnn_train_*_f32 covers: aggr = "max", an active encoder dropout, an empty batch, more layers than the
        C-ABI's fixed arrays hold."""
        from nbd import _lib
        enc = self.node_encoder if isinstance(self.node_encoder, MLP) else None
        if (not self.use_one_call_train or self.aggr == "max" or x_in.shape[0] == 0 or lists.src.numel() == 0
                or x_in.requires_grad            # a gradient with respect to the input: the per-layer Functions provide it
                or (enc is not None and (enc.has_norm or (self.training and enc.dropout > 0) or len(enc.lins) > _lib.TRAIN_MAX_MLP))
                or len(self.gnns) > _lib.GNN_MAX_LAYERS):
            return None
        head = [self.output] if isinstance(self.output, Linear) else [m for m in self.output if isinstance(m, Linear)]
        if len(head) > _lib.TRAIN_MAX_MLP:
            return None
        params = []
        if enc is not None:
            for lin in enc.lins:
                params += [lin.weight, lin.bias]
        for g in self.gnns:
            params += [g.nn[0].weight, g.nn[0].bias, g.nn[2].weight, g.nn[2].bias]
        params += [self.layer_norm.weight, self.layer_norm.bias]
        for lin in head:
            params += [lin.weight, lin.bias]
        if any(p is None for p in params):
            return None
        spec = {"aggr": "mean" if self.aggr == "mean" else "sum", "n_layers": len(self.gnns), "h": self.gnn_dim,
                "enc_dims": enc.channels if enc is not None else None, "ln_eps": self.layer_norm.eps,
                "head_dims": [head[0].in_features] + [lin.out_features for lin in head]}
        return ag.GnnModelFn.apply(x_in, lists, spec, *params)

    def _forward_autograd(self, data):
        """gnn.py:130-148 with every layer a torch.autograd.Function over the HIP kernels."""
        x7 = data.x
        if not x7.is_cuda:
            raise NbdError("GraphModel.forward: data must live on the GPU (no CPU path)")
        n, h = x7.shape[0], self.gnn_dim
        x_in = torch.cat((x7[:, :3], x7[:, 6:]), dim=-1) if self.input_dim == 4 else x7
        x_in = x_in.to(torch.float32).contiguous()
        lists = self._graph_lists(data, n)
        one = self._one_call_train(x_in, lists)
        if one is not None:
            return one
        if isinstance(self.node_encoder, MLP):
            enc, last = x_in, len(self.node_encoder.lins) - 1
            for i, lin in enumerate(self.node_encoder.lins):
                enc = ag.linear(enc, lin.weight, lin.bias, act="tanh" if i < last else None)
                if i < last and self.training and self.node_encoder.dropout > 0:
                    enc = torch.nn.functional.dropout(enc, p=self.node_encoder.dropout, training=True)
        else:
            enc = x_in
        if lists.rowptr is None:
            val = float(lists.fixed_k) if self.aggr != "mean" else (1.0 if lists.fixed_k > 0 else 0.0)
            brs = torch.full((n,), val, dtype=torch.float32, device=x7.device)
        else:
            brs = nnops.degree_scale(lists.rowptr, n, 2 if self.aggr == "mean" else 1, x7.device)
        x = enc
        for g in self.gnns:
            w1, b1 = g.nn[0].weight, g.nn[0].bias
            f = w1.shape[1] // 2
            wpq = torch.cat([w1[:, :f] - w1[:, f:], w1[:, f:]], dim=0)            # rows P then Q
            bpq = torch.cat([b1, torch.zeros_like(b1)])
            pq = ag.linear(x, wpq, bpq)
            if self.aggr == "max":
                msg = ag.linear(ag.EdgeMessagesFn.apply(pq, lists, h), g.nn[2].weight, g.nn[2].bias)
                x = ag.SegmentMaxFn.apply(msg, lists.csr_rowptr(), n)
            else:
                s = ag.EdgeAggregateFn.apply(pq, lists, h, "mean" if self.aggr == "mean" else "sum")
                x = ag.linear(s, g.nn[2].weight, g.nn[2].bias, bias_rowscale=brs)
        z = torch.cat((enc, x), dim=-1)
        z = ag.LayerNormFn.apply(z, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        if isinstance(self.output, Linear):
            return ag.linear(z, self.output.weight, self.output.bias)
        lins = [m for m in self.output if isinstance(m, Linear)]
        for i, lin in enumerate(lins):
            z = ag.linear(z, lin.weight, lin.bias, act="tanh" if i < len(lins) - 1 else None)
        return z

    def compute_loss(self, data):
        """gnn.py:150-161: (RMSE of the scaled accelerations, plain MSE)."""
        acc_pred = self.forward(data)
        loss = torch.sqrt(torch.nn.functional.mse_loss(acc_pred * self.scale_factor, data.y * self.scale_factor,
                                                       reduction="mean"))
        mse_losses = torch.nn.functional.mse_loss(acc_pred, data.y, reduction="mean")
        return loss, mse_losses

    def train_batch(self, optimizer, pos, feat, acc):
        """gnn.py:163-185: a batch of equally sized particle sets -> one k = 50 graph batch -> one step."""
        self.train()
        optimizer.zero_grad()
        batch = torch.cat([torch.full((pos[i].size(0),), i, dtype=torch.long, device=pos.device)
                           for i in range(len(pos))])
        pos = pos.reshape(-1, 3)
        feat = feat.reshape(-1, feat.size(-1))
        acc = acc.reshape(-1, 3)
        data = transform_to_graph(pos, feat, acc, batch=batch, device=self.device)
        loss, mse_loss = self.compute_loss(data)
        loss.backward()
        optimizer.step()
        return loss.item(), mse_loss.item()

    def train_graph_batch(self, optimizer, data):
        """gnn.py:187-191."""
        self.train()
        optimizer.zero_grad()
        loss, mse_loss = self.compute_loss(data)
        loss.backward()
        optimizer.step()
        return loss.item(), mse_loss.item()
