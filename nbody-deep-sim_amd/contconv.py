"""MI355X drop-in for the reference's contconv.py: `ContinuousConv` and `ContinuousConvModel` with
the same constructor arguments, attributes (`neighbors` = 0, `radius`, ...), parameter names
(`contconv.{l}.filters`, `node_encoder.lins/norms.*`, `layer_norm.*`, `output.*`) and inference
methods `forward(data)`, `predict(pos, feat)`, `eval_graph_batch(data)`.

HIP pipeline per forward (csrc/graph.hip, csrc/nn.hip):
  radius_graph(loop=self_loops, max_num_neighbors=32) -> nbd_radius_search_f32 + transpose (CSR by
      edge_index[0], the index the reference aggregates over, contconv.py:82,95)
  node encoder (PyG MLP, BatchNorm folded for eval)   -> nbd_linear_f32 (+tanh)
  ContinuousConv layer  -> nbd_contconv_bin_f32: A[n][cell][i] = sum_e window_e t_cell(e) feat[c_e][i]
                           nbd_linear_f32: tanh( (1/deg_n) * A . filters.reshape(D^3 I, O) )   [fp32 MFMA]
  LayerNorm + decoder   -> nbd_layernorm_f32 + nbd_linear_f32
The trilinear blend of contconv.py:53-78 is linear in the filter, so it is applied to the features
(8 weighted copies per edge) instead of to the (I x O) filters; the contraction with the filters then
becomes one dense GEMM and the (E, I, O) tensor (34 GB at N = 16 384) is never formed.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn as nn

from gnn import MLP, _WeightCache, ensure_eval, head_chain, run_chain, transform_to_graph  # noqa: F401  (same import as contconv.py:6)
from nbd import autograd as ag
from nbd import graphops, nnops
from nbd._lib import NbdError


# Node-chunked bin+contract was measured and rejected: chunks small enough for the Infinity Cache leave
# the GEMM too few row blocks (N=16 384: 3.0 ms un-chunked, 3.7 ms at 512 MB chunks, 7-9 ms at <= 256 MB).
A_CHUNK_BYTES = int(__import__("os").environ.get("NBD_CONTCONV_CHUNK_MB", str(1 << 20))) << 20


class ContinuousConv(nn.Module):
    def __init__(self, in_channels, out_channels, filter_resolution=4, radius=0.5, agg="mean"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.radius, self.agg = radius, agg
        self.filter_resolution = filter_resolution
        if agg not in ("mean", "sum", "add", "max", "min", "mul"):
            raise NotImplementedError(f"agg={agg!r}: torch_scatter's reductions are sum/add, mean, max, min, mul")
        self.filters = nn.Parameter(torch.randn(filter_resolution, filter_resolution, filter_resolution,
                                                in_channels, out_channels))            # contconv.py:20-28

    def ball_to_cube(self, r):
        """contconv.py:30-33: r / (|r| + 1e-8) * tanh |r| (inference helper: no autograd; the layer's forward does the same
        mapping inside its kernels)."""
        return nnops.ball_to_cube(r)

    def trilinear_interpolate(self, coords):
        """contconv.py:53-78: the (N, in, out) filters blended at coords (N, 3) in grid units [0, D - 1] (inference helper;
        the layer's forward applies the same blend to the features instead and never forms this tensor)."""
        return nnops.trilinear_interpolate(self.filters, coords)

    def cells(self):
        """(reachable cell indices int64 [K], cell -> compact index int32 [D^3], K) on the filters' device:
        only grid points within tanh(radius) (D-1)/2 (+ their adjacent unit cubes) of the grid centre can be
        touched by a sample (nnops.reachable_cells); the others are dropped from the binned matrix and from the
        filter matrix alike -- their products are exact zeros in the reference."""
        key = (self.filter_resolution, float(self.radius), str(self.filters.device))
        if getattr(self, "_cells_key", None) != key:
            idx, cmap = nnops.reachable_cells(self.filter_resolution, self.radius)
            self._cells_val = (idx.to(self.filters.device), cmap.to(self.filters.device), int(idx.numel()))
            self._cells_key = key
        return self._cells_val

    def weight_t(self):
        """filters (D,D,D,I,O) -> (O, K*I) over the K reachable cells: the `w` operand of nbd_linear_f32,
        k = compact(cell)*I + i with cell = (z*D+y)*D+x."""
        d, i, o = self.filter_resolution, self.in_channels, self.out_channels
        idx, _, k = self.cells()
        return self.filters.detach().reshape(d * d * d, i, o).index_select(0, idx).reshape(k * i, o).t().contiguous()

    def weight_fused(self):
        """filters in the MFMA fragment order of the fused kernel (nnops.contconv_shuffle_filters)."""
        idx, _, _ = self.cells()
        return nnops.contconv_shuffle_filters(self.filters, idx)

    def fused_ok(self) -> bool:
        # the kernel wants in_channels % 4 == 0; other widths run it on zero-padded feature columns (forward(): one pad
        # kernel; the shuffled filters are zero-padded to % 16 rows anyway, so the product is the same sum)
        ok = self.use_fused and nnops.contconv_fused_supported((self.in_channels + 3) // 4 * 4, self.out_channels, self.cells()[2])
        if self.use_fused and not ok and not getattr(self, "_warned_fallback", False):
            import warnings
            self._warned_fallback = True          # once per layer: the other path is ~2.5x slower and forms A in HBM
            warnings.warn(f"ContinuousConv({self.in_channels} -> {self.out_channels}, D = {self.filter_resolution}, "
                          f"{self.cells()[2]} reachable cells): outside the fused block-sparse kernel's shapes "
                          f"(in_channels <= 128; <= 160 cells) -- using the binned-matrix + GEMM path, which "
                          f"materialises A (nodes x cells x in_channels fp32) in HBM")
        return ok

    use_fused = True         # block-sparse fused kernels (csrc/contconv_fused.hip) where the shape allows
    last_path = None         # which path the last forward took: "fused" (pair lists + stream kernel), "binned" (A in HBM + GEMM),
                             # "fused_train" / "binned_train" under autograd, "extreme" (max / min / mul)

    def trains_fused(self) -> bool:
        """Whether the training step of this layer runs on the pair lists (forward, filter gradient, feature gradient --
        the last is the forward kernel with in / out channels swapped, hence the second shape check)."""
        k = self.cells()[2]
        return (self.fused_ok() and self.in_channels <= 128 and self.out_channels <= 128 and
                nnops.contconv_fused_supported(self.in_channels, self.out_channels, k) and
                nnops.contconv_fused_supported(self.out_channels, self.in_channels, k))

    def forward(self, positions, features, edge_index=None, lists=None, act=None, out=None, wt=None, pairs=None,
                scale=None, graph=None):
        """contconv.py:80-98. Give either the sync-free `lists` (graphops.radius_lists) or a PyG-style
        edge_index [2,E] (row 0 = aggregation target, row 1 = feature source). `pairs`: the pair lists of
        this graph and filter resolution when the caller already has them (layers of one model that share D);
        `scale`: the 1/in-degree row scale of mean aggregation when the caller already has it; `graph`: the
        ag.ConvGraph of this forward pass when the caller built one for all its layers (training)."""
        n = positions.shape[0]
        if lists is not None:
            rowptr, centres = lists.rowptr, lists.centres
        else:
            row, col = edge_index[0], edge_index[1]
            order = torch.sort(row, stable=True).indices
            centres = col[order].to(torch.int32).contiguous()
            rowptr = torch.zeros(n + 1, dtype=torch.int32, device=positions.device)
            rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0).to(torch.int32)
            if centres.numel() == 0:
                centres = torch.zeros(1, dtype=torch.int32, device=positions.device)
        r2 = float(np.float32(self.radius ** 2))                       # contconv.py:86: python double -> fp32
        if self.agg in ("max", "min", "mul"):
            self.last_path = "extreme"
            return self._forward_extreme(positions, features, rowptr, centres, act, out)
        if self.agg != "mean":
            scale = None
        training_path = torch.is_grad_enabled() and (self.filters.requires_grad or features.requires_grad)
        if self.agg == "mean" and scale is None and not training_path:
            scale = nnops.degree_scale(rowptr, n, 0, positions.device)
        if torch.is_grad_enabled() and (self.filters.requires_grad or features.requires_grad):
            if out is not None:
                raise NbdError("ContinuousConv.forward: out= is an inference-only option")
            if self.trains_fused() and n > 0:
                # forward AND backward on the pair lists (ag.ContConvFusedFn): A is formed in neither
                if graph is None:
                    graph = (ag.ConvGraph.from_lists(positions.contiguous(), r2, lists) if lists is not None else
                             ag.ConvGraph.from_edge_index(positions.contiguous(), r2, rowptr, centres, edge_index))
                if self.agg == "mean" and scale is None:
                    _, cmap, n_cells = self.cells()
                    pb, cap_e = graph.pairs(self.filter_resolution, cmap, n_cells)
                    scale = nnops.contconv_pairs_inv_degree(pb, n, cap_e, n_cells)
                self.last_path = "fused_train"
                return ag.ContConvFusedFn.apply(features, self.filters, graph, self.filter_resolution, scale, act,
                                                self.cells())
            if self.agg == "mean" and scale is None:
                scale = nnops.degree_scale(rowptr, n, 0, positions.device)
            if lists is not None:            # the radius search's per-centre lists ARE the by-source grouping
                bwd = dict(tgt_s=lists.nbr, deg=lists.deg, cap=lists.nbr.shape[1])
            else:
                rp, tg = graphops.csr_by_key(edge_index[1], edge_index[0], n)
                bwd = dict(rowptr_s=rp, tgt_s=tg if tg.numel() else torch.zeros(1, dtype=torch.int32, device=tg.device))
            self.last_path = "binned_train"
            return ag.ContConvFn.apply(features, self.filters, positions.contiguous(), (rowptr, centres), bwd,
                                       self.filter_resolution, r2, scale, act, self.cells())
        if self.fused_ok() and n > 0:
            # block-sparse path: pair lists -> fused gather + MFMA + per-node accumulation; A stays on chip
            _, cmap, n_cells = self.cells()
            feats = features if (features.stride(-1) == 1 and features.stride(0) % 2 == 0 and
                                 features.data_ptr() % 8 == 0) else features.contiguous()
            if self.in_channels % 4:
                feats = torch.nn.functional.pad(feats, (0, 4 - self.in_channels % 4))
            if pairs is None:
                pairs = nnops.contconv_pairs(positions.contiguous(), rowptr, centres, centres.numel(),
                                             self.filter_resolution, r2, cmap, n_cells)
            wf = wt if (wt is not None and wt.dim() == 1) else self.weight_fused()
            self.last_path = "fused"
            return nnops.contconv_fused(feats, rowptr, pairs[0], pairs[1], wf, n_cells, self.out_channels,
                                        rowscale=scale, act=act, out=out)
        self.last_path = "binned"
        wt = self.weight_t() if (wt is None or wt.dim() == 1) else wt
        if out is None:
            out = torch.empty((n, self.out_channels), dtype=torch.float32, device=positions.device)
        # bin + contract in node chunks: the chunk's A block (<= A_CHUNK_BYTES) is produced and consumed
        # while it sits in the 256 MiB Infinity Cache, and the buffer is reused chunk after chunk
        _, cmap, n_cells = self.cells()
        kc = n_cells * self.in_channels
        rows = max(128, (A_CHUNK_BYTES // (4 * kc)) // 128 * 128)
        pos_c = positions.contiguous()
        a_buf = torch.empty((min(rows, n), kc), dtype=torch.float32, device=positions.device)
        for lo in range(0, n, rows):
            cnt = min(rows, n - lo)
            nnops.contconv_bin(pos_c, features, rowptr, centres, self.filter_resolution, r2, out=a_buf,
                               node_begin=lo, count=cnt, cell_map=cmap, cells_out=n_cells)
            nnops.linear(a_buf[:cnt], wt, None, act=act, out=out[lo:lo + cnt],
                         rowscale=None if scale is None else scale[lo:lo + cnt])
        return out


    def _forward_extreme(self, positions, features, rowptr, centres, act, out):
        """agg = "max" / "min" / "mul" (scatter's other reductions, contconv.py:95-97; "mul" = the product of a row's
        messages, 1 for a row without any, as torch_scatter's scatter_mul): the feature-side binning needs a
        LINEAR aggregation, so the per-edge messages are materialised -- every edge becomes a row of its own in
        a virtual graph (row N + e at the position of the edge's aggregation target, one edge, sum aggregation)
        through the same kernels -- and reduced per target by nbd_segment_reduce_f32. Under autograd the same construction
        runs through the differentiable layer and ag.SegmentMaxFn (below)."""
        n, dev = positions.shape[0], positions.device
        if torch.is_grad_enabled() and (self.filters.requires_grad or features.requires_grad):
            # Training (contconv.py:95-97 hands `agg` straight to scatter, so the reference trains through max / min too):
            # the same virtual one-edge-per-row graph, through the differentiable sum-aggregation layer (ag.ContConvFn:
            # gradients reach the filters and, through the concatenation, the features), then the segment maximum whose
            # backward sends a row's gradient to the first message attaining it (ag.SegmentMaxFn, as EdgeConv's max).
            e = int(rowptr[-1])
            if e == 0:
                red = torch.full((n, self.out_channels), 1.0 if self.agg == "mul" else 0.0, dtype=torch.float32,
                                 device=dev) + 0.0 * self.filters.sum()
            else:
                tgt = torch.repeat_interleave(torch.arange(n, device=dev), (rowptr[1:] - rowptr[:-1]).to(torch.int64), output_size=e)
                pos_v = torch.cat([positions, positions[tgt]]).contiguous()
                feat_v = torch.cat([features, torch.zeros((e, features.shape[1]), dtype=features.dtype, device=dev)])
                ei_v = torch.stack([n + torch.arange(e, device=dev), centres[:e].to(torch.int64)])
                agg, self.agg = self.agg, "sum"
                try:
                    msgs = self.forward(pos_v, feat_v, edge_index=ei_v)[n:]
                finally:
                    self.agg = agg
                if agg == "mul":
                    red = ag.SegmentMulFn.apply(msgs.contiguous(), rowptr, n)
                else:
                    red = ag.SegmentMaxFn.apply((-msgs if agg == "min" else msgs).contiguous(), rowptr, n)
                    red = -red if agg == "min" else red
            if out is not None:
                raise NbdError("ContinuousConv.forward: out= is an inference-only option")
            return torch.tanh(red) if act == "tanh" else red
        e = int(centres.numel()) if rowptr is None else int(rowptr[-1])       # one read-back: this path is not the hot one
        tgt = torch.repeat_interleave(torch.arange(n, device=dev), (rowptr[1:] - rowptr[:-1]).to(torch.int64),
                                      output_size=e)
        pos_v = torch.cat([positions, positions[tgt]]).contiguous()
        feat_v = torch.cat([features, torch.zeros((e, features.shape[1]), dtype=features.dtype, device=dev)])
        rowptr_v = torch.cat([torch.zeros(n + 1, dtype=torch.int32, device=dev),
                              torch.arange(1, e + 1, dtype=torch.int32, device=dev)])
        cen_v = centres[:e].contiguous() if e else torch.zeros(1, dtype=torch.int32, device=dev)
        agg, self.agg = self.agg, "sum"
        try:
            msgs = self.forward(pos_v, feat_v, lists=graphops.RadiusLists(n=n + e, cap=0, nbr=None, deg=None, last=None,
                                                                            rowptr=rowptr_v, centres=cen_v))[n:]
        finally:
            self.agg = agg
        if agg == "min":
            msgs = -msgs
        red = nnops.segment_reduce(msgs.contiguous(), rowptr, n, "mul" if agg == "mul" else "max")
        if agg == "min":
            red = -red
        if act == "tanh":
            red = torch.tanh(red)
        if out is not None:
            out.copy_(red)
            return out
        return red


class ContinuousConvModel(nn.Module):
    def __init__(self, in_channels=4, out_channels=3, filter_resolution=[4], radius=0.5, agg="mean",
                 self_loops=True, continuous_conv_layers=1, continuous_conv_dim=64, continuous_conv_dropout=0.0,
                 encoder_hiddens=None, encoder_dropout=0.0, decoder_hiddens=None, decoder_dropout=0.0,
                 device="cuda", scale_factor=1):
        super().__init__()
        self.device, self.scale_factor = device, scale_factor
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.encoder_hiddens = encoder_hiddens
        self.encoder_dropout = encoder_dropout
        self.decoder_hiddens = decoder_hiddens
        self.decoder_dropout = decoder_dropout
        self.continuous_conv_layers = continuous_conv_layers
        self.continuous_conv_dim = continuous_conv_dim
        self.continuous_conv_dropout = continuous_conv_dropout
        self.neighbors = 0                                                           # contconv.py:131
        self.radius = radius
        self.self_loops = self_loops
        self.max_num_neighbors = 32            # PyG radius_graph default; the reference does not override it
        if not isinstance(filter_resolution, (list, tuple)):
            # upstream's scalar branch appends to an undefined self.gnns (contconv.py:177,187)
            raise AttributeError("'ContinuousConvModel' object has no attribute 'gnns' (scalar filter_resolution "
                                 "is broken in the reference; pass a list)")
        if encoder_hiddens:                                                          # contconv.py:135-143
            self.node_encoder = MLP([in_channels] + list(encoder_hiddens) + [continuous_conv_dim],
                                    dropout=encoder_dropout)                         # PyG default: batch_norm
        else:
            self.node_encoder = torch.nn.Identity()
        self.contconv = nn.ModuleList()                                              # contconv.py:150-173
        for i in range(continuous_conv_layers):
            cin = in_channels if (i == 0 and encoder_hiddens is None) else continuous_conv_dim
            self.contconv.append(ContinuousConv(cin, continuous_conv_dim, filter_resolution[i], self.radius, agg))
        out_dim = continuous_conv_dim + in_channels if encoder_hiddens is None else continuous_conv_dim * 2
        self.layer_norm = nn.LayerNorm(out_dim)
        if decoder_hiddens:                                                          # contconv.py:206-216
            layers, dims = [], [out_dim] + list(decoder_hiddens) + [out_channels]
            for i in range(len(dims) - 1):
                layers.append(nn.Linear(dims[i], dims[i + 1]))
                if i < len(dims) - 2:
                    layers.append(nn.Tanh())
            self.output = nn.Sequential(*layers)
        else:
            self.output = nn.Linear(out_dim, out_channels)
        self._cache = _WeightCache(self)
        self._radius_cache = None
        self.use_radius_cache = True
        self._side_stream = None
        self.overlap_encoder = True
        self.use_fused_head = True     # LayerNorm + decoder in one launch (nbd_ln_mlp_head_f32) where the shapes allow
        self.to(device)

    def _build_weights(self):
        enc = self.node_encoder.folded() if isinstance(self.node_encoder, MLP) else None
        head = head_chain(self.output)
        return {"enc": enc, "wt": [layer.weight_fused() if layer.fused_ok() else layer.weight_t() for layer in self.contconv],
                "head": head, "head_plan": nnops.ln_mlp_head_plan(self.layer_norm.normalized_shape[0], head, self.layer_norm.weight.detach(),
                                                            self.layer_norm.bias.detach())}

    def forward(self, data):                                                         # contconv.py:218-234
        needs_train_path = self.training and (isinstance(self.node_encoder, MLP) and self.node_encoder.has_norm
                                              or self.continuous_conv_dropout > 0 or self.encoder_dropout > 0)
        if needs_train_path or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return self._forward_autograd(data)
        x7 = data.x
        if not x7.is_cuda:
            raise NbdError("ContinuousConvModel.forward: data must live on the GPU (no CPU path)")
        self._last_path = None
        w = self._cache.get(self._build_weights)
        n = x7.shape[0]
        pre = getattr(data, "_x_pos", None)        # predict() hands over [pos | mass] and pos as it built them
        if pre is not None:
            x, pos = pre
        else:
            x = torch.cat((x7[:, :3], x7[:, 6:]), dim=-1) if self.in_channels == 4 else x7
            x = x.to(torch.float32).contiguous()
            pos = x[:, :3].contiguous()
        c = self.continuous_conv_dim
        enc_dim = self.in_channels if w["enc"] is None else c
        cat_buf = torch.empty((n, enc_dim + c), dtype=torch.float32, device=x7.device)
        enc_view, conv_view = cat_buf[:, :enc_dim], cat_buf[:, enc_dim:]
        # the node encoder needs only x, the graph only the positions: the encoder's three small launches run on a
        # second stream beside the neighbour search and the pair lists (fork / join; captured as such in a hipGraph)
        cur = torch.cuda.current_stream(x7.device)
        side = None
        if self.overlap_encoder and w["enc"] is not None and n > 0:
            if self._side_stream is None or self._side_stream.device != x7.device:
                self._side_stream = torch.cuda.Stream(device=x7.device)
            side = self._side_stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                run_chain(x, w["enc"], out_last=enc_view)
        elif w["enc"] is None:
            enc_view.copy_(x)
        else:
            run_chain(x, w["enc"], out_last=enc_view)
        lists = graphops.radius_lists(pos, self.radius, getattr(data, "batch", None), loop=self.self_loops,
                                      max_num_neighbors=self.max_num_neighbors,
                                      cache=getattr(data, "_radius_cache", None))      # predict(): a rollout's search
        h = enc_view
        # pair lists: one launch for all the filter resolutions of the model (layers share the graph)
        pair_cache = {}
        if n > 0:
            keys, jobs = [], []
            for layer in self.contconv:
                key = (layer.filter_resolution, float(layer.radius))
                if layer.fused_ok() and key not in keys:
                    _, cmap, n_cells = layer.cells()
                    keys.append(key); jobs.append((layer.filter_resolution, cmap, n_cells))
            for lo in range(0, len(jobs), 4):
                got = nnops.contconv_pairs_batch(pos, lists.rowptr, lists.centres, lists.centres.numel(),
                                                 float(np.float32(self.radius ** 2)), jobs[lo:lo + 4])
                pair_cache.update(zip(keys[lo:lo + 4], got))
        # 1 / in-degree, once for all layers: a by-product of the pair kernel when there are pair lists at all
        inv_deg = None
        if n > 0 and pair_cache:
            pb, cap_e = pair_cache[keys[0]]
            inv_deg = nnops.contconv_pairs_inv_degree(pb, n, cap_e, jobs[0][2])
        elif n > 0:
            inv_deg = nnops.degree_scale(lists.rowptr, n, 0, x7.device)
        if side is not None:
            cur.wait_stream(side)                                                          # the layers read the encoder's output
        for li, layer in enumerate(self.contconv):
            last = li == len(self.contconv) - 1
            pairs = pair_cache.get((layer.filter_resolution, float(layer.radius))) if layer.fused_ok() else None
            h = layer(pos, h, lists=lists, act="tanh", out=conv_view if last else None, wt=w["wt"][li], pairs=pairs,
                      scale=inv_deg)
        out = getattr(data, "_out", None)          # _predict_posm(): the caller's acceleration buffer, written directly
        if out is not None and (tuple(out.shape) != (n, self.out_channels) or out.dtype != torch.float32
                                or not out.is_contiguous() or out.device != x7.device):
            out = None
        if w["head_plan"] is not None and self.use_fused_head and n > 0:
            # LayerNorm + decoder (+ the caller's half-kick) in one launch: they sit at the end of the step's dependency chain
            kick = getattr(data, "_kick", None)
            pred = nnops.ln_mlp_head(cat_buf, self.layer_norm.weight.detach(), self.layer_norm.bias.detach(),
                                     self.layer_norm.eps, w["head_plan"], out=out,
                                     kick_vel=kick[0] if kick is not None else None, kick_c=kick[1] if kick is not None else 0.0)
            self._kick_done = kick is not None
            return pred
        ln = nnops.layernorm(cat_buf, self.layer_norm.weight.detach(), self.layer_norm.bias.detach(),
                             self.layer_norm.eps)
        return run_chain(ln, w["head"], out_last=out)

    @property
    def last_path(self):
        """The path every layer's last forward took (ContinuousConv.last_path), e.g. ("fused", "fused"); "one_call_train" when the
        whole training pass ran behind nbd_cc_train_*_f32."""
        return getattr(self, "_last_path", None) or tuple(l.last_path for l in self.contconv)

    @property
    def input_dim(self):
        """The model input's width ([pos | mass] = 4 as published, contconv.py:219-220): what Trainer's captured step asks
        to decide whether the packed {x, y, z, m} rows of its kick-drift kernel ARE the model input."""
        return self.in_channels

    def _predict_posm(self, posm, pos, k=50, out=None, kick=None):
        """predict() for a caller that already holds the packed rows {x, y, z, mass} the kick-drift kernel writes
        (Trainer's captured rollout step): with in_channels == 4 that IS the model input, so neither [vel | mass] nor
        [pos | mass] is concatenated and the prediction lands in the caller's buffer (three launches fewer per step).
        kick = (vel, c): vel += c * prediction in the decoder kernel's epilogue when the fused head runs; `_kick_done`
        tells the caller whether it did (otherwise the caller kicks)."""
        from nbd.data import Data
        ensure_eval(self)
        self._kick_done = False
        with torch.no_grad():
            if getattr(self, "_radius_cache", None) is None:
                self._radius_cache = graphops.RadiusCache()
            n = pos.shape[0]
            data = Data(x=pos, batch=None)
            data._x_pos = (posm[:n], pos)
            data._out = out
            data._kick = kick
            data._radius_cache = self._radius_cache if self.use_radius_cache else None
            return self.forward(data)

    def predict(self, pos, feat):
        """contconv.py:261-271. (The reference also builds a k=50 kNN graph here that forward() then
        ignores; that dead work is not reproduced.)"""
        from nbd.data import Data
        ensure_eval(self)
        with torch.no_grad():
            # predict() is the rollout entry point (Trainer.step): consecutive calls see almost the same configuration,
            # so the radius search re-tests cached candidate lists and runs its O(n^2) scan only when a body has moved
            # far enough to matter (graphops.RadiusCache; the result is exact either way)
            if getattr(self, "_radius_cache", None) is None:
                self._radius_cache = graphops.RadiusCache()
            if (self.in_channels == 4 and pos.dtype == torch.float32 and feat.dtype == torch.float32 and pos.is_cuda
                    and pos.dim() == 2 and pos.shape[1] == 3 and feat.dim() == 2 and feat.shape[1] >= 4
                    and not self.training):
                # the model input is [pos | mass] (contconv.py:219-220): built directly -- the reference's route
                # (cat to 7 columns, slice, cat again, two contiguous copies) is four more launches per step
                data = Data(x=pos, batch=None)                      # .x is only consulted for its device and row count
                data._x_pos = (torch.cat((pos, feat[:, 3:]), dim=-1), pos.contiguous())
            else:
                data = Data(x=torch.cat((pos, feat), dim=-1), batch=None)
            data._radius_cache = self._radius_cache if self.use_radius_cache else None
            return self.forward(data)

    def predict_batched(self, pos, feat, batch):
        """predict() for several independent systems at once (Trainer.test_from_dir): `batch` (sorted int64) names every
        body's system; the radius graph stays inside a system (radius_graph(batch=...)). One set of launches for all."""
        from nbd.data import Data
        ensure_eval(self)
        with torch.no_grad():
            if (self.in_channels == 4 and pos.dtype == torch.float32 and feat.dtype == torch.float32 and pos.is_cuda
                    and pos.dim() == 2 and pos.shape[1] == 3 and feat.dim() == 2 and feat.shape[1] >= 4):
                data = Data(x=pos, batch=batch)
                data._x_pos = (torch.cat((pos, feat[:, 3:]), dim=-1), pos.contiguous())
            else:
                data = Data(x=torch.cat((pos, feat), dim=-1), batch=batch)
            return self.forward(data)

    def eval_graph_batch(self, data):
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

    # ------------------------------------------------------------------ training (contconv.py:236-247)
    def _encoder_autograd(self, x):
        """PyG MLP forward (Linear -> BatchNorm -> tanh -> dropout per hidden layer, plain last Linear) in the
        module's CURRENT mode: batch statistics when self.training, running statistics otherwise -- the
        reference's train_graph_batch never calls train(), so a model that went through eval() keeps
        training with frozen BatchNorm; the same happens here."""
        enc, last = self.node_encoder, len(self.node_encoder.lins) - 1
        for i, lin in enumerate(enc.lins):
            if i == last:
                return ag.linear(x, lin.weight, lin.bias)
            if enc.has_norm:
                bn = enc.norms[i].module
                if self.training:
                    x = ag.batchnorm_act(ag.linear(x, lin.weight, lin.bias), bn, "tanh")
                else:           # eval-mode BatchNorm is affine: fold it into the Linear (tiny torch ops, differentiable)
                    s_ = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    x = ag.linear(x, lin.weight * s_.unsqueeze(1), (lin.bias - bn.running_mean) * s_ + bn.bias, act="tanh")
            else:
                x = ag.linear(x, lin.weight, lin.bias, act="tanh")
            if self.training and enc.dropout > 0:
                x = torch.nn.functional.dropout(x, p=enc.dropout, training=True)
        return x

    use_one_call_train = True    # forward + backward of the whole model through ONE C-ABI call each (csrc/train_model.hip)

    def _one_call_train(self, x, pos, lists):
        """The training forward as one autograd node (ag.ContConvModelFn), or None when the configuration is outside what
        nbd_cc_train_*_f32 covers: a layer off the fused kernels' shapes, max / min aggregation, active dropout, an
        encoder BatchNorm in (sticky) eval mode, fewer than two nodes."""
        from nbd import _lib
        n = x.shape[0]
        enc = self.node_encoder if isinstance(self.node_encoder, MLP) else None
        if (not self.use_one_call_train or n < 2 or x.requires_grad or len(self.contconv) > _lib.GNN_MAX_LAYERS
                or (self.training and (self.continuous_conv_dropout > 0 or self.encoder_dropout > 0))
                or any(not l.trains_fused() or l.agg not in ("mean", "sum", "add") for l in self.contconv)
                or len({("mean" if l.agg == "mean" else "sum") for l in self.contconv}) != 1
                or (enc is not None and (len(enc.lins) > _lib.TRAIN_MAX_MLP or (enc.has_norm and not self.training)))):
            return None
        head = [self.output] if isinstance(self.output, nn.Linear) else [m for m in self.output if isinstance(m, nn.Linear)]
        if len(head) > _lib.TRAIN_MAX_MLP:
            return None
        params, bns = [], []
        if enc is not None:
            for lin in enc.lins:
                params += [lin.weight, lin.bias]
            if enc.has_norm:
                bns = [h.module for h in enc.norms]
                if any(b.weight is None or b.bias is None for b in bns):
                    return None
                for b in bns:
                    params += [b.weight, b.bias]
        layers = []
        for layer in self.contconv:
            idx, cmap, nc = layer.cells()
            layers.append((layer.filter_resolution, idx, cmap, nc))
            params.append(layer.filters)
        params += [self.layer_norm.weight, self.layer_norm.bias]
        for lin in head:
            params += [lin.weight, lin.bias]
        if any(p is None for p in params):
            return None
        graph = ag.ConvGraph.from_lists(pos, float(np.float32(self.radius ** 2)), lists)
        wants = []
        for li, (d, idx, cmap, nc) in enumerate(layers):
            wants.append((d, cmap, nc, False))
            if li > 0 or enc is not None:
                wants.append((d, cmap, nc, True))
        graph.prebuild(wants)
        spec = {"enc_dims": enc.channels if enc is not None else None, "bns": bns, "layers": layers,
                "cdim": self.continuous_conv_dim, "mean": self.contconv[0].agg == "mean", "ln_eps": self.layer_norm.eps,
                "head_dims": [head[0].in_features] + [lin.out_features for lin in head]}
        return ag.ContConvModelFn.apply(x, graph, spec, *params)

    def _forward_autograd(self, data):
        x7 = data.x
        if not x7.is_cuda:
            raise NbdError("ContinuousConvModel.forward: data must live on the GPU (no CPU path)")
        x = torch.cat((x7[:, :3], x7[:, 6:]), dim=-1) if self.in_channels == 4 else x7
        x = x.to(torch.float32).contiguous()
        pos = x[:, :3].contiguous()
        lists = graphops.radius_lists(pos, self.radius, getattr(data, "batch", None), loop=self.self_loops,
                                      max_num_neighbors=self.max_num_neighbors)
        one = self._one_call_train(x, pos, lists)
        self._last_path = "one_call_train" if one is not None else None
        if one is not None:
            return one
        enc = self._encoder_autograd(x) if isinstance(self.node_encoder, MLP) else x
        h = enc
        # one graph object for all layers: forward lists of every resolution and the adjoint lists of the layers whose
        # input carries a gradient, four jobs per launch
        graph = None
        if x7.shape[0] > 0 and any(l.trains_fused() and l.agg not in ("max", "min", "mul") for l in self.contconv):
            graph = ag.ConvGraph.from_lists(pos, float(np.float32(self.radius ** 2)), lists)
            wants, needs_grad = [], enc.requires_grad
            for layer in self.contconv:
                if layer.trains_fused() and layer.agg not in ("max", "min", "mul"):
                    _, cmap, n_cells = layer.cells()
                    wants.append((layer.filter_resolution, cmap, n_cells, False))
                    if needs_grad:
                        wants.append((layer.filter_resolution, cmap, n_cells, True))
                needs_grad = True
            graph.prebuild(wants)
        for layer in self.contconv:
            h = layer(pos, h, lists=lists, act="tanh", graph=graph if layer.agg not in ("max", "min", "mul") else None)
            if self.training and self.continuous_conv_dropout > 0:
                h = torch.nn.functional.dropout(h, p=self.continuous_conv_dropout, training=True)
        z = ag.LayerNormFn.apply(torch.cat((enc, h), dim=-1), self.layer_norm.weight, self.layer_norm.bias,
                                 self.layer_norm.eps)
        if isinstance(self.output, nn.Linear):
            return ag.linear(z, self.output.weight, self.output.bias)
        lins = [m for m in self.output if isinstance(m, nn.Linear)]
        for i, lin in enumerate(lins):
            z = ag.linear(z, lin.weight, lin.bias, act="tanh" if i < len(lins) - 1 else None)
        return z

    def compute_loss(self, data):
        """contconv.py:236-240."""
        acc_pred = self.forward(data)
        return (torch.sqrt(torch.nn.functional.mse_loss(acc_pred * self.scale_factor, data.y * self.scale_factor)),
                torch.nn.functional.mse_loss(acc_pred, data.y))

    def train_graph_batch(self, optimizer, data):
        """contconv.py:242-247 (no self.train() here, as upstream)."""
        optimizer.zero_grad()
        loss, mse_loss = self.compute_loss(data)
        loss.backward()
        optimizer.step()
        return loss.item(), mse_loss.item()
