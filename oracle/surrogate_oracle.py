"""CPU oracle for the learned-surrogate forward passes -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY PARTLY PINNED, the rest UNPINNED: gnn.py / contconv.py / trainer.py import torch_geometric
(2.6.1), torch_cluster (1.6.3) and torch_scatter (2.1.2) (requirements.txt:3-5), none of which exist
in this image, and the reference holds no tests or golden vectors for this path.
  pinned   -- the arithmetic that lives in the reference's OWN files and needs none of those packages:
              ContinuousConv.ball_to_cube / trilinear_interpolate (contconv.py:30-33,53-78) and
              Trainer.step / Trainer.evaluate_rollout (trainer.py:217-344). tests/golden/
              make_golden_surrogate.py compiled those two classes from the reference's source text and ran
              them; tests/test_oracle_golden.py holds this oracle (and tests/test_surrogate_gpu.py the HIP
              path) to the vectors it stored (tests/golden/surrogate_ref_*.npz).
  unpinned -- everything the absent packages compute: neighbour search, PyG MLP / EdgeConv, scatter.
What follows is a
pure-torch restatement of (a) the reference's own lines, cited per function, and (b) the
published semantics of the third-party calls it makes, written down here as the SPECIFICATION
the HIP path is tested against:

  knn_graph(x, k, batch, loop=False)   [gnn.py:13, datautils.py:36]
      torch_cluster 1.6.3 `knn_graph`: `knn(x, x, k if loop else k + 1, batch, batch)` -- for every centre i the
      k (+1) nearest nodes j of the same batch segment, ITSELF INCLUDED as a candidate at distance 0 -- and then,
      if not loop, the entries with row == col are dropped. Squared distance d2 = (dx*dx + dy*dy) + dz*dz in fp32;
      ties -> lower index (the CUDA kernel scans candidates in index order and replaces its current worst only on
      a strictly smaller distance); edges grouped by centre in ascending centre index and, inside a group,
      ascending (d2, j). edge_index[0] = j (neighbour/source), edge_index[1] = i (centre/target)
      ("source_to_target"). Consequence of "search k + 1, then drop self" (as opposed to masking the diagonal,
      which this oracle did until round 3): a centre with at least k + 1 lower-indexed bodies at distance exactly
      0 never sees itself among its k + 1 nearest and KEEPS ALL k + 1. Everywhere else the two rules agree and
      E = sum_i min(k, segment_size_i - 1).
  radius_graph(x, r, batch, loop, max_num_neighbors=32)   [contconv.py:225]
      torch_cluster 1.6.3 `radius_graph`: `radius(x, x, r, batch, batch, max_num_neighbors if loop else
      max_num_neighbors + 1)` -- for every centre i the first max_num_neighbors (+1) nodes j (ascending index,
      same segment, ITSELF INCLUDED) with d2 < r2 strictly -- and then, if not loop, row == col is dropped. So
      without self loops a centre with >= max_num_neighbors + 1 lower-indexed hits returns max_num_neighbors + 1
      neighbours (33), any other centre at most max_num_neighbors. r2 = float(double(r) * double(r)): the kernel
      receives r * r computed in double and cast to fp32 (0.7 -> 0.49000001; the fp32 product of the fp32 radius,
      this oracle's rule until round 3, is 0.48999998). Same edge_index orientation and grouping.
      (torch_cluster's CUDA kernel scans in index order and keeps the first hits; its CPU path
      returns a kd-tree-ordered arbitrary subset when the cap binds -- only the former is
      reproducible)
  EdgeConv(nn, aggr)   [gnn.py:75-93]   x_i' = aggr_{j in N(i)} nn([x_i || x_j - x_i]), aggregated at
      edge_index[1]; nodes without edges get 0; mean divides by the edge count.
  MLP(channels, act="tanh", norm=...)   [gnn.py:57-63, contconv.py:136-141]
      [Linear -> (BatchNorm1d if norm) -> tanh -> dropout] x (L-1) -> Linear (plain_last);
      parameters lins.{i}.weight/bias, norms.{i}.module.{weight,bias,running_mean,running_var}.
      PyG's default norm is "batch_norm": the ContConv encoder has it (norm not passed,
      contconv.py:136-141), the GNN encoder does not (norm=None, gnn.py:62). Inference uses the
      running statistics (predict/eval_graph_batch call eval()).
  scatter(src, index, dim_size, reduce)   [contconv.py:95-97]   sum, or sum / max(count, 1) for mean;
      "max" / "min": per-channel extreme of the messages of a row, 0 for a row without messages;
      "mul": per-channel product of the messages of a row, 1 for a row without messages (torch_scatter 2.1
      scatter(): reduce == "mul" -> scatter_mul, whose output starts as torch.ones).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- neighbour search
def _segments(n: int, batch):
    if batch is None:
        return [(0, n)]
    b = batch.tolist()
    segs, start = [], 0
    for i in range(1, n + 1):
        if i == n or b[i] != b[start]:
            segs.append((start, i))
            start = i
    return segs


def _d2(pos: torch.Tensor, lo: int, hi: int, r0: int | None = None, r1: int | None = None) -> torch.Tensor:
    """d2[i - r0, j - lo] for centres i in [r0, r1) (default: the whole segment) and candidates j in [lo, hi)."""
    p = pos[lo:hi].to(torch.float32)
    c = p if r0 is None else pos[r0:r1].to(torch.float32)
    d = p.unsqueeze(0) - c.unsqueeze(1)            # d[i, j] = p_j - p_i
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    return (dx * dx + dy * dy) + dz * dz           # fp32, this association


_ROW_BLOCK = 2048      # centres per block: bounds the (rows, segment) temporaries; the arithmetic is per pair


def knn_graph(pos: torch.Tensor, k: int, batch=None, loop: bool = False) -> torch.Tensor:
    n = pos.shape[0]
    src, dst = [], []
    for lo, hi in _segments(n, batch):
        m = hi - lo
        d2 = _d2(pos, lo, hi)
        kk = min(k if loop else k + 1, m)             # self is a candidate (distance 0) either way
        if kk <= 0:
            continue
        # stable sort on d2 keeps lower indices first among ties
        order = torch.sort(d2, dim=1, stable=True).indices[:, :kk]
        j = (order + lo).reshape(-1)
        i = (torch.arange(lo, hi).unsqueeze(1).expand(m, kk)).reshape(-1)
        if not loop:                                  # torch_cluster: mask = row != col
            keep = j != i
            j, i = j[keep], i[keep]
        src.append(j)
        dst.append(i)
    if not src:
        return torch.zeros((2, 0), dtype=torch.int64)
    return torch.stack([torch.cat(src), torch.cat(dst)]).to(torch.int64)


def radius_r2(r: float) -> float:
    """r * r in double, then cast to fp32 (what torch_cluster's kernel is handed)."""
    return torch.tensor(float(r) * float(r), dtype=torch.float64).to(torch.float32).item()


def radius_graph(pos: torch.Tensor, r: float, batch=None, loop: bool = False,
                 max_num_neighbors: int = 32) -> torch.Tensor:
    n = pos.shape[0]
    r2 = radius_r2(r)
    cap = max_num_neighbors if loop else max_num_neighbors + 1      # self counts towards the cap either way
    src, dst = [], []
    for lo, hi in _segments(n, batch):
        m = hi - lo
        for b0 in range(lo, hi, _ROW_BLOCK):
            b1 = min(b0 + _ROW_BLOCK, hi)
            ok = _d2(pos, lo, hi, b0, b1) < r2
            rank = torch.cumsum(ok.to(torch.int32), dim=1)          # 1-based rank in index order
            ok = ok & (rank <= cap)
            if not loop:                                            # ... and only then is row == col dropped
                rows = torch.arange(b1 - b0)
                ok[rows, rows + (b0 - lo)] = False
            i_idx, j_idx = torch.nonzero(ok, as_tuple=True)         # row-major: grouped by centre, j ascending
            src.append(j_idx + lo)
            dst.append(i_idx + b0)
    if not src:
        return torch.zeros((2, 0), dtype=torch.int64)
    return torch.stack([torch.cat(src), torch.cat(dst)]).to(torch.int64)


def scatter(src: torch.Tensor, index: torch.Tensor, dim_size: int, reduce: str) -> torch.Tensor:
    out = torch.zeros((dim_size,) + src.shape[1:], dtype=src.dtype)
    out.index_add_(0, index, src)
    if reduce == "mean":
        cnt = torch.zeros(dim_size, dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        out = out / cnt.clamp(min=1).unsqueeze(-1)
    elif reduce not in ("sum", "add"):
        raise ValueError(reduce)
    return out


# --------------------------------------------------------------------------- building blocks
class PygMLP(torch.nn.Module):
    """torch_geometric.nn.MLP as the reference configures it (see module docstring)."""

    def __init__(self, channels, norm="batch_norm"):
        super().__init__()
        self.lins = torch.nn.ModuleList(torch.nn.Linear(a, b) for a, b in zip(channels[:-1], channels[1:]))
        self.norms = torch.nn.ModuleList()
        for c in channels[1:-1]:
            if norm is None:
                self.norms.append(torch.nn.Identity())
            else:
                holder = torch.nn.Module()
                holder.module = torch.nn.BatchNorm1d(c)
                self.norms.append(holder)
        self.has_norm = norm is not None

    def forward(self, x):
        for i, lin in enumerate(self.lins[:-1]):
            x = lin(x)
            if self.has_norm:
                x = self.norms[i].module(x)
            x = torch.tanh(x)
        return self.lins[-1](x)


class EdgeConv(torch.nn.Module):
    def __init__(self, nn, aggr):
        super().__init__()
        self.nn, self.aggr = nn, aggr

    def forward(self, x, edge_index):
        j, i = edge_index[0], edge_index[1]
        msg = self.nn(torch.cat([x[i], x[j] - x[i]], dim=-1))
        if self.aggr == "max":
            out = torch.full((x.shape[0], msg.shape[1]), float("-inf"), dtype=msg.dtype)
            out = out.scatter_reduce(0, i.unsqueeze(1).expand_as(msg), msg, reduce="amax", include_self=True)
            return torch.where(torch.isinf(out), torch.zeros_like(out), out)
        return scatter(msg, i, x.shape[0], "mean" if self.aggr == "mean" else "sum")


# --------------------------------------------------------------------------- GNN (gnn.py:25-148)
class GraphModelOracle(torch.nn.Module):
    def __init__(self, input_dim=1, output_hiddens=None, output_dim=3, node_encoder_dims=None, gnn_dim=128,
                 message_passing_steps=4, aggr="sum", neighbors=50):
        super().__init__()
        self.input_dim, self.neighbors = input_dim, neighbors
        self.node_encoder_dims = node_encoder_dims
        self.node_encoder = (PygMLP([input_dim] + node_encoder_dims + [gnn_dim], norm=None)
                             if node_encoder_dims else torch.nn.Identity())                # gnn.py:56-65
        self.gnns = torch.nn.ModuleList()
        for l in range(message_passing_steps):                                             # gnn.py:71-95
            fin = input_dim if (l == 0 and node_encoder_dims is None) else gnn_dim
            self.gnns.append(EdgeConv(torch.nn.Sequential(
                torch.nn.Linear(2 * fin, gnn_dim), torch.nn.Tanh(), torch.nn.Linear(gnn_dim, gnn_dim)), aggr))
        out_dim = gnn_dim + input_dim if node_encoder_dims is None else 2 * gnn_dim       # gnn.py:97-100
        self.layer_norm = torch.nn.LayerNorm(out_dim)
        if output_hiddens:                                                                 # gnn.py:105-114
            dims = [out_dim] + output_hiddens + [output_dim]
            layers = []
            for a in range(len(dims) - 1):
                layers.append(torch.nn.Linear(dims[a], dims[a + 1]))
                if a < len(dims) - 2:
                    layers.append(torch.nn.Tanh())
            self.output = torch.nn.Sequential(*layers)
        else:
            self.output = torch.nn.Linear(out_dim, output_dim)

    def forward_graph(self, x7, edge_index):
        x = torch.cat((x7[:, :3], x7[:, 6:]), dim=-1) if self.input_dim == 4 else x7       # gnn.py:131-134
        x = self.node_encoder(x)
        enc = x
        for g in self.gnns:
            x = g(x, edge_index)
        return self.output(self.layer_norm(torch.cat((enc, x), dim=-1)))                   # gnn.py:144-148

    def predict(self, pos, feat, k=50):
        """gnn.py:205-215: transform_to_graph never receives `neighbors`, so k = 50 (gnn.py:11)."""
        with torch.no_grad():
            ei = knn_graph(pos, k, loop=False)
            return self.forward_graph(torch.cat((pos, feat), dim=-1), ei)


# --------------------------------------------------------------------------- ContinuousConv (contconv.py:10-98)
class ContinuousConvOracle(torch.nn.Module):
    def __init__(self, in_channels, out_channels, filter_resolution=4, radius=0.5, agg="mean"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.radius, self.agg, self.filter_resolution = radius, agg, filter_resolution
        self.filters = torch.nn.Parameter(torch.randn(filter_resolution, filter_resolution, filter_resolution,
                                                      in_channels, out_channels))

    def trilinear_interpolate(self, coords):
        # contconv.py:53-78 verbatim in behaviour: F.grid_sample is the reference's own call
        D = self.filter_resolution
        g = ((coords / (D - 1)) * 2 - 1).unsqueeze(0).unsqueeze(2).unsqueeze(3)
        vol = self.filters.view(D, D, D, -1).permute(3, 0, 1, 2).unsqueeze(0)
        s = F.grid_sample(vol, g, mode="bilinear", align_corners=True)
        return s.squeeze(0).squeeze(-1).squeeze(-1).transpose(0, 1).view(-1, self.in_channels, self.out_channels)

    def forward(self, positions, features, edge_index, edge_block=4096):
        row, col = edge_index[0], edge_index[1]
        out = torch.zeros((positions.shape[0], self.out_channels), dtype=features.dtype)
        extreme = self.agg in ("max", "min", "mul")    # scatter(reduce="max"/"min"/"mul"): per-edge messages, rows without
        messages = []                                  # edges stay 0 (torch_scatter fills empty segments with 0)
        for e0 in range(0, row.numel(), edge_block):       # blocked only to bound the (E,I,O) temporary
            rw, cl = row[e0:e0 + edge_block], col[e0:e0 + edge_block]
            r = positions[cl] - positions[rw]                                              # :84
            dist2 = (r ** 2).sum(dim=-1)
            window = ((1 - dist2 / (self.radius ** 2)) ** 3) * (dist2 < self.radius ** 2).float()   # :85-87
            norm = torch.norm(r, dim=-1, keepdim=True)
            mapped = r / (norm + 1e-8) * torch.tanh(norm)                                  # :30-33
            grid = (mapped + 1) * ((self.filter_resolution - 1) / 2)                       # :90
            filt = self.trilinear_interpolate(grid)
            conv = torch.einsum("eio,ei->eo", filt, features[cl]) * window.unsqueeze(1)    # :92-93
            if extreme:
                messages.append(conv)
            else:
                out.index_add_(0, rw, conv)
        if extreme and self.agg == "mul":              # scatter_mul: ones, multiplied by every message of the row
            out = torch.ones_like(out)
            if messages:
                out.scatter_reduce_(0, row.unsqueeze(1).expand(-1, self.out_channels), torch.cat(messages),
                                    reduce="prod", include_self=True)
            return out
        if extreme:
            if messages:
                out.scatter_reduce_(0, row.unsqueeze(1).expand(-1, self.out_channels), torch.cat(messages),
                                    reduce="amax" if self.agg == "max" else "amin", include_self=False)
            return out
        if self.agg == "mean":                                                             # :95-97
            cnt = torch.zeros(positions.shape[0]).index_add_(0, row, torch.ones(row.numel()))
            out = out / cnt.clamp(min=1).unsqueeze(-1)
        return out


class ContinuousConvModelOracle(torch.nn.Module):
    """contconv.py:101-234 (list-form filter_resolution only: the scalar branch is broken upstream)."""

    def __init__(self, in_channels=4, out_channels=3, filter_resolution=(4,), radius=0.5, agg="mean",
                 self_loops=True, continuous_conv_layers=1, continuous_conv_dim=64, encoder_hiddens=None,
                 decoder_hiddens=None, max_num_neighbors=32):
        super().__init__()
        self.in_channels, self.radius, self.self_loops = in_channels, radius, self_loops
        self.max_num_neighbors = max_num_neighbors
        self.encoder_hiddens = encoder_hiddens
        self.node_encoder = (PygMLP([in_channels] + list(encoder_hiddens) + [continuous_conv_dim])
                             if encoder_hiddens else torch.nn.Identity())                  # :135-143
        self.contconv = torch.nn.ModuleList()
        for l in range(continuous_conv_layers):                                            # :150-173
            cin = in_channels if (l == 0 and encoder_hiddens is None) else continuous_conv_dim
            self.contconv.append(ContinuousConvOracle(cin, continuous_conv_dim, filter_resolution[l], radius, agg))
        out_dim = continuous_conv_dim + in_channels if encoder_hiddens is None else 2 * continuous_conv_dim
        self.layer_norm = torch.nn.LayerNorm(out_dim)
        if decoder_hiddens:
            dims = [out_dim] + list(decoder_hiddens) + [out_channels]
            layers = []
            for a in range(len(dims) - 1):
                layers.append(torch.nn.Linear(dims[a], dims[a + 1]))
                if a < len(dims) - 2:
                    layers.append(torch.nn.Tanh())
            self.output = torch.nn.Sequential(*layers)
        else:
            self.output = torch.nn.Linear(out_dim, out_channels)

    def forward_x(self, x7, batch=None):
        x = torch.cat((x7[:, :3], x7[:, 6:]), dim=-1) if self.in_channels == 4 else x7     # :219-222
        pos = x[:, :3]
        ei = radius_graph(pos, self.radius, batch, loop=self.self_loops, max_num_neighbors=self.max_num_neighbors)
        x = self.node_encoder(x)
        enc = x
        for layer in self.contconv:
            x = torch.tanh(layer(pos, x, ei))                                              # :228-230
        return self.output(self.layer_norm(torch.cat((enc, x), dim=-1)))                   # :233-234

    def predict(self, pos, feat):
        with torch.no_grad():
            return self.forward_x(torch.cat((pos, feat), dim=-1))


# --------------------------------------------------------------------------- rollout (trainer.py:217-226)
def trainer_step(predict, pos, vel, m, acc, dt):
    vel_ = vel + 0.5 * dt * acc
    pos_ = pos + dt * vel_
    acc_ = predict(pos_, torch.cat([vel_, m], dim=-1))
    vel_ = vel_ + 0.5 * dt * acc_
    return pos_, vel_, acc_
