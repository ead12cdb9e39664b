"""Minimal stand-ins for the two torch_geometric containers the reference passes around
(`Data`, `DataLoader` batches; gnn.py:6,14-20, datautils.py:38-44,51-53): attribute bags of
tensors with `.to(device)`. Only what the hot path reads: x, edge_index, y, batch, scene, step."""
from __future__ import annotations

import torch

from . import graphops


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, batch=None, **extra):
        self.x, self.edge_index, self.edge_attr, self.y, self.batch = x, edge_index, edge_attr, y, batch
        for k, v in extra.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]

    def to(self, device):
        out = Data()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device) if isinstance(v, torch.Tensor) else v)
        return out

    @property
    def num_nodes(self):
        return 0 if self.x is None else self.x.shape[0]

    def __repr__(self):
        parts = [f"{k}={list(v.shape)}" if isinstance(v, torch.Tensor) else f"{k}={v!r}" for k, v in self.__dict__.items()
                 if v is not None]
        return "Data(" + ", ".join(parts) + ")"


def collate(graphs: list[Data]) -> Data:
    """Concatenate graphs the way PyG's DataLoader does: node tensors stacked, edge_index offset by
    the running node count, `batch` = graph id per node. A handful of device ops whatever the number of
    graphs: one cat per field, and the per-edge / per-node graph offsets by repeat_interleave with the
    (host-known) output sizes, so no op waits on the device."""
    out = {}
    node_keys = [k for k in graphs[0].keys() if k != "edge_index" and isinstance(getattr(graphs[0], k), torch.Tensor)]
    dev = graphs[0].x.device
    sizes = [g.num_nodes for g in graphs]
    for k in node_keys:
        out[k] = torch.cat([getattr(g, k) for g in graphs], dim=0)
    n_total = sum(sizes)
    sizes_t = torch.tensor(sizes, dtype=torch.int64).to(dev, non_blocking=True)
    gid = torch.arange(len(graphs), dtype=torch.int64, device=dev)
    with_edges = [g for g in graphs if g.edge_index is not None]
    if with_edges:
        counts = [g.edge_index.shape[1] if g.edge_index is not None else 0 for g in graphs]
        starts = [0]
        for n in sizes[:-1]:
            starts.append(starts[-1] + n)
        e_total = sum(counts)
        ei = torch.cat([g.edge_index for g in with_edges], dim=1)
        shift = torch.repeat_interleave(torch.tensor(starts, dtype=torch.int64).to(dev, non_blocking=True),
                                        torch.tensor(counts, dtype=torch.int64).to(dev, non_blocking=True),
                                        output_size=e_total)
        out["edge_index"] = ei + shift
        if all(graphops.marked(g.edge_index, "_nbd_grouped") for g in with_edges):
            graphops.mark(out["edge_index"], "_nbd_grouped")     # grouped member graphs stay grouped after the offsets
    else:
        out["edge_index"] = None
    out["batch"] = torch.repeat_interleave(gid, sizes_t, output_size=n_total)
    graphops.mark(out["batch"], "_nbd_sorted")  # ascending by construction: graphops need not validate (a host sync)
    return Data(**out)
