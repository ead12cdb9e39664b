"""Minimal stand-ins for the two torch_geometric containers the reference passes around
(`Data`, `DataLoader` batches; gnn.py:6,14-20, datautils.py:38-44,51-53): attribute bags of
tensors with `.to(device)`. Only what the hot path reads: x, edge_index, y, batch, scene, step."""
from __future__ import annotations

import torch


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
    the running node count, `batch` = graph id per node."""
    out, offset = {}, 0
    node_keys = [k for k in graphs[0].keys() if k != "edge_index" and isinstance(getattr(graphs[0], k), torch.Tensor)]
    cols = {k: [] for k in node_keys}
    eis, batch = [], []
    for gi, g in enumerate(graphs):
        n = g.num_nodes
        for k in node_keys:
            cols[k].append(getattr(g, k))
        if g.edge_index is not None:
            eis.append(g.edge_index + offset)
        batch.append(torch.full((n,), gi, dtype=torch.int64, device=g.x.device))
        offset += n
    for k in node_keys:
        out[k] = torch.cat(cols[k], dim=0)
    out["edge_index"] = torch.cat(eis, dim=1) if eis else None
    out["batch"] = torch.cat(batch)
    return Data(**out)
