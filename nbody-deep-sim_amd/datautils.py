"""CSV -> per-(scene, step) particle graphs (the reference's datautils.py:9-53, minus the PyG
containers and the on-disk `processed/*.pt` cache): columns x,y,z,vx,vy,vz,mass,ax,ay,az,scene,step of
the dataset CSV (s01-dataset-generation.py:108-125) become Data(x=[pos|vel|mass], y=acc, scene, step,
edge_index = kNN graph built on the GPU). get_dataloader() yields PyG-style batches (nbd.data.collate).
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch

from nbd import graphops
from nbd.data import Data, collate


_COLS = ["scene", "step", "x", "y", "z", "vx", "vy", "vz", "mass", "ax", "ay", "az"]


def read_rows(csv_path):
    """The twelve columns the graphs are made of, as float64 / int64 arrays in file order. Only these are parsed
    (the reference parses all sixteen, datautils.py:24); pandas' multithreaded pyarrow engine when it is there."""
    try:
        import pyarrow  # noqa: F401
        df = pd.read_csv(csv_path, usecols=_COLS, engine="pyarrow")
    except (ImportError, ValueError):       # no pyarrow, or a pandas that rejects this usecols / engine combination
        # round_trip: the C parser's default fast path can be one fp64 ulp off the correctly rounded value pyarrow
        # gives; with it both engines return the same numbers, so the dataset does not depend on the environment
        df = pd.read_csv(csv_path, usecols=_COLS, float_precision="round_trip")
    return df


class ParticleGraphDataset:
    """One graph per (scene, step), in ascending (scene, step) order, rows of a graph in file order -- what
    `df.groupby(["scene", "step"])` iterates (datautils.py:26-44). Built without a per-group loop: one stable
    sort of the row keys, ONE host->device copy of the node table, ONE batched kNN launch over all graphs
    (csrc/graph.hip searches each node inside its own graph's segment), then per-graph views."""

    def __init__(self, csv_path, k=8, device="cuda"):
        self.csv_path, self.k = csv_path, k
        df = read_rows(csv_path)
        scene = df["scene"].to_numpy(dtype=np.int64)
        step = df["step"].to_numpy(dtype=np.int64)
        order = np.lexsort((step, scene))                                          # stable: file order inside a group
        scene, step = scene[order], step[order]
        rows = scene.shape[0]
        new = np.ones(rows, dtype=bool)
        new[1:] = (scene[1:] != scene[:-1]) | (step[1:] != step[:-1])
        starts = np.flatnonzero(new)
        sizes = np.diff(np.append(starts, rows))
        feat = df[["x", "y", "z", "vx", "vy", "vz", "mass", "ax", "ay", "az"]].to_numpy(dtype=np.float64)[order]
        table = torch.tensor(feat, dtype=torch.float).to(device)                   # fp64 -> fp32 as torch.tensor(dtype=float)
        x_all, y_all = table[:, :7].contiguous(), table[:, 7:].contiguous()
        scene_all = torch.tensor(scene).to(device)
        step_all = torch.tensor(step).to(device)
        n_graphs = starts.shape[0]
        per_graph_edges = sizes * np.minimum(k, np.maximum(sizes - 1, 0)) if k > 0 else np.zeros(n_graphs, dtype=np.int64)
        if k > 0 and rows:
            sizes_t = torch.tensor(sizes).to(device)
            gid = torch.repeat_interleave(torch.arange(n_graphs, device=device), sizes_t, output_size=rows)
            graphops.mark(gid, "_nbd_sorted")
            ei_all = graphops.knn_graph(x_all[:, :3].contiguous(), k=k, batch=gid, loop=False)
            first = torch.repeat_interleave(torch.tensor(starts).to(device), torch.tensor(per_graph_edges).to(device),
                                            output_size=int(per_graph_edges.sum()))
            ei_all = ei_all - first                                                # node ids local to their graph
        else:
            ei_all = torch.zeros((2, 0), dtype=torch.int64, device=device)
        self.graphs = []
        e0 = 0
        for g in range(n_graphs):
            a, b, e1 = int(starts[g]), int(starts[g] + sizes[g]), e0 + int(per_graph_edges[g])
            ei = ei_all[:, e0:e1].contiguous()
            graphops.mark(ei, "_nbd_grouped")
            self.graphs.append(Data(x=x_all[a:b], edge_index=ei, y=y_all[a:b], scene=scene_all[a:b], step=step_all[a:b]))
            e0 = e1

    def __len__(self):
        return len(self.graphs)

    def __getitem__(self, i):
        return self.graphs[i]


class DataLoader:
    def __init__(self, dataset, batch_size=32, shuffle=True):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = torch.randperm(len(self.dataset)).tolist() if self.shuffle else list(range(len(self.dataset)))
        for b in range(0, len(order), self.batch_size):
            yield collate([self.dataset[i] for i in order[b:b + self.batch_size]])


def get_dataloader(csv_path, batch_size=32, k=8, shuffle=True, device="cuda"):
    return DataLoader(ParticleGraphDataset(csv_path, k=k, device=device), batch_size=batch_size, shuffle=shuffle)
