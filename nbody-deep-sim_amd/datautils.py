"""CSV -> per-(scene, step) particle graphs (the reference's datautils.py:9-53, minus the PyG
containers and the on-disk `processed/*.pt` cache): columns x,y,z,vx,vy,vz,mass,ax,ay,az,scene,step of
the dataset CSV (s01-dataset-generation.py:108-125) become Data(x=[pos|vel|mass], y=acc, scene, step,
edge_index = kNN graph built on the GPU). get_dataloader() yields PyG-style batches (nbd.data.collate).
"""
from __future__ import annotations

import pandas as pd
import torch

from nbd import graphops
from nbd.data import Data, collate


class ParticleGraphDataset:
    def __init__(self, csv_path, k=8, device="cuda"):
        self.csv_path, self.k = csv_path, k
        df = pd.read_csv(csv_path)
        self.graphs = []
        for (scene, step), group in df.groupby(["scene", "step"]):                # datautils.py:26
            x = torch.tensor(group[["x", "y", "z", "vx", "vy", "vz", "mass"]].values, dtype=torch.float, device=device)
            y = torch.tensor(group[["ax", "ay", "az"]].values, dtype=torch.float, device=device)
            n = x.shape[0]
            ei = graphops.knn_graph(x[:, :3].contiguous(), k=k, loop=False) if k > 0 else \
                torch.zeros((2, 0), dtype=torch.int64, device=device)
            self.graphs.append(Data(x=x, edge_index=ei, y=y,
                                    scene=torch.full((n,), int(scene), dtype=torch.int64, device=device),
                                    step=torch.full((n,), int(step), dtype=torch.int64, device=device)))

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
