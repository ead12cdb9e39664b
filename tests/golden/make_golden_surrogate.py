"""Golden vectors for the PARTS of the surrogate path whose arithmetic lives in the reference's own
files and needs none of its missing third-party packages (torch_geometric / torch_scatter /
torch_cluster are not installed, so `import contconv` / `import trainer` fail at their first lines):

  contconv.py  ContinuousConv.ball_to_cube, ContinuousConv.trilinear_interpolate   (contconv.py:30-33, 53-78)
  trainer.py   Trainer.step, Trainer.evaluate_rollout                              (trainer.py:217-344)

The two classes are compiled from the reference's source text as it lies under /root/reference (class
definition only, via ast -- nothing is copied into this repository) and run on seeded inputs; the
inputs and the outputs they produced are stored in tests/golden/surrogate_ref_*.npz. What stays
unpinned is what the absent packages compute: neighbour search, PyG's MLP / EdgeConv, scatter.

Run in the build container:  python tests/golden/make_golden_surrogate.py
"""
import ast
import os
import sys
import time

import numpy as np
import pandas as pd
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "nbody-deep-sim_amd"))


def load_class(path, name, namespace):
    tree = ast.parse(open(path).read(), filename=path)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), namespace)
    return namespace[name]


def contconv_vectors():
    cls = load_class(f"{REF}/contconv.py", "ContinuousConv", {"torch": torch, "nn": nn, "F": F})
    out = {}
    for case, (d, i, o) in enumerate([(2, 1, 1), (4, 3, 5), (6, 2, 3), (3, 4, 2)]):
        torch.manual_seed(100 + case)
        layer = cls(i, o, filter_resolution=d, radius=1.0)
        coords = torch.rand(48, 3) * (d - 1)
        coords[:4] = torch.tensor([[0.0, 0.0, 0.0], [d - 1.0, d - 1.0, d - 1.0], [0.0, d - 1.0, 0.5], [1.0, 0.0, d - 1.0]])
        r = torch.randn(48, 3) * torch.logspace(-3, 0.3, 48).unsqueeze(1)
        r[0] = 0.0
        with torch.no_grad():
            out[f"c{case}_filters"] = layer.filters.detach().numpy().copy()
            out[f"c{case}_coords"] = coords.numpy().copy()
            out[f"c{case}_interp"] = layer.trilinear_interpolate(coords).numpy().copy()
            out[f"c{case}_r"] = r.numpy().copy()
            out[f"c{case}_cube"] = layer.ball_to_cube(r).numpy().copy()
            # the composition forward() applies per edge (contconv.py:89-91)
            grid = (layer.ball_to_cube(r) + 1) * ((d - 1) / 2)
            out[f"c{case}_interp_of_r"] = layer.trilinear_interpolate(grid).numpy().copy()
    np.savez_compressed(os.path.join(HERE, "surrogate_ref_contconv.npz"), **out)
    print("contconv:", {k: v.shape for k, v in out.items() if k.startswith("c1")})


class ToyModel:
    """A `model` argument for Trainer: fp32-exact arithmetic (one rounding per op on any IEEE device)."""

    def to(self, device):
        return self

    def predict(self, pos, feat):
        return (feat[:, 3:4] * (-pos)) * 0.5 + feat[:, :3] * 0.25


def trainer_vectors():
    import tqdm
    from datetime import datetime
    from glob import glob
    from nbd.data import Data               # attribute bag with .to(); the reference passes a PyG batch here
    cls = load_class(f"{REF}/trainer.py", "Trainer", {"torch": torch, "pd": pd, "time": time, "os": os, "glob": glob,
                                                       "tqdm": tqdm, "datetime": datetime})
    tr = cls(ToyModel(), optimizer=None, device="cpu", dt=0.01)
    g = torch.Generator().manual_seed(7)
    n, steps = 7, 5
    pos = torch.randn(n, 3, generator=g)
    vel = torch.randn(n, 3, generator=g) * 0.3
    m = torch.rand(n, 1, generator=g) + 0.5
    acc = torch.randn(n, 3, generator=g)
    p1, v1, a1 = tr.step(pos, vel, m, acc, 0.01)
    x = torch.cat([torch.cat([torch.randn(n, 6, generator=g), m], 1) for _ in range(steps)])
    y = torch.randn(n * steps, 3, generator=g)
    step = torch.arange(steps).repeat_interleave(n)
    data = Data(x=x, y=y, step=step)
    df = tr.evaluate_rollout("file.csv", data, 3, steps, 0.01, pd.DataFrame())
    cols = [c for c in df.columns if c not in ("filename", "step_time")]
    np.savez_compressed(os.path.join(HERE, "surrogate_ref_trainer.npz"),
                        pos=pos.numpy(), vel=vel.numpy(), m=m.numpy(), acc=acc.numpy(), dt=np.float64(0.01),
                        step_pos=p1.numpy(), step_vel=v1.numpy(), step_acc=a1.numpy(),
                        data_x=x.numpy(), data_y=y.numpy(), data_step=step.numpy(),
                        rollout_columns=np.array(list(df.columns)), rollout_numeric_columns=np.array(cols),
                        rollout_values=df[cols].to_numpy(dtype=np.float64),
                        rollout_filename=np.array(df["filename"].tolist()))
    print("trainer: rollout frame", df.shape, list(df.columns)[:6], "...")


if __name__ == "__main__":
    contconv_vectors()
    trainer_vectors()
