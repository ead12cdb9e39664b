"""MI355X drop-in for the reference's trainer.py: `Trainer` with `train_from_dir` (trainer.py:20-92),
`step`, `evaluate_rollout`, `evaluate_stepwise`, `test_from_dir` (trainer.py:94-344) -- same
signatures, same DataFrame columns and statistics. The training loop is the reference's (csv files ->
graph batches -> model.train_graph_batch -> scheduler / checkpoints); forward and backward of each
batch run in the HIP kernels (nbd/autograd.py), the optimiser is whatever torch.optim object was passed in.

Differences that do not change outputs: the leapfrog updates run in the HIP kick/drift kernels;
the 18 N `.item()` host syncs per rollout step (trainer.py:286-312) are replaced by one bulk
device->host copy per rollout; step_time is measured with HIP events (the reference's
un-synchronised time.time() would time kernel launches only).
"""
from __future__ import annotations

import os
import time
from glob import glob

import numpy as np
import pandas as pd
import torch

from datautils import get_dataloader
from nbd import direct
from nbd._lib import NbdUnsupported

ROLLOUT_COLUMNS = ["filename", "scene", "step", "x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az",
                   "pred_x", "pred_y", "pred_z", "pred_vx", "pred_vy", "pred_vz", "pred_ax", "pred_ay", "pred_az",
                   "step_time"]


class Trainer:
    def __init__(self, model, optimizer, scheduler=None, device="cpu", dt=0.01):
        self.model = model
        self.optimizer = optimizer
        self.device = device
        self.dt = dt
        self.model = self.model.to(self.device)
        self.scheduler = scheduler
        # rollouts replay ONE captured hipGraph per step (the step is ~17 short launches: launch-bound)
        self.use_hip_graph = True
        self.hip_graph_min_steps = 5       # shorter rollouts do not amortise the capture
        self.pre_advance = True            # captured GNN step: leapfrog bookkeeping in the last layer's epilogue (_capture_step)
        self.last_capture = None           # which form _capture_step captured: "pre_advance" / "packed" / "generic"; None: eager

    def train_from_dir(self, data_path, epochs, batch_size, save_every, save_path=None, create_save_path=False):
        """trainer.py:20-92. Returns (epoch_losses, epoch_mse_losses). Checkpoints are `model_{epoch}.pt`
        state_dicts; an existing `save_path` is resumed from its highest-numbered file. (The reference
        leaves `path` unbound when save_every > 0 with neither save_path nor create_save_path; here nothing
        is saved in that case.)"""
        path = None
        if save_every > 0:
            if save_path:
                path = save_path
            elif create_save_path:
                from datetime import datetime
                path = "./models" + datetime.now().strftime("%Y%m%d%H%M%S")
                os.mkdir(path)
        last_model = 0
        if save_path:
            try:
                models = sorted(os.listdir(save_path), key=lambda x: int(x.split("_")[1].split(".")[0]))
                with torch.no_grad():
                    self.model.load_state_dict(torch.load(f"{save_path}/{models[-1]}", map_location=self.device))
                print(f"Loaded model {models[-1]}")
            except (IndexError, ValueError, OSError):
                print("No model found")
        csv_files = [f.replace("\\", "/") for f in glob(data_path + "/*.csv")]
        # the graphs of a csv file do not change between epochs: build each file's dataset once
        datasets = {f: get_dataloader(csv_path=f, batch_size=batch_size, k=self.model.neighbors, device=self.device)
                    for f in csv_files}
        try:
            import tqdm
            epochs_range = tqdm.trange(epochs)
        except ImportError:
            epochs_range = range(epochs)
        epoch_losses, epoch_mse_losses = [], []
        for epoch in epochs_range:
            epoch_loss, epoch_mse_loss = [], []
            for f in csv_files:
                for data in datasets[f]:
                    loss, mse_loss = self.model.train_graph_batch(self.optimizer, data)
                    epoch_loss.append(loss)
                    epoch_mse_loss.append(mse_loss)
            epoch_losses.append(sum(epoch_loss) / len(epoch_loss))
            epoch_mse_losses.append(sum(epoch_mse_loss) / len(epoch_mse_loss))
            if hasattr(epochs_range, "set_postfix_str"):
                epochs_range.set_postfix_str(f"Epoch {epoch+1}: Loss: {epoch_losses[-1]}, MSE: {epoch_mse_losses[-1]}")
            if self.scheduler:
                self.scheduler.step(epoch_losses[-1])
            if path and save_every > 0 and (epoch + 1) % save_every == 0:
                torch.save(self.model.state_dict(), f"{path}/model_{epoch+1+last_model}.pt")
                print(f"Saved model {epoch+1+last_model}")
        return epoch_losses, epoch_mse_losses

    # ------------------------------------------------------------------ trainer.py:217-226
    def step(self, pos, vel, m, acc, dt, predict=None):
        """Leapfrog with model-predicted accelerations; functional (returns new tensors). predict: the function standing
        in for self.model.predict (the scenes-together rollout passes the model's predict_batched bound to its batch)."""
        half, full = direct.f32(0.5 * dt), direct.f32(dt)
        pos_, vel_ = pos.contiguous().clone(), vel.contiguous().clone()
        direct.kick_drift(pos_, vel_, acc.contiguous(), None, half, full)          # vel_ = vel + .5dt acc ; pos_ = pos + dt vel_
        acc_ = (predict or self.model.predict)(pos_, torch.cat([vel_, m], dim=-1))
        direct.kick(vel_, acc_, half)                                              # vel_ += .5dt acc_
        return pos_, vel_, acc_

    def _capture_step(self, pos, vel, m, acc, dt, predict=None):
        """Capture self.step() on static buffers into a hipGraph; returns a callable that advances the
        static state by one step per call and hands back (pos, vel, acc) copies, or None if capture is
        not possible (then the eager path is used). Same kernels, same arithmetic, fewer launch gaps."""
        try:
            s_pos, s_vel, s_acc = pos.clone(), vel.clone(), acc.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                       # warm-up: allocator pools, weight caches, attributes
                    self.step(s_pos, s_vel, m, s_acc, dt, predict)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            half, full = direct.f32(0.5 * dt), direct.f32(dt)
            # a model whose input is [pos | mass] (GraphModel, input_dim 4) can read the packed {x,y,z,m} rows the
            # kick-drift kernel writes anyway: no concatenations in the captured step
            packed = (predict is None and hasattr(self.model, "_predict_posm") and getattr(self.model, "input_dim", 0) == 4
                      and m.dim() == 2 and m.shape[1] == 1 and m.dtype == torch.float32)
            keep = [m]                 # every buffer the captured kernels touch must outlive the graph
            if packed:
                posm = torch.zeros((direct.padded_len(s_pos.shape[0]), 4), dtype=torch.float32, device=s_pos.device)
                m_flat = m.reshape(-1).contiguous()
                keep += [posm, m_flat]
            # Pre-advancing form (GraphModel, H = 64: include/nbd.h nbd_gnn_layer_args.adv_*): the last layer's epilogue does
            # the leapfrog bookkeeping -- this step's velocity and position out, the next step's half-kick and drift in --
            # so a replay is the search and the layers only. The first half-step is taken here, once, by the usual kernel.
            pre = None
            if packed and self.pre_advance and getattr(self.model, "supports_pre_advance", False):
                vel_half, pos_pre = s_vel.clone(), s_pos.clone()
                direct.kick_drift(pos_pre, vel_half, s_acc, m_flat, half, full, posm=posm)
                # one un-captured step: does this model / shape take the one-call path with the epilogue? Only the dedicated
                # refusal (NbdUnsupported) means "no": any other failure is a real one and goes to the handler below.
                try:
                    t_pos, t_vel, t_acc, t_vh, t_pp, t_pm = (t.clone() for t in (s_pos, s_vel, s_acc, vel_half, pos_pre, posm))
                    self.model._predict_posm(t_pm, t_pp, out=t_acc, kick=(t_vel, half), advance=(t_vh, t_pos, full))
                    if self.model._advance_done:
                        # ... and the probe's step must BE step(): same position, velocity and acceleration (bit for bit:
                        # the epilogue rounds as the separate kernels do)
                        e_pos, e_vel, e_acc = self.step(s_pos, s_vel, m, s_acc, dt)
                        if not (torch.equal(e_pos, t_pos) and torch.equal(e_vel, t_vel) and torch.equal(e_acc, t_acc)):
                            raise RuntimeError("the pre-advancing step does not reproduce Trainer.step()")
                        pre = (vel_half, pos_pre)
                        keep += [vel_half, pos_pre]
                except NbdUnsupported:
                    pre = None
            with torch.cuda.graph(graph):
                # step() on the static state IN PLACE (same kernels and arithmetic; the functional clones and
                # copy-backs of step() would be five more launches per replay)
                if pre is not None:
                    o_acc = self.model._predict_posm(posm, pre[1], out=s_acc, kick=(s_vel, half), advance=(pre[0], s_pos, full))
                    if not self.model._advance_done:
                        raise RuntimeError("pre-advancing step fell off the one-call path during capture")
                elif packed:
                    direct.kick_drift(s_pos, s_vel, s_acc, m_flat, half, full, posm=posm)
                    # the second half-kick rides in the last layer's epilogue when the model's fused path allows it
                    o_acc = self.model._predict_posm(posm, s_pos, out=s_acc, kick=(s_vel, half))   # s_acc is consumed by kick_drift above
                    if not getattr(self.model, "_kick_done", False):
                        direct.kick(s_vel, o_acc, half)
                else:
                    direct.kick_drift(s_pos, s_vel, s_acc, None, half, full)
                    o_acc = (predict or self.model.predict)(s_pos, torch.cat([s_vel, m], dim=-1))
                    direct.kick(s_vel, o_acc, half)
                if o_acc.data_ptr() != s_acc.data_ptr():
                    s_acc.copy_(o_acc)
        except Exception as exc:                          # pragma: no cover - depends on runtime support
            import warnings
            warnings.warn(f"hipGraph capture of the rollout step failed ({exc}); using eager launches")
            torch.cuda.synchronize()
            self.last_capture = None
            return None

        self.last_capture = "pre_advance" if pre is not None else ("packed" if packed else "generic")

        def advance(clone=True):
            """One captured step. clone=False hands back the graph's own state buffers: valid only until
            the next call (evaluate_rollout copies them into its table right away). The buffers are outputs: in the
            pre-advancing form the next step starts from the graph's own pre-advanced copy, not from what they hold."""
            graph.replay()
            return (s_pos.clone(), s_vel.clone(), s_acc.clone()) if clone else (s_pos, s_vel, s_acc)
        cache = getattr(self.model, "_cache", None)
        if cache is not None:          # derived (folded / transposed) weights the captured kernels read
            keep.append(cache.value)
        advance.keep_alive = keep      # (s_pos / s_vel / s_acc and the graph itself live in the closure)
        return advance

    # ------------------------------------------------------------------ trainer.py:228-344
    def _rollout_inputs(self, data, sim_steps):
        """(gt (sim_steps, n, 9), pos, vel, m, feats) of one scene's batch of `sim_steps` graphs."""
        data = data.to(self.device)
        # Ground truth of every step in ONE pass: the reference selects `data.x[data.step == step]` inside the loop
        # (trainer.py:281-284) -- a boolean mask per step, i.e. a device->host count per step here. A stable sort
        # by step gives the same rows in the same order for all steps at once; the only host read-back of the
        # rollout is the per-step row count below (the reference indexes gt rows by prediction row, so every
        # step must hold the n bodies of step 0).
        step_ids = data.step.reshape(-1).to(torch.int64)
        order = torch.argsort(step_ids, stable=True)
        counts = torch.bincount(step_ids.clamp(min=0), minlength=sim_steps)[:sim_steps].cpu()
        n = int(counts[0])
        if not bool((counts == n).all()) or bool((step_ids < 0).any()):
            raise ValueError(f"evaluate_rollout: every step 0..{sim_steps - 1} must hold the {n} bodies of step 0 "
                             f"(rows per step: {counts.tolist()})")
        gt = torch.cat([data.x[:, :6], data.y], dim=1)[order[:sim_steps * n]].reshape(sim_steps, n, 9)
        first = order[:n]
        feats = data.x[first]
        return gt, feats[:, :3].contiguous(), feats[:, 3:6].contiguous(), feats[:, 6:].contiguous(), feats

    @staticmethod
    def _rollout_frame(filename, scene, table, times, sim_steps, n):
        steps = np.repeat(np.arange(sim_steps), n)
        df_new = pd.DataFrame(table.reshape(sim_steps * n, 18), columns=ROLLOUT_COLUMNS[3:21])
        df_new.insert(0, "step", steps)
        df_new.insert(0, "scene", scene)
        df_new.insert(0, "filename", filename)
        df_new["step_time"] = np.repeat(times, n)
        return df_new[ROLLOUT_COLUMNS]

    def evaluate_rollout_scenes(self, filename, datas, sim_steps, dt, df):
        """evaluate_rollout for ALL scenes of a file advanced TOGETHER as one batched system (trainer.py:171-175 runs them
        one after another: at the reference's sizes -- 3 .. 500 bodies -- a step is pure launch latency, the same few
        launches whether they carry one scene or six). Needs model.predict_batched(pos, feat, batch) (GraphModel,
        ContinuousConvModel): neighbours are searched inside a scene only. Rows come out exactly as the per-scene calls
        would append them (scene-major); `step_time` = the batched step's time / number of scenes."""
        ins = [self._rollout_inputs(d, sim_steps) for d in datas]
        sizes = [g.shape[1] for g, *_ in ins]
        pos, vel, m = (torch.cat([x[k] for x in ins]).contiguous() for k in (1, 2, 3))
        feats = torch.cat([x[4] for x in ins])
        batch = torch.repeat_interleave(torch.arange(len(ins), device=pos.device), torch.tensor(sizes, device=pos.device))
        from nbd import graphops
        graphops.mark(batch, "_nbd_sorted")
        predict = lambda p_, f_: self.model.predict_batched(p_, f_, batch)     # noqa: E731
        n_all = pos.shape[0]

        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            return out, (e0, e1)
        acc, ev = timed(lambda: predict(pos, feats[:, 3:].contiguous()))
        events = [ev]
        pred = torch.empty((sim_steps, n_all, 9), dtype=torch.float32, device=pos.device)
        torch.cat([pos, vel, acc], dim=1, out=pred[0])
        graphed = (self._capture_step(pos, vel, m, acc, dt, predict=predict)
                   if (self.use_hip_graph and sim_steps >= self.hip_graph_min_steps) else None)
        torch.cuda.synchronize()
        t_loop = time.perf_counter()
        for step in range(1, sim_steps):
            if graphed is not None:
                (pos, vel, acc), ev = timed(lambda: graphed(clone=False))
            else:
                (pos, vel, acc), ev = timed(lambda: self.step(pos, vel, m, acc, dt, predict))
            events.append(ev)
            torch.cat([pos, vel, acc], dim=1, out=pred[step])
        torch.cuda.synchronize()
        self.last_rollout_timing = {"steps": sim_steps - 1, "loop_wall_s": time.perf_counter() - t_loop,
                                    "captured": graphed is not None, "scenes_together": len(ins), "bodies": n_all}
        times = np.array([a.elapsed_time(b) * 1e-3 for a, b in events]) / len(ins)
        pred_h = pred.cpu().numpy().astype(np.float64)
        frames, lo = [], 0
        for scene, ((gt, *_), n) in enumerate(zip(ins, sizes)):
            table = np.concatenate([gt.cpu().numpy().astype(np.float64), pred_h[:, lo:lo + n]], axis=2)
            frames.append(self._rollout_frame(filename, scene, table, times, sim_steps, n))
            lo += n
        if df is not None and len(df):
            frames.insert(0, df)
        return pd.concat(frames, ignore_index=True)

    def evaluate_rollout(self, filename, data, scene, sim_steps, dt, df):
        gt, pos, vel, m, feats = self._rollout_inputs(data, sim_steps)
        n = gt.shape[1]

        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            return out, (e0, e1)

        pred_accs, ev = timed(lambda: self.model.predict(pos, feats[:, 3:].contiguous()))
        events = [ev]
        # per step [pred_pos pred_vel pred_acc] (n, 9), written into one preallocated device table
        pred = torch.empty((sim_steps, n, 9), dtype=gt.dtype, device=gt.device)
        torch.cat([pos, vel, pred_accs], dim=1, out=pred[0])
        graphed = self._capture_step(pos, vel, m, pred_accs, dt) if (self.use_hip_graph and sim_steps >= self.hip_graph_min_steps) else None
        torch.cuda.synchronize()
        t_loop = time.perf_counter()
        for step in range(1, sim_steps):
            if graphed is not None:
                (pos, vel, pred_accs), ev = timed(lambda: graphed(clone=False))
            else:
                (pos, vel, pred_accs), ev = timed(lambda: self.step(pos, vel, m, pred_accs, dt))
            events.append(ev)
            torch.cat([pos, vel, pred_accs], dim=1, out=pred[step])
        torch.cuda.synchronize()
        # wall time of the stepping loop alone (launches + GPU, no table building): how far the harness is from the
        # captured step's own GPU time
        self.last_rollout_timing = {"steps": sim_steps - 1, "loop_wall_s": time.perf_counter() - t_loop,
                                    "captured": graphed is not None}
        table = torch.cat([gt, pred], dim=2).cpu().numpy().astype(np.float64)      # ONE device->host copy
        times = np.array([a.elapsed_time(b) * 1e-3 for a, b in events])
        df_new = self._rollout_frame(filename, scene, table, times, sim_steps, n)
        return df_new if df is None or len(df) == 0 else pd.concat([df, df_new], ignore_index=True)

    # ------------------------------------------------------------------ trainer.py:202-215
    def evaluate_stepwise(self, filename, loader, df):
        rows = []
        for data in loader:
            data = data.to(self.device)
            loss, mse_loss, step_time = self.model.eval_graph_batch(data)
            rows.append({"filename": filename, "scene": data.scene[0].item(), "step": data.step[0].item(),
                         "loss": loss, "mse_loss": mse_loss, "step_time": step_time})
        if rows:
            new = pd.DataFrame(rows)
            df = new if df is None or len(df) == 0 else pd.concat([df, new], ignore_index=True)
        return df

    # ------------------------------------------------------------------ trainer.py:94-200
    batch_scenes = True      # test_from_dir: all scenes of a file advance together when the model has predict_batched

    def test_from_dir(self, data_path, model_path=None, sim_steps=1000, stepwise=True, rollout=True):
        if model_path:
            models = sorted(os.listdir(model_path), key=lambda x: int(x.split("_")[1].split(".")[0]))
            with torch.no_grad():
                self.model.load_state_dict(torch.load(f"{model_path}/{models[-1]}", map_location=self.device))
            print(f"Loaded model {models[-1]}")
        csv_files = [f.replace("\\", "/") for f in glob(data_path + "/*.csv")]
        df_stepwise = pd.DataFrame(columns=["filename", "scene", "step", "loss", "mse_loss", "step_time"])
        df_rollout = pd.DataFrame(columns=ROLLOUT_COLUMNS)
        self.last_rollout_modes = []       # per file: how its scenes were advanced
        if stepwise:
            for f in csv_files:
                loader = get_dataloader(csv_path=f, batch_size=1, k=self.model.neighbors, shuffle=False,
                                        device=self.device)
                df_stepwise = self.evaluate_stepwise(f.split("/")[-1], loader, df_stepwise)
        if rollout:
            for f in csv_files:
                loader = get_dataloader(csv_path=f, batch_size=sim_steps, k=self.model.neighbors, shuffle=False,
                                        device=self.device)
                scenes = list(loader)
                if self.batch_scenes and len(scenes) > 1 and hasattr(self.model, "predict_batched"):
                    df_rollout = self.evaluate_rollout_scenes(f.split("/")[-1], scenes, sim_steps, self.dt, df_rollout)
                    self.last_rollout_modes.append((f.split("/")[-1], f"{len(scenes)} scenes together"))
                    continue
                self.last_rollout_modes.append((f.split("/")[-1], "scene by scene"))
                for scene, data in enumerate(scenes):
                    df_rollout = self.evaluate_rollout(f.split("/")[-1], data, scene, sim_steps, self.dt, df_rollout)
        cols = ["x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az"]
        for col in cols:                                                           # trainer.py:177-178
            df_rollout[f"error_{col}"] = df_rollout[col].astype(float) - df_rollout[f"pred_{col}"].astype(float)
        df_rollout = df_rollout.groupby(["filename", "scene", "step"])[[f"error_{c}" for c in cols]].mean()
        for name, trio in (("pos", ["x", "y", "z"]), ("vel", ["vx", "vy", "vz"]), ("acc", ["ax", "ay", "az"])):
            errors = torch.tensor(df_rollout[[f"error_{c}" for c in trio]].values.astype(np.float64))
            df_rollout[f"{name}_rmse"] = torch.sqrt((errors ** 2).mean(dim=1)).numpy()   # trainer.py:186-195
        df_stepwise = df_stepwise.astype({"loss": float, "step_time": float}) if len(df_stepwise) else df_stepwise
        return (df_stepwise.groupby(["filename", "scene"]).mean(numeric_only=True)[["loss", "step_time"]],
                df_rollout[["pos_rmse", "vel_rmse", "acc_rmse"]])


def rollout_mse(df_rollout_rows: pd.DataFrame) -> pd.DataFrame:
    """True per-step MSE over particles x xyz for pos / vel / acc (BASELINE.json 'rollout MSE'), from
    the rows evaluate_rollout() returns. The reference's own statistic (mean signed error first,
    trainer.py:179-195) is what test_from_dir() reports."""
    out = {}
    for name, trio in (("pos", ["x", "y", "z"]), ("vel", ["vx", "vy", "vz"]), ("acc", ["ax", "ay", "az"])):
        err2 = sum((df_rollout_rows[c].astype(float) - df_rollout_rows[f"pred_{c}"].astype(float)) ** 2 for c in trio) / 3.0
        out[f"{name}_mse"] = err2.groupby([df_rollout_rows["filename"], df_rollout_rows["scene"], df_rollout_rows["step"]]).mean()
    return pd.DataFrame(out)
