"""Hardware sweep of the force kernel's launch geometry (nbd_accel_tuned_f32): slab count x register variant
for the shapes of the range-sharded step. Writes one JSON line per point; the library's plan_chunks() was
fitted to this table (profiles/r02_plan_sweep.json).  python tools/sweep_accel_plan.py [out.jsonl]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import torch
from nbd import direct
from nbd.plummer import generate_plummer


def time_ms(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return min(ts)


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    n = 65536
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p, dtype=torch.float32, device="cuda")
    mass = torch.tensor(m, dtype=torch.float32, device="cuda")
    posm = direct.pack_posm(pos, mass)
    eps2 = direct.f32(0.01)
    # clock ramp
    ws_full = direct.step_workspace(n, "cuda")
    t_end = torch.cuda.Event(enable_timing=True)
    for _ in range(300):
        direct.accel(posm, n, posm, n, 0, eps2, 1.0, workspace=ws_full)
    torch.cuda.synchronize()
    shapes = [("rank_of_8_all", 8192, (0, 0)), ("rank_of_8_remote", 8192, (3 * 8192, 4 * 8192)),
              ("rank_of_8_local", 8192, None), ("rank_of_4_all", 16384, (0, 0)),
              ("rank_of_4_remote", 16384, (16384, 32768)), ("rank_of_2_all", 32768, (0, 0)),
              ("rank_of_2_remote", 32768, (32768, 65536)), ("single", 65536, (0, 0))]
    for name, n_tgt, ex in shapes:
        lo = 3 * 8192 if n_tgt == 8192 else (16384 if n_tgt == 16384 else (32768 if n_tgt == 32768 else 0))
        tgt = posm[lo:]
        if ex is None:                       # own bodies only
            src = direct.pack_posm(pos[lo:lo + n_tgt].contiguous(), mass[lo:lo + n_tgt].contiguous())
            n_src, ex, tgt_off, tgt = n_tgt, (0, 0), 0, src
        else:
            src, n_src, tgt_off = posm, n, lo
        pairs = float(n_tgt) * (n_src - (ex[1] - ex[0]))
        ws = direct.alloc_bytes(64 * n_tgt * 12, "cuda")
        acc = torch.empty((n_tgt, 3), device="cuda")
        chunks = (n_src - (ex[1] - ex[0])) // 64
        for variant in (0, 1):
            for slabs in range(1, 65):
                if slabs * 4 > chunks:
                    break
                ms = time_ms(lambda: direct.accel_tuned(src, n_src, tgt, n_tgt, tgt_off, eps2, 1.0, slabs, variant,
                                                        exclude=ex, out=acc, workspace=ws))
                rec = {"shape": name, "n_tgt": n_tgt, "n_src_walked": int(n_src - (ex[1] - ex[0])),
                       "variant": variant, "slabs": slabs, "workgroups": ((n_tgt + 127) // 128) * slabs,
                       "chunks_per_wave": -(-chunks // (slabs * 4)), "ms": ms, "pairs_per_s": pairs / (ms * 1e-3)}
                out.write(json.dumps(rec) + "\n")
                out.flush()
        # the library's own plan for this shape
        if name.endswith("_all") or name == "single":
            ms = time_ms(lambda: direct.accel(src, n_src, tgt, n_tgt, tgt_off, eps2, 1.0, out=acc, workspace=ws))
            out.write(json.dumps({"shape": name, "library_plan": direct.accel_plan(n_src, n_tgt), "ms": ms,
                                  "pairs_per_s": pairs / (ms * 1e-3)}) + "\n")
            out.flush()


if __name__ == "__main__":
    main()
