"""Wall time of the dataset CLI (s01-dataset-generation.py) on the reference's own sizes: scenes of 100..500
bodies x 1000 steps, energies on (the CLI hard-codes them).   python tools/cli_timing.py"""
import importlib.util, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nbody-deep-sim_amd")
sys.path[:0] = [PKG, ROOT]
import torch
spec = importlib.util.spec_from_file_location("s01", f"{PKG}/s01-dataset-generation.py")
cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
out = {}
with tempfile.TemporaryDirectory() as tmp:
    for label, env in (("captured_chunks", "1"), ("eager", "0")):
        os.environ["NBD_RUN_GRAPH"] = env
        args = ["--integrator", "leapfrog", "--n-bodies", "100", "300", "500", "--sim-type", "spiral", "--steps", "1000",
                "--dt", "1e-4", "--g", "4.5e-6", "--softening", "0.05", "--seed", "3", "--output", f"{tmp}/{label}.csv", "--device", "cuda"]
        cli.main(args[:-6] + ["--seed", "4", "--output", f"{tmp}/warm.csv", "--device", "cuda"]) if label == "captured_chunks" else None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cli.main(args)
        torch.cuda.synchronize()
        out[label + "_seconds_3_scenes_x_1000_steps"] = time.perf_counter() - t0
        out[label + "_csv_bytes"] = os.path.getsize(f"{tmp}/{label}.csv")
print(json.dumps(out))
