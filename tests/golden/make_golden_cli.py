"""Golden dataset CSV: the REFERENCE's own CLI (src/s01-dataset-generation.py, CPU) run on a tiny case.
The file it writes is the wire format (s01-dataset-generation.py:108-125, 218-241) that datautils /
Trainer.test_from_dir consume; tests/test_direct_gpu.py runs this build's CLI with the same arguments on
the GPU and compares header, row layout, value formatting and values (step_time is a wall-clock reading).

Run in the build container:  python tests/golden/make_golden_cli.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ARGS = ["--integrator", "leapfrog", "--n-bodies", "5", "8", "--sim-type", "spiral", "--steps", "3", "--seed", "42"]

if __name__ == "__main__":
    out = os.path.join(HERE, "ref_cli_spiral_n5_n8.csv")
    subprocess.run([sys.executable, "/root/reference/src/s01-dataset-generation.py", *ARGS, "--device", "cpu",
                    "--output", out], check=True, cwd="/tmp")
    print(out, sum(1 for _ in open(out)), "lines")
