"""nbd_format_f32 against str(np.float32(x)) on EVERY non-negative fp32 bit pattern (the sign only adds '-').
About 55 core-minutes (numpy's formatter is the slow side); no GPU needed.
    python tools/check_f32_format_exhaustive.py [--procs 8] [--out profiles/r02_f32_format_exhaustive.json]"""
import argparse, json, os, sys, time
from multiprocessing import Pool
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-deep-sim_amd"))
CHUNK = 1 << 22


def check(c):
    from nbd import _lib
    L = _lib.lib()
    a = np.arange(c * CHUNK, (c + 1) * CHUNK, dtype=np.uint32).view(np.float32)
    out = np.zeros(a.size, dtype="S24")
    assert L.nbd_format_f32_array(a.ctypes.data, a.size, out.ctypes.data, 24) == 0
    bad = np.nonzero(out.astype(str) != a.astype(str))[0]
    return c, int(bad.size), [(int(a[i:i + 1].view(np.uint32)[0]), out[i].decode(), str(a[i])) for i in bad[:5]]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_f32_format_exhaustive.json"))
    args = ap.parse_args()
    n_chunks = (1 << 31) // CHUNK
    t0 = time.time(); total = 0; examples = []
    with Pool(args.procs) as pool:
        for done, (c, nbad, ex) in enumerate(pool.imap_unordered(check, range(n_chunks))):
            total += nbad; examples += ex
            if done % 32 == 0:
                print(f"{done}/{n_chunks} chunks, mismatches so far {total}, {time.time() - t0:.0f} s", flush=True)
    res = {"patterns_checked": n_chunks * CHUNK, "mismatches": total, "examples": examples[:20], "seconds": time.time() - t0,
           "numpy": np.__version__, "what": "nbd_format_f32 == str(np.float32(x)) for every bit pattern 0 .. 2^31-1 (incl. inf, nan)"}
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res))
