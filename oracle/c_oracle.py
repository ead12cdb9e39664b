"""ctypes loader for oracle/libnbody_oracle.so (C restatement, nbody_oracle.c). TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libnbody_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _lib = ctypes.CDLL(_PATH)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def acc_f32(pos, mass, g, softening, rows=None):
    pos = np.ascontiguousarray(pos, dtype=np.float32); mass = np.ascontiguousarray(mass, dtype=np.float32)
    n = pos.shape[0]
    lo, hi = (0, n) if rows is None else rows
    out = np.empty((hi - lo, 3), dtype=np.float32)
    lib().nbody_oracle_acc_f32(_ptr(pos), _ptr(mass), n, lo, hi, ctypes.c_float(g),
                               ctypes.c_float(np.float32(softening ** 2)), _ptr(out))
    return out


def acc_f64(pos, mass, g, softening, rows=None):
    """fp64 evaluation from the fp32-rounded inputs; eps^2 is the fp32 scalar the reference uses."""
    pos = np.ascontiguousarray(pos, dtype=np.float32); mass = np.ascontiguousarray(mass, dtype=np.float32)
    n = pos.shape[0]
    lo, hi = (0, n) if rows is None else rows
    out = np.empty((hi - lo, 3), dtype=np.float64)
    lib().nbody_oracle_acc_f64(_ptr(pos), _ptr(mass), n, lo, hi, ctypes.c_double(g),
                               ctypes.c_double(float(np.float32(softening ** 2))), _ptr(out))
    return out
