"""The C-ABI shared library loads on a machine WITHOUT a GPU and exports every symbol that
include/*.h declares; the ctypes signature table covers the same set. No compute calls here."""
import ctypes
import glob
import os
import re
import sys

import pytest

from conftest import ROOT
from nbd import _lib


def declared_symbols():
    names = []
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(hdr).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(nbd_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_something():
    assert len(declared_symbols()) >= 10


def test_library_builds_and_loads():
    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    assert _lib.lib().nbd_abi_version() == 1


@pytest.mark.parametrize("sym", declared_symbols())
def test_symbol_exported(sym):
    handle = ctypes.CDLL(_lib.LIB_PATH)
    assert hasattr(handle, sym), f"{sym} declared in include/ but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_host_only_queries():
    L = _lib.lib()
    assert L.nbd_posm_padded_len(0) == 0
    assert L.nbd_posm_padded_len(1) == 64
    assert L.nbd_posm_padded_len(64) == 64
    assert L.nbd_posm_padded_len(65) == 128
    assert L.nbd_strerror(0) == b"ok"
    assert b"workspace" in L.nbd_strerror(-2)
    g, s, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert L.nbd_accel_plan(65536, 65536, g, s, c) == 0
    # 512 target groups x 16 slabs = 8192 workgroups (6.4 residency rounds of 5 per CU), 16 chunks per wave
    assert (g.value, s.value, c.value) == (512, 16, 16)
    assert L.nbd_accel_plan(0, 5, None, None, None) == -1
    assert L.nbd_step_workspace_bytes(65536) == 16 * 65536 * 12
    for n_src, n_tgt in [(3, 3), (1000, 1000), (65536, 8192), (524288, 65536), (100, 7), (65536, 16384), (8192, 8192)]:
        assert L.nbd_accel_plan(n_src, n_tgt, g, s, c) == 0
        chunks = (n_src + 63) // 64
        assert s.value * 4 * c.value >= chunks          # every chunk is covered
        assert 1 <= s.value <= 64 and (c.value - 1) * s.value * 4 < chunks      # ... and no wave is given more than its share
        assert g.value * 128 >= n_tgt
    # the range-sharded split: one of 8 ranks at N = 65 536 (own block + remote block), ragged shards, a lone rank
    a, b, cc, d = (ctypes.c_int() for _ in range(4))
    assert L.nbd_shard_plan(65536, 3 * 8192, 8192, a, b, cc, d) == 0
    assert a.value * 4 * b.value >= 128 and cc.value * 4 * d.value >= 896 and 16 <= cc.value <= 64
    assert L.nbd_shard_workspace_bytes(65536, 3 * 8192, 8192) == (a.value + cc.value) * 8192 * 12
    assert L.nbd_shard_plan(1000, 0, 1000, a, b, cc, d) == 0 and cc.value == 0          # nothing remote
    assert L.nbd_shard_plan(1000, 990, 20, a, b, cc, d) == -1                           # range outside the system
    assert L.nbd_contconv_fused_supported(128, 128, 160) == 1 and L.nbd_contconv_fused_supported(70, 40, 27) == 0
    assert L.nbd_contconv_fused_supported(128, 128, 216) == 0                            # pair-list LDS tables: <= 160 cells


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    L = _lib.lib()
    assert L.nbd_pack_posm_f32(None, None, -1, None, None) == -1
    assert L.nbd_pack_posm_f32(None, None, 5, None, None) == -1
    assert L.nbd_accel_f32(None, -1, None, 1, 0, 0.01, 1.0, None, None, 0, None) == -1
    assert L.nbd_accel_f32(None, 4, None, 4, 0, 0.01, 1.0, None, None, 0, None) == -1
    assert L.nbd_kick_f32(None, None, 3, 0.1, None) == -1
    assert L.nbd_leapfrog_step_f32(None, None, None, None, None, 8, 0.1, 0.1, 0.01, 1.0, None, None, 0, None) == -1
    assert L.nbd_energy_f32(None, None, 4, 0.1, 1.0, None, None, 0, None) == -1
    # n == 0 is a no-op, not an error
    assert L.nbd_pack_posm_f32(None, None, 0, None, None) == 0
    assert L.nbd_kick_f32(None, None, 0, 0.1, None) == 0


def test_no_cpu_path():
    from galaxify import simulation
    import numpy as np
    import torch
    z = np.zeros((4, 3))
    with pytest.raises(ValueError):
        simulation.LeapFrogSimulator(positions=z, velocities=z, masses=np.ones(4), device="tpu")
    with pytest.raises(RuntimeError):
        simulation.LeapFrogSimulator(positions=z, velocities=z, masses=np.ones(4), device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            simulation.LeapFrogSimulator(positions=z, velocities=z, masses=np.ones(4))


def test_fused_contconv_prefetch_survives_the_compiler():
    """The fused ContinuousConv kernel keeps the next cell's filter fragment in flight behind a hand-counted
    `s_waitcnt vmcnt(8)` issued from inline asm, which hipcc's own wait insertion does not track: nothing may touch
    the fragment registers between a load and its wait, and the kernel must stay within 128 VGPRs / its two known
    spills (tools/check_contconv_isa.py disassembles the gfx950 code and checks exactly that; no GPU needed)."""
    import shutil
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_contconv_isa
    ok, report = check_contconv_isa.main()
    assert ok, report
