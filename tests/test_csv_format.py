"""The dataset CSV writer: nbd_format_f32 reproduces str(np.float32(x)) -- what csv.DictWriter prints
for the nine state columns the reference hands it as numpy.float32 scalars (s01-dataset-generation.py:218-241) --
and write_states() produces, byte for byte, the file csv.DictWriter produces from the same states. Host code only:
these run without a GPU. tools/check_f32_format_exhaustive.py covers all 2^32 patterns (profiles/)."""
import csv
import importlib.util
import io

import numpy as np
import pytest
import torch

from conftest import PKG


def _fmt(a):
    from nbd import _lib
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.zeros(a.size, dtype="S24")
    assert _lib.lib().nbd_format_f32_array(a.ctypes.data, a.size, out.ctypes.data, 24) == 0
    return out.astype(str)


def _same(a):
    got, ref = _fmt(a), a.astype(str)
    bad = np.nonzero(got != ref)[0]
    assert bad.size == 0, [(a[i:i + 1].view(np.uint32)[0], got[i], ref[i]) for i in bad[:5]]


def test_special_values_and_notation_switches():
    a = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 1.0, -1.0, 0.1, 1e-4, 9.9999e-5, 1.0001e-4, 1e-5, 1e16,
                  9.9999999e15, 1e15, 123456789.0, 16777216.0, 16777218.0, 33554448.0, 8589973000.0, 3.4028235e38,
                  1e-45, 1.17549435e-38, 1.1754942e-38, 9.5367431640625e-07, 2.0 ** -24, 0.3, 2.5, 1e22, 1e-10, 5e-324],
                 dtype=np.float32)
    _same(a)


def test_every_exponent_at_its_boundaries():
    """For each of the 255 finite exponents: the power of two (the rounding interval is asymmetric there), its
    neighbours, and patterns whose interval bounds are short decimals; both signs."""
    ex = np.arange(0, 255, dtype=np.uint32) << 23
    mant = np.array([0, 1, 2, 3, 0x400000, 0x7ffffe, 0x7fffff, 0x200000, 0x19999a, 0x4ccccd, 0x123456], dtype=np.uint32)
    bits = (ex[:, None] | mant[None, :]).ravel()
    bits = bits[bits != 0]
    _same(bits.view(np.float32))
    _same((bits | np.uint32(1 << 31)).view(np.float32))


@pytest.mark.parametrize("seed", [0, 1])
def test_random_bit_patterns_and_dataset_like_values(seed):
    rng = np.random.default_rng(seed)
    _same(rng.integers(0, 2 ** 32, 300_000, dtype=np.uint64).astype(np.uint32).view(np.float32))
    _same((rng.standard_normal(300_000) * 10.0 ** rng.integers(-9, 4, 300_000)).astype(np.float32))
    # integers and short decimals: trailing zeros, ".0", ties
    _same(rng.integers(-2 ** 26, 2 ** 26, 100_000).astype(np.float32))
    _same((rng.integers(-99999, 99999, 100_000) / 10.0 ** rng.integers(0, 9, 100_000)).astype(np.float32))


def test_single_value_entry_point_and_argument_checks():
    import ctypes
    from nbd import _lib
    L = _lib.lib()
    buf = ctypes.create_string_buffer(24)
    for x in (0.1, -1.5e-5, 123456.7):
        n = L.nbd_format_f32(x, buf)
        assert buf.raw[:n].decode() == str(np.float32(x))
    assert L.nbd_format_f32(1.0, None) < 0
    a = np.ones(4, dtype=np.float32); out = np.zeros(4, dtype="S16")
    assert L.nbd_format_f32_array(a.ctypes.data, 4, out.ctypes.data, 16) != 0          # slot too small
    assert L.nbd_csv_format_state(buf, 24, b"", 0, b"1.0", np.array([0, 3], dtype=np.int32).ctypes.data, a.ctypes.data,
                                  a.ctypes.data, a.ctypes.data, 1, b"", 0) == -1       # cap below the bound


def _reference_style_csv(scenes, fieldnames):
    """What the reference's loop writes (s01-dataset-generation.py:108-125, 218-241): csv.DictWriter, one dict per
    particle per state, values = python int / str / float, numpy.float64 mass, numpy.float32 state components."""
    f = io.StringIO(newline="")
    w = csv.DictWriter(f, fieldnames=fieldnames)
    w.writeheader()
    for scene_id, (scene_type, states, masses) in enumerate(scenes):
        for st in states:
            p, v, a = st.positions.numpy(), st.velocities.numpy(), st.accelerations.numpy()
            for i in range(p.shape[0]):
                w.writerow({"scene": scene_id, "scene_type": scene_type, "step": st.step, "step_time": st.step_time,
                            "mass": masses[i], "x": p[i, 0], "y": p[i, 1], "z": p[i, 2], "vx": v[i, 0], "vy": v[i, 1],
                            "vz": v[i, 2], "ax": a[i, 0], "ay": a[i, 1], "az": a[i, 2], "u": st.u_energy,
                            "k": st.k_energy})
    return f.getvalue().encode()


def test_write_states_equals_csv_dictwriter_byte_for_byte():
    from galaxify.simulation import SimulationState
    spec = importlib.util.spec_from_file_location("s01", f"{PKG}/s01-dataset-generation.py")
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    rng = np.random.default_rng(5)
    scenes = []
    for scene_type, n, steps, energies in (("spiral", 7, 4, True), ("disk", 1, 3, True), ("plummer", 33, 2, False),
                                           ("disk", 0, 2, True)):
        masses = rng.random(n) * 10.0 ** rng.integers(-6, 2, n)
        states = []
        for s in range(steps):
            t = lambda scale: torch.tensor(rng.standard_normal((n, 3)) * 10.0 ** rng.integers(-8, 5, (n, 3)) * scale,
                                           dtype=torch.float32)
            states.append(SimulationState(step=s, step_time=float(rng.random() * 1e-4), positions=t(1.0),
                                          velocities=t(1e-2), accelerations=t(1e-6),
                                          u_energy=float(-rng.random() * 1e-7) if energies else None,
                                          k_energy=float(rng.random() * 1e-9) if energies else None))
        if n:
            states[0].positions[0] = torch.tensor([0.0, -0.0, 1e-4])
        scenes.append((scene_type, states, masses))
    f = io.BytesIO()
    f.write((",".join(cli.FIELDNAMES) + "\r\n").encode())
    for scene_id, (scene_type, states, masses) in enumerate(scenes):
        cli.write_states(f, scene_id, scene_type, states, masses)
    assert f.getvalue() == _reference_style_csv(scenes, cli.FIELDNAMES)
