"""galaxify.galaxies (this repo's generators) against the inputs the REAL reference generated for
the golden vectors (tests/golden/make_golden.py: generate_spiral / generate_disk, seed 42, dataset-CLI
defaults). The goldens hold fp32 copies of positions/velocities and the float64 masses."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden

GAL = dict(total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=4.5e-6, black_hole_mass=0.01)


@pytest.mark.parametrize("name", [c for c in golden_cases() if "spiral" in c or "disk" in c])
def test_generators_reproduce_the_reference_galaxies(name):
    from galaxify import galaxies
    g = load_golden(name)
    n = g["pos"].shape[0]
    if "spiral" in name:
        p, v, m = galaxies.generate_spiral(n_bodies=n, n_arms=2, pitch_angle=-np.pi / 6, arm_strength=0.3, seed=42, **GAL)
    else:
        p, v, m = galaxies.generate_disk(n_bodies=n, seed=42, **GAL)
    assert p.dtype == np.float64 and p.shape == (n, 3) and v.shape == (n, 3) and m.shape == (n,)
    assert np.array_equal(p.astype(np.float32), g["pos"])                 # same RNG stream, same transforms
    assert np.allclose(m, g["mass64"], rtol=1e-13, atol=0)
    # velocities: the disc's enclosed mass is a prefix sum here and a masked sum there (last-bit differences)
    assert np.allclose(v.astype(np.float32), g["vel"], rtol=2e-6, atol=1e-12)
    assert abs(m.sum() - 1.0) < 1e-12 and m[0] == pytest.approx(0.01)


def test_disk_options_and_hernquist():
    from galaxify import galaxies
    base = dict(n_bodies=50, seed=1, **GAL)
    p0, v0, m0 = galaxies.generate_disk(**base)
    p1, v1, _ = galaxies.generate_disk(offset=(1, 2, 3), initial_vel=(0.1, 0, 0), clockwise=False, **base)
    assert np.allclose(p1 - p0, (1, 2, 3)) and np.allclose(v1[:, :2] + v0[:, :2], (0.1, 0)) and p0[0].tolist() == [0, 0, 0]
    p2, _, _ = galaxies.generate_disk(angle=(0, 0, np.pi / 2), **base)
    assert np.allclose(p2[:, 0], -p0[:, 1]) and np.allclose(p2[:, 1], p0[:, 0])
    rho = galaxies.spherical_hernquist_distribution(r=np.array([0.0, 1.0]), r0=1, total_mass=1)
    assert rho[1] == pytest.approx(1 / (2 * np.pi * 8)) and np.isfinite(rho[0])
    with pytest.raises(ValueError):
        galaxies.spherical_hernquist_distribution(r=np.array([0.0]), avoid_distance_zero=False)
    assert galaxies.BodyType.BLACK_HOLE.value == "black hole"
    assert galaxies.generate_spiral(n_bodies=1, seed=0, **GAL)[0].shape == (1, 3)


# ---- the same generators with the post-draw arithmetic on the MI355X (csrc/generators.hip)
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 25, 1024, 70001])
def test_disk_on_device_matches_host_generator(n, gpu_device):
    """generate_disk(device="cuda"): radix sort + prefix sum + lower-bound for the enclosed mass, fp64 kernels for
    the rest, against the host generator (itself seed-for-seed with the reference's goldens) -- same seed, same
    random stream, agreement to fp64 rounding (libm / summation order), incl. rotation, offsets and spin."""
    from galaxify import galaxies
    kw = dict(n_bodies=n, total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=4.5e-6, black_hole_mass=0.01,
              offset=(0.5, -1.0, 2.0), initial_vel=(1e-3, 0.0, -2e-3), clockwise=(n % 2 == 0), angle=(0.3, -1.1, 2.0), seed=42)
    ph, vh, mh = galaxies.generate_disk(**kw)
    pd_, vd, md = galaxies.generate_disk(device="cuda", **kw)
    assert pd_.dtype == torch.float64 and pd_.is_cuda and pd_.shape == (n, 3) and md.shape == (n,)
    for name, got, ref in (("pos", pd_, ph), ("vel", vd, vh), ("mass", md, mh)):
        got = got.cpu().numpy()
        scale = np.abs(ref).max() if ref.size else 1.0
        finite = np.isfinite(ref)
        assert (np.isfinite(got) == finite).all(), name                 # a star at the black hole's radius: same inf/nan
        assert np.abs(got[finite] - ref[finite]).max() <= 1e-11 * max(scale, 1e-300), name
    if n > 1:
        assert abs(float(md.sum()) - 1.0) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 64, 5000])
def test_spiral_on_device_matches_host_generator(n, gpu_device):
    from galaxify import galaxies
    kw = dict(n_bodies=n, total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=4.5e-6, black_hole_mass=0.01, seed=7)
    ph, vh, mh = galaxies.generate_spiral(**kw)
    pd_, vd, md = galaxies.generate_spiral(device="cuda", **kw)
    for name, got, ref in (("pos", pd_, ph), ("vel", vd, vh), ("mass", md, mh)):
        got = got.cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300), name


@pytest.mark.gpu
def test_device_galaxy_feeds_the_simulator(gpu_device):
    """Device-generated float64 tensors go straight into LeapFrogSimulator (fp32 copies, simulation.py:58-65) and
    give the golden accelerations of the host-generated galaxy."""
    from conftest import load_golden, row_rel
    from galaxify import galaxies, simulation
    g = load_golden("direct_disk_n1024")
    p, v, m = galaxies.generate_disk(n_bodies=1024, total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=4.5e-6,
                                     black_hole_mass=0.01, seed=42, device="cuda")
    assert np.array_equal(p.cpu().numpy().astype(np.float32), g["pos"])            # the golden inputs, bit for bit in fp32
    sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, g_const=float(g["g_const"]),
                                       softening=float(g["softening"]), dt=float(g["dt"]), calc_energy=False, device="cuda")
    assert row_rel(sim.accelerations.cpu().numpy(), g["acc0"]) < 1e-5
