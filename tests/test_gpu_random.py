"""GPU: randomized parity sweep of the tiled path against the oracle -- odd image sizes, Gaussian counts
that are not multiples of the workgroup sizes, screen-filling and sub-pixel Gaussians, opacity extremes,
points around the near plane, similarity-transform cameras; and the 60-seed fuzz sweep (random scene, camera,
SH degree, background per seed).  Full-size configs: tests/test_gpu_full_size.py."""
import math

import numpy as np
import pytest

import helpers as hp
from gsaj import synthetic as syn

pytestmark = pytest.mark.gpu

CASES_LARGE_IMAGE = [(3000, 2064, 2000, 9)]  # 129 x 125 = 16 125 tiles > LDS_TILES_MAX (8192): global-atomic binning path


@pytest.mark.parametrize("case", CASES_LARGE_IMAGE, ids=["P3000_2064x2000"])
def test_more_tiles_than_the_lds_histogram_holds(case):
    from gsaj import rasterizer as C
    from oracle import oracle as orc

    P, W, H, seed = case
    cam = hp.small_camera(W, H, f=0.9 * W, orthonormal=True)
    sc = syn.make_scene(P, seed, cam, z_range=(1.0, 4.0), log_scale_range=(math.log(0.004), math.log(0.05)), sh_coeffs=4, margin=0.1)
    (ref, st), kw = hp.oracle_forward(cam, sc, 1)
    out, args = hp.gpu_forward(cam, sc, 1, kw=kw)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    assert R == ref["num_rendered"]
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    hp.assert_image_close(color.cpu().numpy(), ref["color"], hp.IMG_TOL, st=st, tag="large_image/color")
    dLc, dLd = hp.seeds(cam, seed=seed)
    g, gref = hp.check_backward(cam, 1, out, args, st, dLc, dLd, "large_image")


CASES = [
    # P, W, H, seed, deg, coeffs, z_range, log_scale_range, opacity_range, orthonormal
    (1, 33, 17, 1, 0, 1, (1.0, 1.5), (math.log(0.05), math.log(0.1)), (0.5, 0.9), True),
    (63, 47, 31, 2, 1, 4, (0.5, 2.0), (math.log(0.02), math.log(0.3)), (0.01, 0.99), False),
    (65, 16, 16, 3, 3, 16, (1.0, 2.0), (math.log(0.5), math.log(2.0)), (0.2, 0.6), True),      # screen-filling
    (257, 129, 65, 4, 2, 9, (0.15, 1.0), (math.log(0.002), math.log(0.02)), (0.3, 0.95), False),  # near plane, tiny
    (1000, 200, 150, 5, 3, 16, (1.0, 8.0), (math.log(0.003), math.log(0.6)), (0.004, 1.0), True),  # wide mix
    (4097, 96, 96, 6, 0, 1, (1.0, 3.0), (math.log(0.01), math.log(0.05)), (0.9, 0.999), True),   # near-opaque
]


@pytest.mark.parametrize("case", CASES, ids=[f"P{c[0]}_{c[1]}x{c[2]}" for c in CASES])
def test_random_scene_parity(case):
    from gsaj import rasterizer as C
    from oracle import oracle as orc

    P, W, H, seed, deg, coeffs, zr, ls, orng, ortho = case
    cam = hp.small_camera(W, H, f=0.8 * W, orthonormal=ortho)
    sc = syn.make_scene(P, seed, cam, z_range=zr, log_scale_range=ls, opacity_range=orng, sh_coeffs=coeffs, margin=0.2)
    bg = (0.3, 0.1, 0.7)
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg)
    out, args = hp.gpu_forward(cam, sc, deg, bg=bg, kw=kw)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    assert R == ref["num_rendered"]
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    tag = "random/P%d_%dx%d" % (P, W, H)
    hp.assert_counts_close(dbg["n_contrib"], st["n_contrib"], st, tag=tag)
    hp.assert_image_close(color.cpu().numpy(), ref["color"], hp.IMG_TOL, st=st, tag=tag + "/color")
    hp.assert_image_close(depth.cpu().numpy(), ref["depth"], hp.IMG_TOL, st=st, tag=tag + "/depth")
    dLc, dLd = hp.seeds(cam, seed=seed)
    g, gref = hp.check_backward(cam, deg, out, args, st, dLc, dLd, tag)


def _fuzz_case(seed):
    rng = np.random.default_rng(seed)
    P = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 3000, 8000]))
    W, H = int(rng.integers(16, 400)), int(rng.integers(16, 300))
    coeffs = int(rng.choice([1, 4, 9, 16]))
    deg = int(rng.integers(0, int(round(math.sqrt(coeffs)))))
    zlo = float(rng.uniform(0.15, 1.5))
    zr = (zlo, zlo + float(rng.uniform(0.2, 6.0)))
    lo = float(rng.uniform(math.log(0.002), math.log(0.05)))
    ls = (lo, lo + float(rng.uniform(0.1, 3.0)))
    olo = float(rng.uniform(0.003, 0.6))
    orng = (olo, min(1.0, olo + float(rng.uniform(0.05, 0.6))))
    cam = hp.small_camera(W, H, f=float(rng.uniform(0.5, 1.5)) * W, orthonormal=bool(rng.integers(0, 2)))
    sc = syn.make_scene(P, seed, cam, z_range=zr, log_scale_range=ls, opacity_range=orng, sh_coeffs=coeffs,
                        margin=float(rng.uniform(0.0, 0.4)))
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3))
    bits = 16 if seed % 5 == 4 else 32  # every fifth case runs the fp16-storage records against the oracle's fp16 mode
    return P, W, H, deg, cam, sc, bg, bits


def _fuzz_seeds():
    """The suite's 64 seeds; GSAJ_FUZZ_RANGE=lo:hi runs another range (a soak run after a change to the binning kernels)."""
    import os

    lo, hi = (int(x) for x in os.environ.get("GSAJ_FUZZ_RANGE", "5000:5064").split(":"))
    return range(lo, hi)


@pytest.mark.parametrize("seed", _fuzz_seeds())
def test_fuzz_parity(seed):
    """tools/fuzz_parity.py as a test: one random scene / camera / option set per seed (64 seeds; 5052 is the seed whose
    single borderline pixel once widened the image budget -- that pixel is now verified to BE borderline)."""
    from gsaj import rasterizer as C
    from oracle import oracle as orc

    P, W, H, deg, cam, sc, bg, bits = _fuzz_case(seed)
    tag = "fuzz/%d" % seed
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, bg=bg, record_bits=bits)
    out, args = hp.gpu_forward(cam, sc, deg, bg=bg, kw=kw, record_bits=bits)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    assert R == ref["num_rendered"]
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    hp.assert_counts_close(dbg["n_contrib"], st["n_contrib"], st, tag=tag, flip_fraction=5e-4)
    hp.assert_image_close(color.cpu().numpy(), ref["color"], hp.IMG_TOL, st=st, tag=tag + "/color")
    hp.assert_image_close(depth.cpu().numpy(), ref["depth"], hp.IMG_TOL, st=st, tag=tag + "/depth")
    dLc, dLd = hp.seeds(cam, seed=seed)
    g, gref = hp.check_backward(cam, deg, out, args, st, dLc, dLd, tag)


def test_equal_depths_keep_index_order():
    """Every Gaussian four times (ids i, i + 300, i + 600, i + 900: identical depth bits): the reference's stable radix sort of
    (tile | depth) keys leaves equal keys in emission order, i.e. ascending Gaussian index (rasterizer_impl.cu:353-368, 70-111);
    here the index is the low word of the sort key.  point_list must match the oracle bit for bit, ties included."""
    from gsaj import rasterizer as C

    W, H, deg = 150, 97, 1
    cam = hp.small_camera(W, H, f=0.9 * W, orthonormal=True)
    sc1 = syn.make_scene(300, 77, cam, z_range=(0.8, 3.0), log_scale_range=(math.log(0.01), math.log(0.15)), opacity_range=(0.05, 0.6),
                         sh_coeffs=4, margin=0.2)
    sc = {k: (np.concatenate([v] * 4, axis=0) if isinstance(v, np.ndarray) and v.shape[:1] == (300,) else v) for k, v in sc1.items()}
    P = 1200
    (ref, st), kw = hp.oracle_forward(cam, sc, deg)
    out, args = hp.gpu_forward(cam, sc, deg, kw=kw)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    assert R == ref["num_rendered"] and R > 0
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    pl = dbg["point_list"].astype(np.uint32)
    np.testing.assert_array_equal(pl, st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    # (the property itself, independent of the oracle: inside a tile's list the four copies of a Gaussian are adjacent and ascending)
    beg, end = (int(x) for x in dbg["ranges"][np.argmax(dbg["ranges"][:, 1] - dbg["ranges"][:, 0])])
    run = pl[beg:end].astype(np.int64)
    assert (end - beg) % 4 == 0 and np.all(run[0::4] % 300 == run[3::4] % 300) and np.all(np.diff(run.reshape(-1, 4), axis=1) == 300)
    hp.assert_image_close(color.cpu().numpy(), ref["color"], hp.IMG_TOL, st=st, tag="ties/color")
    dLc, dLd = hp.seeds(cam, seed=77)
    hp.check_backward(cam, deg, out, args, st, dLc, dLd, "ties")
