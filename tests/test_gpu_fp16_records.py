"""GPU: the fp16-storage record format (gsaj_set_record_format(16); BASELINE config 5 "fp16 splat with fp32 Jacobian
accumulation").  Integer structure is untouched (the sort keys, lists, ranges, radii, R stay bit-exact against the oracle);
images and gradients are compared with the fp32 oracle at the looser tolerance half-precision conic / opacity / colour imply
(the reference has no fp16 path: this tolerance is this repository's own statement, not a reference fixture)."""
import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fp16_records():
    from gsaj import rasterizer as C

    C.set_record_format(16)
    yield
    C.set_record_format(32)


@pytest.mark.parametrize("name", ["p2000_160x120", "p6000_640x480_sh1"])
def test_fp16_records_parity(fp16_records, name):
    from gsaj import rasterizer as C
    from oracle import oracle as orc

    cam, sc, deg = hp.make(name)
    (ref, st), kw = hp.oracle_forward(cam, sc, deg)
    out, args = hp.gpu_forward(cam, sc, deg, kw=kw)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    P, W, H = sc["means3D"].shape[0], cam["W"], cam["H"]
    assert R == ref["num_rendered"]
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    # half-precision conic moves alpha by ~1e-3 relative: more pixels sit on the other side of a threshold than in fp32
    assert (dbg["n_contrib"].astype(np.int64) != st["n_contrib"].astype(np.int64)).mean() <= 2e-2
    hp.assert_image_close(color.cpu().numpy(), ref["color"], 3e-3, flip_fraction=2e-3, flip_bound=0.05)
    hp.assert_image_close(depth.cpu().numpy(), ref["depth"], 3e-3, flip_fraction=2e-3, flip_bound=0.05)
    dLc, dLd = hp.seeds(cam, seed=5)
    gref = orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
    g = hp.gpu_backward(cam, deg, out, args, dLc, dLd)
    names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot", "dL_dtau"]
    for nm, got in zip(names, g[:9]):
        want = gref[nm]
        if np.abs(want).max() == 0:
            continue
        assert hp.rel_err(got.cpu().numpy().reshape(want.shape), want) < 3e-2, (nm, hp.rel_err(got.cpu().numpy().reshape(want.shape), want))
    assert hp.rel_err(g[9].cpu().numpy(), gref["dL_dtau_sum"]) < 3e-2


def test_fp16_records_do_not_leak_into_fp32_mode(fp16_records):
    """The format is read per frame from the image workspace: switching back gives the fp32 bits again."""
    import torch
    from gsaj import rasterizer as C

    cam, sc, deg = hp.make("p2000_160x120")
    (ref, st), kw = hp.oracle_forward(cam, sc, deg)
    out16, _ = hp.gpu_forward(cam, sc, deg, kw=kw)
    C.set_record_format(32)
    out32a, _ = hp.gpu_forward(cam, sc, deg, kw=kw)
    out32b, _ = hp.gpu_forward(cam, sc, deg, kw=kw)
    assert torch.equal(out32a[1], out32b[1]) and not torch.equal(out16[1], out32a[1])
    hp.assert_image_close(out32a[1].cpu().numpy(), ref["color"], 2e-4)
