"""GPU: the fp16-storage record format (GSAJ_FWD_RECORDS_FP16, a per-call flag; BASELINE config 5 "fp16 splat with fp32
Jacobian accumulation").  Integer structure is untouched (sort keys, lists, ranges, radii, R stay bit-exact against the
oracle).  Images and gradients are compared (a) with the oracle in ITS fp16-record mode -- conic / opacity / colour rounded
to half once, everything else fp32 -- at the same tolerances as the fp32 path, and (b) with the plain fp32 oracle at the
looser tolerance half-precision storage implies (the reference has no fp16 path: (b) is this repository's own statement of
how far config 5 may move from fp32, not a reference fixture).  Full size (10^6 Gaussians): tests/test_gpu_full_size.py.

Also here: the C ABI keeps no process-wide mode -- two host threads rendering fp16-record and fp32-record frames on two
HIP streams at the same time each get their own bits (SURVEY 8(b) "re-entrant per stream")."""
import threading

import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["p2000_160x120", "p6000_640x480_sh1"])
def test_fp16_records_parity(name):
    from gsaj import rasterizer as C
    from oracle import oracle as orc

    cam, sc, deg = hp.make(name)
    (ref, st), kw = hp.oracle_forward(cam, sc, deg, record_bits=16)
    (ref32, st32), _ = hp.oracle_forward(cam, sc, deg)
    out, args = hp.gpu_forward(cam, sc, deg, kw=kw, record_bits=16)
    R, color, radii, geom, binning, img, depth, opacity, n_touched = out
    P, W, H = sc["means3D"].shape[0], cam["W"], cam["H"]
    assert R == ref["num_rendered"] == ref32["num_rendered"]
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    dbg = {k: v.cpu().numpy() for k, v in C.debug_export(P, R, W, H, geom, binning, img).items()}
    np.testing.assert_array_equal(dbg["point_list"].astype(np.uint32), st["point_list"])
    np.testing.assert_array_equal(dbg["ranges"], st["ranges"])
    # (a) against the fp16-record oracle: fp32-path tolerances
    tag = "fp16/" + name
    hp.assert_counts_close(dbg["n_contrib"], st["n_contrib"], st, tag=tag)
    hp.assert_image_close(color.cpu().numpy(), ref["color"], hp.IMG_TOL, st=st, tag=tag + "/color")
    hp.assert_image_close(depth.cpu().numpy(), ref["depth"], hp.IMG_TOL, st=st, tag=tag + "/depth")
    dLc, dLd = hp.seeds(cam, seed=5)
    g, gref = hp.check_backward(cam, deg, out, args, st, dLc, dLd, tag)
    # (b) against the fp32 oracle: half-precision conic moves alpha by ~1e-3 relative
    assert (dbg["n_contrib"].astype(np.int64) != st32["n_contrib"].astype(np.int64)).mean() <= 2e-2
    hp.assert_image_close(color.cpu().numpy(), ref32["color"], 3e-3, flip_fraction=2e-3, flip_bound=0.05)
    g32 = orc.backward(st32, dLc, dLd, cam["projmatrix_raw"])
    for nm, got in zip(hp.GRAD_NAMES, g):
        want = g32[nm]
        if got is None or want.size == 0 or np.abs(want).max() == 0:
            continue
        assert hp.rel_err(got.cpu().numpy().reshape(want.shape), want) < 3e-2, nm


def test_record_format_is_per_call_not_process_wide():
    """The format travels with the call and is latched in that frame's image workspace: interleaving gives each frame its own
    bits, and a backward started after ANOTHER frame's forward of the other format still reads its own frame's records."""
    import torch

    cam, sc, deg = hp.make("p2000_160x120")
    (ref, st), kw = hp.oracle_forward(cam, sc, deg)
    out16, a16 = hp.gpu_forward(cam, sc, deg, kw=kw, record_bits=16)
    out32a, a32 = hp.gpu_forward(cam, sc, deg, kw=kw)
    out16b, _ = hp.gpu_forward(cam, sc, deg, kw=kw, record_bits=16)
    out32b, _ = hp.gpu_forward(cam, sc, deg, kw=kw)
    assert torch.equal(out32a[1], out32b[1]) and torch.equal(out16[1], out16b[1]) and not torch.equal(out16[1], out32a[1])
    hp.assert_image_close(out32a[1].cpu().numpy(), ref["color"], hp.IMG_TOL, st=st)
    dLc, dLd = hp.seeds(cam, seed=6)
    g16 = hp.gpu_backward(cam, deg, out16, a16, dLc, dLd)    # frame 1's backward, three other forwards later
    g16b = hp.gpu_backward(cam, deg, out16b, a16, dLc, dLd)
    g32 = hp.gpu_backward(cam, deg, out32a, a32, dLc, dLd)
    assert torch.equal(g16[9], g16b[9]) and not torch.equal(g16[9], g32[9])


def test_two_threads_two_streams_two_formats():
    """Frontend and backend of a SLAM process share the library: thread A renders fp16-record frames on its own HIP stream while
    thread B renders fp32-record frames on another, both with the per-stage profiler armed.  Each thread must get, every time,
    exactly the bits a single-threaded run of its format gives."""
    import torch
    from gsaj.rasterizer import FrameContext, profile_stages

    cam, sc, deg = hp.make("p6000_640x480_sh1")
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    P, M = sc["means3D"].shape[0], sc["shs"].shape[1]
    fa = dict(bg=torch.zeros(3, device=dev), means3D=t(sc["means3D"]), opacities=t(sc["opacities"]), viewmatrix=t(cam["viewmatrix"]),
              projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=deg,
              shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    dLc, dLd = hp.seeds(cam, seed=9)
    ba = dict(bg=fa["bg"], means3D=fa["means3D"], viewmatrix=fa["viewmatrix"], projmatrix=fa["projmatrix"],
              projmatrix_raw=t(cam["projmatrix_raw"]), campos=fa["campos"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
              dL_dcolor=t(dLc), dL_ddepth=t(dLd), sh_degree=deg, shs=fa["shs"], scales=fa["scales"], rotations=fa["rotations"])

    def run(bits, stream, n, out):
        ctx = FrameContext(P, cam["W"], cam["H"], M, dev, record_bits=bits)
        res = []
        with torch.cuda.stream(stream):
            for i in range(n):
                ctx.forward(**fa, sync=(i == 0))
                g = ctx.backward(**ba)
                res.append((ctx.color.clone(), ctx.bucket.clone(), g["tau_sum"].clone()))
            stream.synchronize()
            ctx.status()
        out[bits] = res

    single = {}
    run(16, torch.cuda.current_stream(dev), 1, single)
    run(32, torch.cuda.current_stream(dev), 1, single)
    assert not torch.equal(single[16][0][0], single[32][0][0])
    torch.cuda.synchronize()
    both, errs = {}, []

    def guarded(bits, stream):
        try:
            run(bits, stream, 25, both)
        except Exception as ex:  # noqa: BLE001
            errs.append((bits, ex))

    sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with profile_stages(max_records=4096) as prof:
        th = [threading.Thread(target=guarded, args=(16, sA)), threading.Thread(target=guarded, args=(32, sB))]
        for x in th:
            x.start()
        for x in th:
            x.join()
    assert not errs, errs
    for bits in (16, 32):
        for color, bucket, tau in both[bits]:
            assert torch.equal(color, single[bits][0][0]) and torch.equal(bucket, single[bits][0][1]) and torch.equal(tau, single[bits][0][2])
    # the profiler saw both threads' launches, none lost or double-closed: 50 frames x one launch of each stage
    assert prof.launches["render_fwd"] == 50 and prof.launches["render_bwd"] == 50 and prof.launches["gaussian_bwd"] == 50
    assert all(v >= 0.0 for v in prof.ms.values())
