"""Torch-tensor front end of the C ABI: the three functions the reference's pybind module
exports (submodules/diff-gaussian-rasterization/ext.cpp:15-19) with the same argument
order and return tuples (rasterize_points.cu:36-246), backed by libgsaj_hip.so.

PyTorch only provides device memory and the current HIP stream here; every computation
happens in the hand-written kernels.
"""
import ctypes
import os

import torch

from . import _lib

_F32 = torch.float32
# tests flip this (or set the variable) to push every tile list longer than 128 entries through the chunk + merge path of the
# tile sort (the path a list longer than the LDS capacity takes)
FORCE_CHUNKED_SORT = bool(int(__import__("os").environ.get("GSAJ_FORCE_CHUNKED_SORT", "0")))
SORT_CAP = 16384  # longest tile list the per-tile sort handles in ONE LDS pass (csrc/gsaj_common.h); longer: chunks + merge passes


def _ptr(t):
    """Device pointer of a tensor, NULL for None / empty tensors (the reference passes empty
    tensors for absent optionals, diff_gaussian_rasterization/__init__.py:229-243)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _prep(t, device, name):
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        t = t.to(device)
    if t.dtype != _F32:
        t = t.to(_F32)
    return t.contiguous()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (HIP device): libgsaj_hip has no CPU path" % what)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height, image_width, sh,
                        degree, campos, prefiltered, debug, record_bits=32):
    """-> (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth, opacity, n_touched)
    (RasterizeGaussiansCUDA, rasterize_points.cu:36-130).  record_bits=16: this frame's sorted instance records store
    conic / opacity / colour as halves (GSAJ_FWD_RECORDS_FP16; a per-call flag, nothing process-wide)."""
    lib = _lib.load()
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_cuda(means3D, "means3D")
    dev = means3D.device
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    f = dict(device=dev, dtype=_F32)
    out_color = torch.zeros((3, H, W), **f)
    out_depth = torch.zeros((1, H, W), **f)
    out_opacity = torch.zeros((1, H, W), **f)
    radii = torch.zeros((P,), device=dev, dtype=torch.int32)
    n_touched = torch.zeros((P,), device=dev, dtype=torch.int32)
    byte = dict(device=dev, dtype=torch.uint8)
    if P == 0:
        e = torch.empty(0, **byte)
        return 0, out_color, radii, e, e.clone(), e.clone(), out_depth, out_opacity, n_touched

    means3D = _prep(means3D, dev, "means3D")
    colors = _prep(colors, dev, "colors")
    opacity = _prep(opacity, dev, "opacity")
    scales = _prep(scales, dev, "scales")
    rotations = _prep(rotations, dev, "rotations")
    cov3D_precomp = _prep(cov3D_precomp, dev, "cov3D")
    sh = _prep(sh, dev, "sh")
    background = _prep(background, dev, "bg")
    viewmatrix = _prep(viewmatrix, dev, "viewmatrix")
    projmatrix = _prep(projmatrix, dev, "projmatrix")
    campos = _prep(campos, dev, "campos")
    M = 0 if sh is None else sh.shape[1]

    geom = torch.empty(lib.gsaj_geom_workspace_bytes(P), **byte)
    img = torch.empty(lib.gsaj_image_workspace_bytes(W, H), **byte)
    st = _stream(dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gsaj_forward_preprocess(
            P, int(degree), M, W, H, _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacity), _ptr(scales),
            float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix),
            _ptr(campos), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), radii.data_ptr(),
            n_touched.data_ptr(), geom.data_ptr(), img.data_ptr(), st), "gsaj_forward_preprocess")
        R, mt = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(lib.gsaj_forward_num_rendered(W, H, img.data_ptr(), st, ctypes.byref(R), ctypes.byref(mt)),
                   "gsaj_forward_num_rendered")
        R = R.value
        nbytes = lib.gsaj_binning_workspace_bytes(R)
        binning = torch.empty(nbytes, **byte)
        _lib.check(lib.gsaj_forward_render(
            P, R, -1 if FORCE_CHUNKED_SORT else mt.value, W, H, _ptr(background), _ptr(colors), radii.data_ptr(), geom.data_ptr(), binning.data_ptr(), nbytes,
            img.data_ptr(), out_color.data_ptr(), out_depth.data_ptr(), out_opacity.data_ptr(), n_touched.data_ptr(),
            _fwd_flags(record_bits), st), "gsaj_forward_render")
        if debug:
            torch.cuda.synchronize(dev)
    return R, out_color, radii, geom, binning, img, out_depth, out_opacity, n_touched


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_depths, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug,
                                 per_gaussian_tau=True):
    """-> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dtau)
    (RasterizeGaussiansBackwardCUDA, rasterize_points.cu:132-223) followed by three extras: dL_dtau_sum [6]
    (= torch.sum(grad_tau.view(-1, 6), dim=0) of diff_gaussian_rasterization/__init__.py:162, reduced in-kernel),
    dL_dconic [P,2,2] and dL_ddepths [P,1] (internal to the reference's binding; exposed for parity tests)."""
    lib = _lib.load()
    _require_cuda(means3D, "means3D")
    dev = means3D.device
    P = means3D.shape[0]
    H, W = dL_dout_color.shape[1], dL_dout_color.shape[2]
    sh = _prep(sh, dev, "sh")
    M = 0 if sh is None else sh.shape[1]
    f = dict(device=dev, dtype=_F32)
    dL_dmeans3D = torch.empty((P, 3), **f)
    dL_dmeans2D = torch.empty((P, 3), **f)
    dL_dcolors = torch.empty((P, 3), **f)
    dL_ddepths = torch.empty((P, 1), **f)
    dL_dconic = torch.empty((P, 2, 2), **f)
    dL_dopacity = torch.empty((P, 1), **f)
    dL_dcov3D = torch.empty((P, 6), **f)
    dL_dsh = torch.empty((P, M, 3), **f)
    dL_dscales = torch.empty((P, 3), **f)
    dL_drotations = torch.empty((P, 4), **f)
    dL_dtau = torch.empty((P, 6), **f) if per_gaussian_tau else None
    dL_dtau_sum = torch.zeros((6,), **f)
    if P == 0:
        return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
                dL_dtau, dL_dtau_sum, dL_dconic, dL_ddepths)
    means3D = _prep(means3D, dev, "means3D")
    colors = _prep(colors, dev, "colors")
    scales = _prep(scales, dev, "scales")
    rotations = _prep(rotations, dev, "rotations")
    cov3D_precomp = _prep(cov3D_precomp, dev, "cov3D")
    background = _prep(background, dev, "bg")
    viewmatrix = _prep(viewmatrix, dev, "viewmatrix")
    projmatrix = _prep(projmatrix, dev, "projmatrix")
    projmatrix_raw = _prep(projmatrix_raw, dev, "projmatrix_raw")
    campos = _prep(campos, dev, "campos")
    dL_dout_color = _prep(dL_dout_color, dev, "dL_dout_color")
    dL_dout_depths = _prep(dL_dout_depths, dev, "dL_dout_depths")
    st = _stream(dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gsaj_rasterize_backward(
            P, int(degree), M, int(R), _ptr(background), W, H, _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(scales),
            float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix),
            _ptr(projmatrix_raw), _ptr(campos), float(tan_fovx), float(tan_fovy), radii.data_ptr(),
            geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr(), _ptr(dL_dout_color),
            _ptr(dL_dout_depths), dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(),
            dL_dcolors.data_ptr(), dL_ddepths.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(),
            _ptr(dL_dsh), _ptr(dL_dscales) if scales is not None else None,
            _ptr(dL_drotations) if rotations is not None else None,
            dL_dtau.data_ptr() if dL_dtau is not None else None, dL_dtau_sum.data_ptr(), st), "gsaj_rasterize_backward")
        if debug:
            torch.cuda.synchronize(dev)
    if scales is None:
        dL_dscales.zero_()
        dL_drotations.zero_()
    return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dtau,
            dL_dtau_sum, dL_dconic, dL_ddepths)


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool [P] (markVisible, rasterize_points.cu:225-246)."""
    lib = _lib.load()
    _require_cuda(means3D, "means3D")
    dev = means3D.device
    P = means3D.shape[0]
    present = torch.zeros((P,), device=dev, dtype=torch.bool)
    if P != 0:
        means3D = _prep(means3D, dev, "means3D")
        viewmatrix = _prep(viewmatrix, dev, "viewmatrix")
        projmatrix = _prep(projmatrix, dev, "projmatrix")
        with torch.cuda.device(dev):
            _lib.check(lib.gsaj_mark_visible(P, _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix), present.data_ptr(),
                                             _stream(dev)), "gsaj_mark_visible")
    return present


def debug_export(P, R, W, H, geomBuffer, binningBuffer, imageBuffer):
    """Internal forward state as tensors (parity tests)."""
    lib = _lib.load()
    dev = geomBuffer.device
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    f = dict(device=dev, dtype=_F32)
    out = dict(
        means2D=torch.zeros((P, 2), **f), depths=torch.zeros((P,), **f), cov3D=torch.zeros((P, 6), **f),
        conic_opacity=torch.zeros((P, 4), **f), rgb=torch.zeros((P, 3), **f),
        clamped=torch.zeros((P, 3), device=dev, dtype=torch.uint8),
        tiles_touched=torch.zeros((P,), device=dev, dtype=torch.int32),
        point_list=torch.zeros((max(R, 1),), device=dev, dtype=torch.int32),
        ranges=torch.zeros((tiles, 2), device=dev, dtype=torch.int32), final_T=torch.zeros((H, W), **f),
        n_contrib=torch.zeros((H, W), device=dev, dtype=torch.int32))
    with torch.cuda.device(dev):
        _lib.check(lib.gsaj_debug_export(
            P, R, W, H, geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr(),
            out["means2D"].data_ptr(), out["depths"].data_ptr(), out["cov3D"].data_ptr(),
            out["conic_opacity"].data_ptr(), out["rgb"].data_ptr(), out["clamped"].data_ptr(),
            out["tiles_touched"].data_ptr(), out["point_list"].data_ptr(), out["ranges"].data_ptr(),
            out["final_T"].data_ptr(), out["n_contrib"].data_ptr(), _stream(dev)), "gsaj_debug_export")
    out["point_list"] = out["point_list"][:R]
    return out


STAGE_NAMES = ("preprocess,scan_blocks,emit_keys,sort,ranges_records,render_fwd,render_bwd,gaussian_bwd,tau_finalize,"
               "dense_bwd,dense_reduce,scatter_instances,tile_sort_records,gather_sums").split(",")


class profile_stages:
    """Context manager around gsaj_profile_begin/_end: per-stage kernel time (ms) and launch
    counts measured with HIP events on the launch stream."""

    def __init__(self, max_records=4096):
        self.max_records = max_records
        self.ms, self.launches = {}, {}

    def __enter__(self):
        _lib.check(_lib.load().gsaj_profile_begin(self.max_records), "gsaj_profile_begin")
        return self

    def __exit__(self, *exc):
        n = len(STAGE_NAMES)
        ms = (ctypes.c_float * n)()
        cnt = (ctypes.c_int * n)()
        _lib.check(_lib.load().gsaj_profile_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p)),
                   "gsaj_profile_end")
        self.ms = {k: float(ms[i]) for i, k in enumerate(STAGE_NAMES)}
        self.launches = {k: int(cnt[i]) for i, k in enumerate(STAGE_NAMES)}
        return False


FWD_RECORDS_FP16 = 1  # GSAJ_FWD_RECORDS_FP16 (include/gsaj.h)


def _fwd_flags(record_bits):
    """16: the sorted instance records store conic / opacity / colour as halves (32-byte records; positions, depth and all
    accumulation stay fp32).  32: the default fp32 records."""
    if record_bits not in (16, 32):
        raise ValueError("record_bits must be 16 or 32, got %r" % (record_bits,))
    return FWD_RECORDS_FP16 if record_bits == 16 else 0


class ArenaWatch:
    """Growing the binning arena BEFORE a frame overflows it, without a host synchronisation.

    An asynchronous forward that needs more instances than the arena holds is aborted on the device and repeated after a
    blocking re-size -- and a SLAM map grows by construction.  After every `every`-th asynchronous forward the K views' frame
    counters (instances R, error flags, longest tile list: 16 bytes per view) are copied to pinned host memory on the stream
    (non-blocking) and an event is recorded; a later call that finds the event complete looks at them -- a query, never a wait --
    and re-allocates the arena (x1.5 of the largest R) once R passes `grow_at` of the capacity.  The figures are a few frames old
    by then: the head room covers a map that grows a few per cent per frame; a jump beyond it still takes the abort path.
    Not used inside a stream capture (a captured graph holds the arena's address)."""

    def __init__(self, img, K, img_stride, counters_offset, every=4, grow_at=0.75):
        self.K, self.every, self.grow_at = int(K), int(every), float(grow_at)
        i32 = img.view(torch.int32)
        self.dev_counters = torch.as_strided(i32, (self.K, 4), (img_stride // 4, 1), counters_offset // 4)
        self.host = torch.zeros((self.K, 4), dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        self.pending, self.calls = False, 0
        self.grown = 0  # how many times the arena was re-allocated ahead of an overflow
        # the very first strided device-to-pinned-host copy costs tens of milliseconds (the copy kernel's code object is loaded
        # lazily): paid here, where blocking is allowed, not in the middle of somebody's loop
        with torch.cuda.device(img.device):
            self.host.copy_(self.dev_counters, non_blocking=True)
            self.event.record()
            self.event.synchronize()

    def post(self):
        """After an asynchronous forward was enqueued."""
        self.calls += 1
        if self.pending or self.calls % self.every or torch.cuda.is_current_stream_capturing():
            return
        self.host.copy_(self.dev_counters, non_blocking=True)
        self.event.record()
        self.pending = True

    def poll(self):
        """Before an asynchronous forward: (largest R, longest tile list) of a frame a few calls back, or None."""
        if not self.pending or torch.cuda.is_current_stream_capturing() or not self.event.query():
            return None
        self.pending = False
        return int(self.host[:, 0].max()), int(self.host[:, 2].max())


class FrameContext:
    """Pre-allocated outputs and workspaces for repeated forward+backward passes over one scene
    size (the tracking / mapping inner loop): no allocation and two C-ABI calls per step.
    The binning workspace grows geometrically when a frame produces more instances."""

    def __init__(self, P, W, H, M, device, has_scales=True, per_gaussian_tau=False, grad_slots=1, n_keyframes=0,
                 keyframe=0, record_bits=32):
        lib = _lib.load()
        self.flags = _fwd_flags(record_bits)  # per context (= per view), handed to every forward call
        self.lib, self.P, self.W, self.H, self.M, self.dev = lib, P, W, H, M, torch.device(device)
        f = dict(device=self.dev, dtype=_F32)
        byte = dict(device=self.dev, dtype=torch.uint8)
        self.color = torch.zeros((3, H, W), **f)
        self.depth = torch.zeros((1, H, W), **f)
        self.opacity = torch.zeros((1, H, W), **f)
        self.radii = torch.zeros((P,), device=self.dev, dtype=torch.int32)
        self.n_touched = torch.zeros((P,), device=self.dev, dtype=torch.int32)
        self.geom = torch.empty(lib.gsaj_geom_workspace_bytes(P), **byte)
        self.img = torch.zeros(lib.gsaj_image_workspace_bytes(W, H), **byte)  # zeroed: holds the sticky abort counter
        self.binning = torch.empty(0, **byte)
        self.R = 0
        self.capacity = 0  # > 0 once an arena has been sized: enables forward(sync=False)
        self.tile_list_capacity = 0  # tile-list length the LDS sort of asynchronous frames is sized for (0: the maximum, 16384);
                                     # a longer list is sorted in chunks + merge passes (slower, never wrong)
        self.watch = ArenaWatch(self.img, 1, self.img.numel(), self.abort_flag_ptr() - 16 - self.img.data_ptr())  # (counters[4] = the abort word)
        self.auto_grow = os.environ.get("GSAJ_ARENA_WATCH", "1") != "0"  # asynchronous frames: grow the arena ahead of an overflow (ArenaWatch)
        # per-Gaussian parameter gradients live in ONE flat bucket (field-major) so that a multi-GPU
        # mapping step can all-reduce it with a single collective (gsaj.keyframe_shard); `grad_slots`
        # buckets let the collective of step i overlap the kernels of step i+1
        from .keyframe_shard import bucket_numel, bucket_views
        self.buckets, self.slots = [], []
        for _ in range(max(1, grad_slots)):
            bucket = torch.zeros(bucket_numel(P, M, has_scales, n_keyframes), **f)
            v = bucket_views(bucket, P, M, has_scales, n_keyframes)
            self.buckets.append(bucket)
            self.slots.append(dict(
                mean2D=torch.zeros((P, 3), **f), conic=torch.zeros((P, 2, 2), **f), opacity=v["opacity"],
                color=torch.zeros((P, 3), **f), depth=torch.zeros((P, 1), **f), mean3D=v["mean3D"],
                cov3D=v["cov3D"] if not has_scales else torch.zeros((P, 6), **f), sh=v["sh"].view(P, M, 3),
                scale=v.get("scale"), rot=v.get("rot"),
                tau=torch.zeros((P, 6), **f) if per_gaussian_tau else None,
                # with keyframes the 6 pose gradients land in this rank's row of the shared tau block
                tau_sum=v["tau_all"][keyframe] if n_keyframes else torch.zeros((6,), **f),
                tau_all=v.get("tau_all")))
        self.bucket, self.g = self.buckets[0], self.slots[0]

    def set_tile_band(self, tile_row_begin, tile_row_end):
        """Render only tile rows [begin, end) from now on (gsaj_set_tile_band; gsaj.tile_band_shard): the frame's other
        pixels come out as background, every gradient is the band's share.  (0, rows) restores the whole frame."""
        _lib.check(self.lib.gsaj_set_tile_band(self.W, self.H, self.img.data_ptr(), int(tile_row_begin), int(tile_row_end),
                                               _stream(self.dev)), "gsaj_set_tile_band")
        self.band = (int(tile_row_begin), int(tile_row_end))

    def abort_flag_ptr(self):
        """Device address of this context's abort word (gsaj_forward_abort_flag): non-zero after an asynchronous forward that did
        not fit the arena.  For PoseTracker.step(skip=...)."""
        return self.lib.gsaj_forward_abort_flag(self.W, self.H, self.img.data_ptr())

    def abort_flag_tensor(self):
        """The same word as a one-element int32 view of the image workspace (for folding it into a collective)."""
        off = self.abort_flag_ptr() - self.img.data_ptr()
        return self.img[off:off + 4].view(torch.int32)

    def _grow_ahead(self):
        """(asynchronous frames) re-allocate the arena if a recent frame came close to filling it; no host synchronisation."""
        seen = self.watch.poll() if self.auto_grow else None
        if seen is None:
            return
        R, longest = seen
        if R > self.watch.grow_at * self.capacity:
            self.capacity = int(R * 1.5) + 1024
            self.binning = torch.empty(self.lib.gsaj_binning_workspace_bytes(self.capacity), device=self.dev, dtype=torch.uint8)
            self.watch.grown += 1
        if longest > self.tile_list_capacity:
            self.tile_list_capacity = min(SORT_CAP, int(1.1 * longest) + 1)

    def _ensure_binning(self, R):
        need = self.lib.gsaj_binning_workspace_bytes(R)
        if self.binning.numel() < need:
            self.capacity = int(R * 1.5) + 1024
            self.binning = torch.empty(self.lib.gsaj_binning_workspace_bytes(self.capacity), device=self.dev,
                                       dtype=torch.uint8)

    def status(self):
        """Blocking: (num_rendered, longest tile list) of the last forward; raises if it, or any asynchronous forward since the
        previous status(), was aborted on the device.  The abort counter is read AND cleared first, so that a caller which
        re-sizes after the exception starts from a clean slate."""
        n = ctypes.c_int(0)
        _lib.check(self.lib.gsaj_forward_aborted_count(self.W, self.H, self.img.data_ptr(), _stream(self.dev),
                                                       ctypes.byref(n)), "gsaj_forward_aborted_count")
        R, mt = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self.lib.gsaj_forward_num_rendered(self.W, self.H, self.img.data_ptr(), _stream(self.dev),
                                                      ctypes.byref(R), ctypes.byref(mt)), "gsaj_forward_num_rendered")
        self.true_R = R.value
        if n.value:
            raise _lib.GsajError("%d asynchronous forward(s) were aborted on the device (binning arena too small for the frame's "
                                 "instances); run forward(sync=True) to re-size" % n.value)
        return R.value, mt.value

    def forward(self, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, sh_degree=0, shs=None,
                colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, sync=True):
        """sync=True: read R back (16-byte D2H, the reference's only sync) and size the arena exactly.
        sync=False: no host round trip at all -- the arena sized by an earlier synchronous frame (x1.5)
        is reused and an overflowing frame aborts on the device; call status() when convenient."""
        with torch.cuda.device(self.dev):
            return self._forward(bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, sh_degree, shs, colors_precomp,
                                 scales, rotations, cov3D_precomp, scale_modifier, sync)

    def _forward(self, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, sh_degree, shs, colors_precomp, scales,
                 rotations, cov3D_precomp, scale_modifier, sync):
        lib, st = self.lib, _stream(self.dev)
        if not sync and self.capacity > 0:
            self._grow_ahead()
            _lib.check(lib.gsaj_rasterize_forward_async(
                self.P, int(sh_degree), self.M, _ptr(bg), self.W, self.H, _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                _ptr(opacities), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tanfovx), float(tanfovy), 0,
                self.color.data_ptr(), self.depth.data_ptr(), self.opacity.data_ptr(), self.radii.data_ptr(),
                self.n_touched.data_ptr(), self.geom.data_ptr(), self.binning.data_ptr(), self.binning.numel(),
                self.capacity, self.tile_list_capacity, self.img.data_ptr(), self.flags, st), "gsaj_rasterize_forward_async")
            self.R = self.capacity  # what the backward must be given (arena carving)
            if self.auto_grow:
                self.watch.post()
            return self.R
        _lib.check(lib.gsaj_forward_preprocess(
            self.P, int(sh_degree), self.M, self.W, self.H, _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
            _ptr(opacities), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix),
            _ptr(projmatrix), _ptr(campos), float(tanfovx), float(tanfovy), 0, self.radii.data_ptr(),
            self.n_touched.data_ptr(), self.geom.data_ptr(), self.img.data_ptr(), st), "gsaj_forward_preprocess")
        R, mt = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(lib.gsaj_forward_num_rendered(self.W, self.H, self.img.data_ptr(), st, ctypes.byref(R), ctypes.byref(mt)),
                   "gsaj_forward_num_rendered")
        self.R, self.max_tile_list = R.value, mt.value
        # asynchronous frames get an LDS sort sized for the longest tile list seen so far + 10 % (the launcher rounds up to a power
        # of two, which is what the bitonic network pads to anyway; more LDS than that only costs resident workgroups: 2x the
        # longest list made cfg5's sort 2.2x slower).  A longer list is still sorted -- in LDS-sized chunks and merge passes
        self.tile_list_capacity = min(SORT_CAP, max(self.tile_list_capacity, int(1.1 * self.max_tile_list) + 1, 256))
        self._ensure_binning(self.R)
        self.true_R = self.R
        _lib.check(lib.gsaj_forward_render(
            self.P, self.R, -1 if FORCE_CHUNKED_SORT else self.max_tile_list, self.W, self.H, _ptr(bg), _ptr(colors_precomp), self.radii.data_ptr(), self.geom.data_ptr(),
            self.binning.data_ptr(), self.binning.numel(), self.img.data_ptr(), self.color.data_ptr(),
            self.depth.data_ptr(), self.opacity.data_ptr(), self.n_touched.data_ptr(), self.flags, st), "gsaj_forward_render")
        return self.R

    # ---- the loss fused into the compositors (gsaj_rasterize_forward_loss / _backward_loss) -----------------
    def forward_loss(self, loss, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, sh_degree=0, shs=None,
                     colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0):
        """Asynchronous forward (the arena must have been sized by a synchronous forward()) whose compositor also sums the loss:
        loss = dict(flags, alpha, rgb_boundary_threshold, gt_color [3,H,W], gt_depth [H,W] or None, grad_mask (uint8 [H*W]) or None,
        exposure_a, exposure_b (device scalars or None), scalars (float32 [5] device: loss, L_rgb, L_depth, dL/da, dL/db))."""
        if self.capacity <= 0:
            raise _lib.GsajError("forward_loss: size the arena with one synchronous forward() first")
        if not hasattr(self, "loss_ws"):
            self.loss_ws = torch.empty(self.lib.gsaj_fused_loss_workspace_bytes(self.W, self.H), device=self.dev, dtype=torch.uint8)
        with torch.cuda.device(self.dev):
            self._grow_ahead()
            _lib.check(self.lib.gsaj_rasterize_forward_loss(
                self.P, int(sh_degree), self.M, _ptr(bg), self.W, self.H, _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                _ptr(opacities), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tanfovx), float(tanfovy), 0,
                self.color.data_ptr(), self.depth.data_ptr(), self.opacity.data_ptr(), self.radii.data_ptr(),
                self.n_touched.data_ptr(), self.geom.data_ptr(), self.binning.data_ptr(), self.binning.numel(),
                self.capacity, self.tile_list_capacity, self.img.data_ptr(), self.flags, int(loss["flags"]), float(loss["alpha"]),
                float(loss["rgb_boundary_threshold"]), _ptr(loss["gt_color"]), _ptr(loss.get("gt_depth")), _ptr(loss.get("grad_mask")),
                _ptr(loss.get("exposure_a")), _ptr(loss.get("exposure_b")), loss["scalars"].data_ptr(), self.loss_ws.data_ptr(),
                _stream(self.dev)), "gsaj_rasterize_forward_loss")
            if self.auto_grow:
                self.watch.post()
        self.R = self.capacity
        return self.R

    def backward_loss(self, loss, bg, means3D, viewmatrix, projmatrix, projmatrix_raw, campos, tanfovx, tanfovy, sh_degree=0, shs=None,
                      colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, slot=0,
                      pose_only=False):
        """backward() with the pixel seeds derived inside the reverse compositor from this context's color / depth / opacity and the
        ground truth of `loss` (same dict as forward_loss): no dL/dcolor, dL/ddepth images exist."""
        g = self.slots[slot]
        if g["tau_all"] is not None:
            g["tau_all"].zero_()
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.gsaj_rasterize_backward_loss(
                self.P, int(sh_degree), self.M, self.R, _ptr(bg), self.W, self.H, _ptr(means3D), _ptr(shs),
                _ptr(colors_precomp), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                _ptr(viewmatrix), _ptr(projmatrix), _ptr(projmatrix_raw), _ptr(campos), float(tanfovx), float(tanfovy),
                self.radii.data_ptr(), self.geom.data_ptr(), self.binning.data_ptr(), self.img.data_ptr(), int(loss["flags"]),
                float(loss["alpha"]), float(loss["rgb_boundary_threshold"]), self.color.data_ptr(), self.depth.data_ptr(),
                self.opacity.data_ptr(), _ptr(loss["gt_color"]), _ptr(loss.get("gt_depth")), _ptr(loss.get("grad_mask")),
                _ptr(loss.get("exposure_a")), _ptr(loss.get("exposure_b")), *([None] * 10 if pose_only else [
                    g["mean2D"].data_ptr(), g["conic"].data_ptr(), g["opacity"].data_ptr(), g["color"].data_ptr(),
                    g["depth"].data_ptr(), g["mean3D"].data_ptr(), g["cov3D"].data_ptr(), _ptr(g["sh"]), _ptr(g["scale"]),
                    _ptr(g["rot"])]), _ptr(g["tau"]), g["tau_sum"].data_ptr(), _stream(self.dev)), "gsaj_rasterize_backward_loss")
        return g

    def backward(self, bg, means3D, viewmatrix, projmatrix, projmatrix_raw, campos, tanfovx, tanfovy, dL_dcolor, dL_ddepth,
                 sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 scale_modifier=1.0, slot=0, pose_only=False):
        """pose_only=True (tracking: only the camera is optimised): the per-Gaussian parameter gradients are neither
        computed to the end nor stored; only g["tau_sum"] (and g["tau"] if allocated) are valid afterwards."""
        g = self.slots[slot]
        if g["tau_all"] is not None:
            g["tau_all"].zero_()  # rows of the other ranks' keyframes must be zero before the sum all-reduce
        with torch.cuda.device(self.dev):
            self._backward(g, bg, means3D, viewmatrix, projmatrix, projmatrix_raw, campos, tanfovx, tanfovy, dL_dcolor, dL_ddepth, sh_degree,
                           shs, colors_precomp, scales, rotations, cov3D_precomp, scale_modifier, pose_only)
        return g

    def _backward(self, g, bg, means3D, viewmatrix, projmatrix, projmatrix_raw, campos, tanfovx, tanfovy, dL_dcolor, dL_ddepth, sh_degree,
                  shs, colors_precomp, scales, rotations, cov3D_precomp, scale_modifier, pose_only):
        _lib.check(self.lib.gsaj_rasterize_backward(
            self.P, int(sh_degree), self.M, self.R, _ptr(bg), self.W, self.H, _ptr(means3D), _ptr(shs),
            _ptr(colors_precomp), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
            _ptr(viewmatrix), _ptr(projmatrix), _ptr(projmatrix_raw), _ptr(campos), float(tanfovx), float(tanfovy),
            self.radii.data_ptr(), self.geom.data_ptr(), self.binning.data_ptr(), self.img.data_ptr(), _ptr(dL_dcolor),
            _ptr(dL_ddepth), *([None] * 10 if pose_only else [
                g["mean2D"].data_ptr(), g["conic"].data_ptr(), g["opacity"].data_ptr(), g["color"].data_ptr(),
                g["depth"].data_ptr(), g["mean3D"].data_ptr(), g["cov3D"].data_ptr(), _ptr(g["sh"]), _ptr(g["scale"]),
                _ptr(g["rot"])]), _ptr(g["tau"]), g["tau_sum"].data_ptr(), _stream(self.dev)),
            "gsaj_rasterize_backward")
        return g

    def interactions(self):
        """sum over pixels of n_contrib = Gaussian-pixel interactions of the last forward."""
        dbg = debug_export(self.P, self.R, self.W, self.H, self.geom, self.binning, self.img)  # only n_contrib is used
        return int(dbg["n_contrib"].to(torch.int64).sum().item())


class BatchContext:
    """K views of ONE Gaussian map through the batched C-ABI entry points (gsaj_rasterize_forward_batch / _backward_batch): the
    mapping window of the reference (utils/slam_backend.py:168-232) in one set of launches.  Per-view outputs: color / depth /
    opacity [K,.,H,W], radii / n_touched [K,P], g["mean2D"] [K,P,3] (densification reads it per view), g["tau_all"] [K,6];
    per-Gaussian parameter gradients summed over the K views in-kernel, in the flat bucket of gsaj.keyframe_shard (one
    all-reduce ships it, tail = the K pose-gradient rows)."""

    def __init__(self, K, P, W, H, M, device, has_scales=True, record_bits=32, per_gaussian_tau=False, grad_slots=1,
                 n_windows=1, window=0, streams=1):
        """n_windows > 1 (multi-GPU: one window per rank): the bucket's tail has a dL/dtau row for every keyframe of every
        window; this context writes the rows of window `window` and keeps the others zero, so that the ONE sum all-reduce of
        the bucket also gathers the pose gradients of all ranks.
        streams = 2: the window is processed as two groups of views on two HIP streams -- the per-Gaussian / binning kernels of
        one group (latency-bound, few workgroups) overlap the compositors of the other; only the two accumulating per-Gaussian
        chain launches are ordered (GSAJ_BWD_ONLY_COMPOSITE / _ONLY_CHAIN / _ACCUMULATE).  Results are identical to streams = 1
        up to the order of the final fp32 additions (group 0's sum + group 1's sum)."""
        from .keyframe_shard import bucket_numel, bucket_views
        lib = _lib.load()
        self.lib, self.K, self.P, self.W, self.H, self.M, self.dev = lib, int(K), int(P), int(W), int(H), int(M), torch.device(device)
        self.flags = _fwd_flags(record_bits)
        f = dict(device=self.dev, dtype=_F32)
        byte = dict(device=self.dev, dtype=torch.uint8)
        self.color = torch.zeros((K, 3, H, W), **f)
        self.depth = torch.zeros((K, 1, H, W), **f)
        self.opacity = torch.zeros((K, 1, H, W), **f)
        self.radii = torch.zeros((K, P), device=self.dev, dtype=torch.int32)
        self.n_touched = torch.zeros((K, P), device=self.dev, dtype=torch.int32)
        self.geom_stride = lib.gsaj_geom_workspace_bytes(P)
        self.img_stride = lib.gsaj_image_workspace_bytes(W, H)
        self.geom = torch.empty(K * self.geom_stride, **byte)
        self.img = torch.zeros(K * self.img_stride, **byte)  # zeroed once: holds the sticky abort counters
        self.binning = torch.empty(0, **byte)
        self.capacity, self.tile_list_capacity, self.bin_stride = 0, 0, 0
        self.watch = ArenaWatch(self.img, self.K, self.img_stride, lib.gsaj_forward_abort_flag(W, H, self.img.data_ptr()) - 16 - self.img.data_ptr())
        self.auto_grow = os.environ.get("GSAJ_ARENA_WATCH", "1") != "0"  # sync=False windows: grow the arena ahead of an overflow (ArenaWatch)
        self.buckets, self.slots = [], []
        for _ in range(max(1, grad_slots)):
            bucket = torch.zeros(bucket_numel(P, M, has_scales, K * n_windows), **f)
            v = bucket_views(bucket, P, M, has_scales, K * n_windows)
            self.buckets.append(bucket)
            self.slots.append(dict(mean2D=torch.zeros((K, P, 3), **f), opacity=v["opacity"], mean3D=v["mean3D"],
                                   cov3D=v["cov3D"] if not has_scales else torch.zeros((P, 6), **f), sh=v["sh"].view(P, M, 3),
                                   scale=v.get("scale"), rot=v.get("rot"), tau=torch.zeros((K, P, 6), **f) if per_gaussian_tau else None,
                                   tau_all=v["tau_all"][window * K:(window + 1) * K], tau_every_window=v["tau_all"] if n_windows > 1 else None))
        self.bucket, self.g = self.buckets[0], self.slots[0]
        # view groups: [(first view, number of views, stream or None = the caller's current stream)]
        if streams >= 2 and K >= 2:
            k0 = (K + 1) // 2
            self.groups = [(0, k0, None), (k0, K - k0, torch.cuda.Stream(self.dev))]
        else:
            self.groups = [(0, K, None)]

    def _size(self, capacity):
        self.capacity = int(capacity)
        self.bin_stride = self.lib.gsaj_binning_workspace_bytes(self.capacity)
        self.binning = torch.empty(self.K * self.bin_stride, device=self.dev, dtype=torch.uint8)

    def set_tile_band(self, tile_row_begin, tile_row_end, views=None):
        """Render only tile rows [begin, end) of the given views (default: all K) from now on -- gsaj_set_tile_band on each view's
        block of the image workspace.  (0, rows) restores whole frames."""
        for v in (range(self.K) if views is None else views):
            _lib.check(self.lib.gsaj_set_tile_band(self.W, self.H, self.img.data_ptr() + int(v) * self.img_stride, int(tile_row_begin),
                                                   int(tile_row_end), _stream(self.dev)), "gsaj_set_tile_band")

    def status(self):
        """Blocking: per view (num_rendered, longest tile list, aborted)."""
        out, st = [], _stream(self.dev)
        for v in range(self.K):
            R, mt = ctypes.c_int(0), ctypes.c_int(0)
            rc = self.lib.gsaj_forward_num_rendered(self.W, self.H, self.img.data_ptr() + v * self.img_stride, st, ctypes.byref(R), ctypes.byref(mt))
            if rc not in (0, -3):
                _lib.check(rc, "gsaj_forward_num_rendered")
            out.append((R.value, mt.value, rc == -3))
        return out

    def abort_flags(self):
        """(address of view 0's abort word, byte stride between the views' words) for PoseTrackerBatch.step(skip=..., skip_stride=...)."""
        return self.lib.gsaj_forward_abort_flag(self.W, self.H, self.img.data_ptr()), self.img_stride

    def clear_aborts(self):
        """Blocking: read and clear every view's count of aborted asynchronous forwards; returns their sum."""
        total, st = 0, _stream(self.dev)
        for v in range(self.K):
            n = ctypes.c_int(0)
            _lib.check(self.lib.gsaj_forward_aborted_count(self.W, self.H, self.img.data_ptr() + v * self.img_stride, st, ctypes.byref(n)),
                       "gsaj_forward_aborted_count")
            total += n.value
        return total

    def _fork(self):
        """side streams wait for everything the caller's stream has enqueued so far (inputs, the previous step's results)"""
        cur = torch.cuda.current_stream(self.dev)
        for _, _, st in self.groups:
            if st is not None:
                st.wait_stream(cur)
        return cur

    def _join(self, cur):
        for _, _, st in self.groups:
            if st is not None:
                cur.wait_stream(st)

    def _launch(self, bg, means3D, opacities, viewmatrices, projmatrices, campos, tanfovx, tanfovy, sh_degree, shs, colors_precomp, scales,
                rotations, cov3D_precomp, scale_modifier):
        HW, P = self.H * self.W, self.P
        cur = self._fork()
        for v0, kv, st in self.groups:
            stream = cur if st is None else st
            _lib.check(self.lib.gsaj_rasterize_forward_batch(
                kv, P, int(sh_degree), self.M, _ptr(bg), self.W, self.H, _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                _ptr(opacities), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                viewmatrices.data_ptr() + 64 * v0, projmatrices.data_ptr() + 64 * v0, None if campos is None else campos.data_ptr() + 12 * v0,
                float(tanfovx), float(tanfovy), 0, self.color.data_ptr() + 12 * HW * v0, self.depth.data_ptr() + 4 * HW * v0,
                self.opacity.data_ptr() + 4 * HW * v0, self.radii.data_ptr() + 4 * P * v0, self.n_touched.data_ptr() + 4 * P * v0,
                self.geom.data_ptr() + self.geom_stride * v0, self.binning.data_ptr() + self.bin_stride * v0, self.bin_stride * kv,
                self.capacity, self.tile_list_capacity, self.img.data_ptr() + self.img_stride * v0, self.flags, stream.cuda_stream),
                "gsaj_rasterize_forward_batch")
        self._join(cur)

    def forward(self, bg, means3D, opacities, viewmatrices, projmatrices, campos, tanfovx, tanfovy, sh_degree=0, shs=None,
                colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, sync=True):
        """viewmatrices / projmatrices [K,4,4] (the rasteriser's transposed matrices), campos [K,3].
        sync=False: no host round trip (arena from an earlier synchronous call; overflowing views abort on the device).
        sync=True: the K instance counts are read back; if a view did not fit, the arena grows and the batch is re-run (tile bands
    set with set_tile_band stay in force: only the abort counters are cleared)."""
        a = (bg, means3D, opacities, viewmatrices, projmatrices, campos, tanfovx, tanfovy, sh_degree, shs, colors_precomp, scales,
             rotations, cov3D_precomp, scale_modifier)
        if self.capacity == 0:
            self._size(max(4096, 12 * self.P))
        elif not sync and self.auto_grow:
            seen = self.watch.poll()
            if seen is not None:
                if seen[0] > self.watch.grow_at * self.capacity:
                    self._size(int(1.5 * seen[0]) + 1024)
                    self.watch.grown += 1
                if seen[1] > self.tile_list_capacity > 0:
                    self.tile_list_capacity = min(SORT_CAP, int(1.1 * seen[1]) + 1)
        self._launch(*a)
        if not sync:
            if self.auto_grow:
                self.watch.post()
            return None
        st = self.status()
        if any(ab for _, _, ab in st):
            self._size(int(1.5 * max(r for r, _, _ in st)) + 1024)
            self.tile_list_capacity = 0  # the maximum (16384) until the lists are known
            self.clear_aborts()          # (NOT img.zero_(): the image workspaces also hold the views' tile bands)
            self._launch(*a)
            st = self.status()
            if any(ab for _, _, ab in st):
                raise _lib.GsajError("the batch was aborted again after re-sizing the arena to %d instances per view: %r" % (self.capacity, st))
        if self.auto_grow and max(r for r, _, _ in st) > self.watch.grow_at * self.capacity:
            # the window fits, but with less head room than the arena watch keeps: re-size HERE, where blocking is allowed, rather than
            # a few asynchronous windows later (an allocation of gigabytes in the middle of a loop)
            self._size(int(1.5 * max(r for r, _, _ in st)) + 1024)
            self._launch(*a)
            st = self.status()
        self.tile_list_capacity = min(SORT_CAP, max(self.tile_list_capacity, int(1.1 * max(m for _, m, _ in st)) + 1, 256))
        return st

    def backward(self, bg, means3D, viewmatrices, projmatrices, projmatrix_raw, campos, tanfovx, tanfovy, dL_dcolor, dL_ddepth,
                 sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, slot=0):
        g = self.slots[slot]
        if g["tau_every_window"] is not None:
            g["tau_every_window"].zero_()  # rows of the other ranks' windows must be zero before the sum all-reduce
        HW, P = self.H * self.W, self.P

        def call(v0, kv, stream, flags):
            _lib.check(self.lib.gsaj_rasterize_backward_batch(
                kv, P, int(sh_degree), self.M, self.capacity, _ptr(bg), self.W, self.H, _ptr(means3D), _ptr(shs),
                _ptr(colors_precomp), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                viewmatrices.data_ptr() + 64 * v0, projmatrices.data_ptr() + 64 * v0, _ptr(projmatrix_raw),
                None if campos is None else campos.data_ptr() + 12 * v0, float(tanfovx), float(tanfovy), self.radii.data_ptr() + 4 * P * v0,
                self.geom.data_ptr() + self.geom_stride * v0, self.binning.data_ptr() + self.bin_stride * v0,
                self.img.data_ptr() + self.img_stride * v0, dL_dcolor.data_ptr() + 12 * HW * v0, dL_ddepth.data_ptr() + 4 * HW * v0,
                g["mean2D"].data_ptr() + 12 * P * v0, None, g["opacity"].data_ptr(), None, None, g["mean3D"].data_ptr(),
                g["cov3D"].data_ptr(), _ptr(g["sh"]), _ptr(g["scale"]), _ptr(g["rot"]),
                None if g["tau"] is None else g["tau"].data_ptr() + 24 * P * v0, g["tau_all"].data_ptr() + 24 * v0, flags,
                stream.cuda_stream), "gsaj_rasterize_backward_batch")

        cur = self._fork()
        if len(self.groups) == 1:
            call(0, self.K, cur, 0)
            return g
        for v0, kv, st in self.groups:      # per-view halves: independent, one stream each
            call(v0, kv, cur if st is None else st, 2)
        self._join(cur)                      # the accumulating per-Gaussian chains: in group order on the caller's stream
        for i, (v0, kv, st) in enumerate(self.groups):
            call(v0, kv, cur, 4 | (1 if i > 0 else 0))
        return g

    def view_sums(self, v):
        """[P,12]: the reverse compositor's 10 sums per Gaussian of view v of the last backward (gsaj_debug_export_view_sums):
        dL/dmean2D x, y | dL/dconic a, b, c | dL/dopacity | dL/dcolor r, g, b | dL/ddepth | 2 pads.  For parity tests."""
        out = torch.zeros((self.P, 12), device=self.dev, dtype=_F32)
        _lib.check(self.lib.gsaj_debug_export_view_sums(self.P, self.geom.data_ptr() + int(v) * self.geom_stride, out.data_ptr(),
                                                        _stream(self.dev)), "gsaj_debug_export_view_sums")
        return out

    def interactions(self):
        """sum over views and pixels of n_contrib (Gaussian-pixel interactions of the last forward)."""
        total = 0
        for v in range(self.K):
            n = torch.zeros((self.H, self.W), device=self.dev, dtype=torch.int32)
            _lib.check(self.lib.gsaj_debug_export(self.P, 0, self.W, self.H, None, None, self.img.data_ptr() + v * self.img_stride,
                                                  None, None, None, None, None, None, None, None, None, None, n.data_ptr(),
                                                  _stream(self.dev)), "gsaj_debug_export")
            total += int(n.to(torch.int64).sum().item())
        return total
