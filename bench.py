#!/usr/bin/env python3
"""Benchmark of the hot path: forward splat + analytical-Jacobian backward at 640x480
(BASELINE.json metric: Gaussian-pixel interactions / second, fwd + Jacobian).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus N ...                    (starts the N ranks itself, one process per GPU: gsaj/launcher.py)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W   (the same ranks, started by torch's launcher)

A step = one pass of the hot path over one MAPPING WINDOW per rank: `--views` (default 8) keyframes with DISTINCT cameras
over one shared Gaussian map -- preprocess -> tile binning -> composite -> reverse composite -> per-Gaussian backward with the
per-Gaussian gradients summed over the keyframes and one dL/dtau per keyframe -- through the batched C ABI of libgsaj_hip.so
(gsaj_rasterize_forward_batch / _backward_batch), every input already resident in HBM, no host synchronisation inside the
timed region.  That is the reference's mapping iteration (utils/slam_backend.py:168-232: window of 8-10 keyframes, one
backward).  For N > 1 the step ends with ONE RCCL all-reduce of the flat gradient bucket (per-Gaussian grads + the
pose-gradient rows), overlapped with the next step; `--scaling weak` (default): every rank owns its own window of `--views`
keyframes of the same (replicated) map; `--scaling strong`: ONE window of `--views` keyframes dealt over the ranks (BASELINE
config 4: "8 keyframes sharded across 8 MI355X").
An interaction = one (pixel, Gaussian) pair the compositor visits: I = sum over views and pixels of n_contrib.
Beside it: the same workload one frame at a time through the single-view entry points (`single_stream`, the latency a
sequential tracking loop sees; the per-kernel roofline numbers of the single-view kernels come from that pass).
Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gs-slam-analytica_jacobian_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

FLOP_FWD, FLOP_BWD = 24.0, 87.0  # fp32 flop per interaction (+ 1 exp fwd, 1 exp + 1 rcp bwd)
PEAK_FP32_TFLOPS = 157.3         # MI355X fp32 vector peak (MI355X_MICROARCH.md; equals the fp32 MFMA dense peak)
PEAK_HBM_GBS = 8000.0
PRIMING_STEPS = 24  # untimed windows before the warm-up the command line asks for: the first sizes the arenas (synchronous), the next
                    # load every code object, and the GPU needs ~15 ms of work to reach its steady clocks (20 timed windows after
                    # 8 untimed ones ran 6 % slower than after 30); reported as priming_steps, never counted as warm-up
PMC_FILE = "profiles/r03_pmc_summary.json"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--views", type=int, default=8, help="keyframes per mapping window, each with its own camera (per rank with "
                    "--scaling weak, in all with --scaling strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one window of --views keyframes per rank; strong = the --views keyframes of ONE window dealt over the ranks")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="HIP streams the window's views are spread over (2: two view groups, the accumulating per-Gaussian chains stay "
                         "ordered; measured SLOWER than 1 at cfg2, K = 8: 1.10 vs 1.03 ms -- the batched grids already fill the chip)")
    ap.add_argument("--skip-single", action="store_true", help="profiling runs: only the batched window, no single-view pass")
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--records", default="fp32", choices=["fp32", "fp16"],
                    help="storage of the per-Gaussian splat rows the compositors gather (fp16: conic / opacity / colour as halves, config 5)")
    a = ap.parse_args(argv)
    if a.gpus < 1 or a.steps < 1 or a.warmup < 0 or a.views < 1:
        ap.error("--gpus, --steps, --views must be >= 1 and --warmup >= 0")
    if a.scaling == "strong" and a.views % a.gpus:
        ap.error("--scaling strong deals the window's %d keyframes evenly: --views must be a multiple of --gpus" % a.views)
    return a


def csrc_hash():
    """Fingerprint of the kernel sources: committed PMC summaries are only quoted if they were taken from THIS code."""
    h = hashlib.sha1()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    a = parse()
    from gsaj import launcher

    # ---- who starts the ranks.  Nothing above or in this block touches a GPU. ----
    bad = launcher.world_mismatch(a.gpus)
    if bad:
        print("bench.py: %s -- refusing to run" % bad, file=sys.stderr)
        return 2
    if a.gpus > 1 and not launcher.under_launcher():
        return launcher.launch_ranks(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:])
    return run(a)


def run(a):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, (world, a.gpus)  # (main() refused anything else)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU product path)"
    share = os.environ.get("GSAJ_SHARE_DEVICE") == "1"  # rehearsal on a 1-GPU box: every rank on cuda:0 (gloo only)
    if share:
        local = 0
    elif local >= torch.cuda.device_count():
        print("bench.py: rank %d has no GPU (%d visible) -- refusing to run" % (rank, torch.cuda.device_count()), file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        backend = os.environ.get("GSAJ_DIST_BACKEND", "gloo" if share else "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    ranks_seen = dist.get_world_size() if world > 1 else 1
    if ranks_seen != a.gpus:
        print("bench.py: the process group has %d ranks, --gpus says %d -- refusing to run" % (ranks_seen, a.gpus), file=sys.stderr)
        return 2

    from gsaj import synthetic as syn
    from gsaj import keyframe_shard as ks
    from gsaj.rasterizer import BatchContext, FrameContext, profile_stages

    bits = 16 if a.records == "fp16" else 32
    cam0, sc = syn.config_scene(a.workload)
    strong = a.scaling == "strong" and world > 1
    K = max(1, a.views // world if strong else a.views)  # keyframes of this rank
    # the window(s): weak -- K of the K * world keyframe cameras on the arc around the configuration's camera per rank;
    # strong -- the ONE window's a.views cameras, K = views / world consecutive ones per rank
    ckw = {k: cam0[k] for k in ("W", "H", "fx", "fy", "cx", "cy")}
    cams = syn.keyframe_cameras(K * world, **ckw)[rank * K:(rank + 1) * K]
    P, W, H = sc["means3D"].shape[0], cam0["W"], cam0["H"]
    M = sc["shs"].shape[1]
    deg = min(a.sh_degree, int(round(M ** 0.5)) - 1)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)  # noqa: E731
    means, opac, shs, scales, rots = t(sc["means3D"]), t(sc["opacities"]), t(sc["shs"]), t(sc["scales"]), t(sc["rotations"])
    views, projs = t(np.stack([c["viewmatrix"] for c in cams])), t(np.stack([c["projmatrix"] for c in cams]))
    campos, proj_raw = t(np.stack([c["campos"] for c in cams])), t(cam0["projmatrix_raw"])
    bg = torch.zeros(3, device=dev)
    rng = np.random.default_rng(1234 + rank)
    dLc = t(rng.normal(size=(K, 3, H, W)) / (3 * H * W))  # pixel-gradient seeds of every keyframe, resident in HBM
    dLd = t(rng.normal(size=(K, 1, H, W)) / (H * W))
    tx, ty = cam0["tanfovx"], cam0["tanfovy"]
    geo = dict(sh_degree=deg, shs=shs, scales=scales, rotations=rots)

    # two gradient buckets: the all-reduce of step i (RCCL stream) overlaps the kernels of step i+1
    ctx = BatchContext(K, P, W, H, M, dev, record_bits=bits, grad_slots=2 if world > 1 else 1, n_windows=world, window=rank,
                       streams=a.streams)
    pending = [None, None]
    counter = [0]

    def step():
        n = counter[0]
        counter[0] += 1
        slot = n & 1 if world > 1 else 0
        if pending[slot] is not None:
            pending[slot].wait()  # stream-level: the bucket is about to be overwritten
            pending[slot] = None
        # first window: synchronous (sizes the binning arenas); afterwards no host round trip
        ctx.forward(bg, means, opac, views, projs, campos, tx, ty, sync=(n == 0), **geo)
        ctx.backward(bg, means, views, projs, proj_raw, campos, tx, ty, dLc, dLd, slot=slot, **geo)
        if world > 1:  # one collective: per-Gaussian grads summed over ranks; bucket tail = this rank's pose-gradient rows
            pending[slot] = ks.allreduce_gaussian_grads(ctx.buckets[slot], async_op=True)

    def fence():
        for i in range(2):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(PRIMING_STEPS):  # untimed and NOT counted as warm-up: reported as priming_steps
        step()
    fence()
    for _ in range(a.warmup):       # the W warm-up steps the command line asks for
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_enqueued = time.perf_counter() - t0  # host time to enqueue the K steps (the GPU runs behind it)
    fence()
    elapsed = time.perf_counter() - t0
    st = ctx.status()
    assert not any(ab for _, _, ab in st), "a view was aborted on the device (arena too small): %r" % (st,)
    inter = ctx.interactions()
    R = sum(r for r, _, _ in st)
    tau_batched = ctx.slots[(counter[0] - 1) & 1 if world > 1 else 0]["tau_all"].clone()  # [K, 6] of the last step (before any all-reduce of ANOTHER step lands: slots alternate)

    # the same steps again with every kernel bracketed by HIP events on its launch stream (outside the timed region)
    fence()
    with profile_stages(max_records=a.steps * 24) as prof:
        for _ in range(a.steps):
            step()
        fence()

    # the gradient exchange on its own (N > 1): blocking collectives of the bucket, outside the timed region
    comm = None
    if world > 1:
        comm = {"bucket_bytes": int(ctx.buckets[0].numel() * 4), "backend": backend}
        for form in ("all_reduce", "reduce_scatter+all_gather"):
            try:
                ks.exchange_gaussian_grads(ctx.buckets[0], form=form)  # (warm)
                fence()
                t1 = time.perf_counter()
                for _ in range(10):
                    ks.exchange_gaussian_grads(ctx.buckets[0], form=form)
                torch.cuda.synchronize(dev)
                comm[form + "_ms"] = 1e2 * (time.perf_counter() - t1)
            except Exception as e:  # (gloo has no reduce-scatter; informational leg, outside the timed region: never fatal)
                comm[form + "_ms"] = None
                comm[form + "_error"] = str(e).splitlines()[0][:120]
        fence()

    # one frame at a time through the single-view entry points: the latency of a sequential tracking loop
    fc = FrameContext(P, W, H, M, dev, record_bits=bits)
    n_single = 0 if a.skip_single else max(a.steps, 20)
    tau_single = []

    def single(n):
        k = n % K
        fc.forward(bg, means, opac, views[k], projs[k], campos[k], tx, ty, sync=(n < K), **geo)
        return fc.backward(bg, means, views[k], projs[k], proj_raw, campos[k], tx, ty, dLc[k], dLd[k], **geo)

    for n in range(K if n_single else 0):  # every camera once, synchronously: sizes the arena for the largest view
        tau_single.append(single(n)["tau_sum"].clone())
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for n in range(K, K + n_single):
        single(n)
    torch.cuda.synchronize(dev)
    elapsed_single = max(time.perf_counter() - t0, 1e-9)
    if n_single:
        fc.status()
    with profile_stages(max_records=max(16, n_single * 16)) as prof1:
        for n in range(K, K + n_single):
            single(n)
        torch.cuda.synchronize(dev)
    inter_1 = inter / K  # mean interactions per frame of the window
    # self-check of the timed path: the batched window's K rows of dL/dtau against the single-view kernels' (different kernels
    # for the per-Gaussian chain and its sums; the parity tests hold both to the oracle)
    selfcheck = None
    if tau_single:
        ts = torch.stack(tau_single)
        tb = tau_batched if world == 1 else tau_batched  # (rows of this rank's keyframes; other ranks' rows are theirs)
        selfcheck = {"tau_max_rel": float((tb - ts).abs().max() / ts.abs().max()), "what": "max |dL/dtau batched - single view| / max |dL/dtau| "
                     "over the window's %d keyframes (rank 0)" % K}

    stats = torch.tensor([elapsed, float(inter), float(R)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed, inter_total, R_total = float(tmax[0]), float(tot[0]), float(tot[1])
    else:
        inter_total, R_total = float(inter), float(R)

    if rank == 0:
        ms_step = 1e3 * elapsed / a.steps
        value = inter_total * a.steps / elapsed

        def per_launch(p, k):
            return p.ms[k] / max(p.launches[k], 1) * 1e-3

        shf = 3 * M * 4
        # algorithmic bytes per launch of the batched kernels (K views), SURVEY 8(d) regime 1
        pre_bytes = K * P * ((12 + 12 + 16 + 4 + shf) + (4 + 4 + 4 + 3 + 4 + 4 + 52 + 8)) + P * 24
        gather_bytes = R * (48 + 1) + K * P * (4 + 4 + 48)   # instance rows + flags in; tiles_touched, offsets in; 48-B sums out
        chain_bytes = K * P * (48 + 4 + 3 + 12 + 24 + shf + 64 + 4 * M + 12) + P * (K * (64 + 4 * M) + 28 + (12 + 4 + 24 + shf + 12 + 16))
        scat_bytes = K * P * 8 * 8 + R * 4                   # the 8-byte rectangles, read by the 8 tile-row classes; one id per instance out
        sort_bytes = R * (4 + 4 + 4 + 1)                     # ids in, gathered depths, sorted ids out, cleared flags
        t_bwd, t_fwd = per_launch(prof, "render_bwd"), per_launch(prof, "render_fwd")
        ach = FLOP_BWD * inter / t_bwd / 1e12 if t_bwd > 0 else 0.0
        pmc = pmc_summary()
        bwd_pmc = (pmc or {}).get("k_render_bwd", {})
        pmc_ok = bool(bwd_pmc) and bwd_pmc.get("csrc_sha1") == csrc_hash()

        def hbm(name, stage, nbytes, p=prof, launches=1):
            tl = per_launch(p, stage) * launches
            return {"bound": "hbm", "achieved": nbytes / tl / 1e9 if tl > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": nbytes / tl / 1e9 / PEAK_HBM_GBS if tl > 0 else 0.0, "avg_launch_ms": tl * 1e3, "algorithmic_bytes": nbytes,
                    "traffic": traffic_of((pmc or {}).get(name))}

        t1_bwd, t1_fwd = per_launch(prof1, "render_bwd"), per_launch(prof1, "render_fwd")
        out = {
            "metric": "Gaussian-pixel interactions/sec (fwd+Jacobian), 640x480",
            "value": value, "unit": "interactions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "priming_steps": PRIMING_STEPS,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32" if a.records == "fp32" else "f32 (the compositors' per-Gaussian rows store conic / opacity / colour as f16; mean2D, depth, every accumulation f32)",
            "data": "synthetic", "ranks_seen": ranks_seen,
            "config": {"workload": "%s: %d Gaussians (SH degree %d, %d coeffs), %dx%d; one step = one mapping window of %d keyframes with "
                                   "distinct cameras: forward splat + analytical-Jacobian backward (dL/dmu, dL/dSigma->conic, per-Gaussian "
                                   "grads summed over the window, one dL/dtau per keyframe)" % (a.workload, P, deg, M, W, H, K * (world if strong else 1)),
                       "views_per_step_per_rank": K, "views_per_step_all_ranks": K * world, "ms_per_frame": ms_step / (K * (world if strong else 1)),
                       "interactions_per_step_rank0": inter, "num_rendered_per_step_rank0": R,
                       "parallelism": ("%s: %d keyframes per GPU (%s); ONE async all-reduce of a %d-float bucket per step (per-Gaussian grads "
                                       "+ pose-gradient rows), overlapped with the next step"
                                       % (a.scaling, K, "one window dealt over the ranks" if strong else "one window per rank", ctx.buckets[0].numel())) if world > 1
                       else "single GPU, batched launches (grids x views), the window's %d views in %d group(s) on %d HIP stream(s)" % (K, a.streams, a.streams)},
            "roofline": {"bound": "valu", "kernel": "k_render_bwd", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_FP32_TFLOPS, "traffic": traffic_of(bwd_pmc),
                         "traffic_unit": "bytes/launch (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes; algorithmic: %d)" % (R * (4 + 48 + 48 + 1) + K * W * H * 32),
                         "traffic_source": {"file": PMC_FILE, "csrc_sha1_of_file": bwd_pmc.get("csrc_sha1"), "csrc_sha1_running": csrc_hash(),
                                            "quoted": pmc_ok, "note": "PMC counters cannot be read inside the timed process: they come from the committed "
                                            "summary and are quoted only when its fingerprint equals the kernel sources being run"},
                         "executed_valu_util": valu_util(bwd_pmc, t_bwd),
                         "inst_mix": inst_mix(bwd_pmc),
                         "note": "fp32 VALU / transcendental-bound reverse compositor (not MFMA, not HBM): NOTIONAL fraction = 87 fp32 flop x "
                                 "interactions per launch / HIP-event launch time / fp32 vector peak; culling skips most lane-operations, so "
                                 "executed_valu_util is the pipe utilisation as a RANGE: SQ_INSTS_VALU (PMC, same kernel sources) x [2.8, 4] cycles per "
                                 "wave-instruction (2.8: plain fma / mul / add measured by tools/op_cost.hip, 4: every other VALU form and the "
                                 "guide's single-wave issue cost) / (launch time x 2.4 GHz x 1024 SIMDs)",
                         "avg_launch_ms": t_bwd * 1e3, "launch_covers_views": K},
            "roofline_other": {
                "k_render_fwd": {"bound": "valu", "achieved": FLOP_FWD * inter / t_fwd / 1e12 if t_fwd > 0 else 0.0, "peak": PEAK_FP32_TFLOPS,
                                 "unit": "TFLOP/s", "frac": FLOP_FWD * inter / t_fwd / 1e12 / PEAK_FP32_TFLOPS if t_fwd > 0 else 0.0,
                                 "avg_launch_ms": t_fwd * 1e3, "executed_valu_util": valu_util((pmc or {}).get("k_render_fwd"), t_fwd)},
                "k_preprocess+k_frame_scan": hbm("k_preprocess", "preprocess", pre_bytes),
                "k_scatter_instances": hbm("k_scatter_instances", "scatter_instances", scat_bytes),
                "k_tile_sort": hbm("k_tile_sort", "tile_sort_records", sort_bytes),
                "k_gather_sums": hbm("k_gather_sums", "gather_sums", gather_bytes),
                "k_chain_window+k_tau_sum": dict(hbm("k_chain_window", "gaussian_bwd", chain_bytes),
                                                 note="timed as one stage: the per-(view, Gaussian) chain with the sum over views (one launch per 8 views) and the fp64 dL/dtau tree")},
            "stage_ms_per_step": {k: v / a.steps for k, v in prof.ms.items() if prof.launches[k]},
            "non_compositor_ms_per_step": sum(v for k, v in prof.ms.items() if prof.launches[k] and k not in ("render_fwd", "render_bwd")) / a.steps,
            "single_stream": None if not n_single else {"ms_per_frame": 1e3 * elapsed_single / n_single, "value_rank0": inter_1 * n_single / elapsed_single,
                              "stage_ms_per_frame": {k: v / n_single for k, v in prof1.ms.items() if prof1.launches[k]},
                              "k_render_bwd_frac": FLOP_BWD * inter_1 / t1_bwd / 1e12 / PEAK_FP32_TFLOPS if t1_bwd > 0 else 0.0,
                              "k_render_fwd_frac": FLOP_FWD * inter_1 / t1_fwd / 1e12 / PEAK_FP32_TFLOPS if t1_fwd > 0 else 0.0,
                              "note": "the window's keyframes one at a time through the single-view entry points on one HIP stream (sequential "
                                      "tracking iterations), no host sync per frame"},
            "selfcheck": selfcheck,
            "gradient_exchange": comm,
            "host_enqueue_ms_per_step": 1e3 * t_enqueued / a.steps,
            "csrc_sha1": csrc_hash(),
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cams[0], sc, deg, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def pmc_summary():
    """The committed rocprofv3 PMC summary (PMC_FILE: FETCH_SIZE / WRITE_SIZE in KB per launch, each from its own --pmc pass,
    and the SQ instruction / cycle counters), or None.  PMC counters cannot be read from inside the timed process."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def valu_util(k, t_launch):
    """VALU issue utilisation of one launch as a range [at 2.8 cycles, at 4 cycles per wave-instruction], from the committed
    PMC summary (only for the same kernel sources)."""
    if not k or k.get("csrc_sha1") != csrc_hash() or "SQ_INSTS_VALU" not in k or t_launch <= 0:
        return None
    return [k["SQ_INSTS_VALU"] * c / (t_launch * 2.4e9 * 1024.0) for c in (2.8, 4.0)]


def inst_mix(k):
    """Wave-instructions per launch by kind (PMC), for reading executed_valu_util."""
    if not k or k.get("csrc_sha1") != csrc_hash():
        return None
    return {n: k.get("SQ_INSTS_" + n) for n in ("VALU", "SALU", "LDS", "VMEM_RD", "VMEM_WR", "SMEM")}


def traffic_of(k):
    """Memory-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE: FETCH_SIZE doubled for 16-B/lane streaming reads as
    MI355X_MICROARCH.md prescribes for gfx950) -- only if the summary was collected from the kernel sources being run."""
    if not k or k.get("csrc_sha1") != csrc_hash() or "FETCH_SIZE" not in k:
        return None
    return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0


def cpu_baseline(cam, sc, deg, budget_s):
    """The CPU oracle (a C port of the reference's rasteriser semantics) on whole frames of the same workload, forward +
    backward: on ONE host core and on ALL host cores (tiles are independent: OpenMP over tiles / Gaussians, results identical),
    and the dense NumPy-semantics oracle (every Gaussian at every pixel) at N = 15 / 640x480 and N = 256 / 160x120."""
    from oracle import dense_oracle as dor
    from oracle import oracle as orc

    W, H = cam["W"], cam["H"]
    rng = np.random.default_rng(1234)
    dLc = (rng.normal(size=(3, H, W)) / (3 * H * W)).astype(np.float32)
    dLd = (rng.normal(size=(1, H, W)) / (H * W)).astype(np.float32)
    ncpu = os.cpu_count() or 1
    # "all cores" = the host cores this job may use: one GPU's share of the node is 16 (more OpenMP threads than that only
    # oversubscribe them: 256 threads measured 1.3x ONE thread); GSAJ_CPU_THREADS overrides
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = ncpu
    nthreads = int(os.environ.get("GSAJ_CPU_THREADS", min(16, avail, ncpu)))

    def tiled(threads, budget):
        used = orc.set_threads(threads)
        reps, inter, t0 = 0, 0, time.perf_counter()
        while True:
            out, st = orc.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], cam["tanfovx"],
                                  cam["tanfovy"], W, H, np.zeros(3, np.float32), shs=sc["shs"], scales=sc["scales"],
                                  rotations=sc["rotations"], sh_degree=deg)
            orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
            reps += 1
            inter += st["interactions"]
            el = time.perf_counter() - t0
            if el >= budget or reps >= 64:
                break
        orc.set_threads(1)
        return {"value": inter / el, "unit": "interactions/s", "cores": used, "kind": "port",
                "sample": "%d full frame(s) of the same workload (forward + backward), %.1f s on %d of %d host cores" % (reps, el, used, ncpu)}

    def dense(N, w, h):
        c = dict(cam)
        m2 = np.stack([rng.uniform(0, w, N), rng.uniform(0, h, N)], 1)
        A = rng.normal(size=(N, 2, 2))
        c2 = A @ A.transpose(0, 2, 1) * (0.02 * w) ** 2 + 4.0 * np.eye(2)
        col, dep, op = rng.uniform(0, 1, (N, 3)), np.sort(rng.uniform(1, 4, N)), rng.uniform(0.3, 0.9, N)
        gc = rng.choice([-1.0, 0.0, 1.0], size=(h, w, 3)).astype(np.float32)
        gd = rng.choice([-1.0, 0.0, 1.0], size=(h, w)).astype(np.float32)
        t0 = time.perf_counter()
        dor.dense_backward(m2, c2, col, dep, op, gc, gd)
        el = time.perf_counter() - t0
        del c
        return {"value": N * w * h / el, "unit": "Gaussian-pixel pairs/s (dense semantics, backward)", "cores": 1,
                "sample": "N=%d at %dx%d, one pass, %.1f s (NumPy elementwise: one core)" % (N, w, h, el)}

    res = tiled(nthreads, 0.35 * budget_s)     # primary: all the host cores of this job
    res["single_core"] = tiled(1, 0.35 * budget_s)
    res["dense_mode"] = {"N15_640x480": dense(15, 640, 480), "N256_160x120": dense(256, 160, 120)}
    return res


if __name__ == "__main__":
    sys.exit(main())
