#!/usr/bin/env python3
"""Benchmark of the hot path: forward splat + analytical-Jacobian backward of one 640x480 frame
(BASELINE.json metric: Gaussian-pixel interactions / second, fwd + Jacobian).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one frame per rank: preprocess -> (tile, depth) sort ->
composite -> reverse composite -> per-Gaussian backward incl. dL/dtau, all through the C ABI of
libgsaj_hip.so, with every input already resident in HBM.  For N > 1 every rank owns one keyframe
of the same (replicated) Gaussian map (weak scaling) and the step ends with the RCCL all-reduce of
the per-Gaussian gradient bucket and the all-gather of the per-keyframe dL/dtau.
An interaction = one (pixel, Gaussian) pair the compositor visits: I = sum over pixels of n_contrib
(SURVEY 8d).  Rank 0 prints ONE JSON line.
"""
import argparse
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
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_FWD, FLOP_BWD = 24.0, 87.0  # fp32 flop per interaction (+ 1 exp fwd, 1 exp + 1 rcp bwd)
PEAK_FP32_TFLOPS = 157.3         # MI355X fp32 vector = fp32 MFMA dense peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync", action="store_true", help="read the instance count back on the host every frame")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="independent frames (keyframes of a mapping window) processed concurrently, one HIP stream each")
    ap.add_argument("--records", default="fp32", choices=["fp32", "fp16"],
                    help="storage of the sorted instance records (fp16: conic / opacity / colour as halves, config 5)")
    ap.add_argument("--sh-coeffs", type=int, default=0, help="experiment: keep only the first N SH coefficients per Gaussian")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU product path)"
    if os.environ.get("GSAJ_SHARE_DEVICE") == "1":  # rehearsal on a 1-GPU box: every rank on cuda:0 (gloo only)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("GSAJ_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == a.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    from gsaj import synthetic as syn
    from gsaj import keyframe_shard as ks
    from gsaj.rasterizer import FrameContext, profile_stages, set_record_format

    set_record_format(16 if a.records == "fp16" else 32)

    cam0, sc = syn.config_scene(a.workload)
    if world > 1:  # one keyframe per rank, on a 0.5 m arc around the cfg camera (cfg4-style window)
        cam = syn.keyframe_cameras(world, W=cam0["W"], H=cam0["H"], fx=cam0["fx"], fy=cam0["fy"], cx=cam0["cx"],
                                   cy=cam0["cy"])[rank]
    else:
        cam = cam0
    if a.sh_coeffs:
        sc["shs"] = np.ascontiguousarray(sc["shs"][:, : a.sh_coeffs])
    P, W, H = sc["means3D"].shape[0], cam["W"], cam["H"]
    M = sc["shs"].shape[1]
    deg = min(a.sh_degree, int(round(M ** 0.5)) - 1)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)  # noqa: E731
    means, opac, shs, scales, rots = t(sc["means3D"]), t(sc["opacities"]), t(sc["shs"]), t(sc["scales"]), t(sc["rotations"])
    view, proj, proj_raw, campos = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["projmatrix_raw"]), t(cam["campos"])
    bg = torch.zeros(3, device=dev)
    rng = np.random.default_rng(1234 + rank)
    dLc = t(rng.normal(size=(3, H, W)) / (3 * H * W))  # pixel-gradient seeds, resident in HBM
    dLd = t(rng.normal(size=(1, H, W)) / (H * W))
    # two gradient buckets: the all-reduce of step i (RCCL stream) overlaps the kernels of step i+1
    S = max(1, a.frames_in_flight)
    ctxs = [FrameContext(P, W, H, M, dev, grad_slots=2 if world > 1 else 1, n_keyframes=world if world > 1 else 0,
                         keyframe=rank) for _ in range(S)]
    ctx = ctxs[0]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    pending = [[None, None] for _ in range(S)]
    counter, seen = [0], [0] * S

    def step(single=False):
        k = 0 if single else counter[0] % S  # frame slot: its own workspaces, gradient buckets and HIP stream
        n = seen[k]                 # how many frames this slot has processed
        seen[k] += 1
        counter[0] += 1
        slot = n & 1 if world > 1 else 0
        c, pend = ctxs[k], pending[k]
        with torch.cuda.stream(streams[k]):
            if pend[slot] is not None:
                pend[slot].wait()  # stream-level: the bucket is about to be overwritten
                pend[slot] = None
            # first frame: synchronous (sizes the binning arena); afterwards no host round trip per frame
            c.forward(bg, means, opac, view, proj, campos, cam["tanfovx"], cam["tanfovy"], sh_degree=deg, shs=shs,
                      scales=scales, rotations=rots, sync=(n == 0) or a.sync)
            c.backward(bg, means, view, proj, proj_raw, campos, cam["tanfovx"], cam["tanfovy"], dLc, dLd,
                       sh_degree=deg, shs=shs, scales=scales, rotations=rots, slot=slot)
            if world > 1:  # one collective: per-Gaussian grads summed, per-keyframe dL/dtau gathered (bucket tail)
                pend[slot] = ks.allreduce_gaussian_grads(c.buckets[slot], async_op=True)

    def fence():
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                for i in range(2):
                    if pending[k][i] is not None:
                        pending[k][i].wait()
                        pending[k][i] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    a.warmup = max(a.warmup, S)  # every frame slot sizes its binning arena on its first (synchronous) frame
    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_enqueued = time.perf_counter() - t0  # host time to enqueue the K steps (the GPU runs behind it)
    fence()
    elapsed = time.perf_counter() - t0
    for c in ctxs:
        R, _ = c.status()  # raises if any asynchronous frame was aborted on the device (arena too small)
    inter = ctx.interactions()

    # the same K steps one frame at a time on one stream: per-frame latency as the tracker sees it
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(single=True)
    fence()
    elapsed_single = time.perf_counter() - t0

    # and once more with every kernel bracketed by HIP events on its launch stream
    fence()
    with profile_stages(max_records=a.steps * 16) as prof:
        for _ in range(a.steps):
            step(single=True)
    fence()

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
        n = max(prof.launches["render_bwd"], 1)
        t_bwd = prof.ms["render_bwd"] / n * 1e-3
        t_fwd = prof.ms["render_fwd"] / max(prof.launches["render_fwd"], 1) * 1e-3
        t_gb = prof.ms["gaussian_bwd"] / max(prof.launches["gaussian_bwd"], 1) * 1e-3
        t_pre = prof.ms["preprocess"] / max(prof.launches["preprocess"], 1) * 1e-3
        pre_bytes = P * ((12 + 4 + 12 + 16 + shf_in(M)) + (4 + 8 + 24 + 16 + 12 + 3 + 4 + 4 + 4 + 48))
        ach = FLOP_BWD * inter / t_bwd / 1e12 if t_bwd > 0 else 0.0
        shf = 3 * M * 4
        gb_bytes = P * ((12 + 24 + 4 + 16 + 12 + shf + 12 + 16 + 3) + (12 + 16 + 4 + 12 + 4 + 12 + 24 + shf + 12 + 16)) + R * (48 + 4)
        out = {
            "metric": "Gaussian-pixel interactions/sec (fwd+Jacobian), 640x480",
            "value": value, "unit": "interactions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.records == "fp32" else "f32 (records stored as f16)", "data": "synthetic",
            "config": {"workload": "%s: %d Gaussians (SH degree %d, %d coeffs), %dx%d, forward splat + analytical "
                                   "Jacobian backward (dL/dmu, dL/dSigma->conic, per-Gaussian grads, dL/dtau)"
                                   % (a.workload, P, deg, M, W, H),
                       "interactions_per_frame_rank0": inter, "num_rendered_rank0": R, "frames_in_flight": S,
                       "parallelism": ("%d independent frames in flight per GPU, one HIP stream each; " % S) + "1 keyframe per GPU; ONE async all-reduce of a %d-float bucket (per-Gaussian grads + per-keyframe dL/dtau rows), overlapped with the next step"
                                      % ctx.buckets[0].numel() if world > 1 else "single GPU; %d independent frames in flight, one HIP stream each" % S},
            "roofline": {"bound": "mfma", "kernel": "k_render_bwd", "achieved": ach, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                         "traffic": pmc_traffic("k_render_bwd") if (a.workload == "cfg2" and M == 16) else None,
                         "traffic_unit": "bytes/launch (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, "
                                         "profiles/r01_pmc_hbm_fetch_write_kb.json; algorithmic: %d)" % (R * 96 + W * H * 32),
                         "note": "fp32 VALU/transcendental-bound reverse compositor: 87 fp32 flop x interactions per "
                                 "launch / HIP-event launch time; peak = fp32 vector = fp32 MFMA dense peak",
                         "avg_launch_ms": t_bwd * 1e3},
            "roofline_other": {
                "k_render_fwd": {"bound": "mfma", "achieved": FLOP_FWD * inter / t_fwd / 1e12 if t_fwd > 0 else 0.0,
                                 "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "avg_launch_ms": t_fwd * 1e3},
                "k_preprocess": {"bound": "hbm", "achieved": pre_bytes / t_pre / 1e9 if t_pre > 0 else 0.0, "peak": PEAK_HBM_GBS,
                                 "unit": "GB/s", "avg_launch_ms": t_pre * 1e3, "algorithmic_bytes": pre_bytes},
                "k_gaussian_bwd": {"bound": "hbm", "achieved": gb_bytes / t_gb / 1e9 if t_gb > 0 else 0.0,
                                   "peak": PEAK_HBM_GBS, "unit": "GB/s", "avg_launch_ms": t_gb * 1e3,
                                   "algorithmic_bytes": gb_bytes}},
            "single_stream": {"ms_per_step": 1e3 * elapsed_single / a.steps, "value_rank0": inter * a.steps / elapsed_single,
                              "note": "same K steps, one frame at a time on one HIP stream (sequential tracking iterations)"},
            "host_enqueue_ms_per_step": 1e3 * t_enqueued / a.steps,
            "stage_ms_per_step": {k: v / a.steps for k, v in prof.ms.items() if prof.launches[k]},
        }
        for k in ("k_render_fwd", "k_preprocess", "k_gaussian_bwd"):
            o = out["roofline_other"][k]
            o["frac"] = o["achieved"] / o["peak"]
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cam, sc, deg, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def shf_in(M):
    """bytes of SH coefficients per Gaussian (fp32, 3 channels)."""
    return 3 * M * 4


def pmc_traffic(kernel):
    """Memory-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes over this same
    command (FETCH_SIZE and WRITE_SIZE, in KB, each collected in its own --pmc run; FETCH_SIZE doubled
    for 16-B/lane streaming reads as MI355X_MICROARCH.md prescribes for gfx950).  None if the summary
    is not in the tree: PMC counters cannot be read from inside the timed process."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_fetch_write_kb.json")
    try:
        with open(path) as f:
            k = json.load(f)[kernel]
        return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(cam, sc, deg, budget_s):
    """The CPU oracle (a C port of the reference's rasteriser semantics, single thread) on whole
    frames of the same workload, repeated until the budget is spent."""
    from oracle import oracle as orc

    W, H = cam["W"], cam["H"]
    rng = np.random.default_rng(1234)
    dLc = (rng.normal(size=(3, H, W)) / (3 * H * W)).astype(np.float32)
    dLd = (rng.normal(size=(1, H, W)) / (H * W)).astype(np.float32)
    reps, inter, t0 = 0, 0, time.perf_counter()
    while True:
        out, st = orc.forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                              cam["tanfovx"], cam["tanfovy"], W, H, np.zeros(3, np.float32), shs=sc["shs"],
                              scales=sc["scales"], rotations=sc["rotations"], sh_degree=deg)
        orc.backward(st, dLc, dLd, cam["projmatrix_raw"])
        reps += 1
        inter += st["interactions"]
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 8:
            break
    return {"value": inter / el, "unit": "interactions/s", "cores": 1, "kind": "port",
            "sample": "%d full frame(s) of the same workload (forward + backward), %.1f s on 1 of %d host cores"
                      % (reps, el, os.cpu_count())}


if __name__ == "__main__":
    main()
