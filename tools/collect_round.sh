#!/bin/bash
# GPU box, repo root: everything profiles/ holds for a round, in one call.  usage: tools/collect_round.sh r03
# -> gpurun_out/<tag>/: bench lines of every named config, rocprofv3 kernel stats + PMC passes of the default bench command
#    (tools/pmc_collect.sh), per-wave traces, tracking / mapping iteration times.  Copy what is to be judged into profiles/.
set -e
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err
for w in cfg3 cfg4 cfg5; do timeout -k 10 300 python bench.py --workload $w --steps 20 --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err; done
timeout -k 10 300 python bench.py --workload cfg5 --records fp16 --steps 20 --no-cpu-baseline > $out/bench_cfg5_fp16.json 2> $out/bench_cfg5_fp16.err
GSAJ_SHARE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline --skip-single > $out/bench_2rank_gloo_shared_gpu_weak.txt 2>&1 || true
GSAJ_SHARE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --scaling strong --steps 5 --warmup 1 --no-cpu-baseline --skip-single > $out/bench_2rank_gloo_shared_gpu_strong.txt 2>&1 || true
bash tools/pmc_collect.sh $out/pmc > $out/pmc.log 2>&1
timeout -k 10 200 python tools/device_tracker_bench.py > $out/device_tracker.json 2> $out/device_tracker.err
timeout -k 10 200 python tools/mapping_iter_bench.py > $out/mapping_iter.json 2> $out/mapping_iter.err || true
if [ -f gs-slam-analytica_jacobian_amd/lib/trace/libgsaj_hip.so ]; then
  GSAJ_LIB_PATH=$GRAFT_REPO_ROOT/gs-slam-analytica_jacobian_amd/lib/trace/libgsaj_hip.so timeout -k 10 200 python tools/batch_trace.py 8 cfg2 > $out/batch_trace_cfg2.txt 2>&1 || true
fi
GSAJ_ERRLOG=1 timeout -k 10 600 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests.log
mv gpurun_out/parity_errors.jsonl $out/ 2>/dev/null || true
tail -3 $out/gpu_tests.log
python3 - $out <<'PY'
import json, sys
o = sys.argv[1]
for n in ("bench", "bench_cfg3", "bench_cfg4", "bench_cfg5", "bench_cfg5_fp16"):
    try:
        d = json.loads(open("%s/%s.json" % (o, n)).read().strip().splitlines()[-1])
        print(n, "ms/step %.4f" % d["ms_per_step"], "value %.3e" % d["value"], {k: round(v * 1e3) for k, v in d["stage_ms_per_step"].items()},
              "single %.4f" % d["single_stream"]["ms_per_frame"])
    except Exception as e:
        print(n, "ERR", e)
PY
