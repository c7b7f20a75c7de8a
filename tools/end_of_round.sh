#!/bin/bash
# GPU box, repo root: full GPU test suite, smoke, default bench, rocprofv3 kernel stats and HBM-traffic PMC passes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/eor && rm -rf gpurun_out/eor/*
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/eor/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/eor/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/eor/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/eor/smoke.log 2>&1 || { tail -20 gpurun_out/eor/smoke.log; exit 1; }
tail -1 gpurun_out/eor/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/eor/bench.json 2> gpurun_out/eor/bench.err
# one frame at a time, so that the per-kernel durations are comparable with bench.py's HIP-event stage times
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/eor/stats -o run --output-format csv -- python3 bench.py --no-cpu-baseline --frames-in-flight 1 > gpurun_out/eor/bench_under_rocprof.json 2> gpurun_out/eor/rocprof.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/eor/pmcF -o run --output-format csv -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2> gpurun_out/eor/pmcF.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/eor/pmcW -o run --output-format csv -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2> gpurun_out/eor/pmcW.err
python3 tools/pmc_summary.py gpurun_out/eor/pmc_hbm.json gpurun_out/eor/pmcF gpurun_out/eor/pmcW > /dev/null
timeout -k 10 400 tools/pmc_run.sh gpurun_out/eor/pmc_sq.json > gpurun_out/eor/pmc_sq.txt 2>&1
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/eor/bench.json"))
print("bench: ms/step %.4f value %.3e single %.4f roofline frac %.3f cpu %.3e" % (d["ms_per_step"], d["value"], d["single_stream"]["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"]))
PY
find gpurun_out/eor/stats -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-160
