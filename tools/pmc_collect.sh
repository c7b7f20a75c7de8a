#!/bin/bash
# GPU box, repo root: rocprofv3 kernel stats + PMC passes over the default bench command -> profiles-ready summaries.
# usage: tools/pmc_collect.sh OUTDIR [bench args]     (each --pmc pass is its own run: kernel-trace only, no other trace domain)
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out" && rm -rf "$out"/*
B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --skip-single $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o run --output-format csv -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --skip-single "$@" > "$out/bench_under_rocprof.json" 2> "$out/rocprof.err"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmcF" -o run --output-format csv -- $B > /dev/null 2> "$out/pmcF.err"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmcW" -o run --output-format csv -- $B > /dev/null 2> "$out/pmcW.err"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d "$out/pmcA" -o run --output-format csv -- $B > /dev/null 2> "$out/pmcA.err"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM -d "$out/pmcB" -o run --output-format csv -- $B > /dev/null 2> "$out/pmcB.err"
python3 tools/pmc_summary.py "$out/pmc_summary.json" "$out/pmcF" "$out/pmcW" "$out/pmcA" "$out/pmcB" > "$out/pmc_summary.txt"
find "$out/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
head -14 "$out/kernel_stats.csv" | cut -c1-150
