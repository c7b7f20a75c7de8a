#!/bin/bash
# container, repo root: copy what tools/collect_round.sh left under gpurun_out/<tag>/ into profiles/r03_* (the files the judge reads).
# usage: tools/adopt_round.sh r03j     (then run `python bench.py` once more on the GPU box and copy its line to profiles/r03_bench.json,
# so that the line quotes the PMC summary taken from the same kernel sources)
set -e
o=gpurun_out/${1:?tag}
for f in bench.json bench_cfg3.json bench_cfg4.json bench_cfg5.json bench_cfg5_fp16.json bench_2rank_gloo_shared_gpu_strong.txt \
         bench_2rank_gloo_shared_gpu_weak.txt device_tracker.json mapping_iter.json; do cp $o/$f profiles/r03_$f; done
cp $o/pmc/bench_under_rocprof.json profiles/r03_bench_under_rocprof.json
cp $o/pmc/kernel_stats.csv profiles/r03_kernel_stats.csv
cp $o/pmc/pmc_summary.json profiles/r03_pmc_summary.json
[ -f $o/batch_trace_cfg2.txt ] && cp $o/batch_trace_cfg2.txt profiles/r03_batch_trace.txt
python tools/parity_summary.py $o/parity_errors.jsonl profiles/r03_parity_errors.json
