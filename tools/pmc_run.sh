#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh OUT.json [bench args]
# Two rocprofv3 --pmc passes (8 SQ counters each) over a short single-stream bench run, averaged per kernel.
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcA gpurun_out/pmcB
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d gpurun_out/pmcA -o runc --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames-in-flight 1 "$@" > gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM -d gpurun_out/pmcB -o runc --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames-in-flight 1 "$@" > gpurun_out/pmcB.log 2>&1
python3 tools/pmc_summary.py "$out" gpurun_out/pmcA gpurun_out/pmcB > /dev/null
python3 - "$out" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, c in d.items():
    print("%-26s VALU %6.2fM SALU %6.2fM LDS %5.2fM  wave_cyc %6.1fM active %5.1fM valu_act %5.1fM wait %5.1fM stall %5.1fM  ldsconf %5.2fM" % (
        k[:26], c["SQ_INSTS_VALU"]/1e6, c["SQ_INSTS_SALU"]/1e6, c["SQ_INSTS_LDS"]/1e6, c["SQ_WAVE_CYCLES"]/1e6, c["SQ_ACTIVE_INST_ANY"]/1e6,
        c["SQ_ACTIVE_INST_VALU"]/1e6, c["SQ_WAIT_ANY"]/1e6, c["SQ_WAIT_INST_ANY"]/1e6, c["SQ_LDS_BANK_CONFLICT"]/1e6))
PY
