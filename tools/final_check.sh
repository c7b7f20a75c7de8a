#!/bin/bash
# GPU box, repo root: what the driver runs at round end -- smoke() and the bench with its flags -- and the fields it reads.
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/final_20_5.json 2> gpurun_out/final_20_5.err
python tools/ab_show.py gpurun_out/final_20_5.json
python - <<'PY'
import json
b=json.loads(open("gpurun_out/final_20_5.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("metric","value","unit","n_gpus","steps","warmup","priming_steps","ms_per_step","scaling","vs_baseline","dtype","data","ranks_seen")})
print(b["roofline"]["frac"], b["roofline"]["traffic"], b["roofline"]["executed_valu_util"], b["cpu_baseline"]["value"], b["cpu_baseline"]["kind"], b["cpu_baseline"]["cores"])
PY
