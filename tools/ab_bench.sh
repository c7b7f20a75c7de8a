#!/bin/bash
# usage (GPU box): tools/ab_bench.sh name1 name2 ...   -- benches tools/ab_<name>.so variants, two runs each
for rep in 1 2; do
for n in "$@"; do
  GSAJ_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab_$n.so timeout -k 10 120 python bench.py --no-cpu-baseline --steps 480 > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/ab_$n.err; exit 1; }
  python - "$n" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
st = d["stage_ms_per_step"]
print("%-12s S3 %.4f  single %.4f  fwd %.1f bwd %.1f gbwd %.1f" % (sys.argv[1], d["ms_per_step"], d["single_stream"]["ms_per_step"], st["render_fwd"]*1e3, st["render_bwd"]*1e3, st["gaussian_bwd"]*1e3))
PY
done; done
