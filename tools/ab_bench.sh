#!/bin/bash
# usage (GPU box): tools/ab_bench.sh name1 name2 ...   -- benches gs-slam-analytica_jacobian_amd/lib/<name>/libgsaj_hip.so variants
# ("base" = the regular library), two runs each; prints the per-stage times of the batched window (us per step).
for rep in 1 2; do
for n in "$@"; do
  lib=$GRAFT_REPO_ROOT/gs-slam-analytica_jacobian_amd/lib/$n/libgsaj_hip.so
  [ "$n" = base ] && lib=$GRAFT_REPO_ROOT/gs-slam-analytica_jacobian_amd/lib/libgsaj_hip.so
  GSAJ_LIB_PATH=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40 $BENCH_ARGS > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/ab_$n.err; exit 1; }
  python - "$n" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
st = d["stage_ms_per_step"]
print("%-10s step %.4f single %.4f | " % (sys.argv[1], d["ms_per_step"], (d["single_stream"] or {"ms_per_frame": 0.0})["ms_per_frame"]) + " ".join("%s %.0f" % (k[:12], v * 1e3) for k, v in st.items()))
PY
done; done
