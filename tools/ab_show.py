"""Print the few numbers of a bench.py JSON line that an A/B comparison looks at.  usage: python tools/ab_show.py FILE..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    st = {k: round(v * 1e3, 1) for k, v in d["stage_ms_per_step"].items()}
    one = d.get("single_stream") or {}
    print(f, "ms/step %.4f" % d["ms_per_step"], "value %.4e" % d["value"], st, "non-compositor %.3f" % d["non_compositor_ms_per_step"],
          "single %.4f" % one.get("ms_per_frame", 0.0), "selfcheck %s" % ("%.2e" % d["selfcheck"]["tau_max_rel"] if d.get("selfcheck") else "-"))
