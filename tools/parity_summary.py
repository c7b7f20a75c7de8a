#!/usr/bin/env python3
"""gpurun_out/parity_errors.jsonl (written by the -m gpu suite under GSAJ_ERRLOG=1, tests/helpers.py) -> the worst value of
every logged check and where it occurred.  usage: parity_summary.py [in.jsonl] [out.json]"""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_errors.jsonl"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_parity_errors.json"
worst, n = {}, 0
for line in open(src):
    r = json.loads(line)
    n += 1
    tag = r.pop("tag", "?")
    for k, v in r.items():
        if isinstance(v, (int, float)) and (k not in worst or v > worst[k]["value"]):
            worst[k] = {"value": v, "where": tag}
NOTE = ("worst value of every logged check over the whole -m gpu suite (GSAJ_ERRLOG=1 python -m pytest tests -m gpu); keys as in tests/helpers.py: "
        "<tensor> = max|err|/max|want| over all Gaussians, <tensor>/clean = over Gaussians no borderline pixel touches, <tensor>/chain_row = "
        "worst row of the per-Gaussian chain on the device's sums vs the fp64 chain (.../chain_row_oracle32: the oracle's own fp32 chain), "
        "<tensor>/chain_row_over_allowed = worst row error / its allowance (< 1 passes; allowance = max of 1e-4 x row scale, 16 x the row's "
        "sensitivity to +-1 ulp inputs, 4 x the oracle's sampled fp32 noise on that row), compositor/err_over_bound = worst |err| / error-model "
        "bound, */sum_of_bands/* = tile-band shares summed vs the whole frame, max_err / max_ok / n_bad / n_diff = image and n_contrib checks "
        "(max_err includes the looser fp16-vs-fp32 comparisons)")
json.dump({"n_records": n, "note": NOTE, "worst": dict(sorted(worst.items()))},
          open(dst, "w"), indent=1)
print(dst, n, "records,", len(worst), "checks")
