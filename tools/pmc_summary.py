#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel name.

usage: pmc_summary.py OUT.json DIR [DIR ...]   (each DIR holds *_counter_collection.csv of one --pmc pass)
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"].split("(")[0]
                    a = acc[name][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    res = {k.replace("void ", "").split("<")[0]: {c: v[0] / v[1] for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())
           if k.startswith(("k_", "void k_"))}
    # fingerprint of the kernel sources these counters were collected from (bench.py quotes them only for the same sources)
    import hashlib
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gs-slam-analytica_jacobian_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    for k, c in res.items():
        c["csrc_sha1"] = h.hexdigest()[:16]
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k, cs in res.items():
        print(k, {c: (round(v, 3) if isinstance(v, float) else v) for c, v in cs.items()})


if __name__ == "__main__":
    main()
