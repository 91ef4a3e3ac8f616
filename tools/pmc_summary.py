"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<name>.json.

usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: both counters are in
KiB; FETCH_SIZE counts half the bytes of a coalesced stream (read bytes = 2 x FETCH_SIZE).
"""
import collections
import csv
import json
import re
import sys


def load(path, counter):
    acc = collections.defaultdict(list)
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        key = (r.get("Dispatch_Id"), r["Kernel_Name"])
        per_dispatch[key] += float(r["Counter_Value"])     # one row per XCD/instance
        names[key] = r["Kernel_Name"]
    for key, v in per_dispatch.items():
        acc[names[key]].append(v)
    return acc


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (python bench.py "
               "--steps 2 --warmup 0 --no-cpu-baseline), per launch, MI355X. Both counters are in KiB. "
               "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes of a "
               "coalesced stream -> read bytes = 2*FETCH_SIZE (calibrated on the Jacobian SpMV, whose "
               "reads are known); WRITE_SIZE is exact.",
       "kernels": {}}
for name in sorted(set(fetch) | set(write)):
    f, w = fetch.get(name, []), write.get(name, [])
    if not f or not w or "fedm::" not in name:
        continue
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    out["kernels"][short(name)] = {"FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa,
                                   "launches_sampled": min(len(f), len(w)),
                                   "traffic_bytes_corrected": (2.0 * fa + wa) * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_corrected"])[:12]:
    print(f'{v["traffic_bytes_corrected"] / 1e6:10.2f} MB  n={v["launches_sampled"]:4d}  {k}')
