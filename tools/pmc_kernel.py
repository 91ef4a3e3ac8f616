"""Average the counters of one kernel from rocprofv3 --pmc counter_collection CSVs."""
import collections, csv, sys
pat = sys.argv[1]
for path in sys.argv[2:]:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in sorted(per.items()):
        v = list(d.values())
        print(f"{c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
