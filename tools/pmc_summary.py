"""Aggregate rocprofv3 counter_collection.csv files per kernel: mean counter value per launch."""
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    n = len(next(iter(cs.values())))
    print("%-62s launches %4d  " % (name, n) + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
