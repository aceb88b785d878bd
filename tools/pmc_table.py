"""Print per-kernel means of every counter found under the given rocprofv3 --pmc output directories."""
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            agg[name.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "conv_ring" not in k and "conv_mfma" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-40s n=%3d mean %.4g" % (c, len(v), sum(v) / len(v)))
