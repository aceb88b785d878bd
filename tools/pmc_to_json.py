"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/<name>.json: mean HBM bytes per launch per kernel.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half of the bytes of wide (16 B/lane) streaming
reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores. Both counters are in KiB."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict
out, dirs = sys.argv[1], sys.argv[2:]
agg = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in agg.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        fe, wr = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        res[k] = dict(launches=len(cs["FETCH_SIZE"]), fetch_kib_raw=fe, write_kib=wr, hbm_bytes_per_launch=(2 * fe + wr) * 1024)
# bench.py reports `roofline.traffic` from this file only while the kernel sources it was measured on are the ones that run
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocr-system_amd", "csrc")
shas = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16] for f in ("conv_ring.hip", "conv_mfma.hip", "conv_pw.hip", "engine.hip")}
json.dump({"source_sha16": shas, "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print("%-80s launches %4d  HBM/launch %.1f MB" % (k[:80], v["launches"], v["hbm_bytes_per_launch"] / 1e6))
