"""Summarise a rocprofv3 kernel_stats.csv: per-kernel ms per bench step."""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print("GPU busy per step: %.2f ms" % tot)
for r in rows[:40]:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    if ms < 0.15: break
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print("%-72s %6d calls/step %8.3f ms  avg %8.1f us" % (name, int(r["Calls"]) / steps, ms, float(r["AverageNs"]) / 1e3))
