# Runs ON THE GPU BOX (gpurun): the round's record set -> gpurun_out/rec/.  tools/collect_profiles.py then copies it into profiles/ (tracked).
# One call, one device: PMC passes (HBM bytes per kernel), the bench line + its rocprofv3 kernel stats, the per-layer table, the
# lines of BASELINE configs 2 / 3 / 5 with their kernel stats, de-skew and JPEG-decode kernel stats.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rec; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/pmc_target.py 64 > $O/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/pmc_target.py 64 > $O/write.log 2>&1 &&
cd $R && python tools/pmc_to_json.py $O/pmc_hbm.json $O/fetch $O/write > $O/pmc.txt 2>&1 && cp $O/pmc_hbm.json profiles/r03_pmc_hbm.json &&
timeout -k 10 500 python bench.py > $O/bench.log 2>&1 && tail -1 $O/bench.log > $O/bench_line.json &&
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $O/ks.log 2>&1 &&
cd $R && timeout -k 10 120 python tools/perf_probe.py 16 2000 1414 16 > $O/per_layer.txt 2>/dev/null &&
for c in 2 3 5; do
  timeout -k 10 200 python bench.py --config $c > $O/c$c.log 2>&1 && tail -1 $O/c$c.log > $O/config${c}_line.json &&
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c$c -- python3 $R/bench.py --config $c > $O/ks_c$c.log 2>&1) || exit 1
done &&
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dk -- python3 $R/tools/deskew_probe.py > $O/dk.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/jd -- python3 $R/tools/jpegdec_probe.py > $O/jd.log 2>&1
echo rc=$?
cd $R && python - <<'PY'
import json, hashlib, os
o = "gpurun_out/rec/"
j = json.load(open(o + "pmc_hbm.json"))
print("PMC source hashes:", j["source_sha16"])
for k in j["source_sha16"]:
    print("  now  ", k, hashlib.sha256(open("ocr-system_amd/csrc/" + k, "rb").read()).hexdigest()[:16])
b = json.loads(open(o + "bench_line.json").read())
print("bench:", b["value"], {k: b[k] for k in b if k.startswith("value_")}, "roofline", b["roofline"]["frac"], "family", b["roofline"]["family"]["frac"], "traffic", b["roofline"]["traffic"])
PY
