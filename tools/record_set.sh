set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rec; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/pmc_target.py 64 > $O/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/pmc_target.py 64 > $O/write.log 2>&1 &&
cd $R && python tools/pmc_to_json.py profiles/r02_pmc_hbm.json $O/fetch $O/write > $O/pmc.txt 2>&1 && cp profiles/r02_pmc_hbm.json $O/ &&
timeout -k 10 400 python bench.py > $O/bench.log 2>&1 && tail -1 $O/bench.log > $O/bench_line.json &&
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/bench.py --no-cpu-baseline > $O/ks.log 2>&1 &&
cd $R && timeout -k 10 120 python tools/perf_probe.py 16 2000 1414 16 > $O/per_layer.txt 2>/dev/null &&
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dk -- python3 $R/tools/deskew_probe.py > $O/dk.log 2>&1
echo rc=$?
tail -3 $O/pmc.txt; cut -c1-220 $O/bench_line.json
