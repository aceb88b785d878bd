"""gpurun_out/rec/ (written by tools/record_set.sh on the GPU box) -> profiles/r03_* (tracked).  Run in the build container after the call."""
import glob, json, shutil, sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
O, P = R / "gpurun_out" / "rec", R / "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
def stats(d):
    import os
    f = glob.glob(str(O / d / "**" / "*kernel_stats.csv"), recursive=True)   # (gpurun MERGES into gpurun_out/: older calls' files may still lie there)
    return max(f, key=os.path.getmtime) if f else None
copies = {"pmc_hbm.json": "%s_pmc_hbm.json", "bench_line.json": "%s_bench_line.json", "per_layer.txt": "%s_det_per_layer.txt",
          "config2_line.json": "%s_config2_det_only_line.json", "config3_line.json": "%s_config3_rec_only_line.json", "config5_line.json": "%s_config5_svtr_base_f16_line.json"}
for src, dst in copies.items():
    shutil.copy(O / src, P / (dst % tag))
for d, dst in {"ks": "%s_bench_kernel_stats.csv", "ks_c2": "%s_config2_kernel_stats.csv", "ks_c3": "%s_config3_kernel_stats.csv", "ks_c5": "%s_config5_kernel_stats.csv",
               "dk": "%s_deskew_kernel_stats.csv", "jd": "%s_jpegdec_kernel_stats.csv"}.items():
    s = stats(d)
    if s:
        shutil.copy(s, P / (dst % tag))
jd = [l for l in (O / "jd.log").read_text().splitlines() if l.startswith(("decode", "one page"))]
(P / ("%s_jpegdec_probe.txt" % tag)).write_text("\n".join(jd) + "\n")
b = json.loads((O / "bench_line.json").read_text())
print(tag, "bench", b["value"], "pages/s; dominant", b["roofline"]["kernel"], b["roofline"]["frac"], "family", b["roofline"]["family"]["frac"], "traffic", b["roofline"]["traffic"])
