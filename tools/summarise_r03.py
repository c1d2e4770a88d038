"""Copy the summaries of tools/collect_r03.sh (gpurun_out/r3/final/, plus the issue-cost microbenchmark and the start-of-round counter
passes) into profiles/:  python tools/summarise_r03.py"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r3", "final")
DST = os.path.join(ROOT, "profiles")
for src, dst in (("bench_default.json", "r03_bench.json"), ("bench_a.json", "r03_bench_steps20_a.json"), ("bench_b.json", "r03_bench_steps20_b.json"),
                 ("bench_under_rocprof.json", "r03_bench_under_rocprof.json")):
    line = open(os.path.join(SRC, src)).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(DST, dst), "w").write(line + "\n")
for f in sorted(glob.glob(os.path.join(SRC, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]:
    shutil.copy(f, os.path.join(DST, "r03_bench_kernel_stats.csv"))
shutil.copy(os.path.join(ROOT, "gpurun_out", "r3", "ub_issue.txt"), os.path.join(DST, "r03_ub_issue.txt"))
# counter summaries: the kernel at the start of the round (tag base) and as shipped (tag final); issue costs from r03_ub_issue.txt at three waves per SIMD
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarise_pmc_r03.py"), "base", "4.0", "3.5", "16.0"], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarise_pmc_r03.py"), "final", "4.0", "3.5", "16.0"], stdout=subprocess.DEVNULL)
a = json.load(open(os.path.join(DST, "r03_scl_pmc_before.json"))); b = json.load(open(os.path.join(DST, "r03_scl_pmc.json")))
for k in ("valu_instructions", "fp64_instructions", "hbm_side_bytes"):
    print(k, a["per_frame"][k], "->", b["per_frame"][k])
print("kernel ms at 2.4 GHz", a["kernel_ms_at_2.4GHz"], "->", b["kernel_ms_at_2.4GHz"])
