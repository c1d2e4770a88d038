# Round-4 evidence in one GPU call: GPU tests, smoke, three driver-style bench lines, kernel-trace stats of the same command.  TAG=name bash tools/collect_r04_final.sh
R=$(pwd)
OUT=$R/gpurun_out/r4/${TAG:-final}
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { tail -20 $OUT/gpu_tests.log; exit 1; }
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }
python bench.py --steps 20 --warmup 5 > $OUT/bench_a.json 2> $OUT/bench_a.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_b.json 2> $OUT/bench_b.err
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cd $R
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/stats -type f ! -name "*kernel_stats.csv" -delete
tail -2 $OUT/gpu_tests.log; tail -1 $OUT/smoke.log
python - <<'PY'
import json, os
for n in ("a","b","default","under_rocprof"):
    try:
        j=json.loads(open("gpurun_out/r4/" + os.environ.get("TAG", "final") + f"/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(j["value"]), round(j["config"]["timed_region_ms"]), {k:(round(v["value"],2) if v.get("value") else None) for k,v in j["legs"].items()}, round(j["roofline"]["frac"],3), round(j["roofline"]["launch_ms"],4), round(j["roofline_scl"]["frac"],3))
    except Exception as e: print(n, "ERR", e)
PY
