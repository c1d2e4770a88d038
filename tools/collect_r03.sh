#!/bin/bash
# Round-3 evidence run on the GPU box (through gpurun from the repository root): bash tools/collect_r03.sh
#   gpurun_out/r3/final/*: two driver-style bench lines (--steps 20 --warmup 5), the default line, the kernel-trace stats of the driver-style
#   command, and the counter passes of the list decoder (tools/collect_pmc_r03.sh final).  tools/summarise_r03.py copies the summaries into profiles/.
R=$(pwd)
OUT=$R/gpurun_out/r3/final
rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_a.json 2> $OUT/bench_a.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_b.json 2> $OUT/bench_b.err
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cd $R
bash tools/collect_pmc_r03.sh final > /dev/null 2>&1
python - <<'PY'
import json,glob
for n in ("a","b","default"):
    try:
        j=json.loads(open(f"gpurun_out/r3/final/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(j["value"]), round(j["config"]["timed_region_ms"]), {k:(round(v["value"],2) if v.get("value") else None) for k,v in j["legs"].items()}, round(j["roofline"]["frac"],3), round(j["roofline_scl"]["frac"],3))
    except Exception as e: print(n, "ERR", e)
PY
