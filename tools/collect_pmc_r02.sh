#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_pmc_r02.sh    -> gpurun_out/r2/pmc/*  (summaries are copied into profiles/ by tools/summarise_pmc_r02.py)
# Counter passes carry no trace flags (gpurun refuses --pmc together with tracing); one rocprofv3 run per counter group.
set -e
R=$(pwd)
OUT=$R/gpurun_out/r2/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $OUT/ub_rowcopy $R/tools/ub/ub_rowcopy.hip
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/rowcopy_$c -- $OUT/ub_rowcopy > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/x32_$c -- python3 $R/tools/xcorr32_c3_launch.py > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/scl_a -- python3 $R/tools/scl_pmc2.py "" 65536 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/scl_b -- python3 $R/tools/scl_pmc2.py "" 65536 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/x32_sq -- python3 $R/tools/xcorr32_c3_launch.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /dev/null
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
for d in rowcopy_FETCH_SIZE rowcopy_WRITE_SIZE; do python tools/pmc_by_grid.py $OUT/$d rowcopy; python tools/pmc_by_grid.py $OUT/$d copy4; done
for d in x32_FETCH_SIZE x32_WRITE_SIZE x32_sq; do python tools/pmc_by_grid.py $OUT/$d es_xcorr32; done
for d in scl_a scl_b; do python tools/pmc_by_grid.py $OUT/$d es_scl_; done
