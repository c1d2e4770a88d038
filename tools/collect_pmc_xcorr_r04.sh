#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the two correlation kernels on the config-3 launch shape (65 536 windows of 2 048): es_xcorr32_kernel<17,2048,false>
# (stand-alone screen, the HBM-graded kernel) and <17,2048,true> (fused sync).  One counter per run, counters only.
#   bash tools/collect_pmc_xcorr_r04.sh  -> gpurun_out/r4/pmc_xcorr/* ; then python tools/summarise_pmc_xcorr_r04.py -> profiles/r04_xcorr32_pmc_traffic.json
R=$(pwd)
OUT=$R/gpurun_out/r4/pmc_xcorr
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/xcorr32_c3_launch.py > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
cd $R
for d in fetch write sq; do echo "== $d"; python tools/pmc_by_grid.py $OUT/$d es_xcorr32 2>&1 | tail -24; done > $OUT/summary.txt
