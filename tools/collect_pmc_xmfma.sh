#!/bin/bash
# Counters of the two stand-alone correlation kernels on the config-3 launch shape (65 536 windows of 2 048): the packed-vector kernel
# (es_xcorr32_kernel<17,2048>) and the matrix-pipe one (es_xcorr32_mfma_kernel<2048>).  bash tools/collect_pmc_xmfma.sh -> gpurun_out/r3/pmc_xmfma/*
R=$(pwd)
OUT=$R/gpurun_out/r3/pmc_xmfma
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i -o "SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*\|SQ_[A-Z_]*MOPS[A-Z0-9_]*" $OUT/counters.txt | sort -u > $OUT/mfma_counters.txt
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/xcorr32_mfma_launch.py > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq_a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq_b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU
run sq_c SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
for d in sq_a sq_b sq_c fetch write; do echo "== $d"; python tools/pmc_by_grid.py $OUT/$d es_xcorr32 2>&1 | tail -24; done > $OUT/summary.txt
