#!/bin/bash
# Counters of the front-end kernels on the headline's windows (65 536 x 2 048): bash tools/collect_pmc_front.sh -> gpurun_out/r3/pmc_front/*
R=$(pwd)
OUT=$R/gpurun_out/r3/pmc_front
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/c3_scl_alone.py > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq_a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq_b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU
run f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU
run f32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED
cd $R
for k in es_llr_wave es_xcorr32 es_bpf; do for d in sq_a sq_b f64 f32; do echo "== $k $d"; python tools/pmc_by_grid.py $OUT/$d $k 2>&1 | tail -9; done; done > $OUT/summary.txt
