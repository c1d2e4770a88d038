#!/bin/bash
# Counters of the lone-batch list decoder (es_scl_kernel<8>, one frame per wave, B = 1024 = one wave per SIMD):
#   bash tools/collect_pmc_single.sh   -> gpurun_out/r3/pmc_single/*
R=$(pwd)
OUT=$R/gpurun_out/r3/pmc_single
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/scl_pmc3.py "" 1024 8 frame > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq_a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq_b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU
run f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU
run waits SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU
cd $R
for d in sq_a sq_b f64 waits; do echo "== $d"; python tools/pmc_by_grid.py $OUT/$d es_scl_kernel 2>&1 | tail -12; done > $OUT/summary.txt
