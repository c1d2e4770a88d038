#!/bin/bash
# Counter collection (round 4 on; ROUND=r3 reproduces the round-3 layout) for es_scl_wide_kernel<64,8> on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_pmc_r04.sh [TAG]   -> gpurun_out/${ROUND:-r4}/pmc_TAG/*
# Counter passes carry no trace flags; one rocprofv3 run per counter group (TCC: FETCH_SIZE and WRITE_SIZE never together).
R=$(pwd)
TAG=${1:-base}
OUT=$R/gpurun_out/${ROUND:-r4}/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/scl_pmc3.py "${BUILD:-}" 65536 8 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq_a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq_b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU
run f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU
run f32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED
run waits SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
cd $R
for d in fetch write sq_a sq_b f64 f32 waits tcc tcc2 tcp; do echo "== $d"; python tools/pmc_by_grid.py $OUT/$d es_scl_wide 2>&1 | tail -12; done > $OUT/summary.txt
