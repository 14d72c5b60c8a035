#!/bin/bash
# PMC passes over one kernel of the bench (default: the register-tile factor kernel); run on the GPU box from the repo root.
#   bash tools/gpu_pmc_wave.sh [kernel-substring]      -> gpurun_out/pmcw/summary.txt
set -u
K=${1:-wave_front_kernel}
R=$PWD
OUT=$R/gpurun_out/pmcw${TAG:-}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_BRANCH SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32" \
           "SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pw$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pw$i -o p -- python $R/bench.py --no-cpu-baseline --graph off --steps 2 --warmup 1 ${BENCH_ARGS:-} > $OUT/pass$i.log 2>&1
  python $R/tools/pmc_kernel.py $(find /tmp/pw$i -name "*.db" | head -1) $K >> $OUT/summary.txt 2>> $OUT/err.txt
done
cd $R
cat $OUT/summary.txt
