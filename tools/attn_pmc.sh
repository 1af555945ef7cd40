#!/bin/bash
# dev: counters of the attention kernels in tools/attn_bench.py
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for pass in "a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "b SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
  set -- $pass; tag=$1; shift
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/atpmc_$tag -- python3 $ROOT/tools/attn_bench.py > $OUT/atpmc_$tag.log 2>&1) || { echo "pass $tag failed"; tail -3 $OUT/atpmc_$tag.log; continue; }
  for k in "attn_bwd_q" "attn_fwd_mfma"; do echo "-- $k"; python tools/pmc_dump.py $OUT/atpmc_$tag "$k"; done
  rm -rf $OUT/atpmc_$tag
done
