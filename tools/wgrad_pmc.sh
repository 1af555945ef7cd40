#!/bin/bash
# dev: HBM-side traffic and MFMA busy share of the weight-gradient kernels in tools/wgrad_bench.py (separate --pmc passes)
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "s SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  set -- $pass; tag=$1; shift
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/wgpmc_$tag -- python3 $ROOT/tools/wgrad_bench.py > $OUT/wgpmc_$tag.log 2>&1) || { echo "pass $tag failed"; tail -3 $OUT/wgpmc_$tag.log; continue; }
  for k in "wgrad_batched_kernel<5>" "wgrad_batched_kernel<1>" "weightnorm_bwd_batched"; do echo "-- $k"; python tools/pmc_dump.py $OUT/wgpmc_$tag "$k"; done
  rm -rf $OUT/wgpmc_$tag
done
