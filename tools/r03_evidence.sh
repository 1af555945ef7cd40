#!/bin/bash
# dev: the round's evidence runs on the GPU box, one call (rocprofv3 kernel stats of the bench command, PMC passes of the decoder's fused
# kernels, bench lines of the four workloads, the un-profiled timeline).  Outputs under gpurun_out/r03_*; copy what is judged to profiles/.
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 30 --warmup 5 > $OUT/r03_trainstep_bench.json 2> $OUT/r03_trainstep_bench.err || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --marks > /dev/null 2> $OUT/r03_trainstep_marks.txt || exit 1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r03_prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03_prof.log 2>&1) || exit 1
python tools/prof_summary.py $OUT/r03_prof 25 50 --csv $OUT/r03_trainstep_kernel_stats.csv > $OUT/r03_trainstep_summary.txt 2>&1
python tools/timeline.py $OUT/r03_prof >> $OUT/r03_trainstep_summary.txt 2>&1
rm -rf $OUT/r03_prof
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "s SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  set -- $pass; tag=$1; shift
  (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace -d $OUT/r03_pmc_$tag -- python3 $ROOT/tools/decoder_pmc.py > $OUT/r03_pmc_$tag.log 2>&1) || { echo "pmc pass $tag failed"; tail -5 $OUT/r03_pmc_$tag.log; }
done
python tools/decoder_pmc.py --parse $OUT/r03_pmc_f $OUT/r03_pmc_w $OUT/r03_pmc_s > $OUT/r03_decoder_pmc.json 2> $OUT/r03_decoder_pmc.err
rm -rf $OUT/r03_pmc_f $OUT/r03_pmc_w $OUT/r03_pmc_s
for wl in cfg3 cfg4 cfg5; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03_${wl}_bench.json 2> $OUT/r03_${wl}_bench.err || echo "$wl failed"
done
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --marks > /dev/null 2> $OUT/r03_cfg5_marks.txt
echo done
