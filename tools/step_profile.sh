#!/bin/bash
# dev: rocprofv3 kernel trace of the bench step as it runs (both branches), with the chronological listing of one replayed step
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/step_prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/step_prof.log 2>&1) || exit 1
python tools/prof_summary.py $OUT/step_prof 25 60 > $OUT/step_summary.txt 2>&1
python tools/timeline.py $OUT/step_prof --list >> $OUT/step_summary.txt 2>&1
rm -rf $OUT/step_prof
echo done
