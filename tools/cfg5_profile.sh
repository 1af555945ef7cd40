#!/bin/bash
# dev: rocprofv3 kernel stats of the cfg 5 step
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/cfg5_prof -- python3 $ROOT/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/cfg5_prof.log 2>&1) || exit 1
python tools/prof_summary.py $OUT/cfg5_prof 13 60 > $OUT/cfg5_summary.txt 2>&1
python tools/timeline.py $OUT/cfg5_prof --list >> $OUT/cfg5_summary.txt 2>&1
rm -rf $OUT/cfg5_prof
echo done
