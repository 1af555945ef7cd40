#!/bin/bash
# dev: rocprofv3 kernel stats of the bench step with the encoder branch OFF (GT_ENC_STREAM=0: one stream, every kernel alone on the
# machine) — the isolated duration of every launch of the step, i.e. the CU-work the two branches share when they overlap.
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp GT_ENC_STREAM=0
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/serial_bench.json 2> $OUT/serial_bench.err || exit 1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/serial_prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/serial_prof.log 2>&1) || exit 1
python tools/prof_summary.py $OUT/serial_prof 25 70 > $OUT/serial_summary.txt 2>&1
python tools/timeline.py $OUT/serial_prof --list >> $OUT/serial_summary.txt 2>&1
rm -rf $OUT/serial_prof
echo done
