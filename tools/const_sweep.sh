#!/bin/bash
# dev: step time per experiment library under glow-tts_amd/build/exp (two passes, interleaved with the default)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in "" glow-tts_amd/build/exp/libglowtts_*.so; do
  if [ -n "$lib" ]; then export GT_LIB=$PWD/$lib; else unset GT_LIB; fi
  echo -n "${lib:-default}  "
  python bench.py --workload ${WL:-cfg2} --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done; done
