#!/bin/bash
# dev: time the WaveNet stack kernels of every experiment library under glow-tts_amd/build/exp (tools/exp_variant.py)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for lib in "" glow-tts_amd/build/exp/libglowtts_*.so; do
  echo "== ${lib:-default}"
  if [ -z "$lib" ]; then python tools/wn_layer_bench.py stack; else python tools/wn_layer_bench.py stack "$lib"; fi 2>&1 | grep "us / launch"
done
