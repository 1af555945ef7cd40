#!/bin/bash
# dev: weight-gradient kernel variants (tools/exp_variant.py builds) x forced slab counts, per-kernel durations from rocprofv3
cd "$(dirname "$0")/.."
for lib in glow-tts_amd/build/exp/libglowtts_nopipe.so glow-tts_amd/build/exp/libglowtts_kb64.so; do
  for S in 1 2 3; do
    echo "== lib ${lib:-default} slabs $S"
    if [ -z "$lib" ]; then GT_WGRAD_SLABS=$S bash tools/wgrad_prof.sh; else GT_WGRAD_SLABS=$S bash tools/wgrad_prof.sh --lib $PWD/$lib; fi
  done
done
