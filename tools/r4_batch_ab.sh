#!/bin/bash
# batched launch sequences: the split between the tall data-flow launch, the one update and the closing launch (GPE_BATCH_TAIL_MAX)
for tm in 2560 1536 1280 1024 768; do
  echo "##### GPE_BATCH_TAIL_MAX=$tm"
  GPE_BATCH_TAIL_MAX=$tm timeout 120 python tools/r4_ab.py batch 2>&1 | grep -v amdgpu.ids
done
for tm in 1536 1280 1024; do
  echo "##### GPE_BATCH_TAIL_MAX=$tm GPE_BATCH_TAIL_TILES=100000"
  GPE_BATCH_TAIL_MAX=$tm GPE_BATCH_TAIL_TILES=100000 timeout 120 python tools/r4_ab.py batch 2>&1 | grep -v amdgpu.ids
done
