#!/bin/bash
# GPU call 3: dispatch order of the data-flow launches (GPE_TAIL_W, GPE_TAIL_LAG; process-wide switches -> one child per setting)
out=gpurun_out/r4c; mkdir -p $out
for wl in "0 0" "0 2" "0 4" "6 3" "8 2" "8 4" "12 3" "16 3"; do
  set -- $wl
  GPE_TAIL_W=$1 GPE_TAIL_LAG=$2 timeout 100 python tools/r4_ab.py single2 >> $out/single.log 2>&1
done
for wb in 0 3 4 6 8 12; do
  echo "##### GPE_TAIL_W_BATCH=$wb" >> $out/batch.log
  GPE_TAIL_W_BATCH=$wb GPE_TAIL_LAG=3 timeout 120 python tools/r4_ab.py batch >> $out/batch.log 2>&1
done
GPE_TAIL_W=8 GPE_TAIL_LAG=3 timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "tiled_tail or data_flow_buffers or c4_batch or hand_over_timeout" > $out/tests.log 2>&1
tail -3 $out/tests.log
cat $out/single.log $out/batch.log
