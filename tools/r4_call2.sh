#!/bin/bash
# GPU call 2: cached first look (default library) against device-scope-only polling (tools/tmp/libgpe_nc.so)
out=gpurun_out/r4b; mkdir -p $out
for so in "" tools/tmp/libgpe_nc.so; do
  echo "##### R4_SO=$so" >> $out/ab.log
  R4_SO=$so timeout 120 python tools/r4_ab.py single >> $out/ab.log 2>&1
  R4_SO=$so timeout 60 python tools/r4_ab.py phases >> $out/ab.log 2>&1
  R4_SO=$so timeout 120 python tools/r4_ab.py batch >> $out/ab.log 2>&1
  R4_SO=$so GPE_TAIL_MAX=4096 timeout 120 python tools/r4_ab.py batch >> $out/ab.log 2>&1
done
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "tiled_tail or data_flow_buffers or c4_batch or hand_over_timeout or polled_buffers or contention" > $out/tests.log 2>&1
tail -3 $out/tests.log
