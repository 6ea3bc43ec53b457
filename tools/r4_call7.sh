#!/bin/bash
# GPU call 7: (a) gen mode (data-flow launches generate their own K tiles) correctness + A/B; (b) the chain-critical consumer
# watching its own words (default) vs poll_one for everybody (libgpe_nocrit.so: built before gen mode existed, so compare with
# GPE_TAIL_GEN=0); (c) the k = 1536 update through 64 x 64 tiles; (d) hp_objective under the tall schedule vs GPE_TALL=0
out=gpurun_out/r4g; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "tiled_tail or data_flow_buffers or c4_batch or small_parity or golden or alternate or hand_over or batch_hp" > $out/tests.log 2>&1
tail -3 $out/tests.log
echo "##### default (gen on, crit poll on)" >> $out/ab.log
timeout 100 python tools/r4_ab.py single2 >> $out/ab.log 2>&1
echo "##### GPE_TAIL_GEN=0" >> $out/ab.log
GPE_TAIL_GEN=0 timeout 100 python tools/r4_ab.py single2 >> $out/ab.log 2>&1
echo "##### GPE_TAIL_GEN=0 R4_SO=libgpe_nocrit.so" >> $out/ab.log
GPE_TAIL_GEN=0 R4_SO=tools/tmp/libgpe_nocrit.so timeout 100 python tools/r4_ab.py single2 >> $out/ab.log 2>&1
echo "##### default again" >> $out/ab.log
timeout 100 python tools/r4_ab.py single2 >> $out/ab.log 2>&1
echo "##### GPE_GEMM_TILE=64" >> $out/ab.log
GPE_GEMM_TILE=64 timeout 100 python tools/r4_ab.py single2 >> $out/ab.log 2>&1
echo "##### hp_objective: default / GPE_TALL=0 / GPE_INV_OVERLAP=0" >> $out/ab.log
timeout 60 python tools/hp_try.py >> $out/ab.log 2>&1
GPE_TALL=0 timeout 60 python tools/hp_try.py >> $out/ab.log 2>&1
GPE_INV_OVERLAP=0 timeout 60 python tools/hp_try.py >> $out/ab.log 2>&1
timeout 100 python tools/r4_ab.py batch >> $out/ab.log 2>&1
cat $out/ab.log
