#!/bin/bash
# Is this MI355X box one of the slow ones for latency-bound launches?  (Round 4: one box in eight ran the data-flow launches 2x
# slower — single diagonal blocks took 22-30 us instead of 7 — while its matrix-core and HBM rates were normal.)  Prints the
# performance level / clocks rocm-smi reports, the N = 2048 / 4096 evaluation times, then tries `--setperflevel high`.
out=${1:-gpurun_out/boxcheck.log}
{
  echo "##### rocm-smi before"; rocm-smi --showperflevel --showclocks --showpower 2>&1 | grep -v "^$" | head -40
  echo "##### evaluation times (default)"; timeout 100 python tools/r4_ab.py single2 2>&1 | tail -5
  (for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>&1 | grep -iE "sclk|fclk|mclk" | head -4; sleep 0.5; done) > /tmp/clk_during.log &
  timeout 60 python tools/r4_ab.py single2 > /dev/null 2>&1
  wait
  echo "##### clocks sampled while evaluations ran"; cat /tmp/clk_during.log | sort | uniq -c | sort -rn | head -12
  echo "##### rocm-smi --setperflevel high"; rocm-smi --setperflevel high 2>&1 | tail -3
  echo "##### evaluation times (perf level high)"; timeout 100 python tools/r4_ab.py single2 2>&1 | tail -5
  rocm-smi --showperflevel 2>&1 | grep -i perf
  echo "##### rocm-smi --setperflevel auto"; rocm-smi --setperflevel auto 2>&1 | tail -2
} > $out 2>&1
cat $out
