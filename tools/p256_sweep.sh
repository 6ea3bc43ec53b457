#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_time_spread']['median_ms'], d['log_lik'])"; }
run GPE_TAIL_MAX=2560
run GPE_TAIL_MAX=3072
run GPE_TAIL_MAX=4096
run GPE_TAIL_MAX=2048
