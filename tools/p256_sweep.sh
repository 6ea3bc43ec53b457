#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_time_spread']['median_ms'])"; }
run GPE_ROWS_TAIL=0
run GPE_ROWS_TAIL=1
run GPE_ROWS_TAIL=0
run GPE_ROWS_TAIL=1
