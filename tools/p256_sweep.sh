#!/bin/bash
# headline at N = 4096 under GEMM-variant knobs
run() { echo "== $*"; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run GPE_X=0
run GPE_GLDS64_VARIANT=1
run GPE_GLDS64_VARIANT=2
run GPE_GLDS_VARIANT=1
run GPE_GLDS_VARIANT=0
run GPE_X=0
