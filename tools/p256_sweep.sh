#!/bin/bash
# headline at N = 4096 under scheduling knobs
run() { echo "== $*"; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run GPE_BULK_FREE_TILES=0
run GPE_BULK_FREE_TILES=0 GPE_NEAR_WGS=192
run GPE_BULK_FREE_TILES=40
run GPE_BULK_FREE_TILES=100
run GPE_BULK_FREE_TILES=0
run GPE_X=0
for v in 0 250; do echo "FREE_TILES=$v"; GPE_BULK_FREE_TILES=$v python tools/p256_try.py; done
