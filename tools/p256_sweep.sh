#!/bin/bash
# headline at N = 4096 under look-ahead knobs (k_panel256 on)
run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run GPE_PANEL256=1
run GPE_PANEL256=0
run GPE_BULK_FREE_TILES=100000
run GPE_BULK_FREE_TILES=100000 GPE_BULK_WGS=160
run GPE_BULK_FREE_TILES=100000 GPE_BULK_WGS=208
run GPE_BULK_FREE_TILES=100000 GPE_BULK_WGS=224
run GPE_BULK_FREE_TILES=400
run GPE_NEAR_WGS=192
run GPE_NEAR_WGS=128
