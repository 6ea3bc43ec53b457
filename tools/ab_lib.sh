#!/bin/bash
# A/B of two builds of the engine on ONE box: limbo_amd/libgpengine.so (the tree's) against limbo_amd/libgpengine_head.so (an
# older build put there by hand); the command given runs under each.   bash tools/ab_lib.sh python tools/tail_try.py 4096
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
for round in 1 2; do
  echo "== new"; "$@"
  if [ -f limbo_amd/libgpengine_head.so ]; then
    mv limbo_amd/libgpengine.so limbo_amd/libgpengine_new.so; cp limbo_amd/libgpengine_head.so limbo_amd/libgpengine.so
    echo "== old"; "$@"
    mv limbo_amd/libgpengine_new.so limbo_amd/libgpengine.so
  fi
done
