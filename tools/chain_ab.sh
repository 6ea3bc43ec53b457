#!/bin/bash
# The round-5 chain study on ONE box (run through gpurun from the repo root): stamps of the chain workgroups of a 1024-column
# data-flow launch and compute()+log_lik by size, this tree's build against an older one put at limbo_amd/libgpengine_head.so and
# tools/kbench_t_head by hand (round 4's potrf.hip at commit fec4892).   bash tools/chain_ab.sh <round tag>
set -u
tag=${1:-r05}
root=$(pwd)
out=$root/gpurun_out/prof
mkdir -p $out
{
  echo "# tools/kbench_t: k_tail over the leading 1024 x 1024 block of K alone (16 tile columns), wall_clock64 stamps of the chain workgroups"
  echo "## this tree"
  timeout 60 tools/kbench_t 3 2>&1 | grep -A70 "^k_tail"
  if [ -x tools/kbench_t_head ]; then
    echo "## round 4's potrf.hip (commit fec4892)"
    timeout 60 tools/kbench_t_head 3 2>&1 | grep -A40 "^k_tail"
  fi
} > $out/${tag}_chain_stamps.log 2>&1
{
  echo "# compute()+log_lik by size (tools/tail_try.py: N, rc, log-lik, hand-over re-runs, ms), two rounds of new | old on one box"
  echo "# new = this tree; old = the same library with round 4's potrf.hip (commit fec4892)"
  bash tools/ab_lib.sh python tools/tail_try.py 512 1024 1100 1700 2048 3072 4096 8192
} > $out/${tag}_chain_ab.log 2>&1
tail -22 $out/${tag}_chain_ab.log
