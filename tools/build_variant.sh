#!/bin/bash
# An A/B copy of libgpengine.so with extra compiler flags on some translation units (run where hipcc is; the result travels to
# the GPU box with the snapshot):   tools/build_variant.sh <name> "<flags>" potrf.hip [gemm.hip ...]
# -> tools/tmp/libgpe_<name>.so   (tools/r4_ab.py loads it when R4_SO points at it)
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/limbo_amd/csrc
mkdir -p $root/tools/tmp/$name
objs=""
for f in engine kbuild gemm potrf solve grad microbench sparsify inv solve_mp small; do
  if [[ " $* " == *" $f.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -ffp-contract=on $flags -c $cs/$f.hip -o $root/tools/tmp/$name/$f.o
    objs="$objs $root/tools/tmp/$name/$f.o"
  else
    objs="$objs $cs/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/tmp/libgpe_$name.so $objs
echo built $root/tools/tmp/libgpe_$name.so
