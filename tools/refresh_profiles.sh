#!/bin/bash
# Regenerates the artefacts under profiles/ on an MI355X box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh <round tag, e.g. r01>
# rocprofv3 passes run with GPE_STOP_EVENT=0 (its kernel trace delays dispatches that carry their own
# completion event by ~100 us each, see engine.hip) and the PMC passes with GPE_LOOKAHEAD=0 so that every
# trailing update runs alone on the chip.  --pmc is never combined with any trace domain but the kernel trace.
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
GPE_STOP_EVENT=0 rocprofv3 --kernel-trace -d /tmp/p_kt -o p -- $B > /dev/null 2>&1
db=$(find /tmp/p_kt -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace (GPE_STOP_EVENT=0) over: bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline  (N=4096 D=6 SE-ARD, MI355X)"; python $root/tools/kstats.py $db; } > $out/${tag}_rocprofv3_kernel_stats.txt
{ echo "# last evaluation of the same trace: busy time per kernel; negative gap = overlap (two streams)"; python $root/tools/ktimeline.py $db; } > $out/${tag}_kernel_trace_timeline.txt
for cnt in FETCH_SIZE WRITE_SIZE; do
  GPE_LOOKAHEAD=0 rocprofv3 --kernel-trace --pmc $cnt -d /tmp/p_$cnt -o p -- $B > /dev/null 2>&1
done
GPE_LOOKAHEAD=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 -d /tmp/p_MFMA -o p -- $B > /dev/null 2>&1
f=$(find /tmp/p_FETCH_SIZE -name "*.db" | head -1); w=$(find /tmp/p_WRITE_SIZE -name "*.db" | head -1); m=$(find /tmp/p_MFMA -name "*.db" | head -1)
python $root/tools/pmc_summary.py $f $w $m > $out/${tag}_pmc_trailing_update.json
{ echo "# rocprofv3 --pmc passes (GPE_LOOKAHEAD=0 so every update runs alone) over: bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"; echo "# separate passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64"; KSTATS_GRID=1 python $root/tools/kpmc.py $f $w $m; } > $out/${tag}_pmc_bench_n4096.txt
cd $root
python bench.py > $out/${tag}_bench_n4096.json 2> $out/bench.err
python bench_extra.py > $out/${tag}_bench_extra.json 2> $out/bench_extra.err
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $out/${tag}_s1_gpu_tests.log
tests/cpp/test_gp_dropin >> $out/${tag}_s1_gpu_tests.log 2>&1
ls -la $out
