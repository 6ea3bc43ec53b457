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
B="python $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --headline-only"
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
tests/cpp/test_mixed_tree >> $out/${tag}_s1_gpu_tests.log 2>&1
python tools/small_bench.py > $out/${tag}_small_path_latency.json 2> $out/small_bench.err
python tools/c4bench.py 8 64 > $out/${tag}_c4bench.log 2>&1
GPE_BATCH=0 python tools/c4bench.py 8 64 >> $out/${tag}_c4bench.log 2>&1
python - > $out/${tag}_hbm_write_stream.log 2>&1 <<'PY'
import ctypes as C, sys
sys.path.insert(0, ".")
from limbo_amd import _capi
e = _capi.load_engine()
g = C.c_double()
e.fn("hbm_stream_peak")(0, C.byref(g))
print(f"write-only stream, 1 GiB, 16 B per lane, best of 5: {g.value:.0f} GB/s  (the kernel-matrix build writes 67.3 MB per evaluation at N = 4096)")
PY
cd /tmp
for v in 2 0; do GPE_KBUILD=$v GPE_STOP_EVENT=0 rocprofv3 --kernel-trace -d /tmp/p_kb$v -o p -- $B > /dev/null 2>&1; db=$(find /tmp/p_kb$v -name "*.db" | head -1); echo "GPE_KBUILD=$v" >> $out/${tag}_kernel_build_trace.txt; python $root/tools/kstats.py $db | grep -i "k_build" >> $out/${tag}_kernel_build_trace.txt; done
for cnt in WRITE_SIZE FETCH_SIZE; do rocprofv3 --kernel-trace --pmc $cnt -d /tmp/p_kbp_$cnt -o p -- $B > /dev/null 2>&1; db=$(find /tmp/p_kbp_$cnt -name "*.db" | head -1); KSTATS_GRID=0 python $root/tools/kpmc.py $db | grep -i "k_build" >> $out/${tag}_kernel_build_trace.txt; done
cd $root
# in-kernel stamps: the diagonal block alone (data-flow form, and the barrier rounds it replaced), the four steps of an outer
# panel (workgroup 0 / last workgroup), with and without the head-tile hand-over
python tools/make_k64.py > /dev/null
{ echo "# tools/diagflow (diag_flow.h): k_diag alone on tools/tmp/K64.bin if present"; tools/diagflow; } > $out/${tag}_diag_flow_stamps.log 2>&1
{ echo "# tools/kbench_t: k_panel_step at the first panel of N = 4096, nt = head tiles of the step (3, 2, 1, 0); s_memtime cycles at 2.38 GHz"; echo "## head tiles handed over (default)"; tools/kbench_t 1 | grep -A2 "step with nt"; } > $out/${tag}_panel_step_stamps.log 2>&1
{ echo "# tools/kbench_t: k_panel256 (all steps of the first outer panel of N = 4096 in one launch) alone; wall_clock64 stamps of strips 0-3 and the last one"; tools/kbench_t 1 | grep -A7 "^k_panel256"; } > $out/${tag}_panel256_stamps.log 2>&1
{ echo "# tools/kbench_t: k_tail (a 1024-column tail = the leading 1024 x 1024 block of K as one tiled data-flow launch) alone; wall_clock64 stamps of the diagonal workgroups"; tools/kbench_t 1 | grep -A18 "^k_tail"; echo "# compute()+log_lik by size, default (tail 2560) and GPE_TAIL_MAX=0"; python tools/tail_try.py 150 520 1024 1100 1700 2048 2500 3072 4096 8192; GPE_TAIL_MAX=0 python tools/tail_try.py 150 520 1024 1100 1700 2048 2500 3072 4096 8192; } > $out/${tag}_tail_stamps.log 2>&1
# round 3: the production schedule traced by the library itself (every launch with its own start/stop events), the
# stream-k study of the trailing update, the resident small-path workgroup
python tools/trace_eval.py compute 4096 > $out/${tag}_production_timeline.txt 2> $out/trace.err
python tools/trace_eval.py hp 4096 > $out/${tag}_production_timeline_hp_objective.txt 2>> $out/trace.err
# (tools/updbench, tools/updbench_t: built where hipcc is, make -C tools updbench updbench_t; the binaries travel with the snapshot)
{ tools/updbench 4096 5; } > $out/${tag}_updbench_trailing_update.log 2>&1
{ tools/updbench_t 4096 2 | head -120; } > $out/${tag}_updbench_stream_k_stamps.log 2>&1
ls -la $out
