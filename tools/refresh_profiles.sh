#!/bin/bash
# Regenerates the artefacts under profiles/ on an MI355X box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh <round tag, e.g. r05>      (LEAN=1: the kernel stats, the PMC passes, the bench line and the
#   production timelines only — what bench.py and DESIGN quote; the round-4 studies keep their r04_ files)
# rocprofv3 passes run with GPE_STOP_EVENT=0 (its kernel trace delays dispatches that carry their own completion event by
# ~100 us each, see engine.hip) and the PMC passes with GPE_LOOKAHEAD=0 so that every trailing update runs alone on the
# chip (at N = 4096 the round-4 schedule is three launches on one stream either way).  --pmc is never combined with any
# trace domain but the kernel trace.  Round-3 studies that did not change (stream-k, diagonal-block stamps, kernel-build
# variants, the write-stream rate) keep their r02_/r03_ files.
set -u
tag=${1:-r05}
root=$(pwd)
out=$root/gpurun_out/prof
mkdir -p $out
# One box in about ten runs the latency-bound launches at half speed (profiles/r04_slow_box_observation.log): look before
# spending ten minutes on it (exit 3: try again on another box)
probe=$(cd $root && python tools/r4_ab.py probe 2>/dev/null | tail -1)
echo "probe: $probe evaluations/s"
if [ "${probe%.*}" -lt "${MIN_PROBE:-700}" ]; then echo "slow box: nothing refreshed" | tee $out/SLOW_BOX; exit 3; fi
rm -f $out/SLOW_BOX
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --headline-only"
GPE_STOP_EVENT=0 rocprofv3 --kernel-trace -d /tmp/p_kt -o p -- $B > /dev/null 2>&1
db=$(find /tmp/p_kt -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace (GPE_STOP_EVENT=0) over: bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --headline-only  (N=4096 D=6 SE-ARD, MI355X)"; python $root/tools/kstats.py $db; } > $out/${tag}_rocprofv3_kernel_stats.txt
{ echo "# last evaluation of the same trace: busy time per kernel; negative gap = overlap (two streams)"; python $root/tools/ktimeline.py $db; } > $out/${tag}_kernel_trace_timeline.txt
for cnt in FETCH_SIZE WRITE_SIZE; do
  GPE_LOOKAHEAD=0 GPE_TAIL_GEN=0 rocprofv3 --kernel-trace --pmc $cnt -d /tmp/p_$cnt -o p -- $B > /dev/null 2>&1
done
GPE_LOOKAHEAD=0 GPE_TAIL_GEN=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 -d /tmp/p_MFMA -o p -- $B > /dev/null 2>&1
f=$(find /tmp/p_FETCH_SIZE -name "*.db" | head -1); w=$(find /tmp/p_WRITE_SIZE -name "*.db" | head -1); m=$(find /tmp/p_MFMA -name "*.db" | head -1)
python $root/tools/pmc_summary.py $f $w $m > $out/${tag}_pmc_bench.json   # (bench.py reads profiles/r05_pmc_bench.json: roofline.traffic)
{ echo "# rocprofv3 --pmc passes (GPE_LOOKAHEAD=0 GPE_TAIL_GEN=0) over: bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --headline-only"; echo "# separate passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64   (FETCH_SIZE / WRITE_SIZE in KiB per dispatch; FETCH to be doubled, MI355X_MICROARCH.md)"; KSTATS_GRID=1 python $root/tools/kpmc.py $f $w $m; } > $out/${tag}_pmc_bench_n4096.txt
cd $root
python bench.py > $out/${tag}_bench_n4096.json 2> $out/bench.err
if [ -z "${SKIP_TESTS:-}" ]; then
{ python -m pytest tests -m gpu -q 2>&1 | tail -5; tests/cpp/test_gp_dropin; LIMBO_AMD_MIN_N_FOR_GPU=0 tests/cpp/test_gp_dropin | tail -3; tests/cpp/test_mixed_tree; } > $out/${tag}_s1_gpu_tests.log 2>&1
fi
# the production schedule traced by the library itself (every launch with its own start/stop events)
python tools/trace_eval.py compute 4096 > $out/${tag}_production_timeline.txt 2> $out/trace.err
python tools/trace_eval.py hp 4096 > $out/${tag}_production_timeline_hp_objective.txt 2>> $out/trace.err
if [ -n "${LEAN:-}" ]; then ls -la $out; exit 0; fi
python tools/trace_eval.py compute 2048 > $out/${tag}_production_timeline_n2048.txt 2>> $out/trace.err
python tools/small_bench.py > $out/${tag}_small_path_latency.json 2> $out/small_bench.err
{ python tools/c4bench.py 8 64; GPE_BATCH_TAIL_TILES=0 python tools/c4bench.py 8 64; } > $out/${tag}_c4bench.log 2>&1
# the schedule A/B of round 4 (tall launch on / off, closing-launch widths, phases alone on the chip, other sizes, batches)
{ python tools/r4_ab.py single; python tools/r4_ab.py phases; python tools/r4_ab.py sizes; python tools/r4_ab.py batch; GPE_BATCH_TAIL_TILES=0 python tools/r4_ab.py batch; } > $out/${tag}_schedule_ab.log 2>&1
{ echo "# tools/kbench_t: k_tail (the leading 1024 x 1024 block of K as one tiled data-flow launch) alone; wall_clock64 stamps of the diagonal workgroups"; tools/kbench_t 1 | grep -A18 "^k_tail"; echo "# compute()+log_lik by size: default, GPE_TALL=0 (panels in front of the closing launch), GPE_TAIL_MAX=0 (panels to the end)"; python tools/tail_try.py 150 520 1024 1100 1700 2048 2500 3072 4096 8192; GPE_TALL=0 python tools/tail_try.py 3072 4096 8192; GPE_TAIL_MAX=0 python tools/tail_try.py 1024 2048 4096; } > $out/${tag}_tail_stamps.log 2>&1
ls -la $out
