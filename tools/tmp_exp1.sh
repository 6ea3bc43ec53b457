cd /root/repo
mkdir -p gpurun_out/r06
timeout 120 tools/kbench_s2 1 2>&1 | grep -A70 "^k_trsv_bwd_m" > gpurun_out/r06/sweep_m2_stamps.log
head -24 gpurun_out/r06/sweep_m2_stamps.log; tail -8 gpurun_out/r06/sweep_m2_stamps.log
{
for r in 1 2; do for v in 1 0; do echo "== GPE_SWEEP_M=$v"; GPE_SWEEP_M=$v timeout 300 python tools/tail_try.py 256 520 1024 1100 2048 4096 4000; done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_path or update_alpha or vs_oracle" 2>&1 | tail -3
} > gpurun_out/r06/sweep_m2.log 2>&1
cat gpurun_out/r06/sweep_m2.log
