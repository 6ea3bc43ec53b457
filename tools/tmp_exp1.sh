cd /root/repo
mkdir -p gpurun_out/r06
timeout 120 tools/kbench_s2 1 2>&1 | grep -A70 "^k_trsv_bwd_m" > gpurun_out/r06/sweep_m3_stamps.log
head -3 gpurun_out/r06/sweep_m3_stamps.log; sed -n 20,40p gpurun_out/r06/sweep_m3_stamps.log
{
for r in 1 2; do for v in 1 0; do echo "== GPE_SWEEP_M=$v"; GPE_SWEEP_M=$v timeout 300 python tools/tail_try.py 1024 2048 4096; done; done
} > gpurun_out/r06/sweep_m3.log 2>&1
cat gpurun_out/r06/sweep_m3.log
