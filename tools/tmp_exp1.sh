cd /root/repo
mkdir -p gpurun_out/r06
KB_TAIL_T=1024 timeout 120 tools/kbench_t 1 2>&1 | grep -A24 "^k_tail, the leading" > gpurun_out/r06/defer_stamps.log
head -22 gpurun_out/r06/defer_stamps.log | cut -c1-220
bash tools/ab_lib.sh python tools/tail_try.py 520 1024 2048 3072 4096 > gpurun_out/r06/defer_ab.log 2>&1
cat gpurun_out/r06/defer_ab.log
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x -k "tiled_tail or full_path or vs_oracle or c2_full_size_gradient" 2>&1 | tail -3
