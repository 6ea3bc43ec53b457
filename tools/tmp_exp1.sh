cd /root/repo
mkdir -p gpurun_out/r06
GPE_SWEEP_DBG=1 timeout 300 python tools/tail_try.py 4096 > gpurun_out/r06/sweep2_stamps.log 2>&1
cat gpurun_out/r06/sweep2_stamps.log
