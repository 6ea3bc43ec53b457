cd /root/repo
mkdir -p gpurun_out/r06
bash tools/ab_lib.sh python tools/tail_try.py 520 1024 2048 3072 4096 8192 > gpurun_out/r06/nopair_ab.log 2>&1
cat gpurun_out/r06/nopair_ab.log
