cd /root/repo
mkdir -p gpurun_out/r06
for r in 1 2; do
for d in 0 4 6 8 12 16; do echo "== GPE_TAIL_DLEAD=$d"; GPE_TAIL_DLEAD=$d timeout 120 python tools/tail_try.py 4096 3072 2560 1536 3500; done
done > gpurun_out/r06/dlead2.log 2>&1
cat gpurun_out/r06/dlead2.log | grep -v retries.*ms.*xx | awk '/==/{d=$2} /retries/{print d, $1, $NF}' | sort -k2,2n -k1,1 | head -80
