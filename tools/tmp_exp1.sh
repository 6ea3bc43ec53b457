cd /root/repo
mkdir -p gpurun_out/r06
{
for tm in 2816 3072 3328 3584; do echo "== GPE_TAIL_MAX=$tm"; GPE_TAIL_MAX=$tm timeout 300 python tools/tail_try.py 2880 3072 3200 3328 3584 3840 4096; done
for cfg in "GPE_TALL=1024 GPE_TAIL_MAX=3072" "GPE_TALL=1536 GPE_TAIL_MAX=2560" "GPE_TALL=1536 GPE_TAIL_MAX=2816"; do echo "== $cfg"; env $cfg timeout 300 python tools/tail_try.py 3584 3840 4096 4352; done
} > gpurun_out/r06/retune.log 2>&1
awk '/==/{d=$0} /retries/{print $1, $NF, d}' gpurun_out/r06/retune.log | sort -k1,1n -k2,2n
