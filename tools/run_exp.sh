cd /root/repo
VARIANTS=0,10,11 timeout -k 10 300 python tools/ab_spmm.py banded 2>&1 | tail -4
