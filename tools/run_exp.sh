cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_oell_device.py tests/test_gpu_spmv.py tests/test_gpu_f3.py -x -q 2>&1 | tail -3
EXP_PATTERNS=near2048,banded timeout -k 10 300 python tools/exp_tile.py D 10000000 uniform 2>&1 | grep "^D "
