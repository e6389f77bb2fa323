cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_oell_device.py tests/test_gpu_spmv.py tests/test_gpu_f3.py -x -q 2>&1 | tail -5
echo "--- mild"
EXP_FORMS=tile2,ragged0,ragged1,ragged2,ragged3,raggedg timeout -k 10 300 python tools/exp_tile.py D 10000000 mild 2>&1 | grep "^D "
for cap in 128 256; do
echo "--- power-law cap $cap"
EXP_PATTERNS=near EXP_ONLY_WINDOWED=1 EXP_ORDERS=1024:256,2048:256,4096:256 EXP_FORMS=tile2,ragged0,ragged1,ragged2,ragged3 SPGPU_DEEP_CAP=$cap timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D "
done
