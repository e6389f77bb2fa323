cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_oell_device.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
for cap in 256 128; do
EXP_PATTERNS=near EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256,4096:256 EXP_FORMS=ragged0,ragged1 SPGPU_DEEP_CAP=$cap timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D " | sed "s/^/cap $cap /"
done
