cd /root/repo
SPGPU_RAGGED=2 timeout -k 10 600 python -m pytest tests/test_gpu_oell_device.py -x -q -k "ragged" 2>&1 | tail -2
EXP_PATTERNS=near EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256,4096:256 EXP_FORMS=ragged0,raggedp0,raggedp1 timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D "
EXP_FORMS=ragged0,raggedp0,raggedp1 timeout -k 10 300 python tools/exp_tile.py D 10000000 mild 2>&1 | grep "^D " | grep -v plain
