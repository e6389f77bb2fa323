#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_oell_device.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_spmv.py -q -m gpu -x 2>&1 | tail -4 || exit 1
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_PATTERNS=band,near EXP_FORMS=auto,ragged4,ragged0
for al in "" 1; do
echo "== EXP_ALIGNED=$al"
EXP_ALIGNED=$al timeout -k 10 600 python3 tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep -E "^D |MISMATCH" || exit 1
done
EXP_ALIGNED=1 EXP_FORMS=auto timeout -k 10 600 python3 tools/exp_tile.py S 10000000 powerlaw 2>&1 | grep -E "^S |MISMATCH" || exit 1
