set -x
mkdir -p gpurun_out/r04a
(rocm-smi --showclocks --showpower --showmemuse --showcomputepartition --showmemorypartition --showperflevel 2>&1 | head -80) > gpurun_out/r04a/rocm_smi.txt
(rocm-smi -a 2>&1 | head -200) > gpurun_out/r04a/rocm_smi_all.txt
EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band,near EXP_ONLY_WINDOWED=1 EXP_ALIGNED=1 timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw > gpurun_out/r04a/base_aligned.txt 2>&1
EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band EXP_IDENTITY_FOR_PLAIN=1 timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw > gpurun_out/r04a/plain_identity.txt 2>&1
EXP_ORDERS=2048:256 EXP_FORMS=auto EXP_PATTERNS=band timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw > gpurun_out/r04a/plain_default.txt 2>&1
tail -n 5 gpurun_out/r04a/*.txt
