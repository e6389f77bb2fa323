cd /root/repo
export EXP_PATTERNS=band,near EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_FORMS=ragged0
for cap in 256 512 768 1024 1280 1536 256 1024; do
echo "cap $cap"
SPGPU_DEEP_CAP=$cap timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D " | cut -c1-110
done
