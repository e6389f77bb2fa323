cd /root/repo
export EXP_PATTERNS=band,near EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256,4096:256
for i in 1 2; do
EXP_FORMS=ragged0,ragged0x2,ragged0x4,ragged0x8,ragged0x16,ragged0x32,ragged0 timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D " | cut -c1-118
done
