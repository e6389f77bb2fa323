cd /root/repo
export EXP_PATTERNS=band,near EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256
for i in 1 2 3; do
EXP_FORMS=ragged0,ragged0n,ragged0,ragged0n timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D " | cut -c1-118
done
EXP_FORMS=ragged0,ragged0n,ragged0,ragged0n timeout -k 10 300 python tools/exp_tile.py D 10000000 mild 2>&1 | grep "^D " | grep -v plain | cut -c1-118
