cd /root/repo
export EXP_PATTERNS=band,near EXP_ONLY_WINDOWED=1
for al in "" 1; do
export EXP_ALIGNED=$al
EXP_ORDERS=1024:128,1024:256 EXP_FORMS=ragged0 timeout -k 10 500 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D "
EXP_ORDERS=2048:256 EXP_FORMS=ragged0,ragged1 timeout -k 10 500 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D "
done
