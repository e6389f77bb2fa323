cd /root/repo
export EXP_PATTERNS=random EXP_ONLY_WINDOWED=1 EXP_WINDOWS_FOR_ALL=1 EXP_ORDERS=2048:256 EXP_FORMS=ragged0,raggedg,ragged0,raggedg
timeout -k 10 300 python tools/exp_tile.py D 10000000 powerlaw 2>&1 | grep "^D " | cut -c1-125
