cd /root/repo
for i in 1 2 3; do timeout -k 10 400 python tools/exp_placement.py band 2>&1 | grep -v amdgpu.ids; done
