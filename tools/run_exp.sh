cd /root/repo
start=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/bench_r02d.json 2> gpurun_out/bench_r02d.err; echo "bench rc $? in $(( $(date +%s) - start )) s"
