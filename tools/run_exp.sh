cd /root/repo
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 &&
timeout -k 10 600 python bench.py > gpurun_out/bench_r02f.json 2> gpurun_out/bench_r02f.err; echo "bench rc $?"
