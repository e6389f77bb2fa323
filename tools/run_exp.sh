cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_sharded_c.py tests/test_gpu_bench_rehearsal.py -x -q 2>&1 | tail -8
echo "--- bench spmm 1 rank, RCCL initialised, forced split, C driver"
SPGPU_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --workload spmm --force-split --steps 20 --warmup 5 2>&1 | tail -2 | cut -c1-1500
echo "--- bench spmm 1 rank, no dist"
timeout -k 10 300 python bench.py --workload spmm --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-1200
