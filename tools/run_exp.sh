cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_bench_rehearsal.py tests/test_gpu_sharded_c.py -x -q 2>&1 | tail -3 &&
timeout -k 10 300 python bench.py --workload spmm --steps 20 --warmup 3 --no-extras 2>gpurun_out/spmm1.err | tail -1 | cut -c1-400 &&
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --workload spmm --steps 20 --warmup 3 --no-extras 2>gpurun_out/spmm1t.err | tail -1 | cut -c1-400
tail -3 gpurun_out/spmm1t.err
