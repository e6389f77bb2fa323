cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_oell_device.py -x -q -k "replayed_graph or two_handles" 2>&1 | tail -5
