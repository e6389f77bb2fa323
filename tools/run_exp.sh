cd /root/repo
gcc -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/mock_rccl.c -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,/opt/rocm/lib -o /tmp/libmock_rccl.so
SPGPU_RCCL_LIBRARY=/tmp/libmock_rccl.so timeout -k 10 300 python tests/run_sharded_ranks.py 8 banded needed uneven 2>&1 | grep -v amdgpu.ids
SPGPU_RCCL_LIBRARY=/tmp/libmock_rccl.so timeout -k 10 300 python tests/run_sharded_ranks.py 3 random allgather uneven 2>&1 | grep -v amdgpu.ids
