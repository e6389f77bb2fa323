mkdir -p gpurun_out/r04i
timeout -k 10 300 python tools/exp_alloc_which.py 10000000 4 > gpurun_out/r04i/which_powerlaw.txt 2>&1; cat gpurun_out/r04i/which_powerlaw.txt
EXP_KIND=uniform timeout -k 10 300 python tools/exp_alloc_which.py 10000000 3 > gpurun_out/r04i/which_uniform.txt 2>&1; cat gpurun_out/r04i/which_uniform.txt
