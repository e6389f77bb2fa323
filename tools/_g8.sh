mkdir -p gpurun_out/r04h
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04h/gpu_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04h/gpu_tests.txt
tail -n 6 gpurun_out/r04h/gpu_tests.txt
timeout -k 10 900 python bench.py > gpurun_out/r04h/bench_line.json 2> gpurun_out/r04h/bench_err.log; echo "bench rc=$?"
tail -n 5 gpurun_out/r04h/bench_err.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04h/bench_line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ['value','ms_per_step']}, d['roofline']['frac'], d['roofline'].get('kernel_ms_blocks'))
for k,v in d['target'].items(): print(k, v)
print(d['device'])
pl=d['configs']['powerlaw_fp64']
for k in pl:
    if isinstance(pl[k], dict) and 'ms' in pl[k]: print(k, pl[k]['ms'], pl[k]['frac'], pl[k].get('as_built_ms'), pl[k].get('placements_ms'), pl[k].get('plan_counts_uses_builds_stales'))
print(d['spmm_1gpu']['ms_per_step'], d['spmm_1gpu']['roofline_frac'])
PY
SPGPU_LIB=spgpu_amd/lib_trace/libspgpu.so SPGPU_RAGGED_SHAPE=4 EXP_PATTERN=band timeout -k 10 200 python tools/exp_ragged_trace.py 10000000 2048:256 powerlaw > gpurun_out/r04h/trace_plan.txt 2>&1
cat gpurun_out/r04h/trace_plan.txt
