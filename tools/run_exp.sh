cd /tmp && export TMPDIR=/tmp
export EXP_ONLY_WINDOWED=1 EXP_ORDERS=2048:256 EXP_FORMS=ragged0
for pat in near band; do
export EXP_PATTERNS=$pat
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_r02c_${pat}/$c -- python3 /root/repo/tools/exp_tile.py D 10000000 powerlaw > /root/repo/gpurun_out/pmc_r02c_${pat}_$c.log 2>&1 || exit 1
done
echo "== $pat"; grep "^D " /root/repo/gpurun_out/pmc_r02c_${pat}_WRITE_SIZE.log
python3 /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_r02c_${pat} raggedSpmvKernel deepItemsKernel deepFinishKernel
done
