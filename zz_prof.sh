cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout -k 10 200 python3 bench.py > gpurun_out/prof/r02_bench.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-copy > gpurun_out/prof/r02_bench_profiled.log 2>&1 || exit 1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) gpurun_out/prof/r02_kernel_stats.csv
timeout -k 10 600 python3 profiles/collect_pmc.py r02 > gpurun_out/prof/pmc.log 2>&1 || exit 1
rm -rf gpurun_out/pmc_pass*
timeout -k 10 200 python3 bench.py --config C5 --no-host-copy > gpurun_out/prof/c5.log 2>&1 || exit 1
timeout -k 10 200 python3 bench.py --config C5 --turbulence 1 --no-host-copy --no-cpu-baseline > gpurun_out/prof/c5t.log 2>&1 || exit 1
for c in C1 C2 C3; do timeout -k 10 200 python3 bench.py --config $c --no-host-copy --no-cpu-baseline > gpurun_out/prof/$c.log 2>&1 || exit 1; done
echo done
