#!/bin/bash
# Regenerates the round-3 files of profiles/ on the GPU box.  Run from the repository root:
#     bash profiles/run_all.sh [headline|secondary|pmc]...
# Everything is written under gpurun_out/ (the only directory that travels back from the box) and copied into
# profiles/ afterwards.  Every step that uses the GPU runs under its own time limit; the script stops at the first
# failure.
set -e -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
parts="$*"
[ -z "$parts" ] && parts="headline secondary pmc"
B="timeout -k 10 200 python3 bench.py"
for part in $parts; do
case $part in
headline)
    $B > $OUT/r03_bench.json 2> $OUT/r03_bench.err
    # kernel trace and statistics of a short run; the profiler's preloaded library initialises the GPU, so the
    # program itself follows the "--"
    rm -rf $OUT/prof_r03
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r03 -o r03 -- \
        python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-copy > $OUT/r03_bench_profiled.json 2> $OUT/prof_r03.err
    cp $OUT/prof_r03/r03_kernel_stats.csv $OUT/r03_kernel_stats.csv
    ;;
secondary)
    NB="--no-cpu-baseline"
    for c in C1 C2 C3 C5; do $B --config $c $NB > $OUT/r03_bench_$c.json 2>> $OUT/r03_bench.err; done
    $B --config C5 --turbulence 1 $NB > $OUT/r03_bench_C5_turbulence.json 2>> $OUT/r03_bench.err
    $B --doppler 3 --microphysics 1 --steps 3 --warmup 1 $NB > $OUT/r03_bench_C4_spectrum.json 2>> $OUT/r03_bench.err
    $B --microphysics 1 $NB > $OUT/r03_bench_C4_1mom.json 2>> $OUT/r03_bench.err
    $B --gh 9 --steps 5 --warmup 1 $NB > $OUT/r03_bench_gh9.json 2>> $OUT/r03_bench.err
    $B --batch-units 4050000 $NB --no-host-copy > $OUT/r03_bench_eight_batches.json 2>> $OUT/r03_bench.err
    $B --rehearse-gather $NB --no-host-copy > $OUT/r03_bench_gather_rehearsal.json 2>> $OUT/r03_bench.err
    for n in 2 4 8; do $B --shard-of $n $NB --no-host-copy > $OUT/r03_bench_shard_of_$n.json 2>> $OUT/r03_bench.err; done
    ;;
pmc)
    timeout -k 10 1000 python3 profiles/collect_pmc.py r03
    ;;
esac
done
echo "all done: $parts"
