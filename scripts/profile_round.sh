#!/bin/bash
# Round profiling recipe (GPU box): rocprofv3 kernel stats of the bench command, then the two HBM
# counter passes (separate --pmc runs, no trace domains besides the kernel trace).  Outputs under
# gpurun_out/prof_<tag>/ ; summaries are copied into profiles/ by hand.
tag=${1:-cl}
steps=${2:-20}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd $PWD
rocprofv3 --kernel-trace --stats -d $out/stats -o r -- python bench.py --steps $steps --no-cpu-baseline --no-saturated --no-side > $out/bench_stats.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o r -- python bench.py --steps 4 --no-cpu-baseline --no-saturated --no-side > $out/bench_fetch.json 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o r -- python bench.py --steps 4 --no-cpu-baseline --no-saturated --no-side > $out/bench_write.json 2> $out/write.err
python scripts/summarize_rocprof.py $(find $out -name "*_results.db" | sort) > $out/summary.txt 2>&1
tail -2 $out/bench_stats.json | cut -c1-400
cat $out/summary.txt | cut -c1-220
