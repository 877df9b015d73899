#!/bin/bash
# Round profiling recipe for the POOLED dense-metric path (GPU box): rocprofv3 kernel stats of `bench.py --config 4 --pooled-metric`, then the HBM read counter
# in a pass of its own (no trace domains besides the kernel trace).  Outputs under gpurun_out/prof_pooled/; summaries are copied into profiles/ by hand.
out=$PWD/gpurun_out/prof_pooled
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --config 4 --pooled-metric --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $out/stats -o r -- $B > $out/bench_stats.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o r -- $B > $out/bench_fetch.json 2> $out/fetch.err
python scripts/summarize_rocprof.py $(find $out -name "*_results.db" | sort) > $out/summary.txt 2>&1
cut -c1-200 $out/summary.txt | head -70
