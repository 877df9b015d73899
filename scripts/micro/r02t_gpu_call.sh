#!/bin/bash
mkdir -p gpurun_out/r02t
export TMPDIR=/tmp
out=$PWD/gpurun_out/r02t
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o r -- python bench.py --config 4 --steps 4 --warmup 0 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
head -2 $f | cut -c1-600
python scripts/micro/trace_by_grid.py $f k_dn_symv > $out/by_grid.txt 2>&1
cat $out/by_grid.txt
rm -rf $out/trace
