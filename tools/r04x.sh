#!/bin/bash
# round 4: sub-pipelines per batch re-measured with this round's prefilter (round 2: lanes = 2 lost 8 %), at the default size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python bench.py --db-scale 7 --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job --steps 20 --warmup 5 \
  --ab lanes=2 --ab lanes=3 --ab lanes=2,sweep_blocks=6 --ab lanes=2,oversub=2 --ab lanes=4 > $O/r04x_bench.json 2> $O/r04x_bench.err
echo "exit $?"; grep "^\[bench\] ab" $O/r04x_bench.err | cut -c1-220; python tools/bsum.py lanes < $O/r04x_bench.json
