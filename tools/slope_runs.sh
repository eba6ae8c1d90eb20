#!/bin/bash
# bench lines over several database sizes (GPU box): ms per batch against accelerator records per read -> tools/slope_fit.py
#   bash tools/slope_runs.sh "0.5 2 4 6"      (db-scale values; 1 = 4.5 Gbp; no CPU baseline, the .acx for the reference would not fit the disk)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out
for S in ${1:-"0.5 2 4"}; do
  W=/tmp/burst_slope_$S
  BHIP_DEBUG=1 python $R/bench.py --db-scale $S --no-cpu-baseline --steps 20 --warmup 5 --drop-refs --no-end-to-end --workdir $W > $R/gpurun_out/slope_$S.json 2> $R/gpurun_out/slope_$S.err
  grep "^\[bench\]\|accelerator built" $R/gpurun_out/slope_$S.err | cut -c1-400
  python $R/tools/bsum.py scale_$S < $R/gpurun_out/slope_$S.json
  rm -rf $W
done
