#!/bin/bash
# round 4, third GPU call: the whole default bench at the metric's own database size (31.5 GB .edx), timed as the driver would run it
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
S=${1:-11.37}
T0=$SECONDS
BHIP_DEBUG=1 timeout 2700 python bench.py --db-scale $S --no-continuity > $O/r04c_bench.json 2> $O/r04c_bench.err
echo "bench exit $? after $((SECONDS - T0)) s"
grep "^\[bench\]\|accelerator built\|lane masks" $O/r04c_bench.err | grep -v "^\[bench\] ab" | cut -c1-500
grep "prefilter kernel:" $O/r04c_bench.err | sort | uniq -c
python tools/bsum.py full < $O/r04c_bench.json
tail -3 $O/r04c_bench.err | cut -c1-600
df -h /dev/shm | tail -1
