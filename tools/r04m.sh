#!/bin/bash
# round 4: A/B of the scheduling knobs at the default database size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python bench.py --db-scale 7 --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job --ab seed_ahead_blocks=1 --ab seed_ahead_blocks=4 --ab seed_ahead_blocks=8 --ab seed_ahead=0 --ab peq_ahead_blocks=4 --ab oversub=1 --ab oversub=4 --ab sweep_blocks=4 > $O/r04m_bench.json 2> $O/r04m_bench.err
echo "bench exit $?"
grep "^\[bench\] ab" $O/r04m_bench.err | cut -c1-200
python tools/bsum.py s7 < $O/r04m_bench.json
