#!/bin/bash
# round 4: reads per batch (the scheduler's batch size) at the default database size: 2 M (default), 4 M, 8 M
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
C="--db-scale 7 --workdir /dev/shm/bsz --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job"
timeout 900 python bench.py $C --reads 4000000 --steps 10 --warmup 3 > $O/r04y_4m.json 2> $O/r04y_4m.err; echo "4M exit $?"; python tools/bsum.py 4M < $O/r04y_4m.json
timeout 900 python bench.py $C --reads 8000000 --pool 2 --steps 6 --warmup 2 > $O/r04y_8m.json 2> $O/r04y_8m.err; echo "8M exit $?"; python tools/bsum.py 8M < $O/r04y_8m.json
timeout 900 python bench.py $C --reads 2000000 --steps 20 --warmup 5 > $O/r04y_2m.json 2> $O/r04y_2m.err; echo "2M exit $?"; python tools/bsum.py 2M < $O/r04y_2m.json
rm -rf /dev/shm/bsz
