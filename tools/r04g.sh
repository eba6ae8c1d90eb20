#!/bin/bash
# round 4: the whole -m gpu suite, then the rocprofv3 evidence (kernel trace + PMC passes) at the bench's default database size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$SECONDS
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r04g_gputests.txt 2>&1; echo "tests exit $? after $((SECONDS - T0)) s" >> $O/r04g_gputests.txt
tail -6 $O/r04g_gputests.txt
PROFILE_COMMIT=$1 bash tools/profile_round.sh r04g --db-scale 7 --workdir /dev/shm/prof7 2>&1 | tail -30
rm -rf /dev/shm/prof7
