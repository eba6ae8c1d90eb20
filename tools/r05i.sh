#!/bin/bash
# round 5: the whole -m gpu suite (new: -x, ANY goldens, two-plan accelerator build, deterministic leg), the rocprofv3 evidence at the metric's size,
# the command line's phases, the strains line again with its JSON kept
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
TAG=r05i bash tools/run.sh tests
python bench.py --workdir /dev/shm/burst_amd_bench --db-scale 11.37 --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job --steps 3 --warmup 1 > /dev/null 2> $O/r05i_setup.err
TAG=r05i COMMIT=$1 bash tools/run.sh profile --db-scale 11.37
TAG=r05i bash tools/run.sh cli 11.37
BHIP_ACX_NO_PREMAP=1 TAG=r05i_nopremap bash tools/run.sh cli 11.37
BURST_HOST_SERIAL_INGEST=1 TAG=r05i_serial bash tools/run.sh cli 11.37
rm -rf /dev/shm/burst_amd_bench
