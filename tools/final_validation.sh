#!/bin/bash
# round 5, final validation: the whole -m gpu suite, the driver's command line in full, the rocprofv3 evidence at the metric's size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
LIMIT=1500 TAG=r05u bash tools/run.sh tests
TAG=r05u bash tools/run.sh bench --keep-files
TAG=r05u COMMIT=7d099f9 bash tools/run.sh profile --db-scale 11.37
rm -rf /dev/shm/burst_amd_bench
