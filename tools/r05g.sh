#!/bin/bash
# round 5: the whole -m gpu suite with k_prefilter_cq as default, the driver's command line, the rocprofv3 evidence at the metric's size
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
TAG=r05g bash tools/run.sh tests
TAG=r05g bash tools/run.sh bench --keep-files
TAG=r05g COMMIT=$1 bash tools/run.sh profile
rm -rf /dev/shm/burst_amd_bench
