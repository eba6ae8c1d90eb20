#!/bin/bash
# round 5: the driver's command line with the job's memory broken down beside it, the rocprofv3 evidence at the metric's size, occupancy and
# sub-pipeline A/B, the command line's phases with the record area mapped beside the first pass, the strains database profile
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
TAG=r05h bash tools/run.sh bench --keep-files
TAG=r05h COMMIT=$1 bash tools/run.sh profile
TAG=r05h_cq5 bash tools/run.sh variant cq5 --db-scale 11.37
TAG=r05h bash tools/run.sh ab 11.37 lanes=2 prefilter_waves=14
TAG=r05h bash tools/run.sh cli 11.37
TAG=r05h_again bash tools/run.sh cli 11.37
rm -rf /dev/shm/burst_amd_bench
TAG=r05h_strains bash tools/run.sh bench --workdir /dev/shm/bb_strains --db-profile strains --no-continuity --no-end-to-end
rm -rf /dev/shm/bb_strains
