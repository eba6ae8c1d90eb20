#!/bin/bash
# round 5, final validation: the whole -m gpu suite (a test that hangs fails after 10 minutes instead of taking the run with it), the
# driver's command line in full, the rocprofv3 evidence at the metric's size.  STEPS="tests bench profile" picks a part.
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
TAG=${TAG:-final}; STEPS=${STEPS:-tests bench profile}
for s in $STEPS; do case $s in
tests) LIMIT=${LIMIT:-1500} TAG=$TAG bash tools/run.sh tests ${TESTS:-tests} --timeout 600 ;;
bench) TAG=$TAG bash tools/run.sh bench --keep-files ;;
profile) TAG=$TAG COMMIT=${COMMIT:-unknown} bash tools/run.sh profile --db-scale 11.37 ;;
esac; done
rm -rf /dev/shm/burst_amd_bench
