#!/bin/bash
# round 4: end-to-end and full-size GPU tests (all failures listed), the driver's bench command line with the host A/B, the rocprofv3 evidence
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$SECONDS
timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -m gpu > $O/r04h_gputests.txt 2>&1; echo "tests exit $? after $((SECONDS - T0)) s" >> $O/r04h_gputests.txt
grep -n "^FAILED\|^ERROR\|passed\|failed\|tests exit" $O/r04h_gputests.txt | tail -12
T0=$SECONDS
BHIP_DEBUG=1 timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 --ab-host > $O/r04h_bench.json 2> $O/r04h_bench.err
echo "bench exit $? after $((SECONDS - T0)) s"
grep "^\[bench\]" $O/r04h_bench.err | cut -c1-400
python tools/bsum.py default < $O/r04h_bench.json
PROFILE_COMMIT=$1 bash tools/profile_round.sh r04h --db-scale 7 --workdir /dev/shm/prof7 2>&1 | grep -v "rocprim\|k_acx\|fillBuffer\|k_qs_" | head -24
rm -rf /dev/shm/prof7
