#!/bin/bash
# round 5: what makes the cooperative test fail once in a while -- the same test over and over with the sort buffers as ordinary allocations
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
for i in 1 2 3 4 5 6 7 8; do BHIP_TEST_NO_VMM_SORT=1 LIMIT=300 TAG=r05s_plain$i bash tools/run.sh tests tests/test_gpu_acx.py -k "cooperative_build_equals"; done
for i in 1 2 3 4; do LIMIT=300 TAG=r05s_vmm$i bash tools/run.sh tests tests/test_gpu_acx.py -k "cooperative_build_equals"; done
grep -h "differ at" $O/r05s_*_gputests.txt | grep -v "assert np" | cut -c1-300
