#!/bin/bash
# burst_hip --gpus 2 --devices 0,0 -ad on the bench's 2.77 GB database (both ranks on the one device of the box): the ranks build the
# accelerator together -- against every rank building alone (BURST_HIP_SOLO_BUILD=1) and against one rank; the three .b6 must be one file
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
W=/dev/shm/burst_amd_coop; TAG=${TAG:-coopcli}
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --workdir $W --db-scale 1 --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job --steps 2 --warmup 1 > /dev/null 2> $O/${TAG}_setup.err
EDX=$(ls $W/db_*.edx | head -1); RD=$(ls $W/reads_*.fa | head -1)
run() { BHIP_DEBUG=1 BURST_HOST_DEBUG=1 timeout 600 burst_amd/burst_hip -r $EDX -ad -k 15 -q $RD -m BEST -i 0.98 "$@" 2>&1; }
run -o $W/one.b6 > $O/${TAG}_one_rank.txt
run -o $W/coop.b6 --gpus 2 --devices 0,0 > $O/${TAG}_two_ranks_together.txt
BURST_HIP_SOLO_BUILD=1 run -o $W/solo.b6 --gpus 2 --devices 0,0 > $O/${TAG}_two_ranks_alone.txt
for f in one_rank two_ranks_together two_ranks_alone; do echo "== $f"; grep "accelerator built\|built by\|device database upload\|Alignment time\|rank . of" $O/${TAG}_$f.txt | cut -c1-330; done
wc -l $W/one.b6 $W/coop.b6 $W/solo.b6; cmp $W/one.b6 $W/coop.b6 && cmp $W/one.b6 $W/solo.b6 && echo "THREE OUTPUTS IDENTICAL"
rm -rf $W
