# BASELINE configs[3]'s job shape (10 M 100-bp reads, BEST, -i 0.98, 8 ranks) through the native multi-rank search with every
# rank on ONE device (a 1-GPU box is all there is): per-rank align times, and the .b6 compared byte for byte with the 1-rank run.
#   bash tools/multi_rank_one_device.sh [out file]         (after a bench.py run: uses its database and reference FASTA)
# Rank layouts: 4 ranks query-sharded with the database replicated; 8 ranks as 2 database shards x 4 replica groups (what a
# database of twice a device's capacity would take on 8 GPUs).  The ranks share one device, so the times say how even the
# shares are and what the exchange costs -- not how fast 8 devices are.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/multi_rank_one_device.txt}
D=${BURST_BENCH_DIR:-/tmp/burst_amd_bench}
N=${MR_READS:-10000000}
EDX=$(ls $D/db_*_k15.edx 2>/dev/null | head -1)
if [ -z "$EDX" ]; then python $R/bench.py --no-cpu-baseline --no-end-to-end --steps 2 --warmup 1 > /dev/null 2>&1; EDX=$(ls $D/db_*_k15.edx | head -1); fi
READS=$D/mr_reads_$N.fa
[ -f $READS ] || python3 -c "
import sys, glob; sys.path.insert(0, '$R')
from burst_amd import host
host.synth_reads(glob.glob('$D/refs_*_k15.fa')[0], '$READS', $N, 100, [0, 1, 2], rc=False, iupac=0.0, seed=977)"
run() {   # name, flags...
  local name=$1; shift
  echo "## burst_hip -r $(basename $EDX) -ad -k 15 -q <$N reads> -m BEST -i 0.98 $*"
  BURST_HOST_DEBUG=1 timeout 600 $R/burst_amd/burst_hip -r $EDX -ad -k 15 -q $READS -o $D/mr_$name.b6 -m BEST -i 0.98 "$@" > $D/mr_$name.log 2>&1
  local rc=$?
  grep -E "^Rank|gather:|bh_search_multi|Search complete|search \(all|device database upload|Alignment time|Wrote" $D/mr_$name.log
  if [ $rc -ne 0 ]; then echo "exit code $rc; last lines:"; tail -5 $D/mr_$name.log; fi
  echo
}
{
  run 1
  run 4q --gpus 4 --devices 0,0,0,0 --gather host
  run 8s --gpus 8 --devices 0,0,0,0,0,0,0,0 --gather host --shards 2
  for n in 4q 8s; do
    if cmp -s $D/mr_1.b6 $D/mr_$n.b6; then echo "output of $n == output of the 1-rank run, byte for byte ($(wc -l < $D/mr_1.b6) lines)"; else echo "output of $n DIFFERS from the 1-rank run"; fi
  done
} > $OUT 2>&1
cat $OUT
