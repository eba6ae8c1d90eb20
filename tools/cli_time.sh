# end-to-end timing of the burst_hip command line on the bench workload (database + read pool made by bench.py):
#   bash tools/cli_time.sh [out file]      -> per-phase wall times printed by the CLI, and its search-phase reads/s
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/cli_phases.txt}
D=${BURST_BENCH_DIR:-/tmp/burst_amd_bench}
python $R/bench.py --no-cpu-baseline --no-end-to-end --steps 2 --warmup 1 > /dev/null 2>&1
EDX=$(ls $D/db_*_k15.edx | head -1); READS=$(ls $D/reads_8000000_l100_*.fa | head -1)
if [ -n "$CLI_READS" ]; then      # a longer job: CLI_READS distinct synthetic reads of the same kind (other seed)
  BIG=$D/cli_reads_$CLI_READS.fa
  [ -f $BIG ] || python3 -c "
import sys; sys.path.insert(0, '$R')
from burst_amd import host
import glob
host.synth_reads(glob.glob('$D/refs_*_k15.fa')[0], '$BIG', $CLI_READS, 100, [0, 1, 2], rc=False, iupac=0.0, seed=4711)"
  READS=$BIG
fi
{
  echo "# burst_hip -r $(basename $EDX) -ad -k 15 -q $(basename $READS) -m BEST -i 0.98   (accelerator built on the device, 2 M-read batches)"
  for i in 1 2; do
    BURST_HOST_DEBUG=1 $R/burst_amd/burst_hip -r $EDX -ad -k 15 -q $READS -o $D/cli.b6 -m BEST -i 0.98 2>&1 | grep -E "^ \[|Search complete|Parsed|Alignment time|Wrote|bh_queries\]|query sort"
    echo
  done
  NR=$(grep -c '^>' $READS)
  echo "reads: $NR"
} > $OUT 2>&1
python3 - "$OUT" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
n = int(re.search(r"reads: (\d+)", t).group(1))
s = [float(x) for x in re.findall(r"\[search \(all batches\)\s+([\d.]+) s\]", t)]
if s:
    open(sys.argv[1], "a").write("search phase (second run): %.4f s -> %.1f M reads/s\n" % (s[-1], n / s[-1] / 1e6))
print(open(sys.argv[1]).read())
PY
