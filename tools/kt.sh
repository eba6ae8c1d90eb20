# per-kernel time summary of one bench configuration: tools/kt.sh <outdir> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$O -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$O.log 2>&1
python - "$R/gpurun_out/$O" <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:22]:
        print('%-60s calls %6s avg %10.1f us total %8.2f ms %5.1f%%'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
