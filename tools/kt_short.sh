#!/bin/bash
# kernel trace of a short bench run: tools/kt_short.sh <tag> <bench args...>   (top kernels to gpurun_out/<tag>_kt.txt)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-continuity --no-strains --no-short-job --no-prime "$@" > $O/${TAG}_kt.log 2>&1
python - "$O/${TAG}_kt" <<'PY' | tee $O/${TAG}_kt.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    for r in [x for x in rows if "k_myers" in x["Name"] or "k_prefilter" in x["Name"] or "k_rescore_reg<0>" in x["Name"]][:8]:
        print('%-60s %6s %11.1f us %10.2f ms %5.1f%%' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
grep '^{' $O/${TAG}_kt.log | tail -1 | python $R/tools/bsum.py $TAG
rm -rf $O/${TAG}_kt
