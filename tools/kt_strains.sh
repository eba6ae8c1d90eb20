cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; W=/dev/shm/burst_amd_bench_s
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06v_kt -- python $R/bench.py --db-profile strains --db-scale 1 --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-continuity --no-strains --no-short-job --no-prime --workdir $W > $O/r06v_kt.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('/root/repo/gpurun_out/r06v_kt/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:22]:
        print('%-60s %6s %11.1f us %10.2f ms %5.1f%%' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
rm -rf $W $O/r06v_kt
