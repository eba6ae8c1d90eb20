#!/bin/bash
# round 4: the whole -m gpu suite (all failures listed), then configs[4]'s shape at 18 GB again (expansions of ambiguous query words)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$SECONDS
timeout 2400 python -m pytest tests -q -m gpu > $O/r04n_gputests.txt 2>&1; echo "tests exit $? after $((SECONDS - T0)) s" >> $O/r04n_gputests.txt
grep -n "^FAILED\|^ERROR\|passed\|failed\|tests exit" $O/r04n_gputests.txt | tail -12
if [ "$1" = "configs4" ]; then
T0=$SECONDS
BHIP_DEBUG=1 timeout 900 python bench.py --db-scale 5 --read-len 320 --mode FORAGE --id 0.95 --fr --iupac 0.001 --edits 0,2,4,8,12 --reads 500000 --steps 6 --warmup 2 --cpu-sample 600 --no-continuity --no-end-to-end --no-short-job > $O/r04n_configs4.json 2> $O/r04n_configs4.err
echo "configs4 bench exit $? after $((SECONDS - T0)) s"
grep "^\[bhip\] lane 0 class" $O/r04n_configs4.err | sed -n 12,15p | cut -c1-330
python tools/bsum.py configs4 < $O/r04n_configs4.json
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('/root/repo/gpurun_out/r04n_configs4.json') if l.startswith('{')][-1])
    for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference"):
        print(k, json.dumps(d.get(k))[:700])
except Exception as e: print("no line", e)
PY
fi
