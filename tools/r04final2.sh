#!/bin/bash
# round 4, final state after the accelerator-build change: the whole -m gpu suite, the driver's command line, configs[3]'s job end to end at the metric's size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$SECONDS
timeout 2400 python -m pytest tests -q -m gpu > $O/r04final2_gputests.txt 2>&1; echo "tests exit $? after $((SECONDS - T0)) s" >> $O/r04final2_gputests.txt
grep -n "^FAILED\|^ERROR\|passed\|failed\|tests exit" $O/r04final2_gputests.txt | tail -8
T0=$SECONDS
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04final2_bench.json 2> $O/r04final2_bench.err
echo "default bench exit $? after $((SECONDS - T0)) s"; python tools/bsum.py default < $O/r04final2_bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04final2_bench.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference","gpu_over_cpu","end_to_end"): print(k, json.dumps(d.get(k))[:600])
PY
T0=$SECONDS
timeout 1700 python bench.py --db-scale 11.37 --pool 5 --no-cpu-baseline --no-continuity --no-short-job --steps 20 --warmup 5 > $O/r04final2_bench_full_e2e.json 2> $O/r04final2_bench_full_e2e.err
echo "full exit $? after $((SECONDS - T0)) s"; python tools/bsum.py full < $O/r04final2_bench_full_e2e.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04final2_bench_full_e2e.json') if l.startswith('{')][-1])
print(json.dumps(d.get("end_to_end"))[:900])
PY
