#!/bin/bash
# round 4, final state: the whole -m gpu suite, the rocprofv3 evidence at the default size, the driver's command line
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$SECONDS
timeout 2400 python -m pytest tests -q -m gpu > $O/r04final_gputests.txt 2>&1; echo "tests exit $? after $((SECONDS - T0)) s" >> $O/r04final_gputests.txt
grep -n "^FAILED\|^ERROR\|passed\|failed\|tests exit" $O/r04final_gputests.txt | tail -8
PROFILE_COMMIT=$1 bash tools/profile_round.sh r04final --db-scale 7 --workdir /dev/shm/prof7 2>&1 | grep "k_prefilter_cf\|k_myers_prefix_task\|k_rescore_reg<0>" | head -6
rm -rf /dev/shm/prof7
T0=$SECONDS
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04final_bench.json 2> $O/r04final_bench.err
echo "default bench exit $? after $((SECONDS - T0)) s"; python tools/bsum.py default < $O/r04final_bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04final_bench.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference","gpu_over_cpu"): print(k, json.dumps(d.get(k))[:500])
print("pmc_source", d["roofline"].get("pmc_source"))
PY
