#!/bin/bash
# round 4, final state: the whole -m gpu suite, the rocprofv3 evidence at the default size, the driver's command line, configs[4]'s shape, the metric's size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$SECONDS
timeout 2400 python -m pytest tests -q -m gpu > $O/r04v_gputests.txt 2>&1; echo "tests exit $? after $((SECONDS - T0)) s" >> $O/r04v_gputests.txt
grep -n "^FAILED\|^ERROR\|passed\|failed\|tests exit" $O/r04v_gputests.txt | tail -8
PROFILE_COMMIT=$1 bash tools/profile_round.sh r04v --db-scale 7 --workdir /dev/shm/prof7 2>&1 | grep "k_prefilter_cf\|k_myers_prefix_task\|k_rescore_reg<0>" | head -6
rm -rf /dev/shm/prof7
T0=$SECONDS
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04v_bench.json 2> $O/r04v_bench.err
echo "default bench exit $? after $((SECONDS - T0)) s"; python tools/bsum.py default < $O/r04v_bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04v_bench.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","parity_vs_reference","gpu_over_cpu"): print(k, json.dumps(d.get(k))[:600])
PY
timeout 900 python bench.py --db-scale 5 --read-len 320 --mode FORAGE --id 0.95 --fr --iupac 0.001 --edits 0,2,4,8,12 --reads 500000 --steps 6 --warmup 2 --cpu-sample 2000 --no-continuity --no-end-to-end --no-short-job > $O/r04v_configs4.json 2> $O/r04v_configs4.err
echo "configs4 exit $?"; python tools/bsum.py configs4 < $O/r04v_configs4.json
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('/root/repo/gpurun_out/r04v_configs4.json') if l.startswith('{')][-1])
    for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference"): print(k, json.dumps(d.get(k))[:600])
except Exception as e: print("no line", e)
PY
timeout 1500 python bench.py --db-scale 11.37 --no-cpu-baseline --no-end-to-end --no-continuity --ab prefilter_rb=3 > $O/r04v_bench_full.json 2> $O/r04v_bench_full.err
echo "full exit $?"; grep "^\[bench\] ab" $O/r04v_bench_full.err | cut -c1-200; python tools/bsum.py full < $O/r04v_bench_full.json
