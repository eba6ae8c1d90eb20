#!/bin/bash
# round 4: BASELINE configs[4]'s shape at its database size (320-bp reads with IUPAC codes, both strands, FORAGE at 95 %, ~20 GB .edx) on one device,
# then the driver's default command line once more (final state of bench.py)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
( while true; do echo "$(date +%s) $(cat /sys/fs/cgroup/memory.current 2>/dev/null)"; sleep 2; done ) > $O/r04j_mem.txt &
MON=$!
T0=$SECONDS
BHIP_DEBUG=1 timeout 2400 python bench.py --db-scale 5 --read-len 320 --mode FORAGE --id 0.95 --fr --iupac 0.01 --edits 0,2,4,8,12 --reads 500000 --steps 8 --warmup 2 --cpu-sample 1200 --no-continuity --no-end-to-end --no-short-job > $O/r04j_configs4.json 2> $O/r04j_configs4.err
echo "configs4 bench exit $? after $((SECONDS - T0)) s"
grep "^\[bench\]\|accelerator built\|overflowed" $O/r04j_configs4.err | sort | uniq -c | sort -rn | head -14 | cut -c1-420
grep "prefilter kernel:" $O/r04j_configs4.err | sort | uniq -c
python tools/bsum.py configs4 < $O/r04j_configs4.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04j_configs4.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference","gpu_over_cpu"):
    print(k, json.dumps(d.get(k))[:900])
print(d["config"]["workload"])
PY
T0=$SECONDS
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04j_bench.json 2> $O/r04j_bench.err
echo "default bench exit $? after $((SECONDS - T0)) s"
kill $MON
python tools/bsum.py default < $O/r04j_bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04j_bench.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","parity_vs_reference","consolidation","one_rank_share_of_configs3"):
    print(k, json.dumps(d.get(k))[:500])
print(json.dumps({k:v for k,v in d["roofline"].items() if k!="per_kernel" and k!="note"})[:900])
PY
awk '{ if ($2>m) m=$2 } END { printf "peak memory.current %.1f GB\n", m/1e9 }' $O/r04j_mem.txt
