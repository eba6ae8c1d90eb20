#!/bin/bash
# round 4: the bench at the metric's own database size (31.5 GB .edx) on one device, memory sampled beside it
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
S=${1:-11.37}
( while true; do echo "$(date +%s) $(cat /sys/fs/cgroup/memory.current 2>/dev/null) $(df --output=used -B1 /dev/shm | tail -1) $(rocm-smi --showmemuse --csv 2>/dev/null | tail -1)"; sleep 2; done ) > $O/r04e_mem.txt &
MON=$!
T0=$SECONDS
BHIP_DEBUG=1 timeout 2700 python bench.py --db-scale $S --no-continuity --steps 20 --warmup 5 > $O/r04e_bench.json 2> $O/r04e_bench.err
echo "bench exit $? after $((SECONDS - T0)) s"
kill $MON
grep "^\[bench\]\|accelerator built" $O/r04e_bench.err | grep -v "^\[bench\] ab" | cut -c1-700
grep "prefilter kernel:" $O/r04e_bench.err | sort | uniq -c
python tools/bsum.py full < $O/r04e_bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04e_bench.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference","end_to_end"):
    print(k, json.dumps(d.get(k))[:900])
print(d["config"]["workload"])
PY
awk '{ if ($2>m) m=$2; if ($3>s) s=$3 } END { printf "peak memory.current %.1f GB, peak /dev/shm %.1f GB\n", m/1e9, s/1e9 }' $O/r04e_mem.txt
tail -3 $O/r04e_bench.err | cut -c1-500
