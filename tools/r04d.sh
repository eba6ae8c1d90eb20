#!/bin/bash
# round 4, GPU call: the driver's own bench command line (auto scale), with the job's memory use sampled beside it; then kernel-level tests
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
( while true; do echo "$(date +%s) $(cat /sys/fs/cgroup/memory.current 2>/dev/null) $(df --output=used -B1 /dev/shm | tail -1)"; sleep 2; done ) > $O/r04d_mem.txt &
MON=$!
T0=$SECONDS
BHIP_DEBUG=1 timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04d_bench.json 2> $O/r04d_bench.err
echo "bench exit $? after $((SECONDS - T0)) s"
kill $MON
grep "^\[bench\]\|accelerator built" $O/r04d_bench.err | grep -v "^\[bench\] ab" | cut -c1-600
python tools/bsum.py default < $O/r04d_bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04d_bench.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference","gpu_over_cpu","end_to_end","continuity_small_db"):
    print(k, json.dumps(d.get(k))[:700])
print(d["config"]["workload"])
PY
awk '{ if ($2>m) m=$2; if ($3>s) s=$3 } END { printf "peak memory.current %.1f GB, peak /dev/shm %.1f GB\n", m/1e9, s/1e9 }' $O/r04d_mem.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_acx.py -x -q -m gpu > $O/r04d_tests.txt 2>&1; echo "tests exit $?" >> $O/r04d_tests.txt
tail -4 $O/r04d_tests.txt
