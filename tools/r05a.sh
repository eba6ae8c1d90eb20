#!/bin/bash
# round 5, first call: the reference fed through a named pipe (small database first), then the driver's own command line (auto -> the
# metric's 31.5 GB database with the reference beside it) under a memory watcher, then the rocprofv3 passes at that size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/cpu.max > $O/r05a_box.txt 2>&1; df -h /dev/shm /tmp >> $O/r05a_box.txt; cat /proc/sys/fs/pipe-max-size >> $O/r05a_box.txt
T0=$SECONDS
timeout 600 python bench.py --db-scale 1 --steps 5 --warmup 1 --no-end-to-end --no-continuity --no-short-job --cpu-sample 100000 --workdir /dev/shm/bb1 > $O/r05a_small.json 2> $O/r05a_small.err
echo "small exit $? after $((SECONDS - T0)) s"; tail -3 $O/r05a_small.err | cut -c1-300
python - <<'PY' || exit 1
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r05a_small.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference"): print(k, json.dumps(d.get(k))[:700])
assert d["cpu_baseline"] and d["parity_vs_reference"]["identical"]
PY
rm -rf /dev/shm/bb1
(while true; do echo "$(date +%s) $(cat /sys/fs/cgroup/memory.current) $(df --output=used -B1 /dev/shm | tail -1)"; sleep 2; done) > $O/r05a_mem.txt &
MW=$!
T0=$SECONDS
timeout 2000 python bench.py --gpus 1 --steps 20 --warmup 5 --keep-files > $O/r05a_bench.json 2> $O/r05a_bench.err
echo "default bench exit $? after $((SECONDS - T0)) s"; python tools/bsum.py default < $O/r05a_bench.json
kill $MW
grep "^\[bench\]" $O/r05a_bench.err | cut -c1-330
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r05a_bench.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference","gpu_over_cpu","end_to_end"): print(k, json.dumps(d.get(k))[:700])
print(d["config"]["extrapolation"]["this_run_is_at_metric_size"], d["config"]["workload"])
PY
sort -k2 -n $O/r05a_mem.txt | tail -1
T0=$SECONDS
PROFILE_COMMIT=$1 timeout 1500 bash tools/profile_round.sh r05a 2>&1 | grep -v "rocprim\|k_acx\|fillBuffer\|k_qs_" | head -24
echo "profile exit $? after $((SECONDS - T0)) s"
rm -rf /dev/shm/burst_amd_bench
