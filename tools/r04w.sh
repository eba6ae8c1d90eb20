#!/bin/bash
# round 4: the driver's command line once more (cpu_baseline without a pseudo-terminal: stdbuf -oL on a pipe)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
which stdbuf; ls /dev/pts 2>&1 | head -3
T0=$SECONDS
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04w_bench.json 2> $O/r04w_bench.err
echo "default bench exit $? after $((SECONDS - T0)) s"; python tools/bsum.py default < $O/r04w_bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04w_bench.json') if l.startswith('{')][-1])
for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference","gpu_over_cpu"): print(k, json.dumps(d.get(k))[:700])
PY
