#!/bin/bash
# round 4: configs[3]'s own job end to end on ONE device: burst_hip -r <31.5 GB .edx> -ad -k 15 -q <10 M reads> (bench.py's end_to_end at the metric's size)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1700 python bench.py --db-scale 11.37 --pool 5 --no-cpu-baseline --no-continuity --no-short-job --steps 20 --warmup 5 > $O/r04zz_bench_full_e2e.json 2> $O/r04zz_bench_full_e2e.err
echo "exit $?"; python tools/bsum.py full < $O/r04zz_bench_full_e2e.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04zz_bench_full_e2e.json') if l.startswith('{')][-1])
print(json.dumps(d.get("end_to_end"))[:900])
PY
