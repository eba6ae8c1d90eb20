#!/bin/bash
# round 4: BASELINE configs[4]'s shape at its database size (320-bp reads with IUPAC codes, both strands, FORAGE at 95 %, 18 GB .edx) on one device.
# IUPAC rate 0.1 %: a read with five or more ambiguous symbols has no guaranteed seed word and is aligned against EVERY clump (as the reference's
# "bad" bin is, burst.c:3130-3131) -- at 1 % that is a fifth of the reads and 3 M clumps each: the first attempt did not finish in 40 minutes
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$SECONDS
BHIP_DEBUG=1 timeout 1100 python bench.py --db-scale 5 --read-len 320 --mode FORAGE --id 0.95 --fr --iupac 0.001 --edits 0,2,4,8,12 --reads 500000 --steps 6 --warmup 2 --cpu-sample 600 --no-continuity --no-end-to-end --no-short-job > $O/r04k_configs4.json 2> $O/r04k_configs4.err
echo "configs4 bench exit $? after $((SECONDS - T0)) s"
grep "^\[bench\]\|accelerator built\|overflowed" $O/r04k_configs4.err | sort | uniq -c | sort -rn | head -14 | cut -c1-420
grep "^\[bhip\] lane 0 class" $O/r04k_configs4.err | tail -2 | cut -c1-400
grep "prefilter kernel:" $O/r04k_configs4.err | sort | uniq -c
python tools/bsum.py configs4 < $O/r04k_configs4.json
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('/root/repo/gpurun_out/r04k_configs4.json') if l.startswith('{')][-1])
    for k in ("cpu_baseline","cpu_baseline_skipped","parity_vs_reference","gpu_over_cpu"):
        print(k, json.dumps(d.get(k))[:900])
    print(d["config"]["workload"])
except Exception as e: print("no line", e)
PY
