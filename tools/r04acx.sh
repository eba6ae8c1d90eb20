#!/bin/bash
# round 4: accelerator build with 48-bit packed tuples (keys-only sort, six radix passes): the .acx tests, then the build time at the default size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_acx.py -q -m gpu > $O/r04acx_tests.txt 2>&1; echo "acx tests exit $?"; tail -4 $O/r04acx_tests.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x -k "golden or device or ad" > $O/r04acx_e2e.txt 2>&1; echo "e2e exit $?"; tail -3 $O/r04acx_e2e.txt
BHIP_DEBUG=1 timeout 1200 python bench.py --db-scale 7 --no-cpu-baseline --no-continuity --no-short-job --steps 20 --warmup 5 > $O/r04acx_bench.json 2> $O/r04acx_bench.err
echo "bench exit $?"; grep -a "accelerator built on the device\|device database upload\|upload" $O/r04acx_bench.err | cut -c1-260 | head -6; python tools/bsum.py acx < $O/r04acx_bench.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04acx_bench.json') if l.startswith('{')][-1])
print(json.dumps(d.get("end_to_end"))[:700])
PY
