#!/bin/bash
# round 4: the driver's N = 2 command line with both ranks on the one device of the box (BURST_BENCH_DEVICE=0), 2.5 units of database (6.9 GB .edx per rank)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
BURST_BENCH_DEVICE=0 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 --db-scale 2.5 \
  > $O/r04two_bench_two_ranks_one_device.json 2> $O/r04two_bench_two_ranks_one_device.err
echo "exit $?"; python tools/bsum.py two < $O/r04two_bench_two_ranks_one_device.json
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04two_bench_two_ranks_one_device.json') if l.startswith('{')][-1])
for k in ("n_gpus","scaling","value","ms_per_step","weak_scaling","configs3_job","rccl","handover"): print(k, json.dumps(d.get(k))[:400])
PY
