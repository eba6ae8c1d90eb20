#!/bin/bash
# round 4: the RCCL extra of bench.py at its new place (last, under a watchdog): one process under torch.distributed.run, then the watchdog itself
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
C="--db-scale 1 --n-base 20000 --reads 100000 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-continuity --workdir /tmp/rcclw"
BURST_BENCH_DIST1=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 $C > $O/r04rccl_1.json 2> $O/r04rccl_1.err
echo "exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04rccl_1.json') if l.startswith('{')][-1])
print("value", d["value"], "rccl", json.dumps(d.get("rccl"))[:300])
PY
BURST_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29672 bench.py --gpus 2 $C > $O/r04rccl_2.json 2> $O/r04rccl_2.err
echo "exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r04rccl_2.json') if l.startswith('{')][-1])
print("value", d["value"], "n_gpus", d["n_gpus"], "rccl", json.dumps(d.get("rccl"))[:200], "configs3", json.dumps(d.get("configs3_job"))[:120])
PY
