# The driver's N = 2 command line (python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2) on a ONE-GPU box:
# BURST_BENCH_DEVICE=0 puts both ranks on device 0 (gloo plumbing; the records take the shared-memory hand-over, bh_node.c).
# What it shows: the multi-process path end to end at the bench's full size, the record count of the single-process run, and
# what the hand-over costs -- not a scaling number (the two ranks share one device).
#   bash tools/bench_two_ranks_one_device.sh [out file] [bench args...]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/bench_two_ranks_one_device.json}; shift
BURST_BENCH_DEVICE=0 BURST_HOST_DEBUG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 \
  $R/bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end "$@" > $OUT 2> $OUT.err
grep -E "bh_search_multi|^\[bench\]|rror" $OUT.err | tail -12
python3 - "$OUT" <<'PY'
import json, sys
ln = [l for l in open(sys.argv[1]) if l.startswith("{")]
if ln:
    d = json.loads(ln[-1])
    print("n_gpus", d["n_gpus"], "value %.1f M reads/s" % (d["value"] / 1e6), "ms_per_step %.3f" % d["ms_per_step"], "records", d["work"]["records"], d["config"]["parallelism"][:120])
PY
