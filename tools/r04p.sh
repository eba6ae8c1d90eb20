#!/bin/bash
# round 4: the other BASELINE shapes on their (small) family database with this round's kernels: configs[1] (100 bp, CAPITALIST, -i 0.97, DB12) and
# configs[2] (292 bp, ALLPATHS, -i 0.97, DB12); lane-set codes against round 3's full masks (option lane_masks 0 = clump-level for comparison)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
C="--db-scale 1 --K 12 --n-base 3300 --n-variants 30 --no-continuity --no-end-to-end --no-short-job --workdir /tmp/bb_fam"
BHIP_DEBUG=1 timeout 900 python bench.py $C --mode CAPITALIST --id 0.97 --edits 0,1,2,3 --cpu-sample 300000 > $O/r04p_configs1.json 2> $O/r04p_configs1.err
echo "configs1 exit $?"; python tools/bsum.py configs1 < $O/r04p_configs1.json
BHIP_DEBUG=1 timeout 900 python bench.py $C --read-len 292 --mode ALLPATHS --id 0.97 --edits 0,2,4,6,8 --cpu-sample 60000 > $O/r04p_configs2.json 2> $O/r04p_configs2.err
echo "configs2 exit $?"; python tools/bsum.py configs2 < $O/r04p_configs2.json
python - <<'PY'
import json
for n in ("configs1","configs2"):
    try:
        d=json.loads([l for l in open('/root/repo/gpurun_out/r04p_%s.json'%n) if l.startswith('{')][-1])
        print(n, "value %.1f M"%(d["value"]/1e6), "cpu", d.get("cpu_baseline") and round(d["cpu_baseline"]["value"]), json.dumps(d.get("parity_vs_reference"))[:500])
    except Exception as e: print(n, "no line", e)
PY
grep "lane 0 class" $O/r04p_configs2.err | tail -1 | cut -c1-300
