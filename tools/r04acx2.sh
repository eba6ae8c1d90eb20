#!/bin/bash
# round 4: accelerator build A/B on one box: packed 48-bit tuples (new) vs (word << 24 | clump, lane bit) pairs (old: the two libraries built from
# bhip_acx.hip of commit 0d1864f^ -- the parent of the packed-tuple change -- into burst_amd/acxold/, which is not kept in the tree)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
C="--db-scale 7 --workdir /dev/shm/acxab --keep-files --no-cpu-baseline --no-continuity --no-short-job --no-end-to-end --steps 3 --warmup 1"
for round in 1 2 3; do
for v in old new; do
  if [ $v = old ]; then export BURST_AMD_LIBDIR=$R/burst_amd/acxold; else unset BURST_AMD_LIBDIR; fi
  BHIP_DEBUG=1 timeout 900 python bench.py $C > $O/r04acx2_$v$round.json 2> $O/r04acx2_$v$round.err
  echo "$v $round exit $?: $(grep -a 'accelerator built on the device' $O/r04acx2_$v$round.err | cut -c1-250 | head -1)"
  grep -a "device upload + accelerator build" $O/r04acx2_$v$round.err | sed 's/.*(\.edx/(.edx/' | cut -c1-140
done; done
rm -rf /dev/shm/acxab
