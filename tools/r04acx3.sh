#!/bin/bash
# round 4: where the command line's "device database upload" goes at 19.4 GB: the build's own timer inside burst_hip, with and without the ingest thread beside it
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
W=/dev/shm/acx3
timeout 900 python bench.py --db-scale 7 --workdir $W --keep-files --no-cpu-baseline --no-continuity --no-short-job --no-end-to-end --steps 3 --warmup 1 > $O/r04acx3_prime.json 2> $O/r04acx3_prime.err
EDX=$(ls $W/db_*.edx | head -1); RD=$(ls $W/reads_*.fa | head -1); echo "$EDX $RD"
for v in threaded threaded serial serial; do
  if [ $v = serial ]; then export BURST_HOST_SERIAL_INGEST=1; else unset BURST_HOST_SERIAL_INGEST; fi
  BHIP_DEBUG=1 timeout 600 burst_amd/burst_hip -r $EDX -ad -k 15 -q $RD -o $W/out.b6 -m BEST -i 0.98 > $O/r04acx3_$v.txt 2>&1
  echo "$v: $(grep -a 'accelerator built on the device' $O/r04acx3_$v.txt | sed 's/.*per entry; //' | cut -c1-120) | $(grep -a 'device database upload\|database read\|queries parsed' $O/r04acx3_$v.txt | tr -s ' ' | tr '\n' ' ' | cut -c1-200)"
done
rm -rf $W
