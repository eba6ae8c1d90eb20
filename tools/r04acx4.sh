#!/bin/bash
# round 4: the command line's upload phase with the query sort on the device / on the host / after the upload
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
W=/dev/shm/acx4
timeout 900 python bench.py --db-scale 7 --workdir $W --keep-files --no-cpu-baseline --no-continuity --no-short-job --no-end-to-end --steps 3 --warmup 1 > $O/r04acx4_prime.json 2> $O/r04acx4_prime.err
EDX=$(ls $W/db_*.edx | head -1); RD=$(ls $W/reads_*.fa | head -1)
for v in devsort hostsort devsort hostsort serial; do
  unset BURST_HOST_SERIAL_INGEST BURST_HOST_SORT
  [ $v = serial ] && export BURST_HOST_SERIAL_INGEST=1
  [ $v = hostsort ] && export BURST_HOST_SORT=1
  T0=$(date +%s.%N)
  BHIP_DEBUG=1 BURST_HOST_DEBUG=1 timeout 600 burst_amd/burst_hip -r $EDX -ad -k 15 -q $RD -o $W/out.b6 -m BEST -i 0.98 > $O/r04acx4_$v.txt 2>&1
  T1=$(date +%s.%N)
  echo "$v: wall $(echo "$T1 - $T0" | bc) s | build $(grep -a 'accelerator built on the device' $O/r04acx4_$v.txt | sed 's/.*per entry; //' | cut -c1-40) | $(grep -a 'device database upload\|database read\|queries parsed\|Alignment time' $O/r04acx4_$v.txt | tr -s ' ' | tr '\n' ' ' | cut -c1-230)"
done
rm -rf $W
