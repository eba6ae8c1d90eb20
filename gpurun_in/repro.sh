W=/tmp/burst_amd_bench
REFS=$(ls $W/refs_*.fa | head -1); READS=$(ls $W/reads_*_r0.fa | head -1)
grep -A1 -E "^>ref_b99_v|^>ref_b3200_v" $REFS | grep -v "^--" > gpurun_out/repro_refs.fa
grep -A1 -E "^>read129108_|^>read28216_" $READS | grep -v "^--" > gpurun_out/repro_reads.fa
grep -c ">" gpurun_out/repro_refs.fa gpurun_out/repro_reads.fa
for exe in oracle/_ref/burst12 burst_amd/burst_hip; do
  $exe -r gpurun_out/repro_refs.fa -d QUICK 110 -s 500 -i 0.97 -o /tmp/rp.edx -a /tmp/rp.acx > /dev/null 2>&1
  for acc in "" "-a /tmp/rp.acx"; do
    $exe -r /tmp/rp.edx $acc -q gpurun_out/repro_reads.fa -o /tmp/rp.b6 -m ALLPATHS -i 0.97 > /dev/null 2>&1; echo "== $exe $acc"; sort /tmp/rp.b6 | grep -E "v12|v1	" | head -6
  done
done
