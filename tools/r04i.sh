#!/bin/bash
# round 4: bench test fix, clean rocprofv3 evidence at the default size, then the metric's database: bench line + serial shards against one device
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k bench_multi_rank > $O/r04i_test.txt 2>&1; tail -3 $O/r04i_test.txt
PROFILE_COMMIT=$1 bash tools/profile_round.sh r04i --db-scale 7 --workdir /dev/shm/prof7 2>&1 | grep -v "rocprim\|k_acx\|fillBuffer\|k_qs_" | head -16
rm -rf /dev/shm/prof7
W=/dev/shm/bbi
BHIP_DEBUG=1 timeout 1500 python bench.py --workdir $W --db-scale 11.37 --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --ab seed_min_need=3 > $O/r04i_bench_full.json 2> $O/r04i_bench_full.err
echo "bench exit $?"; grep "^\[bench\]" $O/r04i_bench_full.err | cut -c1-300; python tools/bsum.py full < $O/r04i_bench_full.json
EDX=$(ls $W/db_*.edx | head -1); RD=$(ls $W/reads_*.fa | head -1)
head -n 4000000 $RD > $W/sample2m.fa
( time ./burst_amd/burst_hip -r $EDX -ad -k 15 -q $W/sample2m.fa -o $W/one.b6 -m BEST -i 0.98 ) > $O/r04i_cli_one.txt 2>&1
( time ./burst_amd/burst_hip -r $EDX -ad -k 15 -q $W/sample2m.fa -o $W/shards.b6 -m BEST -i 0.98 --gpus 1 --shards 2 ) > $O/r04i_cli_shards.txt 2>&1
grep "serial shards\|Search complete\|\[\|real" $O/r04i_cli_shards.txt | cut -c1-300
grep "Search complete\|\[\|real" $O/r04i_cli_one.txt | cut -c1-200
cmp $W/one.b6 $W/shards.b6 && echo "serial shards .b6 == one device .b6: $(wc -l < $W/one.b6) lines, sha256 $(sha256sum < $W/one.b6 | cut -c1-16)"
( time ./burst_amd/burst_hip -r $EDX -ad -k 15 -q $W/sample2m.fa -o $W/shards_ap.b6 -m ALLPATHS -i 0.98 --gpus 1 --shards 2 ) > $O/r04i_cli_shards_ap.txt 2>&1
./burst_amd/burst_hip -r $EDX -ad -k 15 -q $W/sample2m.fa -o $W/one_ap.b6 -m ALLPATHS -i 0.98 > /dev/null 2>&1
cmp $W/one_ap.b6 $W/shards_ap.b6 && echo "ALLPATHS: serial shards == one device: $(wc -l < $W/one_ap.b6) lines"
rm -rf $W
