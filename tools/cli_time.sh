# end-to-end timing of the burst_hip command line on the bench workload (database + 1 M reads made by bench.py)
cd /root/repo
python bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>&1
D=/tmp/burst_amd_bench
R=$(ls $D/reads_1000000_l100_*_r0.fa | head -1)
E=$(ls $D/db_*_q110_*.edx | head -1); A=${E%.edx}.acx
for t in 1 2; do time burst_amd/burst_hip -r $E -a $A -q $R -o /tmp/out.b6 -m CAPITALIST -i 0.97 2>&1 | tail -12; done
wc -l /tmp/out.b6
