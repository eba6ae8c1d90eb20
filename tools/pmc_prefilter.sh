cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/lds_counters.txt
for i in 1 2; do
  if [ $i = 1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAVES"; fi
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pf$i -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --lanes 1 > $R/gpurun_out/pmc_pf$i.log 2>&1
done
python - <<'PY'
import csv,glob,collections
for i in (1,2):
    for f in glob.glob('/root/repo/gpurun_out/pmc_pf%d/**/*counter_collection.csv'%i, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:40]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        for k,v in agg.items():
            if 'prefilter_mask' in k or 'rescore' in k or 'myers' in k: print(k, dict(v))
PY
