#!/bin/bash
# Parity at the size of the bench workload (GPU box): the compiled reference (all host threads, accelerator) and burst_hip
# on the same reads / database.  BEST does not depend on the order in which the reference's threads find hits and must be
# identical; in ALLPATHS the reference's DUPE_HUNT keeps the first placement it met among overlapping shears
# (burst.c:4563-4570, 4604), so there the contract is the relaxed one of tests/goldenlib.py: same number of lines, and
# every reference line is one of the placements burst_hip computes (--no-dupe-hunt prints all of them).
#   bash tools/scale_diff.sh [reads]      (after bench.py has built /tmp/burst_amd_bench)
set -u
N=${1:-200000}
W=${BURST_BENCH_DIR:-/tmp/burst_amd_bench}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
EDX=${SD_EDX:-$(ls $W/db_*_q110_*.edx | head -1)}; ACX=${EDX%.edx}.acx          # other shapes: SD_EDX / SD_READS / SD_IDS / SD_EXTRA (e.g. -fr)
READS=${SD_READS:-$(ls $W/reads_*_l100_*_r0.fa | head -1)}
IDS=${SD_IDS:-"0.97 0.98"}
EXTRA=${SD_EXTRA:-}
REFBIN=${SD_REF:-$ROOT/oracle/_ref/burst12}      # DB15 accelerators: SD_REF=oracle/_ref/burst15 SD_HIP_EXTRA="-k 15"
HIPX=${SD_HIP_EXTRA:-}
HIPACC=${SD_HIP_ACCEL:-"-a $ACX"}                # burst_hip without the file: SD_HIP_ACCEL="-ad -k 15" (accelerator built on the device)
REFACC="-a $ACX"; THREADS=${SD_THREADS:-$(nproc)}
T=${SD_TAG:-}                                    # several runs side by side in one work directory: a tag of their own for the scratch files
# SD_EXHAUSTIVE=1: both programs without an accelerator; with SD_THREADS=1 that is the reference's DETERMINISTIC configuration (one hit list
# per query, clumps ascending: bh_report.c) -- every mode must then be identical byte for byte
if [ "${SD_EXHAUSTIVE:-0}" = 1 ]; then HIPACC=""; REFACC=""; fi
head -n $((2 * N)) $READS > $W/sd_reads$T.fa
secs() { awk -v a=$1 -v b=$2 'BEGIN { printf "%.2f", b - a }'; }
for MODE in ${SD_MODES:-BEST ALLPATHS}; do
  for ID in $IDS; do
    T0=$(date +%s.%N); $REFBIN -r $EDX $REFACC -q $W/sd_reads$T.fa -o $W/sd_ref$T.b6 -m $MODE -i $ID $EXTRA -t $THREADS --noprogress > $W/sd_ref$T.log 2>&1; T1=$(date +%s.%N)
    $ROOT/burst_amd/burst_hip -r $EDX $HIPACC -q $W/sd_reads$T.fa -o $W/sd_hip$T.b6 -m $MODE -i $ID $EXTRA $HIPX > $W/sd_hip$T.log 2>&1; T2=$(date +%s.%N)
    sort $W/sd_ref$T.b6 > $W/sd_ref$T.s; sort $W/sd_hip$T.b6 > $W/sd_hip$T.s
    NR=$(wc -l < $W/sd_ref$T.s); NH=$(wc -l < $W/sd_hip$T.s)
    if cmp -s $W/sd_ref$T.s $W/sd_hip$T.s; then R=IDENTICAL
    else
      ND=$(diff $W/sd_ref$T.s $W/sd_hip$T.s | grep -c '^<')
      R="$ND of $NR lines differ"
      if [ $MODE = CAPITALIST ]; then
        QD=$(diff <(cut -f1 $W/sd_ref$T.s | uniq) <(cut -f1 $W/sd_hip$T.s | uniq) | wc -l)
        # every reference line must be one of the query's minimum-edit-distance placements (what -m ALLPATHS --no-dupe-hunt prints):
        # which of several equally voted ones is kept depends on the reference's hit order (burst.c:4763-4776)
        $ROOT/burst_amd/burst_hip -r $EDX $HIPACC -q $W/sd_reads$T.fa -o $W/sd_nd$T.b6 -m ALLPATHS -i $ID $EXTRA $HIPX --no-dupe-hunt > /dev/null 2>&1
        sort -u $W/sd_nd$T.b6 > $W/sd_nd$T.s
        MISSING=$(comm -23 $W/sd_ref$T.s $W/sd_nd$T.s | wc -l)
        R="$R (equally voted placements, decided by the reference's hit order); reference lines that are not a placement burst_hip computed: $MISSING; queries reported by only one program: $QD; line counts $NR / $NH"
      elif [ $MODE != BEST ]; then
        $ROOT/burst_amd/burst_hip -r $EDX $HIPACC -q $W/sd_reads$T.fa -o $W/sd_nd$T.b6 -m $MODE -i $ID $EXTRA $HIPX --no-dupe-hunt > /dev/null 2>&1
        sort -u $W/sd_nd$T.b6 > $W/sd_nd$T.s
        MISSING=$(comm -23 $W/sd_ref$T.s $W/sd_nd$T.s | wc -l)
        QD=$(diff <(cut -f1 $W/sd_ref$T.s | uniq) <(cut -f1 $W/sd_hip$T.s | uniq) | wc -l)
        R="$R; reference lines that are not a placement burst_hip computed: $MISSING; queries reported by only one program: $QD; line counts $NR / $NH"
      fi
    fi
    echo "$MODE -i $ID: $N reads, $NR reference lines, $NH burst_hip lines: $R   [reference $(secs $T0 $T1) s on $THREADS threads, burst_hip $(secs $T1 $T2) s, both incl. database load]"
  done
done
