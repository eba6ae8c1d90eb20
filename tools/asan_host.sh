# The C host (libburst_host.so) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU tests that exercise it
# (report, database builders, ingest, record ordering, the shared-memory exchange between processes):
#   bash tools/asan_host.sh [pytest args...]
# Builds a sanitized copy of the library beside the normal one, runs the tests with it preloaded, puts the normal one back.
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/burst_asan.XXXXXX)
for f in $R/burst_amd/csrc/host/bh_*.c; do
  gcc -std=gnu11 -O1 -g -fPIC -fopenmp -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -I$R/include -I$R/burst_amd/csrc/host -c $f -o $W/$(basename $f .c).o || exit 1
done
gcc -shared -fPIC -fopenmp -fsanitize=address,undefined -o $W/libburst_host.so $W/*.o -L$R/burst_amd -lburst_hip -Wl,-rpath,$R/burst_amd -lm -lz || exit 1
cp $R/burst_amd/libburst_host.so $W/normal.so
trap 'cp $W/normal.so $R/burst_amd/libburst_host.so; rm -rf $W' EXIT
cp $W/libburst_host.so $R/burst_amd/libburst_host.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest $R/tests/test_host_cpu.py $R/tests/test_distributed_cpu.py -x -q "$@"
