#!/bin/bash
# round 4: reads beyond 1 024 symbols (k_myers_long) -- the new tests first, then the two test files they live in
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -m gpu -x -k "beyond_1024" > $O/r04z_long.txt 2>&1; echo "long exit $?"; tail -30 $O/r04z_long.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -m gpu > $O/r04z_files.txt 2>&1; echo "files exit $?"; grep -n "^FAILED\|^ERROR\|passed\|failed" $O/r04z_files.txt | tail -8
