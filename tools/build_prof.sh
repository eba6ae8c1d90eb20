#!/bin/bash
# a second build of the libraries with the prefilter's phase timers (-DPFM_PROF=2: s_memrealtime between the phases of k_prefilter_cf,
# summed per phase; read with bhip_debug_prof) into burst_amd/prof/ -- used as BURST_AMD_LIBDIR=burst_amd/prof BHIP_PROF=1 python bench.py ...
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/burst_amd && cp -r $R/burst_amd/csrc $T/burst_amd/ && cp -r $R/include $T/
(cd $T/burst_amd/csrc && rm -f *.o host/*.o && make -s all EXTRA_HIPFLAGS=-DPFM_PROF=2) || exit 1
mkdir -p $R/burst_amd/prof && cp $T/burst_amd/libburst_hip.so $T/burst_amd/libburst_host.so $R/burst_amd/prof/
rm -rf $T
