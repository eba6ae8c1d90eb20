#!/bin/bash
# a second build of the two libraries with extra compiler flags into burst_amd/<name>/ (selected at run time with BURST_AMD_LIBDIR):
#   tools/build_variant.sh prof -DPFM_PROF=2        the prefilter's phase timers (BHIP_PROF=1 python bench.py ... prints the shares)
#   tools/build_variant.sh barriers -DCF_BARRIERS=1  k_prefilter_cf with workgroup barriers between its phases (A/B of the default)
# Only the prefilter translation units (the ones the build switches PFM_PROF, CQ_*, CF_* reach) are compiled again; the others' objects
# are taken from the tree (run `make` there first).
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
T=$(mktemp -d)
mkdir -p $T/burst_amd && cp -r $R/burst_amd/csrc $T/burst_amd/ && cp -r $R/include $T/
KO=""; case "$*" in *WB_*) KO=bhip_kernels.o ;; esac      # (switches of the sweep kernels: that translation unit as well)
(cd $T/burst_amd/csrc && rm -f bhip_prefilter.o bhip_prefilter_alt.o bhip_prefilter_legacy.o $KO && make -s -j3 all EXTRA_HIPFLAGS="$*") || exit 1
mkdir -p $R/burst_amd/$N && cp $T/burst_amd/libburst_hip.so $T/burst_amd/libburst_host.so $R/burst_amd/$N/
rm -rf $T
