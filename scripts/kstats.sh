#!/bin/bash
# print per-kernel register / scratch / spill metadata of the built library (dev tool)
LIB=$(readlink -f ${1:-milagro_bls_amd/libmbls_hip.so})
T=$(mktemp -d); cd $T; cp $LIB lib.so
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null
for f in lib.so.*gfx950; do echo "== $f"; /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f | grep -E "^    \.name:|private_segment_fixed|\.vgpr_count|vgpr_spill|agpr_count" | paste - - - - - | sed 's/ \+/ /g'; done
rm -rf $T
