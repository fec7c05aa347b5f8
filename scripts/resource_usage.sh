#!/bin/bash
# Compiler view of every kernel of libavp_hip.so (no GPU needed): registers, spills, scratch, occupancy, static LDS.
# Writes profiles/r06_kernel_resource_usage.txt (or the file given as $1), stamped with the hash of the kernel sources (bench.source_hash).
cd "$(dirname "$0")/.."
H=$(python -c "import bench; print(bench.source_hash())")
OUT=${1:-profiles/r06_kernel_resource_usage.txt}
echo "# hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage on automatedvaletparking_amd/csrc/avp_capi.hip; source_hash $H" > $OUT
(cd automatedvaletparking_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-result -DAVP_BUILD -Rpass-analysis=kernel-resource-usage -o /tmp/avp_resource_usage.so avp_capi.hip 2>&1) \
  | grep -E "Function Name|SGPRs:|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* //;s/ \[-Rpass.*//;s/Function Name: //' | paste - - - - - - - - | sed 's/ \+/ /g' \
  | while IFS= read -r line; do n=$(echo "$line" | cut -f1 | sed 's/Name: //'); d=$(echo "$n" | c++filt | sed 's/(.*//;s/^void //'); echo "$d | $(echo "$line" | cut -f2-)"; done >> $OUT
rm -f /tmp/avp_resource_usage.so
cat $OUT | cut -c1-220
