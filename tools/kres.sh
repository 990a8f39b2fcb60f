#!/bin/bash
# compact per-kernel resource table: tools/kres.sh <file.hip> [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kres.o 2>&1 \
 | grep -E "remark: +(Function Name|VGPRs:|ScratchSize|Occupancy|VGPRs Spill|TotalSGPRs)" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n, v, sg, s, o, sp; n=$3; next} /^VGPRs:/{v="vgpr="$2} /^TotalSGPRs/{sg="sgpr="$2} /ScratchSize/{s="scratch="$3} /Occupancy/{o="occ="$3} /VGPRs Spill/{sp="spill="$3} END{print n, v, sg, s, o, sp}' \
 | while read n rest; do echo "$(echo $n | c++filt | sed -E 's/\(.*//; s/void star:://') $rest"; done
