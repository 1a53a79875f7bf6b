#!/bin/bash
# Per-kernel VGPR / SGPR / scratch / LDS / occupancy of one HIP source (compiler view, gfx950).
#   tools/kernel_resources.sh pytorch_toolbelt_amd/csrc/ptb_losses.hip [name-filter]
SRC=$1; FILTER=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c "$SRC" -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs:" \
 | sed -E 's/.*remark: [^ ]* *//; s/\[-Rpass-analysis=kernel-resource-usage\]//' \
 | paste - - - - - - | grep -E "$FILTER" | sed -E 's/Function Name: //' | c++filt | awk '{print}' | cut -c1-230
