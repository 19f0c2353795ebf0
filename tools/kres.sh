#!/bin/bash
# developer tool: the compiler's resource summary of both region-stage builds (registers, scratch, spills, LDS) and per-function scratch / spills    tools/kres.sh [extra flags]
S=linesegmentdetector-slam_amd/csrc/k_region.hip
for v in "4 -DLSD_REGION_WAVES_PER_SIMD=3" "8"; do set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DLSD_REGION_NW=$1 $2 $KRES_FLAGS -S --cuda-device-only -o /tmp/kres_w$1.s $S -Rpass-analysis=kernel-resource-usage 2> /tmp/kres_w$1.log
  echo "== w$1"; grep -A12 "Function Name: .*k_region" /tmp/kres_w$1.log | grep -E "VGPRs:|Spill|ScratchSize|Occupancy|LDS Size|SGPRs:" | tr '\n' ' '; echo
  grep -E "Function Name|ScratchSize|VGPRs Spill" /tmp/kres_w$1.log | paste - - - | sed 's/remark: [^ ]* //g' | awk '{print $0}' | grep -v "k_region" | sed 's/.*Function Name: //' | cut -c1-160
done
