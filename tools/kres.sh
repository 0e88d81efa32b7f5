#!/bin/bash
# kernel resource usage of one .hip file (gfx950): name, SGPRs, VGPRs, scratch, occupancy, LDS, code size
# usage: tools/kres.sh drprg_amd/csrc/read_verify.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Iinclude --offload-arch=gfx950 -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
  | grep -E "Function Name|TotalSGPRs|VGPRs:|ScratchSize|Occupancy|LDS Size" \
  | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - \
  | sed -E 's/Function Name: //' | while IFS=$'\t' read -r n rest; do echo "$(echo "$n" | c++filt | cut -c1-90) | $rest"; done
