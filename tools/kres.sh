#!/bin/bash
# Per-kernel register / LDS / occupancy summary of one HIP source (compile-only, no GPU needed).
# Usage: bash tools/kres.sh phant_amd/csrc/mpt_verify_v3.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I phant_amd/csrc -c "$src" -o /tmp/kres.o "$@" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|Occupancy|ScratchSize|LDS Size" \
  | sed -E 's/^.*remark: +//; s/ \[-Rpass.*//; s/Function Name: /\n/' | tr '\n' ' ' | sed 's/ _Z/\n_Z/g'; echo
