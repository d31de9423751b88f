#!/bin/bash
# Rebuild libphant_gpu.so (a stale in-tree .so is what travels to the GPU box), then gpurun the given command.
# Usage: bash tools/g.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.." && python -c "from phant_amd import build as B; B.build(force=True)" && exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
