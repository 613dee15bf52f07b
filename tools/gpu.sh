#!/bin/bash
# build in-tree (hipcc cross-compiles here), then run a command on the GPU box:  tools/gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
