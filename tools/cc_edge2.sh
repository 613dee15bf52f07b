#!/bin/bash
# compile one kernel source with the library's flags and print the compiler's register / spill report:  tools/cc_edge2.sh [file.hip] [extra flags]
f=${1:-ba_edge2.hip}; shift
cd /root/repo/batrack_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -x hip -c "$f" -o /tmp/cc_edge2.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|ScratchSize|warning" | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//' -e 's/^.*remark: *//'
