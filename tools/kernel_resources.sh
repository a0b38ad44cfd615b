#!/bin/bash
# kernel resource usage (VGPRs, AGPRs, scratch, LDS, occupancy) of one csrc/*.hip, from the compiler's remarks
# usage: tools/kernel_resources.sh dm_p2p [extra hipcc flags]
f=$1; shift
cd "$(dirname "$0")/../densematcher_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -I . -c $f.hip -o /tmp/kr_$f.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
 awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /LDS Size/ {printf "%-110s vgpr %3s agpr %3s scratch %4s occ %s lds %s\n", substr(name,1,110), v, a, s, o, $6}'
