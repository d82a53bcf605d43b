#!/bin/bash
# Print VGPR/SGPR/scratch/occupancy per kernel of one .hip file (cross-compiles for gfx950, no GPU needed).
# usage: tools/kernel_resources.sh multi_stylegan_amd/csrc/upfirdn2d.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: /\n/' | tr '\n' ' ' | sed 's/ _Z/\n_Z/g' | c++filt | cut -c1-220
echo
