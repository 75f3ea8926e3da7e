#!/bin/bash
# prints VGPR / scratch / occupancy per kernel of rt_device.hip
SRC=$(cd $(dirname $0)/.. && pwd)/raytracer_amd/csrc/rt_device.hip; cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Rpass-analysis=kernel-resource-usage --cuda-device-only -o /tmp/x.o $SRC 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - | sed -E 's/Function Name: _ZN12_GLOBAL__N_1//' | cut -c1-150
