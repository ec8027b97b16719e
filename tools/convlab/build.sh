#!/bin/bash
# builds the standalone convolution laboratory (cross-compiles for gfx950; the binary travels to the GPU box)
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/convlab/convlab.hip \
    -Lvolume-segmantics_amd/lib -lvolseg_hip -Wl,-rpath,'$ORIGIN/../../../volume-segmantics_amd/lib' -o tools/convlab/bin/convlab "$@"
