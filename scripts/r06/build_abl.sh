#!/bin/bash
# probe libraries of the chain kernel: scripts/r06/build_abl.sh "1 10 4 112 126"   -> hudiff_amd/libhudiff_abl<N>.so (git-ignored, travel with gpurun)
cd "$(dirname "$0")/../.."
for a in $1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -Wno-unused-value -mllvm -pragma-unroll-threshold=200000 -DHD_CHAIN_ABL=$a hudiff_amd/csrc/hd_api.hip -o hudiff_amd/libhudiff_abl$a.so &
done
wait
ls -la hudiff_amd/libhudiff_abl*.so
