// hd_chain.hip -- DEVELOPMENT AID, not part of the build (hudiff_amd/build.py compiles hd_api.hip, which includes hd_chain.hip.h):
// compiles the chain kernels alone, e.g. to read their register counts:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --save-temps -mllvm -pragma-unroll-threshold=200000 hd_chain.hip
#include "../../hudiff_amd/csrc/hd_chain.hip.h"
namespace hd { hipError_t chain_dev_prepare() { return bn_chain_prepare(); } void chain_dev_launch(ChainP p, int DH, int D, hipStream_t st) { launch_bn_chain(p, DH, D, st); } bool chain_dev_ok() { return bn_chain_supported(384, 768, ACT_RELU) && bn_chain_tile_rows(384) == 128; } }
