// Probe (scripts only): which XCD does workgroup b of a 1-D grid run on?  Reads HW_REG_XCC_ID (s_getreg id 20, bits 3:0), the register
// the ln_sync epilogue checks (hd_kernels.hip.h), and prints the map blockIdx % 8 -> XCC ids seen.  Expected on MI355X: block b on XCD b % 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(6164) & 15;
    // keep the block alive for a moment so that many blocks are resident together
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
    const int n = 8192;
    int* d; hipMalloc(&d, n * sizeof(int));
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
    std::vector<int> h(n);
    hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
    int seen[8][16] = {};
    for (int b = 0; b < n; ++b) seen[b & 7][h[b] & 15]++;
    bool ok = true;
    for (int r = 0; r < 8; ++r) {
        printf("blockIdx %% 8 = %d ->", r);
        for (int x = 0; x < 16; ++x) if (seen[r][x]) { printf(" xcc %d x %d", x, seen[r][x]); if (x != r) ok = false; }
        printf("\n");
    }
    printf("block b runs on XCD b %% 8: %s\n", ok ? "yes" : "NO");
    return 0;
}
