// Probe: can fp32 MFMA and fp32 packed VALU FMA run concurrently on gfx950 (separate pipes)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mfma_work(int iters, float* out) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}
__device__ __forceinline__ void valu_work(int iters, float* out) {
    f32x2 c[16];
    for (int i = 0; i < 16; ++i) { c[i][0] = i; c[i][1] = -i; }
    f32x2 a = {1.0001f, 0.9999f}, b = {1e-6f, -1e-6f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_elementwise_fma(c[i], a, b);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1];
    if (s == 12345.678f) out[1] = s;
}
// mode 0: all waves MFMA, 1: all waves VALU, 2: waves 0-3 MFMA, 4-7 VALU
__global__ void __launch_bounds__(512) k(int mode, int mi, int vi, float* out) {
    const int wave = threadIdx.x >> 6;
    if (mode == 0 || (mode == 2 && wave < 4)) mfma_work(mi, out);
    if (mode == 1 || (mode == 2 && wave >= 4)) valu_work(vi, out);
}
static float timeit(int mode, int mi, int vi, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256 * 2), dim3(512), 0, 0, mode, mi, vi, out);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(256 * 2), dim3(512), 0, 0, mode, mi, vi, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
    float* out; hipMalloc(&out, 64);
    const int mi = 4000, vi = 4000;
    // flops: mfma: per wave per iter 16 mfma * 2*32*32*2 = 65536 ; valu: per wave per iter 64 pk_fma * 64 lanes * 4 flop = 16384
    const double blocks = 512, mf = 65536.0 * mi, vf = 16384.0 * vi;
    float t0 = timeit(0, mi, vi, out), t1 = timeit(1, mi, vi, out), t2 = timeit(2, mi, vi, out);
    printf("MFMA only (8 waves/block): %.3f ms  %.1f TF\n", t0, blocks * 8 * mf / t0 / 1e9);
    printf("VALU only (8 waves/block): %.3f ms  %.1f TF\n", t1, blocks * 8 * vf / t1 / 1e9);
    printf("hybrid 4 MFMA + 4 VALU waves: %.3f ms  MFMA %.1f TF + VALU %.1f TF = %.1f TF\n", t2, blocks * 4 * mf / t2 / 1e9,
           blocks * 4 * vf / t2 / 1e9, blocks * 4 * (mf + vf) / t2 / 1e9);
    // balance the hybrid so both halves take about the same time
    int vi2 = (int)(vi * ((double)t0 / 2 / ((double)t1 / 2)) );
    float t3 = timeit(2, mi, vi2, out);
    printf("hybrid balanced (vi=%d): %.3f ms  total %.1f TF\n", vi2, t3, blocks * 4 * (mf + 16384.0 * vi2) / t3 / 1e9);
    return 0;
}
