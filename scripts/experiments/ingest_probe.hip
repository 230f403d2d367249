// ingest_probe.hip -- how fast can a CU pull L2-resident data in?  (a) buffer_load_dwordx4 -> VGPRs, (b) buffer_load_dwordx4 ... lds (LDS DMA),
// (c) global loads + ds_write_b128.  Every workgroup streams over its own 128 KiB window of a 64 MiB buffer (L2 / Infinity-Cache resident after the
// first pass) `iters` times.   hipcc --offload-arch=gfx950 -O3 -o ingest_probe ingest_probe.hip && ./ingest_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
constexpr int WIN = 128 * 1024;

template <int MODE>
__global__ void __launch_bounds__(256) probe_k(const float* src, float* out, int iters, int nwin) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const char* base = reinterpret_cast<const char*>(src) + (size_t)(blockIdx.x % nwin) * WIN;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, WIN, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        // 128 KiB per pass = 32 wave-instructions of 1 KiB per wave
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
            const int off = (j * 4 + wave) * 1024 + lane * 16;
            if (MODE == 0) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                acc += v;
            } else if (MODE == 1) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vp)(lds + ((j & 15) * 4 + wave) * 1024), 16, off, 0, 0, 0);
            } else {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                *reinterpret_cast<f32x4*>(lds + ((j & 15) * 4 + wave) * 1024 + lane * 16) = v;
            }
        }
        if (MODE != 0) { __builtin_amdgcn_s_waitcnt(0x0F70); }
    }
    if (MODE != 0) { __syncthreads(); acc[0] = *reinterpret_cast<float*>(lds + tid * 4); }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[tid] = acc[0];
}

template <int MODE>
static void run(const char* name, const float* src, float* out, int wgs, int threads_note) {
    const int iters = 200, nwin = 64 * 1024 * 1024 / WIN;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe_k<MODE>, dim3(wgs), dim3(256), 0, 0, src, out, 4, nwin);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe_k<MODE>, dim3(wgs), dim3(256), 0, 0, src, out, iters, nwin);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)wgs * iters * WIN;
    printf("%-28s %5d workgroups: %8.1f GB/s chip-wide, %6.1f GB/s per CU (256 CUs), %5.1f B/clk/CU at 2.1 GHz\n", name, wgs, bytes / ms * 1e-6, bytes / ms * 1e-6 / 256,
           bytes / ms * 1e-6 / 256 / 2.1);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 64 << 20); hipMalloc(&out, 4096);
    hipMemset(src, 0, 64 << 20);
    for (int wgs : {256, 512, 1024, 2048}) {
        run<0>("buffer_load -> VGPR", src, out, wgs, 0);
        run<1>("buffer_load ... lds (DMA)", src, out, wgs, 0);
        run<2>("buffer_load + ds_write_b128", src, out, wgs, 0);
    }
    return 0;
}
