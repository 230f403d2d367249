// Probe: time gemm_k on the bench shapes with ablations (scripts only; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/gemm_probe.hip -o gpurun_out/gemm_probe && ./gemm_probe
#include "../hudiff_amd/csrc/hd_kernels.hip.h"
#include <cstdio>
#include <vector>
using namespace hd;

template <int BM, int BN, int WM, int WN, bool CONV, int PRO, int ABL, int NBUF = 1, int BK = 32>
static float run(GemmP q, int iters) {
    q.tiles0 = (q.sg.B * q.sg.len[0] + BM - 1) / BM;
    q.tiles_m = q.tiles0;
    q.tiles_n = (q.N + BN - 1) / BN;
    dim3 grid(((q.tiles_m + 7) / 8) * 8 * q.tiles_n), blk(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_k<BM, BN, WM, WN, CONV, PRO, ABL, NBUF, BK>), grid, blk, 0, 0, q);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_k<BM, BN, WM, WN, CONV, PRO, ABL, NBUF, BK>), grid, blk, 0, 0, q);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int M = 74496;
    const bool pc = argc > 1;      // any argument: only the production QKV kernel, for PC sampling / thread trace
    struct Shape { const char* name; int K, N, taps; } shapes[] = {
        {"qkv   K=768  N=1536", 768, 1536, 1}, {"oproj K=512  N=768", 512, 768, 1}, {"ff2   K=256  N=768", 256, 768, 1},
        {"g1    K=768  N=384", 768, 384, 1}, {"conv  K=7x384 N=384", 384, 384, 7}};
    float *A, *W, *C, *g; float2* st;
    hipMalloc(&A, (size_t)M * 1536 * 4); hipMalloc(&W, (size_t)2688 * 1536 * 4); hipMalloc(&C, (size_t)M * 1536 * 4);
    hipMalloc(&g, 4096 * 4); hipMalloc(&st, (size_t)M * 8);
    std::vector<float> h((size_t)M * 768);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)2688 * 1536 * 4, hipMemcpyHostToDevice);
    hipMemset(g, 0, 4096 * 4); hipMemset(st, 0, (size_t)M * 8);
    for (auto& s : shapes) {
        GemmP p{};
        if (pc && argv[1][0] == 'p' && (s.K != 768 || s.N != 1536)) continue;
        p.A = A; p.lda = s.K; p.W = W; p.bias = g; p.C = C; p.ldc = s.N; p.N = s.N; p.Kc = s.K; p.taps = s.taps; p.dil = 4;
        p.stats = st; p.gamma = g; p.beta = g; p.ldw = s.N;
        p.a_bytes = (uint32_t)((size_t)M * s.K * 4); p.w_bytes = (uint32_t)((size_t)s.taps * s.K * s.N * 4);
        p.sg.nseg = 1; p.sg.B = M / 291; p.sg.L = 291; p.sg.len[0] = 291;
        const double gf = 2.0 * M * s.K * s.taps * s.N * 1e-9;
        if (pc) {       // "pc": QKV only (profiling); anything else: the production BK = 16 kernels on every shape
            float t0 = s.taps == 1 ? run<128, 128, 2, 2, false, 0, 0, 1, 16>(p, 30) : run<128, 128, 2, 2, true, 0, 0, 1, 16>(p, 30);
            float t1 = s.taps == 1 ? run<128, 128, 2, 2, false, 1, 0, 1, 16>(p, 30) : 0.f;
            printf("%-22s BK=16: %8.1f us %6.1f TF | LN prologue %6.1f TF\n", s.name, t0 * 1e3, gf / t0, t1 > 0 ? gf / t1 : 0.0);
            continue;
        }
        float t[4];
        if (s.taps == 1) {
            t[0] = run<128, 128, 2, 2, false, 0, 0>(p, 5); t[1] = run<128, 128, 2, 2, false, 0, 1>(p, 5);
            t[2] = run<128, 128, 2, 2, false, 0, 2>(p, 5); t[3] = run<128, 128, 2, 2, false, 0, 3>(p, 5);
        } else {
            t[0] = run<128, 128, 2, 2, true, 2, 0>(p, 5); t[1] = run<128, 128, 2, 2, true, 2, 1>(p, 5);
            t[2] = run<128, 128, 2, 2, true, 2, 2>(p, 5); t[3] = run<128, 128, 2, 2, true, 2, 3>(p, 5);
        }
        if (s.taps == 1) {
            float u0 = run<128, 128, 2, 2, false, 0, 0, 2>(p, 5), u1 = run<128, 128, 2, 2, false, 0, 1, 2>(p, 5);
            float v0 = run<128, 128, 2, 2, false, 0, 0, 2, 16>(p, 5), v1 = run<128, 128, 2, 2, false, 0, 0, 1, 16>(p, 5);
            printf("   2-buffer 128x128: full %7.1f us %6.1f TF no-gload %6.1f TF | BK=16 2-buffer: full %7.1f us %6.1f TF | BK=16 1-buffer %6.1f TF\n",
                   u0 * 1e3, gf / u0, gf / u1, v0 * 1e3, gf / v0, gf / v1);
        } else {
            float u0 = run<128, 128, 2, 2, true, 2, 0, 2>(p, 5), v0 = run<128, 128, 2, 2, true, 2, 0, 2, 16>(p, 5);
            printf("   2-buffer 128x128: full %7.1f us %6.1f TF | BK=16 2-buffer %6.1f TF\n", u0 * 1e3, gf / u0, gf / v0);
        }
        printf("%-22s %7.1f GF | full %7.1f us %6.1f TF | no-gload %7.1f us %6.1f TF | +no-commit/barrier %7.1f us %6.1f TF | mfma-only %7.1f us %6.1f TF\n",
               s.name, gf, t[0] * 1e3, gf / t[0], t[1] * 1e3, gf / t[1], t[2] * 1e3, gf / t[2], t[3] * 1e3, gf / t[3]);
    }
    return 0;
}
