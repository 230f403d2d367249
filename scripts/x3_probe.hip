// Probe (scripts only): gemm_x3_k alone on the Q|K|V shape with its ablation knobs (GemmP::x3_abl), 0.6 s per variant, plus a
// monitor wave that measures the shader clock (scripts/x3_probe.sh SMI=1 also polls rocm-smi).  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/x3_probe.hip -o scripts/x3_probe.bin
#include "../hudiff_amd/csrc/hd_kernels.hip.h"
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>
using namespace hd;

// One wave with raised priority running a dependent SALU chain: iterations per 100 MHz tick follow the shader clock
// (calibrated against the idle run), independent of what rocm-smi reports.
__global__ void clockmon(int* stop, unsigned long long* out) {
    __builtin_amdgcn_s_setprio(3);
    unsigned long long c0 = clock64(), w0 = wall_clock64(), it = 0;
    uint32_t x = 1;
    while (wall_clock64() - w0 < 50000000ull && it < 4000000ull) {     // 1.2 s of the 100 MHz counter
#pragma unroll 1
        for (int i = 0; i < 256; ++i)
            asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                         "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(x) : : "scc");
        ++it;
    }
    out[0] = clock64() - c0; out[1] = wall_clock64() - w0; out[2] = it; out[3] = x;
}
static int* g_stop; static unsigned long long* g_out; static hipStream_t g_ms;
static void mon_start() { hipLaunchKernelGGL(clockmon, dim3(1), dim3(64), 0, g_ms, g_stop, g_out); }
static void mon_stop(const char* tag) {
    hipStreamSynchronize(g_ms);
    unsigned long long o[4]; hipMemcpy(o, g_out, 32, hipMemcpyDeviceToHost);
    printf("MON   %s clock64/wall=%.3f  salu iters per us=%.2f\n", tag, (double)o[0] / o[1], o[2] / (o[1] * 0.01));
}

typedef float pf4 __attribute__((ext_vector_type(4)));
__global__ void fill_k(pf4* p, long n, int nt) {
    const pf4 v = {1.f, 2.f, 3.f, 4.f};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (nt) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
static void fill_test(float* C, size_t bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
        for (int blocks : {512, 2048, 8192, 65536}) {
            hipLaunchKernelGGL(fill_k, dim3(blocks), dim3(256), 0, 0, (pf4*)C, (long)(bytes / 16), nt);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fill_k, dim3(blocks), dim3(256), 0, 0, (pf4*)C, (long)(bytes / 16), nt);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("END   fill nt=%d blocks=%d: %.1f us = %.2f TB/s\n", nt, blocks, 1e3 * ms / 20, bytes / (ms / 20 * 1e-3) * 1e-12);
        }
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipMemsetAsync(C, 0, bytes, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("END   hipMemset: %.1f us = %.2f TB/s\n", 1e3 * ms / 20, bytes / (ms / 20 * 1e-3) * 1e-12);
}

template <int BM, int BN, int WM, int WN, int NS>
static float run(GemmP q, int abl, double seconds, const char* tag) {
    q.x3_abl = abl;
    const int rows = q.sg.B * q.sg.len[0];
    q.tiles0 = (rows + BM - 1) / BM; q.tiles_m = q.tiles0; q.tiles_n = q.N / BN;
    dim3 grid(((q.tiles_m + 7) / 8) * 8 * q.tiles_n), blk(64 * WM * WN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_x3_k<BM, BN, WM, WN, false, NS>), grid, blk, 0, 0, q);
    hipDeviceSynchronize();
    mon_start();
    auto t0 = std::chrono::steady_clock::now();
    int iters = 0; float ms_total = 0;
    if (0) printf("BEGIN %s abl=%d t=%.3f\n", tag, abl, std::chrono::duration<double>(t0.time_since_epoch()).count()); fflush(stdout);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((gemm_x3_k<BM, BN, WM, WN, false, NS>), grid, blk, 0, 0, q);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms_total += ms; iters += 50;
    }
    printf("END   %s abl=0x%x  %.1f us per launch\n", tag, abl, 1e3 * ms_total / iters); fflush(stdout);
    mon_stop(tag);
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
    return ms_total / iters;
}

int main(int argc, char** argv) {
    const int M = 74496, K = 768, N = 1536;
    float *A, *C; uint16_t* Wx;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    const int nkt = K / 32, nt = N / 128;
    hipMalloc(&Wx, (size_t)nt * nkt * X3_TILE_HALFS * 2);
    std::vector<uint16_t> h((size_t)M * K * 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3000 + (uint16_t)((i * 2654435761u >> 20) & 0x3FF);     // halfs in [0.125, 0.25)
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(Wx, h.data(), (size_t)nt * nkt * X3_TILE_HALFS * 2, hipMemcpyHostToDevice);
    GemmP p{};
    p.A = A; p.lda = K; p.Wx = Wx; p.wx_stride = 0; p.acc_scale = 1.f; p.C = C; p.ldc = N; p.N = N; p.Kc = K; p.taps = 1; p.dil = 1;
    p.a_bytes = (uint32_t)((size_t)M * K * 4);
    p.sg.nseg = 1; p.sg.B = M / 291; p.sg.L = 291; p.sg.len[0] = 291;
    hipMalloc(&g_stop, 4); hipMalloc(&g_out, 64); printf("alloc\n"); fflush(stdout); hipStreamCreateWithFlags(&g_ms, hipStreamNonBlocking);
    printf("start\n"); fflush(stdout); mon_start(); mon_stop("idle"); fflush(stdout);
    const double gf = 2.0 * M * K * N * 1e-9;
    if (argc > 1 && std::string(argv[1]) == "pmc") {
        // one launch per epilogue configuration, for `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA ...` (scripts/x3_epi_pmc.sh):
        // what each epilogue feature costs in vector-ALU instructions.  N = 768 (the out-projection / FF2 / PFF3 shape, K = 512).
        const int N2 = 768, K2 = 512;
        float *bias, *C2; float2 *part, *stats;
        hipMalloc(&bias, N2 * 4); hipMemset(bias, 0, N2 * 4);
        hipMalloc(&C2, (size_t)M * N2 * 4);
        hipMalloc(&part, (size_t)16 * M * 8); hipMalloc(&stats, (size_t)M * 8); hipMemset(stats, 0, (size_t)M * 8);
        GemmP q = p; q.N = N2; q.Kc = K2; q.lda = K2; q.ldc = N2; q.a_bytes = (uint32_t)((size_t)M * K2 * 4);
        q.tiles0 = M / 128; q.tiles_m = q.tiles0; q.tiles_n = N2 / 128;
        auto go = [&](const char* name, GemmP g) {
            printf("PMC %s\n", name);
            hipLaunchKernelGGL((gemm_x3_k<128, 128, 2, 2, false, 2>), dim3(((g.tiles_m + 7) / 8) * 8 * g.tiles_n), dim3(256), 0, 0, g);
            hipDeviceSynchronize();
        };
        GemmP g = q; go("0 plain (scale, fp32 store)", g);
        g = q; g.x3_abl = 8; go("1 no epilogue at all", g);
        g = q; g.bias = bias; go("2 + bias", g);
        g = q; g.bias = bias; g.resid = C; g.ldr = N2; go("3 + bias + residual", g);
        g = q; g.bias = bias; g.resid = C; g.ldr = N2; g.part = part; g.part_rows = M; go("4 + bias + residual + LayerNorm partials", g);
        g = q; g.bias = bias; g.resid = C; g.ldr = N2; g.part = part; g.part_rows = M; g.C2 = C2; go("5 + bias + residual + partials + split copy (out-projection / FF2)", g);
        g = q; g.bias = bias; g.c_split = 1; go("6 bias + split output only (FF1 without ReLU)", g);
        g = q; g.bias = bias; g.c_split = 1; g.epi_act = ACT_RELU; go("7 bias + ReLU + split output (FF1)", g);
        g = q; g.bias = bias; g.c_split = 1; g.ln_fold = 1; g.stats = stats; go("8 folded LayerNorm + bias + split output (Q|K|V)", g);
        g = q; g.bias = bias; g.part = part; g.part_rows = M; go("9 bias + partials (PFF1 / tap GEMM)", g);
        return 0;
    }
    const double secs = 0.6;
    // the table of DESIGN.md section 9: x3_abl 0 = everything, 8 = no epilogue, 8+1 = operand DMA only, 8+2 = MFMAs + fragment
    // reads only, 8+3 = barriers / set-up only, 3 = epilogue only, 32+3 = epilogue without its stores
    for (int abl : {0, 8, 9, 10, 11, 3, 32 + 3}) run<128, 128, 2, 2, 2>(p, abl, secs, "128x128x2");
    for (int abl : {0, 8, 9, 10, 11, 3, 32 + 3}) run<256, 256, 2, 4, 2>(p, abl, secs, "256x256x2");
    p.st_nt = 1;
    run<128, 128, 2, 2, 2>(p, 0, secs, "128x128x2 non-temporal stores");
    run<256, 256, 2, 4, 2>(p, 0, secs, "256x256x2 non-temporal stores");
    p.st_nt = 0;
    printf("GF %.1f\n", gf);
    return 0;
}
