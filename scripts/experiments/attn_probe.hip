// Probe: time attn_k<19> (B = 256 antibody rows, 8 heads) with ablations (scripts only; not part of the product).
#include "../hudiff_amd/csrc/hd_kernels.hip.h"
#include <cstdio>
#include <vector>
using namespace hd;
template <int ABL>
static float run(const float* QKV, const float* rc, const float* rs, float* O, Segs sg, int iters, const RunState* st) {
    const int A = 512, nhead = 8;
    const size_t smem = (size_t)sg.L * (ATT_KS + ATT_VS) * sizeof(float);
    hipFuncSetAttribute((const void*)attn_k<19, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(sg.B * nhead);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((attn_k<19, ABL>), grid, dim3(ATT_THREADS), smem, 0, QKV, 3 * A, A, rc, rs, O, A, nhead, sg, 0, st);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((attn_k<19, ABL>), grid, dim3(ATT_THREADS), smem, 0, QKV, 3 * A, A, rc, rs, O, A, nhead, sg, 0, st);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
}
int main() {
    const int B = 256, L = 291, A = 512;
    Segs sg{}; sg.nseg = 2; sg.B = B; sg.L = L; sg.len[0] = 152; sg.len[1] = 139; sg.off[0] = 0; sg.off[1] = 152;
    sg.base[0] = 0; sg.base[1] = B * 152;
    const size_t rows = (size_t)B * L;
    float *QKV, *O, *rc, *rs;
    RunState* st; hipMalloc(&st, sizeof(RunState)); hipMemset(st, 0, sizeof(RunState));
    hipMalloc(&QKV, rows * 3 * A * 4); hipMalloc(&O, rows * A * 4); hipMalloc(&rc, L * 32 * 4); hipMalloc(&rs, L * 32 * 4);
    std::vector<float> h(rows * 3 * A);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.f - 0.5f;
    hipMemcpy(QKV, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(rc, h.data(), L * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(rs, h.data() + 7, L * 32 * 4, hipMemcpyHostToDevice);
    const double gf = 4.0 * L * L * A * B * 1e-9;     // QK^T + PV
    float t0 = run<0>(QKV, rc, rs, O, sg, 20, st), t1 = run<1>(QKV, rc, rs, O, sg, 20, st), t2 = run<2>(QKV, rc, rs, O, sg, 20, st),
          t3 = run<3>(QKV, rc, rs, O, sg, 20, st), t4 = run<4>(QKV, rc, rs, O, sg, 20, st);
    printf("attn_k<19> B=256: full %.1f us (%.1f TF) | no softmax %.1f | no S mfma %.1f | no PV mfma %.1f | no staging %.1f\n",
           t0, gf / t0 * 1e3, t1, t2, t3, t4);
    return 0;
}
