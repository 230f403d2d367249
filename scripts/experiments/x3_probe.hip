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

namespace hd {
// ------------------------------------------------------------------------------------------------
// gemm_x3h_k: the split-precision GEMM with 16-deep k tiles ("h" = half k tile), round 4.
// MEASURED AND NOT ADOPTED (round 4): kept here, in the probe, as the record of the experiment.  Bit-identical
// to gemm_x3_k on the Q|K|V shape, but 432 vs 404 us per isolated launch: the loop's operand DMA is bound by the L2 -> LDS fabric
// (~16-19 TB/s chip-wide: 256 x 256 tiles move 2.8 GB per launch in 168 us, 256 x 128 tiles 4.2 GB in 271 us), so the 1.5 x bytes of
// the smaller tile cost the loop more (350 -> 378 us) than the overlapped epilogue returns (87 us alone, 53 us exposed).
// gemm_x3_k's widest launch (Q|K|V, 256 x 256 tiles) owns a CU alone: nothing computes while its epilogue drains 458 MB of
// stores (VERDICT r3 "Next" #3 i).  A k tile of 16 makes a stage of a 256 x 128 tile 24 KB (A 256 rows x 64 B, W 2 planes x 128
// rows x 32 B), so that two or three stages fit TWICE per CU beside the epilogue's staging: one block's epilogue runs under the
// other's K loop, at 0.75 of the operand bytes per flop of 128 x 128 tiles.  Same operand formats as gemm_x3_k: A in X16 rows
// (per 32 columns 32 high parts then 32 low parts: a 16-column k tile takes 32 B of each), W from the [128][32] tile images
// (half of every 64-byte row).  LDS images: A rows of 64 B = [hi k 0-7 | hi k 8-15 | lo k 0-7 | lo k 8-15], chunks swizzled by
// (row >> 2) & 3; W planes of 32-byte rows, chunks swizzled by (row >> 3) & 1 -- conflict-free ds_read_b128 fragments.
// The epilogue is gemm_x3_k's.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool CONV, int NS = 2>
__global__ void __launch_bounds__(64 * WM * WN, 2) gemm_x3h_k(const GemmP p) {
    constexpr int BK = 16, NW = WM * WN, NT = 64 * NW;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int ES = WTN + 4;
    constexpr int EPI_FLOATS = NW * 32 * ES, PART_FLOATS = NW * WTM * 2;
    constexpr int A_BYTES = BM * 64, W_PLANE = BN * 32, W_BYTES = 2 * W_PLANE;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int LOOP_FLOATS = NS * STAGE_BYTES / 4;
    constexpr int WORK_FLOATS = LOOP_FLOATS > EPI_FLOATS + PART_FLOATS ? LOOP_FLOATS : EPI_FLOATS + PART_FLOATS;
    constexpr int SM_FLOATS = WORK_FLOATS + 2 * BM;
    constexpr int A_PIECES = A_BYTES / 1024 / NW, W_PIECES = W_BYTES / 1024 / NW;   // 1 KiB DMA pieces per wave and tile
    static_assert(A_BYTES / 1024 % NW == 0 && W_BYTES / 1024 % NW == 0 && BN % X3_BN == 0 && W_PLANE % 1024 == 0, "tile / wave split");
    static_assert(lds_fill_ok(SM_FLOATS * 4, NT), "co-resident blocks of this kernel would fill the CU's LDS (see LDS co-residency rule)");
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
    char* St = reinterpret_cast<char*>(smem);          // stage s at St + s * STAGE_BYTES: A rows, W hi plane, W lo plane

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int bx, by, seg = 0;
    {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;        // all N tiles of an M tile on one XCD (see gemm_k)
        by = slot % p.tiles_n;
        bx = (slot / p.tiles_n) * 8 + xcd;
        if (bx >= p.tiles_m) return;
    }
    if (p.sg.nseg > 1 && bx >= p.tiles0) { seg = 1; bx -= p.tiles0; }
    const int abl_mode = p.x3_abl & 7;                 // probes only (scripts/x3_probe.hip); bit 3 = no epilogue, bit 5 = no stores
    const int Lc = p.sg.len[seg];
    const int seg_rows = p.sg.B * Lc;
    const int rbase = p.sg.base[seg];
    const int m0 = bx * BM, n0 = by * BN;
    const int Kc = p.Kc;
    const int nkt_tap = Kc / BK;                       // Kc % 32 == 0: an even number of half tiles per tap
    const int nkt = nkt_tap * p.taps;
    const int nkt32 = nkt / 2;                         // 32-deep tiles of the weight image
    const int half = (p.taps - 1) / 2;
    const uint16_t* __restrict__ Wx = p.Wx + (long)seg * p.wx_stride + (long)by * (BN / X3_BN) * nkt32 * X3_TILE_HALFS;

    if (p.ln_fold) {                                   // folded LayerNorm: the epilogue needs rstd of every row of the tile
        float2* rowst = reinterpret_cast<float2*>(smem + WORK_FLOATS);
        for (int r = tid; r < BM; r += NT) {
            const int lrow = m0 + r;
            rowst[r] = gemm_row_stat(p, lrow < seg_rows ? (long)rbase + lrow : (long)rbase);
        }
    }
    // A pieces of 1 KiB = 16 rows x 64 B: lane l lands at row l >> 2, slot l & 3 and fetches logical chunk c = slot ^ swizzle(row):
    // c < 2 the high parts k 8c .. 8c+7 of this half tile, c >= 2 the low parts -- 32 B of each half of the row's 128-byte group.
    // W pieces of 1 KiB = 32 rows x 32 B of one plane: lane l lands at row l >> 1, slot l & 1 and fetches the image's chunk
    // 2 (kt & 1) + (slot ^ swizzle(row)), which the image keeps at position chunk ^ ((row >> 2) & 3): the lane offset of an odd
    // half tile is that of an even one with bit 5 flipped.
    constexpr uint32_t BUF_OOB = 0x80000000u;
    int a_pos[A_PIECES];
    long a_row[A_PIECES];
    bool a_ok[A_PIECES];
    uint32_t a_in[A_PIECES], a_vo[A_PIECES], w_vo[W_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const int piece = NW * i + wave;
        const int r = 16 * piece + (lane >> 2);
        const int lrow = m0 + r;
        a_ok[i] = lrow < seg_rows;
        a_pos[i] = CONV ? (lrow % Lc) : 0;
        a_row[i] = a_ok[i] ? (long)rbase + lrow : 0;
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        a_in[i] = (uint32_t)((c & 1) * 16 + (c >> 1) * 64);
        a_vo[i] = a_ok[i] ? (uint32_t)(a_row[i] * p.lda * 4) + a_in[i] : BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
        const int piece = NW * i + wave;                                 // LDS image: hi plane of all BN rows, then lo plane
        const int plane = piece / (W_PLANE / 1024), rg = piece % (W_PLANE / 1024);      // row group of 32 rows
        const int r = 32 * rg + (lane >> 1);                             // row of the block's BN columns
        const int t128 = r / X3_BN, r128 = r % X3_BN;
        const int c = (lane & 1) ^ ((r >> 3) & 1);                       // logical chunk of the half tile
        w_vo[i] = (uint32_t)(t128 * nkt32 * X3_TILE_BYTES + plane * (X3_TILE_BYTES / 2) + r128 * 64 + ((c ^ ((r128 >> 2) & 3)) << 4));
    }
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Wx), 0, (BN / X3_BN) * nkt32 * X3_TILE_BYTES, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_vp;
    auto dma = [&](int kt, int st) {
        const int tap = CONV ? kt / nkt_tap : 0;
        const int kh = CONV ? kt - tap * nkt_tap : kt;                   // half tile within the tap
        if (CONV && kh == 0) {                         // first k tile of a tap: row shift + zero padding at the chain ends
            const int shift = (tap - half) * p.dil;
#pragma unroll
            for (int i = 0; i < A_PIECES; ++i) {
                const int sp = a_pos[i] + shift;
                const bool v = a_ok[i] && sp >= 0 && sp < Lc;
                a_vo[i] = v ? (uint32_t)((a_row[i] + shift) * p.lda * 4) + a_in[i] : BUF_OOB;
            }
        }
        char* dst = St + st * STAGE_BYTES;
        const int a_so = (kh >> 1) * 128 + (kh & 1) * 32;
        const uint32_t flip = (uint32_t)(kt & 1) << 5;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_vp)(dst + (NW * i + wave) * 1024), 16, (int)a_vo[i], a_so, 0, 0);
#pragma unroll
        for (int i = 0; i < W_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_vp)(dst + A_BYTES + (NW * i + wave) * 1024), 16, (int)(w_vo[i] ^ flip),
                                                     (kt >> 1) * X3_TILE_BYTES, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments of row r = (wave part) + 32 t + (lane & 31), k octet g = lane >> 5: A high parts = chunk g, low parts = chunk 2 + g
    // at position chunk ^ ((r >> 2) & 3) of the 64-byte row; W chunk g at position g ^ ((r >> 3) & 1) of the plane's 32-byte row
    const int fg = lane >> 5;
    const int aoff = (lane & 31) * 64 + ((fg ^ ((lane >> 2) & 3)) << 4);
    const int woff = (lane & 31) * 32 + ((fg ^ ((lane >> 3) & 1)) << 4);
    auto mma = [&](int st) {
        const char* At = St + st * STAGE_BYTES + wm * WTM * 64;
        const char* Wt = St + st * STAGE_BYTES + A_BYTES + wn * WTN * 32;
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(At + aoff + 32 * 64 * i);
            al[i] = *reinterpret_cast<const f16x8*>(At + (aoff ^ 32) + 32 * 64 * i);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[j] = *reinterpret_cast<const f16x8*>(Wt + woff + 32 * 32 * j);
            bl[j] = *reinterpret_cast<const f16x8*>(Wt + W_PLANE + woff + 32 * 32 * j);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    };
    constexpr int PER_TILE = A_PIECES + W_PIECES, KEEP = (NS - 2) * PER_TILE;
    static_assert(KEEP < 64, "vmcnt is a 6-bit counter");
    constexpr int WAIT_STEADY = (KEEP & 0xF) | ((KEEP >> 4) << 14) | 0x0F70, WAIT_ALL = 0x0F70;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nkt) dma(t, t);
    if (NS - 1 <= nkt) __builtin_amdgcn_s_waitcnt(WAIT_STEADY); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);   // tile 0 has landed
    lds_barrier();
    int st = 0, st_in = NS - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + NS - 1 < nkt;
        if (more && abl_mode != 2 && abl_mode != 3) dma(kt + NS - 1, st_in);
        if (abl_mode != 1 && abl_mode != 3) mma(st);
        if (more) __builtin_amdgcn_s_waitcnt(WAIT_STEADY); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
        lds_barrier();
        st_in = st;
        st = st + 1 == NS ? 0 : st + 1;
    }
    if (p.x3_abl & 8) {                                // probe: no epilogue (the accumulators stay live through a never-true store)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 1.2345678f) p.C[tid] = s;
        return;
    }
    const float2* rowst = reinterpret_cast<const float2*>(smem + WORK_FLOATS);
    const int need = epi_needs(p);
#define HD_EPI(F) gemm_epilogue<BM, BN, WM, WN, (F) | EPI_X3>(p, acc, smem, rowst, seg, seg_rows, rbase, Lc, m0, n0, by)
    if (!(need & ~EPI_PART)) HD_EPI(EPI_PART);
    else if (!(need & ~EPI_FOLD)) HD_EPI(EPI_FOLD);
    else if (!(need & ~(EPI_FOLD | EPI_ACT | EPI_CSPLIT))) HD_EPI(EPI_FOLD | EPI_ACT | EPI_CSPLIT);
    else if (!(need & ~(EPI_RESID | EPI_PART | EPI_C2))) HD_EPI(EPI_RESID | EPI_PART | EPI_C2);
    else HD_EPI(EPI_ALL);
#undef HD_EPI
}

}  // namespace hd

template <int BM, int BN, int WM, int WN, int NS>
static float runh(GemmP q, int abl, double seconds, const char* tag) {
    q.x3_abl = abl;
    const int rows = q.sg.B * q.sg.len[0];
    q.tiles0 = (rows + BM - 1) / BM; q.tiles_m = q.tiles0; q.tiles_n = q.N / BN;
    dim3 grid(((q.tiles_m + 7) / 8) * 8 * q.tiles_n), blk(64 * WM * WN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_x3h_k<BM, BN, WM, WN, false, NS>), grid, blk, 0, 0, q);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int iters = 0; float ms_total = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((gemm_x3h_k<BM, BN, WM, WN, false, NS>), grid, blk, 0, 0, q);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms_total += ms; iters += 50;
    }
    printf("END   %s abl=0x%x  %.1f us per launch\n", tag, abl, 1e3 * ms_total / iters); fflush(stdout);
    return ms_total / iters;
}

// results of two kernels on the same operands must be bit-identical in their K-sequential fp32 accumulation?  No: the k grouping
// differs (16 vs 32 per MFMA pass order), so compare within a tolerance against each other
static void compare(const float* dC, size_t n, std::vector<float>& ref, const char* tag) {
    std::vector<float> got(n);
    hipMemcpy(got.data(), dC, n * 4, hipMemcpyDeviceToHost);
    if (ref.empty()) { ref = got; printf("CMP   %s reference taken (C[0] = %g, C[last] = %g)\n", tag, got[0], got[n - 1]); return; }
    double mx = 0, mxr = 0;
    for (size_t i = 0; i < n; ++i) { const double d = fabs((double)got[i] - ref[i]); if (d > mx) mx = d; if (fabs(ref[i]) > mxr) mxr = fabs(ref[i]); }
    printf("CMP   %s max |diff| = %g of max |value| %g\n", tag, mx, mxr);
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
    if (argc > 1 && std::string(argv[1]) == "half") {
        // gemm_x3h_k (16-deep k tiles, two blocks per CU) against the shipping kernels on the Q|K|V shape
        const double secs = 0.4;
        p.st_nt = 1;
        std::vector<float> ref;
        const size_t ncmp = (size_t)4096 * N;
        hipMemset(C, 0, (size_t)M * N * 4);
        run<256, 256, 2, 4, 2>(p, 0, 0.05, "256x256x32 (shipping)"); compare(C, ncmp, ref, "256x256x32");
        hipMemset(C, 0, (size_t)M * N * 4);
        runh<256, 128, 2, 2, 2>(p, 0, 0.05, "256x128x16 NS=2"); compare(C, ncmp, ref, "256x128x16 NS=2");
        hipMemset(C, 0, (size_t)M * N * 4);
        runh<256, 128, 2, 2, 3>(p, 0, 0.05, "256x128x16 NS=3"); compare(C, ncmp, ref, "256x128x16 NS=3");
        for (int abl : {0, 8, 9, 10, 3}) run<256, 256, 2, 4, 2>(p, abl, secs, "256x256x32 waves 2x4 (shipping)");
        for (int abl : {0, 8, 9, 10, 3}) runh<256, 128, 2, 2, 2>(p, abl, secs, "256x128x16 NS=2 4 waves");
        for (int abl : {0, 8, 9, 10, 3}) runh<256, 128, 2, 2, 3>(p, abl, secs, "256x128x16 NS=3 4 waves");
        for (int abl : {0, 8}) runh<128, 128, 2, 2, 3>(p, abl, secs, "128x128x16 NS=3 4 waves");
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "tiles") {
        // tile / wave-shape sweep on the Q|K|V shape: everything, no epilogue, epilogue only (round 4: where does the epilogue's time go)
        const double secs = 0.4;
        p.st_nt = 1;
        for (int abl : {0, 8, 3, 32 + 3}) run<256, 256, 2, 4, 2>(p, abl, secs, "256x256 waves 2x4 (shipping)");
        for (int abl : {0, 8, 3, 32 + 3}) run<256, 256, 4, 2, 2>(p, abl, secs, "256x256 waves 4x2 (wave tile 64 x 128)");
        for (int abl : {0, 8, 3, 32 + 3}) run<128, 128, 2, 2, 2>(p, abl, secs, "128x128 waves 2x2");
        for (int abl : {0, 8, 3, 32 + 3}) run<128, 128, 4, 1, 2>(p, abl, secs, "128x128 waves 4x1 (wave tile 32 x 128)");
        fill_test(C, (size_t)M * N * 4);
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
