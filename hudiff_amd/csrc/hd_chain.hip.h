// hd_chain.hip.h -- "row-owner" ByteNet chain kernel of the split-precision route (round 6, VERDICT r5 "Next" #1).
//
// Reference: ByteNetBlock (sequence_models, restated in oracle/ref_import.py:124-147) inside DualConv / NanoConv / ByteNetTime
// (model/encoder/model.py:118-180, 249-304):  x -> x + PFF3(act(LN3(conv7(act(LN2(PFF1(act(LN1(x))))))))), then dropout.
// Rounds 2-5 ran a block as three gemm_x3_k launches of 128 x 128 tiles whose N tiles met at an L2 counter to normalise their rows
// (ln_sync) and which passed h1, h2, S = act(LN(out)) and a second copy of every N = 768 output through HBM: 0.16-0.45 of the split
// MFMA peak, bound by the L2 -> LDS operand path and by epilogues nothing overlapped (NOTES.md D).
//
// Here a WAVE owns 32 whole rows (64 in the 128-wide token encoder) through everything that is row-local, and only the 7-tap convolution -- which needs neighbouring rows
// -- is a launch boundary:
//   phase A  h2^T[DH, rows] = Wc^T . shift_tap(h1)^T     the k = 7 dilated conv as 7 row-shifted tap GEMMs, computed TRANSPOSED: the weights
//            are the MFMA's A operand (from LDS, shared by the workgroup's four waves), the wave's own activation rows its B operand
//            (straight from global memory into registers, zero padding at the chain ends from the buffer descriptor's range check).  A lane
//            then holds 16 channels of ONE row per 32 x 32 tile and all DH channels of its row over the CT tiles: LayerNorm 3 is a
//            lane-local reduction plus one exchange with lane ^ 32 -- no partials, no meeting, no second pass.
//   phase B  out^T[32-channel chunk, rows] = W3^T . act(LN3(h2))^T + b3 + x, dropout, (+ extra): act(LN3(h2)) never leaves the registers --
//            split (hi, lo) in place, it IS the B operand (the k order of W3's image is permuted at hd_finalize to the order the
//            accumulators hold channels in).  Per chunk: fp32 rows out (one copy), LayerNorm statistics of the finished row accumulated
//            on the way (Chan et al.), residual rows staged by LDS DMA.
//   phase C  h1'^T[DH, rows] = W1'^T . act(LN1'(out))^T of the NEXT block: the wave reads back the rows it just wrote (past the L1),
//            normalises + splits them in its operand path ONCE (it computes all DH output channels of its rows: no per-N-tile
//            prologue), LayerNorm 2' lane-locally again, and writes act(LN2'(h1')) in X16 split form for the next launch's conv.
// A stack of n blocks is n + 1 launches (C | A B C | ... | A B) instead of 3 n + 1, h2 / S / the second copies never exist, and the
// weights cross L2 -> LDS once per 128 rows for ALL DH output channels (gemm_x3_k: once per 128 rows and 128 channels).  One workgroup (4 waves, one
// per SIMD, 440-470 registers) per CU.
// STATUS (round 6, NOTES.md E, profiles/r06): correct on every width, trace and test -- and NOT faster than the launches it replaces (a dual block
// 1 058-1 114 us against 920-960 us): with one wave per SIMD nobody covers that wave's DMA issue, fragment latency, vector-ALU phases and barriers
// (everything that is not an MFMA takes 773 us of the launch by itself), and 582 one-per-CU workgroups are three rounds for 2.27 rounds of work.
// Option bn_chain (HD_OPT_BN_CHAIN) therefore defaults to 0; the kernel stays as a tested alternative and as the record of the experiment.
// Arithmetic: the same three-term split products with fp32 accumulation, k tiles in the same order as gemm_x3_k (taps outer); phase A's
// accumulators are bit-identical to the tap GEMM's, LayerNorm statistics and phase B's in-step k order differ in the last ulp.
#pragma once
#include "hd_kernels.hip.h"

// Probe builds only (-DHD_CHAIN_ABL=bits, a separate .so through HUDIFF_LIB; scripts/r06/chain_abl.sh): what the kernel leaves out, at compile
// time (a run-time branch in the loops would change their register allocation): 1 MFMAs, 2 weight DMA inside the loops, 4 weight fragment reads
// (one fragment serves all), 8 activation / row / residual loads inside the loops, 16 LayerNorm finish arithmetic, 32 phase B's epilogue,
// 64 phase C's normalisation.  Results are wrong in such a build; only its times are read.
#ifndef HD_CHAIN_ABL
#define HD_CHAIN_ABL 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define HD_CHAIN_KEEP(...) asm volatile("" :: __VA_ARGS__)
#else
#define HD_CHAIN_KEEP(...) ((void)0)
#endif

namespace hd {

struct ChainP {
    Segs sg; int tiles0, tiles;          // workgroup tiles (128 RT rows) of segment 0 / of both segments
    const RunState* rs;
    int act;                             // activation behind every LayerNorm of the stack (ACT_RELU DualConv, ACT_GELU NanoConv / token encoder)
    int phases;                          // bit 0: A + B (finish the block whose h1 is H1), bit 1: C (open the next block), bit 2 (tests): A only, h2 out
    // phase A
    const float* H1; uint32_t h1_bytes; int taps, dil;
    const uint16_t* Wc; long wc_seg; float sc_c;
    const float* bc; const float* g3; const float* be3;          // [DH] per segment
    float* H2dbg;                        // phases bit 2: act(LN3(conv)) as X16 rows [rows, DH]
    // phase B
    const uint16_t* W3; long w3_seg; float sc_3; const float* b3; // W3: k-permuted image (X3Packer::add kperm); b3 [D] per segment
    const float* X; int ldx; uint32_t x_bytes;                    // residual rows (block input), fp32
    float* Y; int ldy;                   // block output rows, fp32 (may alias X: a wave touches its own rows only)
    float* YX;                           // optional X16 split copy of the output rows [rows, D] (last block: the attention's operand)
    float2* ST;                          // optional (mean, rstd) of the output rows
    int drop_mode; uint32_t drop_thresh; float drop_scale; uint32_t drop_site; const uint8_t* drop_mask;
    const float* extra; int lde;         // added after dropout (token encoder's last block)
    // phase C
    const float* Yin; int ldyin; uint32_t yin_bytes; const float2* STin;     // C-only launch: input rows and their (mean, rstd)
    const uint16_t* W1; long w1_seg; float sc_1;
    const float* b1; const float* g1; const float* be1;           // b1 [DH]; LN1 (gamma, beta) [D] per segment
    const float* g2; const float* be2;   // LN2 [DH]
    float* H1out;                        // act(LN2(PFF1(..))) as X16 rows [rows, DH]
};

constexpr int CH_THREADS = 256;

template <int CT, int DT, int RT>
struct ChainGeom {
    static constexpr int WROWS = 32 * RT, ROWS = 4 * WROWS;        // rows of a wave / of a workgroup
    static constexpr int DH = 32 * CT, D = 32 * DT, NTH = CT / 4;
    static constexpr int STAGE = CT * 4096;                      // one k tile of all DH weight rows (phases A, C) = one 32-channel chunk over all k (phase B)
    // LDS: three weight stages in phases A and C (two tiles in flight: a tile's DMA round trip under load is longer than the 72 MFMAs of a
    // k tile); phase B runs on two and keeps the residual tiles of its four waves (32 RT rows x 128 B each) in the third when they fit.
    // Behind them 2 D floats of vectors the loops read (b3 | gamma1, beta1); the vectors of a LayerNorm finish (3 DH floats) are loaded
    // into stage 0 once the loop in front of it has ended.
    static constexpr int RES_WAVE = WROWS * 128;
    static constexpr int RES_OFF = 4 * RES_WAVE <= STAGE ? 2 * STAGE : 3 * STAGE;
    static constexpr int VEC_OFF = RES_OFF + 4 * RES_WAVE > 3 * STAGE ? RES_OFF + 4 * RES_WAVE : 3 * STAGE;
    static constexpr int SMEM = VEC_OFF + 2 * D * 4;
    static_assert(CT % 4 == 0 && SMEM <= LDS_PER_CU && 3 * DH * 4 <= STAGE, "LDS");
};

__device__ __forceinline__ float lane_xor32(float v) { return __shfl_xor(v, 32); }

// FL (compile time): optional outputs / operands of phase B -- bit 0 the X16 copy YX, bit 1 the addend `extra`.
// INJ: dropout keep-masks are injected (parity tests) instead of generated.  ACT (compile time): the activation behind every LayerNorm of the
// stack.  PH (compile time): the phases of this instantiation -- 1 = A + B, 2 = C alone, 3 = A + B + C, 4 = A alone (tests: h2 out)
template <int CT, int DT, int RT, int PH, int ACT, bool INJ, int FL>
__global__ void __launch_bounds__(CH_THREADS, 1) bn_chain_k(const ChainP p) {
    typedef ChainGeom<CT, DT, RT> G;
    constexpr int DH = G::DH, D = G::D, NTH = G::NTH, STAGE = G::STAGE;
    extern __shared__ __attribute__((aligned(16))) char chs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    int bx = blockIdx.x, seg = 0;
    if (bx >= p.tiles) return;
    if (p.sg.nseg > 1 && bx >= p.tiles0) { seg = 1; bx -= p.tiles0; }
    const int Lc = p.sg.len[seg], seg_rows = p.sg.B * Lc, rbase = p.sg.base[seg];
    const int m0 = bx * G::ROWS + wave * G::WROWS;               // first row of this wave inside the segment
    float* vec = reinterpret_cast<float*>(chs + G::VEC_OFF);     // loop vectors
    float* fin = reinterpret_cast<float*>(chs);                  // finish vectors (bias, gamma, beta of a DH-wide LayerNorm): stage 0, between loops
    typedef __attribute__((address_space(3))) void* lds_vp;
    constexpr uint32_t BUF_OOB = 0x80000000u;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    using std::integral_constant;
    auto cp_vec = [&](float* dst, const float* src, int n, int stride) {
        for (int i = tid; i < n; i += CH_THREADS) dst[i] = src[seg * stride + i];
    };

    // rows of this lane: tile i -> segment row m0 + 32 i + l31
    int lrow[RT]; bool rok[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) { lrow[i] = m0 + 32 * i + l31; rok[i] = lrow[i] < seg_rows; }

    // fragment offsets inside a 32-row x 64-byte slab of a weight plane: chunk (2 s + kh) ^ ((row >> 2) & 3)
    const int fsw = (lane >> 2) & 3;
    const int woff0 = l31 * 64 + (((0 + kh) ^ fsw) << 4), woff1 = l31 * 64 + (((2 + kh) ^ fsw) << 4);

    f32x16 acc[CT][RT];
    f16x8 Hh[CT][RT][2], Hl[CT][RT][2];                // act(LN3(h2)) split, [channel tile][row tile][k step]: phase B's B operand
    float st_mean[RT], st_rstd[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) { st_mean[i] = 0.f; st_rstd[i] = 1.f; }      // statistics of the rows phase B finished (phase C's LayerNorm 1)

    // ---- weight tiles of the natural image (phases A, C): k tile q of all DH rows = NTH tiles of 16 KiB, copied linearly --------------
    constexpr int WPIECES = NTH * 4;                             // DMA instructions per wave and weight stage (all three phases: = CT)
    auto dma_w_tile = [&](const uint16_t* W, int nkt, int q, int st) {
        const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(W), 0, NTH * nkt * X3_TILE_BYTES, 0x00020000);
        char* dst = chs + st * STAGE;
#pragma unroll
        for (int j = 0; j < WPIECES; ++j) {
            const int piece = 4 * j + wave;                      // 0 .. NTH * 16 - 1
            const int nt = piece >> 4, within = piece & 15;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_vp)(dst + piece * 1024), 16, (int)((nt * nkt) * X3_TILE_BYTES + within * 1024 + lane * 16),
                                                     q * X3_TILE_BYTES, 0, 0);
        }
    };

    // MFMAs of one k step on the RT x CT tiles: W fragments from stage `Wt` (natural image; pairs of channel tiles, the next pair's fragments
    // requested before this pair's MFMAs), activations xh / xl (B operand).  Per pair: cross terms first, the leading term last (gemm_x3_k's order).
    auto mma_step = [&](const char* Wt, int wo, const f16x8 (&xh)[RT], const f16x8 (&xl)[RT]) {
        f16x8 wh[2][2], wl[2][2];                                // [buffer][tile of the pair]
        auto ldw = [&](int tp, int b) {
            if ((HD_CHAIN_ABL & 4) && tp > 0) { wh[b][0] = wh[b ^ 1][0]; wh[b][1] = wh[b ^ 1][1]; wl[b][0] = wl[b ^ 1][0]; wl[b][1] = wl[b ^ 1][1]; return; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * tp + u;
                const char* a = Wt + (t >> 2) * 16384 + (t & 3) * 2048 + wo;
                wh[b][u] = *reinterpret_cast<const f16x8*>(a);
                wl[b][u] = *reinterpret_cast<const f16x8*>(a + 8192);
            }
        };
        ldw(0, 0);
#pragma unroll
        for (int tp = 0; tp < CT / 2; ++tp) {
            const int b = tp & 1;
            if (tp + 1 < CT / 2) ldw(tp + 1, b ^ 1);
            if (HD_CHAIN_ABL & 1) { HD_CHAIN_KEEP("v"(wh[b][0]), "v"(wl[b][0]), "v"(wh[b][1]), "v"(wl[b][1]), "v"(xh[0]), "v"(xl[0])); continue; }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[2 * tp + u][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][u], xl[i], acc[2 * tp + u][i], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[2 * tp + u][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[b][u], xh[i], acc[2 * tp + u][i], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[2 * tp + u][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][u], xh[i], acc[2 * tp + u][i], 0, 0, 0);
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
    };
    auto vm_wait = [&](auto n_c) {                               // s_waitcnt vmcnt(n)
        constexpr int N = decltype(n_c)::value;
        static_assert(N < 64, "vmcnt is a 6-bit counter");
        __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0F70);
    };

    // Scheduling pattern of a region that holds NM MFMAs beside other work that does not depend on them (hipcc left alone issues all MFMAs,
    // then everything else: with ONE wave per SIMD nothing then covers the other instructions' issue time or the matrix pipe's idle time):
    // per MFMA at most ND LDS reads, NV vector-ALU instructions and NX vector-memory instructions, in that order.
    auto interleave = [&](auto nm_c, auto nd_c, auto nv_c, auto nx_c) {
        constexpr int NM = decltype(nm_c)::value, ND = decltype(nd_c)::value, NV = decltype(nv_c)::value, NX = decltype(nx_c)::value;
#pragma unroll
        for (int k = 0; k < NM; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (ND) __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
            if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            if (NX) __builtin_amdgcn_sched_group_barrier(0x010, NX, 0);
        }
    };
    constexpr int MM_STEP = 3 * CT * RT;                         // MFMAs of one k step (mma_step)

    // Finish of a DH-wide accumulator set (phases A and C): scale + bias, LayerNorm over the DH channels of each row (lane-local over the
    // CT tiles + one exchange with lane ^ 32), activation, (hi, lo) split into Hh / Hl.  (bias, gamma, beta) [DH] each come into stage 0
    // first: every wave has left the K loop (barrier), nothing is in flight into that stage.
    auto finish_ln_split = [&](float sc, const float* bsrc, const float* gsrc, const float* esrc) {
        vm_wait(integral_constant<int, 0>{});
        lds_barrier();
        cp_vec(fin, bsrc, DH, DH); cp_vec(fin + DH, gsrc, DH, DH); cp_vec(fin + 2 * DH, esrc, DH, DH);
        lds_barrier();
        float vmax = 0.f;
        if (HD_CHAIN_ABL & 16) {
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) { Hh[t][i][s2] = __builtin_bit_cast(f16x8, f32x4{acc[t][i][8 * s2], acc[t][i][8 * s2 + 1], acc[t][i][8 * s2 + 2], acc[t][i][8 * s2 + 3]}); Hl[t][i][s2] = __builtin_bit_cast(f16x8, f32x4{acc[t][i][8 * s2 + 4], acc[t][i][8 * s2 + 5], acc[t][i][8 * s2 + 6], acc[t][i][8 * s2 + 7]}); }
            lds_barrier();
            return;
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(fin + 32 * t + 8 * q + 4 * kh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[t][i][4 * q + e] * sc + b4[e];
                        acc[t][i][4 * q + e] = v;
                        s += v;
                    }
                    if (q == 3) __builtin_amdgcn_sched_barrier(0);       // (one tile's vector reads at a time: left alone the scheduler hoists all of them)
                }
            s += lane_xor32(s);
            const float mean = s * (1.0f / (float)DH);
            float m2 = 0.f;
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = acc[t][i][r] - mean; m2 += d * d; }
            m2 += lane_xor32(m2);
            const float rstd = 1.0f / sqrtf(m2 * (1.0f / (float)DH) + 1e-5f);
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                f32x4 w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(fin + DH + 32 * t + 8 * q + 4 * kh);
                    const f32x4 e4 = *reinterpret_cast<const f32x4*>(fin + 2 * DH + 32 * t + 8 * q + 4 * kh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[q][e] = act_f((acc[t][i][4 * q + e] - mean) * rstd * g4[e] + e4[e], ACT);
                    if (rok[i]) vmax = absmax4(vmax, w[q]);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {                 // k step s2 of channel tile t: accumulator registers 8 s2 .. 8 s2 + 7
                    f16x4 h0, l0, h1, l1;
                    split4(w[2 * s2], h0, l0);
                    split4(w[2 * s2 + 1], h1, l1);
                    Hh[t][i][s2] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                    Hl[t][i][s2] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        raise_range_flag(p.rs, vmax);
        lds_barrier();                                           // every wave has read the finish vectors: stage 0 may be filled again
    };
    // X16 rows out: register j of (tile t, k step s2) holds channel 32 t + 16 s2 + 8 (j >> 2) + 4 kh + (j & 3)
    auto store_x16 = [&](float* dstp) {
        const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(dstp, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const uint32_t rb = rok[i] ? (uint32_t)(rbase + lrow[i]) * (uint32_t)(DH * 4) : BUF_OOB;
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                    const f16x8 hv = Hh[t][i][s2], lv = Hl[t][i][s2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int c = 32 * t + 16 * s2 + 8 * q;                    // + 4 kh: four consecutive channels
                        const h16x4 h4 = q ? __builtin_shufflevector(hv, hv, 4, 5, 6, 7) : __builtin_shufflevector(hv, hv, 0, 1, 2, 3);
                        const h16x4 l4 = q ? __builtin_shufflevector(lv, lv, 4, 5, 6, 7) : __builtin_shufflevector(lv, lv, 0, 1, 2, 3);
                        const uint32_t o = rb == BUF_OOB ? BUF_OOB : rb + (uint32_t)(x16_hi(c) * 2) + (uint32_t)kh * 8u;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), d_rs, (int)o, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, l4), d_rs, (int)(o == BUF_OOB ? BUF_OOB : o + (uint32_t)X16_LO * 2u), 0, 0);
                    }
                }
        }
    };

    // =================================================================================================================================
    // phase A: the dilated convolution as taps x CT k tiles.  Three stages: at the top of tile q the wave waits for ITS pieces of tile q
    // (tile q + 1's stay in flight), meets the others, then sends the activations of tile q + 1 and the weights of tile q + 2 on their way
    // and multiplies tile q.  The loop body is ONE basic block (a run-time branch makes the register allocator copy accumulators around
    // every block boundary: 300 v_accvgpr moves per k tile in the first version); behind the last tile the prefetches re-fetch it.
    // =================================================================================================================================
    if constexpr ((PH & 5) != 0) {
        const uint16_t* Wc = p.Wc + (long)seg * p.wc_seg;
        const int nkt = p.taps * CT, half = (p.taps - 1) / 2;
        const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.H1), 0, (int)p.h1_bytes, 0x00020000);
        int pos[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) pos[i] = rok[i] ? lrow[i] % Lc : -(1 << 24);
        // activations of k tile q (two k steps: X16 groups 2 kt, 2 kt + 1 of tap q / CT): hi 16 B at + 16 kh, lo 32 B further; a row the tap
        // shifts out of its chain gets an offset the descriptor's range check rejects (zeros)
        struct XT { f16x8 h[2][RT], l[2][RT]; };
        auto ldx = [&](int q, XT& x) {
            const int tap = q / CT, kt = q - tap * CT;
            const int shift = (tap - half) * p.dil;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const uint32_t vo = (unsigned)(pos[i] + shift) < (unsigned)Lc ? (uint32_t)(rbase + lrow[i] + shift) * (uint32_t)(DH * 4) + (uint32_t)kh * 16u : BUF_OOB;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    x.h[s2][i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(h_rs, (int)vo, (2 * kt + s2) * 64, 0));
                    x.l[s2][i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(h_rs, (int)vo, (2 * kt + s2) * 64 + 32, 0));
                }
            }
        };
        zero_acc();
        XT xa, xb;
        dma_w_tile(Wc, nkt, 0, 0);
        ldx(0, xa);
        dma_w_tile(Wc, nkt, 1, 1);
        int st = 0;                                              // stage of tile q
        auto body = [&](int q, XT& xc, XT& xn) {
            vm_wait(integral_constant<int, WPIECES>{});          // tile q's weights and activations have landed (tile q + 1's weights may travel)
            lds_barrier();                                       // ... everybody's; everybody is done with tile q - 1 (its stage is free)
            const int st2 = st == 0 ? 2 : st - 1;                // (q + 2) % 3
            __builtin_amdgcn_sched_barrier(0);
            if (!(HD_CHAIN_ABL & 8)) ldx(min(q + 1, nkt - 1), xn);
            if (!(HD_CHAIN_ABL & 2)) dma_w_tile(Wc, nkt, min(q + 2, nkt - 1), st2);
            const char* Wt = chs + st * STAGE;
            mma_step(Wt, woff0, xc.h[0], xc.l[0]);
            mma_step(Wt, woff1, xc.h[1], xc.l[1]);
            // the loads and DMA pieces of the NEXT tiles are dealt out one per MFMA over the first ones: a piece's issue then costs the
            // matrix pipe a gap, not the whole burst in front of the tile
            interleave(integral_constant<int, 2 * MM_STEP>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{}, integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            st = st == 2 ? 0 : st + 1;
        };
        for (int q = 0; q < nkt; q += 2) { body(q, xa, xb); body(q + 1, xb, xa); }      // (taps x CT is even for every shipped width)
        finish_ln_split(p.sc_c, p.bc, p.g3, p.be3);
        if constexpr ((PH & 4) != 0) { store_x16(p.H2dbg); return; }
    }

    // =================================================================================================================================
    // phase B: out = dropout(x + PFF3(act(LN3(h2)))) (+ extra), 32 output channels at a time.  The epilogue of chunk c - 1 (vector ALU: bias,
    // residual, dropout hash, statistics) sits in the same scheduling region as the MFMAs of chunk c, which do not depend on it.
    // =================================================================================================================================
    if constexpr ((PH & 1) != 0) {
        const uint16_t* W3 = p.W3 + (long)seg * p.w3_seg;
        const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(W3), 0, (D / X3_BN) * CT * X3_TILE_BYTES, 0x00020000);
        const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, 0x7FFFFFFF, 0x00020000);
        // chunk c of the k-permuted image: rows (c & 3) * 32 .. of n tile c >> 2, every k tile: CT x (2 planes x 2 KiB) -> stage [kt][plane][32 rows x 64 B]
        auto dma_chunk = [&](int c, int stg) {
            char* dst = chs + stg * STAGE;
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int piece = 4 * j + wave;                  // 0 .. 4 CT - 1: k tile piece >> 2, plane (piece >> 1) & 1, half piece & 1
                const int ktile = piece >> 2, plane = (piece >> 1) & 1, hf = piece & 1;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_vp)(dst + piece * 1024), 16,
                                                         (int)(ktile * X3_TILE_BYTES + plane * 8192 + hf * 1024 + lane * 16),
                                                         ((c >> 2) * CT) * X3_TILE_BYTES + (c & 3) * 2048, 0, 0);
            }
        };
        // residual tile of chunk c: this wave's rows x 32 channels (128 B per row) as pieces of 8 rows; 16-byte chunks swizzled by (row >> 1) & 7
        char* res = chs + G::RES_OFF + wave * G::RES_WAVE;
        constexpr int RP = 4 * RT;                                   // residual pieces of 8 rows per wave
        uint32_t r_vo[RP];
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            const int r = 8 * j + (lane >> 3);
            const int lr = m0 + r;
            r_vo[j] = lr < seg_rows ? (uint32_t)(rbase + lr) * (uint32_t)(p.ldx * 4) + (uint32_t)((((lane & 7) ^ ((r >> 1) & 7))) << 4) : BUF_OOB;
        }
        auto dma_res = [&](int c) {
#pragma unroll
            for (int j = 0; j < RP; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_vp)(res + j * 1024), 16, (int)r_vo[j], c * 128, 0, 2);
        };
        // dropout keys (a launch without dropout passes threshold 0 and scale 1: every hash is kept -- the epilogue has no branch)
        uint32_t dk0, dk1, drow0;
        {
            uint32_t o[4];
            philox4x32_10(0u, 0u, p.rs->step, p.drop_site, p.rs->seed_lo, p.rs->seed_hi, o);
            dk0 = o[0]; dk1 = o[1]; drow0 = p.rs->row0;
        }
        const uint32_t drop_thresh = p.drop_mode == DROP_GEN ? p.drop_thresh : 0u;
        const float drop_scale = p.drop_mode != DROP_NONE ? p.drop_scale : 1.0f;
        const __amdgpu_buffer_rsrc_t e_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((FL & 2) ? p.extra : p.X), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t yx_rs = __builtin_amdgcn_make_buffer_rsrc((FL & 1) ? p.YX : p.Y, 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t m_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(INJ ? p.drop_mask : reinterpret_cast<const uint8_t*>(p.X)), 0, 0x7FFFFFFF, 0x00020000);
        uint32_t rk[RT]; int bq[RT], slot[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int lr = rok[i] ? lrow[i] : 0;
            bq[i] = lr / Lc; slot[i] = p.sg.off[seg] + (lr - bq[i] * Lc);
            rk[i] = mix32(dk0 ^ mix32(drow0 + (uint32_t)bq[i] + dk1));
        }
        // running LayerNorm statistics of the finished rows: per lane over its own channels, merged with lane ^ 32 at the end
        float rn = 0.f, rmean[RT], rm2[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) { rmean[i] = 0.f; rm2[i] = 0.f; }
        float vmaxy = 0.f;
        const int asw = (lane >> 1) & 7;
        typedef f32x16 A2[2][RT];                                // [k tile parity][row tile]: two independent chains per row tile, summed in the epilogue
        // MFMAs of chunk c: W3 fragments from the stage, act(LN3(h2)) from the registers
        auto mma_chunk = [&](const char* Wt, A2& a2) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { a2[0][i][r] = 0.f; a2[1][i][r] = 0.f; }
            f16x8 wh[2], wl[2];
            auto ldw = [&](int ks, int b) {                      // k step ks = 2 kt + s
                const char* a = Wt + (ks >> 1) * 4096 + ((ks & 1) ? woff1 : woff0);
                wh[b] = *reinterpret_cast<const f16x8*>(a);
                wl[b] = *reinterpret_cast<const f16x8*>(a + 2048);
            };
            ldw(0, 0);
#pragma unroll
            for (int ks = 0; ks < 2 * CT; ++ks) {
                const int b = ks & 1, t = ks >> 1, s2 = ks & 1;
                if (ks + 1 < 2 * CT && !((HD_CHAIN_ABL & 4) && ks > 0)) ldw(ks + 1, b ^ 1);
                if (HD_CHAIN_ABL & 1) { HD_CHAIN_KEEP("v"(wh[b]), "v"(wl[b]), "v"(Hl[t][0][s2]), "v"(Hh[t][0][s2])); continue; }
#pragma unroll
                for (int i = 0; i < RT; ++i) a2[t & 1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b], Hl[t][i][s2], a2[t & 1][i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT; ++i) a2[t & 1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[b], Hh[t][i][s2], a2[t & 1][i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT; ++i) a2[t & 1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b], Hh[t][i][s2], a2[t & 1][i], 0, 0, 0);
            }
        };
        // epilogue arithmetic of chunk c from its accumulators and the residual tile in LDS -> finished values v
        auto epi_math = [&](int c, const A2& a2, f32x4 (&v)[RT][4]) {
            rn += 16.f;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = 32 * c + 8 * q + 4 * kh;
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(vec + col);
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(res + (32 * i + l31) * 128 + (((2 * q + kh) ^ asw) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][q][e] = (a2[0][i][4 * q + e] + a2[1][i][4 * q + e]) * p.sc_3 + b4[e] + x4[e];
                    if constexpr (INJ) {
                        const uint32_t mk = __builtin_amdgcn_raw_buffer_load_b32(m_rs, (int)(rok[i] ? (uint32_t)((bq[i] * p.sg.L + slot[i]) * D + col) : BUF_OOB), 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[i][q][e] = ((mk >> (8 * e)) & 0xFFu) ? v[i][q][e] * drop_scale : 0.f;
                    } else {
                        const uint32_t h0 = rk[i] + (uint32_t)(slot[i] * D + col) * 0x9E3779B9U;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t w = mix32(h0 + (uint32_t)e * 0x9E3779B9U);
                            v[i][q][e] = (w >= drop_thresh) ? v[i][q][e] * drop_scale : 0.f;
                        }
                    }
                    if constexpr ((FL & 2) != 0)
                        v[i][q] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(e_rs, (int)(rok[i] ? (uint32_t)(rbase + lrow[i]) * (uint32_t)(p.lde * 4) + (uint32_t)col * 4u : BUF_OOB), 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) s += v[i][q][e];
                }
                // Chan et al.: merge this chunk's 16 values (mean16, M2_16) into the lane's running (n, mean, M2)
                const float m16 = s * (1.0f / 16.0f);
                float q16 = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[i][q][e] - m16; q16 += d * d; }
                const float delta = m16 - rmean[i];
                rmean[i] += delta * (16.0f / rn);
                rm2[i] += q16 + delta * delta * (16.0f * (rn - 16.0f) / rn);
            }
        };
        constexpr int NST = 4 * RT + ((FL & 1) ? 8 * RT : 0);    // store instructions of a chunk: fp32 rows + the X16 copy
        auto epi_store = [&](int c, const f32x4 (&v)[RT][4]) {
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const uint32_t yb = rok[i] ? (uint32_t)(rbase + lrow[i]) * (uint32_t)(p.ldy * 4) : BUF_OOB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = 32 * c + 8 * q + 4 * kh;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i][q]), y_rs, (int)(yb == BUF_OOB ? BUF_OOB : yb + (uint32_t)col * 4u), 0, 0);
                    if constexpr ((FL & 1) != 0) {
                        f16x4 h4, l4;
                        split4(v[i][q], h4, l4);
                        if (rok[i]) vmaxy = absmax4(vmaxy, v[i][q]);
                        const uint32_t o = rok[i] ? (uint32_t)(rbase + lrow[i]) * (uint32_t)(D * 4) + (uint32_t)(x16_hi(col) * 2) : BUF_OOB;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), yx_rs, (int)o, 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, l4), yx_rs, (int)(o == BUF_OOB ? BUF_OOB : o + (uint32_t)X16_LO * 2u), 0, 2);
                    }
                }
            }
        };
        // (phase A's finish ended with a barrier: the stages are free, nothing is in flight)
        cp_vec(vec, p.b3, D, D);
        dma_chunk(0, 0);
        dma_res(0);
        dma_chunk(1, 1);
        A2 aa, ab;
        // chunk 0's MFMAs alone; then per iteration: MFMAs of chunk c beside the epilogue of chunk c - 1; the last epilogue alone
        vm_wait(integral_constant<int, CT>{});
        lds_barrier();
        mma_chunk(chs, aa);
        auto body = [&](int c, A2& acur, const A2& aprev) {
            // chunk c's weights (and, older, the residual tile of chunk c - 1) have landed; in flight at most: the stores of chunk c - 2 ... no:
            // issue order of an iteration is  weights(c + 1) | stores(c - 1) | residual(c), so everything but the last NST + RP is awaited
            vm_wait(integral_constant<int, NST + RP>{});
            lds_barrier();                                       // everybody has chunk c; everybody is done with chunk c - 1's stage
            if (!(HD_CHAIN_ABL & 2)) dma_chunk(min(c + 1, DT - 1), (c + 1) & 1);
            f32x4 v[RT][4];
            vm_wait(integral_constant<int, CT>{});               // the residual tile of chunk c - 1 (older than these CT pieces) has landed
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk(chs + (c & 1) * STAGE, acur);
            if (HD_CHAIN_ABL & 32) { HD_CHAIN_KEEP("v"(aprev[0][0]), "v"(aprev[1][0])); } else {
            epi_math(c - 1, aprev, v);
            // (the epilogue's vector ALU work -- bias, residual, dropout hash, statistics: ~10 instructions per MFMA -- between the MFMAs)
            interleave(integral_constant<int, 2 * MM_STEP>{}, integral_constant<int, 1>{}, integral_constant<int, 10>{}, integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            epi_store(c - 1, v);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's reads of its residual tile are done ...
            if (!(HD_CHAIN_ABL & 8)) dma_res(c);                 // ... so the next one may land (private to the wave: no barrier)
        };
        static_assert(DT % 2 == 0, "chunks are walked in pairs");
        {   // c = 1 (issue order so far: weights(0) | residual(0) | weights(1): the uniform wait of body() would leave residual(0) un-awaited)
            vm_wait(integral_constant<int, 0>{});
            body(1, ab, aa);
        }
        for (int c = 2; c < DT; c += 2) { body(c, aa, ab); body(c + 1, ab, aa); }
        {
            f32x4 v[RT][4];
            vm_wait(integral_constant<int, 0>{});
            epi_math(DT - 1, ab, v);
            epi_store(DT - 1, v);
        }
        if constexpr ((FL & 1) != 0) raise_range_flag(p.rs, vmaxy);
        // row statistics: lane and lane ^ 32 each hold D / 2 channels of the row
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float om = lane_xor32(rmean[i]), oq = lane_xor32(rm2[i]);
            const float mean = 0.5f * (rmean[i] + om);
            const float d = rmean[i] - om;
            const float m2 = rm2[i] + oq + d * d * (0.25f * (float)D);
            st_mean[i] = mean;
            st_rstd[i] = 1.0f / sqrtf(m2 * (1.0f / (float)D) + 1e-5f);
            if (p.ST && rok[i] && kh == 0) p.ST[rbase + lrow[i]] = make_float2(st_mean[i], st_rstd[i]);
        }
    }

    // =================================================================================================================================
    // phase C: h1' = act(LN2'(PFF1'(act(LN1'(out))))).  Pipeline as phase A; the rows come as fp32 (past the L1: the wave's own stores of phase B)
    // and are normalised + split one k step ahead of the MFMAs that consume them, in the same scheduling region.
    // =================================================================================================================================
    if constexpr ((PH & 2) != 0) {
        const uint16_t* W1 = p.W1 + (long)seg * p.w1_seg;
        constexpr bool chained = (PH & 1) != 0;
        const float* Ysrc = chained ? p.Y : p.Yin;
        const int ldy = chained ? p.ldy : p.ldyin;
        const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ysrc), 0, chained ? 0x7FFFFFFF : (int)p.yin_bytes, 0x00020000);
        if (!chained) {
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const float2 stv = p.STin[rbase + (rok[i] ? lrow[i] : 0)];
                st_mean[i] = stv.x; st_rstd[i] = stv.y;
            }
        }
        uint32_t yvo[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) yvo[i] = rok[i] ? (uint32_t)(rbase + lrow[i]) * (uint32_t)(ldy * 4) + (uint32_t)kh * 32u : BUF_OOB;
        // raw rows of k tile q: per k step g = 2 q + s the channels 16 g + 8 kh .. + 7 of each of the lane's rows
        struct RW { f32x4 v[2][RT][2]; };
        auto ldy_raw = [&](int q, RW& r) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    r.v[s2][i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y_rs, (int)yvo[i], (2 * q + s2) * 64, 16));
                    r.v[s2][i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y_rs, (int)yvo[i], (2 * q + s2) * 64 + 16, 16));
                }
        };
        float vmaxc = 0.f;
        auto norm_split = [&](int g, const f32x4 (&raw)[RT][2], f16x8 (&xh)[RT], f16x8 (&xl)[RT]) {
            if (HD_CHAIN_ABL & 64) {
#pragma unroll
                for (int i = 0; i < RT; ++i) { xh[i] = __builtin_bit_cast(f16x8, raw[i][0]); xl[i] = __builtin_bit_cast(f16x8, raw[i][1]); }
                return;
            }
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(vec + 16 * g + 8 * kh), g1 = *reinterpret_cast<const f32x4*>(vec + 16 * g + 8 * kh + 4);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(vec + D + 16 * g + 8 * kh), e1 = *reinterpret_cast<const f32x4*>(vec + D + 16 * g + 8 * kh + 4);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                f32x4 w0, w1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    w0[e] = rok[i] ? act_f((raw[i][0][e] - st_mean[i]) * st_rstd[i] * g0[e] + e0[e], ACT) : 0.f;
                    w1[e] = rok[i] ? act_f((raw[i][1][e] - st_mean[i]) * st_rstd[i] * g1[e] + e1[e], ACT) : 0.f;
                }
                vmaxc = absmax4(absmax4(vmaxc, w0), w1);
                f16x4 h0, l0, h1, l1;
                split4(w0, h0, l0);
                split4(w1, h1, l1);
                xh[i] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                xl[i] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        vm_wait(integral_constant<int, 0>{});                    // (chained: this wave's output rows have reached the L2)
        lds_barrier();                                           // (chained: every wave is past its last residual / vector read of phase B)
        cp_vec(vec, p.g1, D, D); cp_vec(vec + D, p.be1, D, D);
        zero_acc();
        constexpr int nkt = DT;
        static_assert(nkt % 2 == 0, "k tiles are walked in pairs");
        RW ra, rb;
        f16x8 xh0[RT], xl0[RT], xh1[RT], xl1[RT];
        // issue order, prologue and every iteration alike: raw rows(q + 2) | weights(q + 2)
        ldy_raw(0, ra);
        dma_w_tile(W1, nkt, 0, 0);
        ldy_raw(1, rb);
        dma_w_tile(W1, nkt, 1, 1);
        vm_wait(integral_constant<int, 4 * RT + WPIECES>{});     // tile 0: rows and weights
        lds_barrier();                                           // (the loop vectors are visible; tile 0 is everybody's)
        norm_split(0, ra.v[0], xh0, xl0);
        int st = 0;
        // iteration q: k step 1 of tile q is normalised (its raw rows are free then), [raw rows | weights](q + 2) go out, MFMAs of k step 0;
        // then the normalisation of tile q + 1's k step 0 (requested a whole tile ago) beside the MFMAs of k step 1
        auto body = [&](int q, RW& rc, RW& rn_) {
            const int st2 = st == 0 ? 2 : st - 1;
            const char* Wt = chs + st * STAGE;
            __builtin_amdgcn_sched_barrier(0);
            norm_split(2 * q + 1, rc.v[1], xh1, xl1);
            if (!(HD_CHAIN_ABL & 8)) ldy_raw(min(q + 2, nkt - 1), rc);
            if (!(HD_CHAIN_ABL & 2)) dma_w_tile(W1, nkt, min(q + 2, nkt - 1), st2);
            mma_step(Wt, woff0, xh0, xl0);
            interleave(integral_constant<int, MM_STEP>{}, integral_constant<int, 1>{}, integral_constant<int, 4>{}, integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            vm_wait(integral_constant<int, 2 * WPIECES + 4 * RT>{});     // raw rows of tile q + 1 have landed (in flight: weights(q + 1), [rows | weights](q + 2))
            norm_split(2 * min(q + 1, nkt - 1), rn_.v[0], xh0, xl0);
            mma_step(Wt, woff1, xh1, xl1);
            interleave(integral_constant<int, MM_STEP>{}, integral_constant<int, 1>{}, integral_constant<int, 4>{}, integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            vm_wait(integral_constant<int, WPIECES + 4 * RT>{}); // tile q + 1's weights have landed
            lds_barrier();
            st = st == 2 ? 0 : st + 1;
        };
        for (int q = 0; q < nkt; q += 2) { body(q, ra, rb); body(q + 1, rb, ra); }
        raise_range_flag(p.rs, vmaxc);
        finish_ln_split(p.sc_1, p.b1, p.g2, p.be2);
        store_x16(p.H1out);
    }
}

#ifdef HD_CHAIN_EXPERIMENT
// ---- experiment (round 6, NOTES.md E): phase A with TWO waves per SIMD -- eight waves per workgroup, a PAIR of waves owns 32 rows and each
// wave of the pair half of the hidden channels (CT / 2 tiles: 96 accumulator registers at 384 channels), so that a wave stalled in the issue
// of a DMA piece or waiting for a fragment has a neighbour that feeds the matrix pipe.  Probe libraries only (-DHD_CHAIN_EXPERIMENT,
// HUDIFF_CHAIN_EXP=2 launches it in place of every block: timing, not results).
template <int CT, int ACT>
__global__ void __launch_bounds__(512, 1) bn_pair_a_k(const ChainP p) {
    constexpr int DH = 32 * CT, NTH = CT / 4, STAGE = CT * 4096, CTW = CT / 2, NW = 8;
    extern __shared__ __attribute__((aligned(16))) char chs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 1, hf = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    int bx = blockIdx.x, seg = 0;
    if (bx >= p.tiles) return;
    if (p.sg.nseg > 1 && bx >= p.tiles0) { seg = 1; bx -= p.tiles0; }
    const int Lc = p.sg.len[seg], seg_rows = p.sg.B * Lc, rbase = p.sg.base[seg];
    const int m0 = bx * 128 + grp * 32;
    typedef __attribute__((address_space(3))) void* lds_vp;
    constexpr uint32_t BUF_OOB = 0x80000000u;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    using std::integral_constant;
    const int lrow = m0 + l31; const bool rok = lrow < seg_rows;
    const int fsw = (lane >> 2) & 3;
    const int woff0 = l31 * 64 + (((0 + kh) ^ fsw) << 4), woff1 = l31 * 64 + (((2 + kh) ^ fsw) << 4);
    f32x16 acc[CTW];
    constexpr int WPIECES = NTH * 16 / NW;
    const uint16_t* Wc = p.Wc + (long)seg * p.wc_seg;
    const int nkt = p.taps * CT, half = (p.taps - 1) / 2;
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Wc), 0, NTH * nkt * X3_TILE_BYTES, 0x00020000);
    auto dma_w_tile = [&](int q, int st) {
        char* dst = chs + st * STAGE;
#pragma unroll
        for (int j = 0; j < WPIECES; ++j) {
            const int piece = NW * j + wave;
            const int nt = piece >> 4, within = piece & 15;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_vp)(dst + piece * 1024), 16, (int)((nt * nkt) * X3_TILE_BYTES + within * 1024 + lane * 16), q * X3_TILE_BYTES, 0, 0);
        }
    };
    auto mma_step = [&](const char* Wt, int wo, const f16x8 xh, const f16x8 xl) {
        f16x8 wh[2][2], wl[2][2];
        auto ldw = [&](int tp, int b) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = hf * CTW + 2 * tp + u;
                const char* a = Wt + (t >> 2) * 16384 + (t & 3) * 2048 + wo;
                wh[b][u] = *reinterpret_cast<const f16x8*>(a);
                wl[b][u] = *reinterpret_cast<const f16x8*>(a + 8192);
            }
        };
        ldw(0, 0);
#pragma unroll
        for (int tp = 0; tp < CTW / 2; ++tp) {
            const int b = tp & 1;
            if (tp + 1 < CTW / 2) ldw(tp + 1, b ^ 1);
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[2 * tp + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][u], xl, acc[2 * tp + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[2 * tp + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[b][u], xh, acc[2 * tp + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[2 * tp + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b][u], xh, acc[2 * tp + u], 0, 0, 0);
        }
    };
    auto vm_wait = [&](auto n_c) {
        constexpr int N = decltype(n_c)::value;
        __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0F70);
    };
    const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.H1), 0, (int)p.h1_bytes, 0x00020000);
    const int pos = rok ? lrow % Lc : -(1 << 24);
    struct XT { f16x8 h[2], l[2]; };
    auto ldx = [&](int q, XT& x) {
        const int tap = q / CT, kt = q - tap * CT;
        const int shift = (tap - half) * p.dil;
        const uint32_t vo = (unsigned)(pos + shift) < (unsigned)Lc ? (uint32_t)(rbase + lrow + shift) * (uint32_t)(DH * 4) + (uint32_t)kh * 16u : BUF_OOB;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            x.h[s2] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(h_rs, (int)vo, (2 * kt + s2) * 64, 0));
            x.l[s2] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(h_rs, (int)vo, (2 * kt + s2) * 64 + 32, 0));
        }
    };
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    XT xa, xb;
    dma_w_tile(0, 0);
    ldx(0, xa);
    dma_w_tile(1, 1);
    int st = 0;
    auto body = [&](int q, XT& xc, XT& xn) {
        vm_wait(integral_constant<int, WPIECES>{});
        lds_barrier();
        const int st2 = st == 0 ? 2 : st - 1;
        ldx(min(q + 1, nkt - 1), xn);
        dma_w_tile(min(q + 2, nkt - 1), st2);
        const char* Wt = chs + st * STAGE;
        mma_step(Wt, woff0, xc.h[0], xc.l[0]);
        mma_step(Wt, woff1, xc.h[1], xc.l[1]);
        __builtin_amdgcn_sched_barrier(0);
        st = st == 2 ? 0 : st + 1;
    };
    for (int q = 0; q < nkt; q += 2) { body(q, xa, xb); body(q + 1, xb, xa); }
    // finish: scale + bias, LayerNorm over the pair's DH channels (each wave its half: (mean, M2) of the halves merged through LDS), activation, split, X16 rows out
    vm_wait(integral_constant<int, 0>{});
    lds_barrier();
    float* fin = reinterpret_cast<float*>(chs);
    for (int i = tid; i < DH; i += 512) { fin[i] = p.bc[seg * DH + i]; fin[DH + i] = p.g3[seg * DH + i]; fin[2 * DH + i] = p.be3[seg * DH + i]; }
    float2* xch = reinterpret_cast<float2*>(chs + 3 * DH * 4);          // [wave][32 rows] (mean, M2) of a wave's channel half
    lds_barrier();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(fin + 32 * (hf * CTW + t) + 8 * q + 4 * kh);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float v = acc[t][4 * q + e] * p.sc_c + b4[e]; acc[t][4 * q + e] = v; s += v; }
        }
    s += __shfl_xor(s, 32);
    const float mh = s * (1.0f / (float)(DH / 2));
    float m2 = 0.f;
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[t][r] - mh; m2 += d * d; }
    m2 += __shfl_xor(m2, 32);
    if (kh == 0) xch[wave * 32 + l31] = make_float2(mh, m2);
    lds_barrier();
    const float2 o = xch[(wave ^ 1) * 32 + l31];
    const float mean = 0.5f * (mh + o.x), dlt = mh - o.x;
    const float rstd = 1.0f / sqrtf((m2 + o.y + dlt * dlt * (0.25f * (float)DH)) * (1.0f / (float)DH) + 1e-5f);
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(p.H2dbg, 0, 0x7FFFFFFF, 0x00020000);
    const uint32_t rb = rok ? (uint32_t)(rbase + lrow) * (uint32_t)(DH * 4) : BUF_OOB;
    float vmax = 0.f;
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        const int tg = hf * CTW + t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(fin + DH + 32 * tg + 8 * q + 4 * kh);
            const f32x4 e4 = *reinterpret_cast<const f32x4*>(fin + 2 * DH + 32 * tg + 8 * q + 4 * kh);
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = act_f((acc[t][4 * q + e] - mean) * rstd * g4[e] + e4[e], ACT);
            if (rok) vmax = absmax4(vmax, w);
            f16x4 h4, l4;
            split4(w, h4, l4);
            const int c = 32 * tg + 8 * q + 4 * kh;
            const uint32_t oo = rb == BUF_OOB ? BUF_OOB : rb + (uint32_t)(x16_hi(c) * 2);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), d_rs, (int)oo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, l4), d_rs, (int)(oo == BUF_OOB ? BUF_OOB : oo + (uint32_t)X16_LO * 2u), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    raise_range_flag(p.rs, vmax);
}
static void launch_bn_pair_a(ChainP p, int DH, hipStream_t st) {
    const int rows0 = p.sg.B * p.sg.len[0], rows1 = p.sg.nseg > 1 ? p.sg.B * p.sg.len[1] : 0;
    p.tiles0 = (rows0 + 127) / 128; p.tiles = p.tiles0 + (rows1 + 127) / 128;
    static bool prep = false;
    if (!prep) { prep = true; hipFuncSetAttribute((const void*)bn_pair_a_k<12, ACT_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 12 * 4096 + 8192);
                 hipFuncSetAttribute((const void*)bn_pair_a_k<8, ACT_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 8 * 4096 + 8192); }
    if (DH == 384) hipLaunchKernelGGL((bn_pair_a_k<12, ACT_RELU>), dim3(p.tiles), dim3(512), 3 * 12 * 4096 + 8192, st, p);
    else if (DH == 256) hipLaunchKernelGGL((bn_pair_a_k<8, ACT_GELU>), dim3(p.tiles), dim3(512), 3 * 8 * 4096 + 8192, st, p);
}
#endif

// ---- host side: instantiations and launcher ----------------------------------------------------------------------------------------
// (CT, DT) = (hidden width / 32, block width / 32): 12 x 24 DualConv (768 / 384), 8 x 16 NanoConv (512 / 256), 4 x 8 token encoder (256 / 128)
static bool bn_chain_supported(int DH, int D, int act) {
    return (DH == 384 && D == 768 && act == ACT_RELU) || (DH == 256 && D == 512 && act == ACT_GELU) || (DH == 128 && D == 256 && act == ACT_GELU);
}
// rows of a workgroup tile of the instantiation that serves (DH, D)
static int bn_chain_tile_rows(int DH) { return DH == 128 ? 256 : 128; }

// Instantiations (kept few: each costs ~10 s of hipcc): per width PH = 2 (open a stack), PH = 3 (a middle block) and PH = 1 with the FL bits
// the LAST block of that stack always carries (token encoder: the addend `extra`; Dual / NanoConv: the X16 copy for the attention layers).
// Injected keep-masks (parity tests) and PH = 4 (phase A alone) exist only in -DHD_CHAIN_EXPERIMENT builds; the host keeps launches with
// injected masks on the gemm_x3_k path (bn_chain_use, hd_api.hip).
template <int CT, int DT, int RT, int ACT, int LASTFL>
static hipError_t bn_chain_prep1() {
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)bn_chain_k<CT, DT, RT, 1, ACT, false, LASTFL>, hipFuncAttributeMaxDynamicSharedMemorySize, ChainGeom<CT, DT, RT>::SMEM)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)bn_chain_k<CT, DT, RT, 2, ACT, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, ChainGeom<CT, DT, RT>::SMEM)) != hipSuccess) return e;
#ifdef HD_CHAIN_EXPERIMENT
    if ((e = hipFuncSetAttribute((const void*)bn_chain_k<CT, DT, RT, 4, ACT, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, ChainGeom<CT, DT, RT>::SMEM)) != hipSuccess) return e;
#endif
    return hipFuncSetAttribute((const void*)bn_chain_k<CT, DT, RT, 3, ACT, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, ChainGeom<CT, DT, RT>::SMEM);
}
static hipError_t bn_chain_prepare() {
    hipError_t e;
    if ((e = bn_chain_prep1<12, 24, 1, ACT_RELU, 1>()) != hipSuccess) return e;
    if ((e = bn_chain_prep1<8, 16, 1, ACT_GELU, 1>()) != hipSuccess) return e;
    return bn_chain_prep1<4, 8, 2, ACT_GELU, 2>();
}

static void launch_bn_chain(ChainP p, int DH, int D, hipStream_t st) {
    const int rows0 = p.sg.B * p.sg.len[0], rows1 = p.sg.nseg > 1 ? p.sg.B * p.sg.len[1] : 0;
    auto go = [&](auto ct, auto dt, auto rt, auto ac, auto lf) {
        constexpr int CT = decltype(ct)::value, DT = decltype(dt)::value, RT = decltype(rt)::value, ACT = decltype(ac)::value, LASTFL = decltype(lf)::value;
        constexpr int SM = ChainGeom<CT, DT, RT>::SMEM, ROWS = ChainGeom<CT, DT, RT>::ROWS;
        p.tiles0 = (rows0 + ROWS - 1) / ROWS;
        p.tiles = p.tiles0 + (rows1 + ROWS - 1) / ROWS;
        const dim3 grid(p.tiles), blk(CH_THREADS);
        switch (p.phases) {
            case 1: hipLaunchKernelGGL((bn_chain_k<CT, DT, RT, 1, ACT, false, LASTFL>), grid, blk, SM, st, p); break;     // (the caller sets YX / extra: bytenet_stack_chain)
            case 2: hipLaunchKernelGGL((bn_chain_k<CT, DT, RT, 2, ACT, false, 0>), grid, blk, SM, st, p); break;
            case 3: hipLaunchKernelGGL((bn_chain_k<CT, DT, RT, 3, ACT, false, 0>), grid, blk, SM, st, p); break;
#ifdef HD_CHAIN_EXPERIMENT
            default: hipLaunchKernelGGL((bn_chain_k<CT, DT, RT, 4, ACT, false, 0>), grid, blk, SM, st, p); break;
#else
            default: break;
#endif
        }
    };
    using std::integral_constant;
    if (DH == 384 && D == 768) go(integral_constant<int, 12>{}, integral_constant<int, 24>{}, integral_constant<int, 1>{}, integral_constant<int, ACT_RELU>{}, integral_constant<int, 1>{});
    else if (DH == 256 && D == 512) go(integral_constant<int, 8>{}, integral_constant<int, 16>{}, integral_constant<int, 1>{}, integral_constant<int, ACT_GELU>{}, integral_constant<int, 1>{});
    else go(integral_constant<int, 4>{}, integral_constant<int, 8>{}, integral_constant<int, 2>{}, integral_constant<int, ACT_GELU>{}, integral_constant<int, 2>{});
}

}  // namespace hd
