// hd_attn_fused.hip.h -- the Q|K|V projection of an attention layer fused into its attention core (round 5, VERDICT r4 "Next" #1a).
//
// Reference: AttLayer.forward, model/encoder/cross_attention.py:149-173 -- Q, K, V = Linear(d -> 512)(x), 8 heads x 64, RoPE on Q and K,
// softmax(Q K^T / 8) V, then the out-projection (which stays a separate GEMM).  Rounds 2-4 ran it as two launches: gemm_x3_k wrote
// fp32 Q|K|V rows [rows, 1536] (458 MB per launch at 256 antibodies, an epilogue nothing overlapped: one 256 x 256 block per CU) and
// attn_x3_k read them straight back, rotated and split K / V while staging them into LDS.  That round trip was 9.2 GB of the 46 GB a
// denoiser step moved.
//
// Here ONE workgroup per (sequence, head group) does both:
//   1. projection   C[rows of the sequence, 192 per head] = X_seq[ROWS, D] . W_head[D, 192]   (Q | K | V columns of its heads), the
//                   split-precision K loop of gemm_x3_k: X in X16 rows and the pre-tiled (hi, lo) weight images by LDS DMA through
//                   two stages, three v_mfma_f32_32x32x16_f16 per product.  Twelve waves, five 32 x 32 tiles each; the K columns are
//                   multiplied as W^T X^T (operands swapped) so that a lane ends up with four consecutive head dimensions of ONE key
//                   -- the RoPE pairs and an 8-byte piece of the K plane row -- while Q and V come out rows x d, which is what the
//                   fp32 Q rows and the key-major V^T planes want.
//   2. hand-over    accumulators -> [rstd of a folded LayerNorm] -> + bias -> Q: rotated, scaled, split, to global as the core's MFMA fragments
//                   (the workgroup reads them back past the L1, 76 KB per head); K: rotated, split, into the K planes; V: split, into the permuted V^T planes -- the
//                   very LDS images attn_x3_k stages (AxGeom), laid over the operand stages once every wave has left the K loop.
//   3. attention    attn_x3_tiles (hd_kernels.hip.h), unchanged: S^T = K Q^T, fp32 softmax, O^T = V^T P^T, O rows out in X16 form.
// Antibody model (L = 291): one head per workgroup, 304 x 192 tile (stages 2 x 64 KB, planes 156 KB); nanobody model (L = 152): two
// heads per workgroup, 160 x 384 tile (stages 2 x 68 KB, planes 2 x 78 KB).  One workgroup per CU either way.
// Arithmetic: the same three-term split products and fp32 accumulation in the same k order as gemm_x3_k (k tiles of 32, two k steps,
// cross terms first), the same epilogue formula, the same RoPE, split and softmax code: results differ from the two-launch route only
// where the orientation changes the order inside an MFMA (nowhere: each output element is one dot product over k in both orientations).
#pragma once
#include "hd_kernels.hip.h"

namespace hd {

struct QkvAttnP {
    const float* X; int ldx; uint32_t x_bytes;     // layer input in X16 split form [rows, D] (ldx = D = K of the projection), its extent in bytes
    const uint16_t* Wx; float acc_scale;           // fused Q|K|V weight images (X3Packer: tiles [3 att / 128][D / 32] of 16 KiB), 2^-shift
    const float* bias;                             // [3 att] (beta W + b when the LayerNorm is folded)
    int ln_fold; const float2* stats; const float2* spart; int spw; long spart_rows;     // folded LayerNorm: rstd of every input row
    float* QKV; int ldq; int att;                  // Q rows out, fp32, columns [0, att) of [rows, ldq]
    const float* rope_cos; const float* rope_sin;
    const float* rope_cs;                          // the same table interleaved, [L][32] x (cos, sin): one 16-byte load per two RoPE pairs
    float* O; int ldo;                             // attention output rows, X16 split form [rows, att]
    int nhead; Segs sg; const RunState* rs;
#ifdef HD_PROBES
    int abl;                                       // probe builds only (-DHD_PROBES, HUDIFF_QA_ABL): 1 no MFMAs in the projection, 2 no operand DMA after the first tile,
                                                   // 4 no attention core, 8 no hand-over of K / V into the planes, 32 phase time stamps (100 MHz
                                                   // ticks since kernel entry, per wave of the head-0 workgroups) into the unused V third of QKV
#endif
};

constexpr int QA_THREADS = 768, QA_WAVES = 12;

template <int KT, int NH>
struct QaGeom {
    typedef AxGeom<KT> G;
    static constexpr int RT = (16 * KT + 31) / 32;             // 32-row tiles of the sequence (10 / 5)
    static constexpr int ROWS = 32 * RT, COLS = 192 * NH, CT = COLS / 32;
    static constexpr int RG = QA_WAVES / CT;                   // row groups (2 / 1)
    static constexpr int TM = RT / RG;                         // row tiles per wave (5)
    static constexpr int A_BYTES = ROWS * 128, W_BYTES = COLS * 128, STAGE = A_BYTES + W_BYTES;
    static constexpr int A_PIECES = A_BYTES / 1024, W_PIECES = W_BYTES / 1024;
    static constexpr int A_PER_WAVE = (A_PIECES + QA_WAVES - 1) / QA_WAVES, W_PER_WAVE = W_PIECES / QA_WAVES;
    static constexpr int RSTD_OFF = 2 * STAGE;                 // float rstd[ROWS] behind the two stages
    // K / V planes of head hh at hh * PLANES (exact-length K planes for the short model, as attn_x3_k<10> has them)
    __host__ __device__ static constexpr int planes(int L) { return 2 * (KT <= 10 ? L : G::KROWS) * 128 + 2 * G::VPLANE; }
    static constexpr int BIAS_OFF = RSTD_OFF + ROWS * 4;       // float bias[COLS] of the workgroup's columns, in the order of the weight rows in LDS
    __host__ __device__ static constexpr int smem(int L) {
        const int a = NH * planes(L), b = BIAS_OFF + COLS * 4;
        return a > b ? a : b;
    }
    static_assert(RT % RG == 0 && CT * RG == QA_WAVES && W_PIECES % QA_WAVES == 0 && TM == 5, "tile / wave split");
    static_assert(BIAS_OFF + COLS * 4 <= LDS_PER_CU && NH * (2 * G::KROWS * 128 + 2 * G::VPLANE) <= LDS_PER_CU + (KT <= 10 ? 4096 : 0), "LDS");
};

template <int KT, int NH>
__global__ void __launch_bounds__(QA_THREADS, 3) qkv_attn_x3_k(const QkvAttnP p) {
    typedef QaGeom<KT, NH> Q;
    typedef AxGeom<KT> G;
    constexpr int TM = Q::TM, A_BYTES = Q::A_BYTES, W_BYTES = Q::W_BYTES, STAGE = Q::STAGE;
    constexpr bool EXACT = KT <= 10;
    extern __shared__ __attribute__((aligned(16))) char qas[];
    const int L = p.sg.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (sequence, head group): the head groups of a sequence run on ONE XCD in consecutive slots, so that the sequence's
    // input rows (0.9 MB / 0.3 MB) come from HBM once and from that XCD's L2 for the other heads (the tile order of gemm_k)
    const int HG = p.nhead / NH;
    int b, h0;
    {
        const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        b = (slot / HG) * 8 + xcd;
        h0 = (slot % HG) * NH;
        if (b >= p.sg.B) return;
    }
    const int rA0 = p.sg.base[0] + b * p.sg.len[0] - p.sg.off[0];
    const int rA1 = p.sg.nseg > 1 ? p.sg.base[1] + b * p.sg.len[1] - p.sg.off[1] : rA0;
    const int roff1 = p.sg.nseg > 1 ? p.sg.off[1] : 0x7fffffff;
    const int nkt = p.ldx / X3_BK;
    const unsigned long long t_entry = (HD_QABL(p) & 32) ? wall_clock64() : 0ull;
    auto stamp = [&](int k) {                          // probe (abl bit 5): phase k of this wave, head-0 workgroups only
        if ((HD_QABL(p) & 32) && h0 == 0 && lane == 0)
            p.QKV[(long)(rA0 + p.sg.off[0]) * p.ldq + 2 * p.att + wave * 8 + k] = (float)(wall_clock64() - t_entry);
    };
    float* rstd_s = reinterpret_cast<float*>(qas + Q::RSTD_OFF);
    // rstd of every row of the sequence for a folded LayerNorm (1 without one) and the bias of the workgroup's columns, into LDS (visible
    // after the first barrier; called once the first tile's DMA is in flight): step 1 of the hand-over then reads them with vector LDS
    // reads instead of opening with a global round trip (bias) and 80 dependent scalar reads behind a branch each (V waves, round 5 stamps:
    // 6.0 us of step 1 against 2.3 us for the K waves)
    float* bias_s = reinterpret_cast<float*>(qas + Q::BIAS_OFF);
    auto row_rstd = [&]() {
        if (tid < Q::COLS) {
            const int hh_ = tid / 192, part_ = (tid % 192) / 64, within = tid % 64;
            bias_s[tid] = p.bias[part_ * p.att + (h0 + hh_) * ATT_HD + within];
        }
        for (int r = tid; r < Q::ROWS; r += QA_THREADS) {
            float v = 1.f;
            if (p.ln_fold && r < L) {
                const long grow = r + (r >= roff1 ? rA1 : rA0);
                v = p.spart ? merge_row_stat(p.spart, p.spw, p.spart_rows, p.ldx, grow).y : p.stats[grow].y;
            }
            rstd_s[r] = v;
        }
    };

    // ---- operand DMA (pieces of 1 KiB, as gemm_x3_k) -------------------------------------------------------------------------
    constexpr uint32_t BUF_OOB = 0x80000000u;
    uint32_t a_vo[Q::A_PER_WAVE], w_vo[Q::W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < Q::A_PER_WAVE; ++i) {
        const int piece = QA_WAVES * i + wave;                                   // rows 8 piece .. of the sequence
        const int r = 8 * piece + (lane >> 3);
        const uint32_t a_in = (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) << 4);    // swizzled chunk of the row's 128-byte (hi | lo) group
        const long grow = r + (r >= roff1 ? rA1 : rA0);
        a_vo[i] = (piece < Q::A_PIECES && r < L) ? (uint32_t)(grow * p.ldx * 4) + a_in : BUF_OOB;       // rows >= L: the DMA writes zeros
    }
#pragma unroll
    for (int i = 0; i < Q::W_PER_WAVE; ++i) {
        const int piece = QA_WAVES * i + wave;                                   // LDS image: hi plane of all COLS rows, then the lo plane
        const int plane = piece / (Q::COLS / 16), rg16 = piece % (Q::COLS / 16);
        const int R = rg16 * 16, hh = R / 192, part = (R % 192) / 64, within = R % 64;
        const int col = part * p.att + (h0 + hh) * ATT_HD + within;              // column of the fused [D, 3 att] matrix
        w_vo[i] = (uint32_t)((col / X3_BN) * nkt * X3_TILE_BYTES + plane * (X3_TILE_BYTES / 2) + (col % X3_BN) * 64 + lane * 16);
    }
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.Wx), 0, (3 * p.att / X3_BN) * nkt * X3_TILE_BYTES, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_vp;
    auto dma = [&](int kt, int st) {
        char* dst = qas + st * STAGE;
#pragma unroll
        for (int i = 0; i < Q::A_PER_WAVE; ++i)
            if (QA_WAVES * i + wave < Q::A_PIECES)                               // (wave-uniform)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_vp)(dst + (QA_WAVES * i + wave) * 1024), 16, (int)a_vo[i], kt * X3_BK * 4, 0, 0);
#pragma unroll
        for (int i = 0; i < Q::W_PER_WAVE; ++i) {
            const int piece = QA_WAVES * i + wave;
            const int plane = piece / (Q::COLS / 16), rg16 = piece % (Q::COLS / 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_vp)(dst + A_BYTES + plane * (W_BYTES / 2) + rg16 * 1024), 16, (int)w_vo[i],
                                                     kt * X3_TILE_BYTES, 0, 0);
        }
    };

    // ---- wave -> five 32 x 32 tiles: column tile ct (32 of the COLS output columns), row tiles TM rg .. ----------------------
    const int ct = wave % Q::CT, rg = wave / Q::CT;
    const int hh = ct / 6, part = (ct % 6) >> 1, half = ct & 1;                  // head of the group, 0 Q / 1 K / 2 V, which 32 of its 64 columns
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int fsw = (lane >> 2) & 3, fg = lane >> 5, asw = (lane >> 1) & 7;
    const int foff0 = (lane & 31) * 64 + (((0 + fg) ^ fsw) << 4), foff1 = (lane & 31) * 64 + (((2 + fg) ^ fsw) << 4);
    const int aoff0 = (lane & 31) * 128 + (((0 + fg) ^ asw) << 4), aoff1 = (lane & 31) * 128 + (((4 + fg) ^ asw) << 4);      // (X16: hi chunk 4 ks + g)
    // The K loop, software-pipelined by hand.  A k tile is four steps -- (k step 0 | 1) x (row tiles 0-2 | 3-4) -- and the fragments of
    // step s + 1 are requested from LDS BEFORE the MFMAs of step s are issued (across the tile's closing barrier too: wave_program);
    // the batches of 3 + 2 tiles keep at most 56 fragment registers live beside
    // the 80 accumulators (three waves per SIMD: 168 registers).  hipcc left to itself read every fragment right in front of its MFMA
    // (s_waitcnt lgkmcnt between them) once the whole-tile form had spilled the accumulators.
    // Per step: the two cross terms first, the leading term last (gemm_x3_k's order), independent accumulators inside a term.
    // SW (compile time): the K columns' orientation.  The whole loop exists once per orientation (a run-time choice inside it splits it
    // into basic blocks the register allocator then spills an accumulator around).
    constexpr int XROW = 128;                          // bytes of an LDS row of the A image (a k tile of a row: one X16 line)
    auto load_x = [&](const char* At, int oa, auto i0_c, auto n_c, f16x8* xh, f16x8* xl) {
        constexpr int I0 = decltype(i0_c)::value, N = decltype(n_c)::value;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            xh[i] = *reinterpret_cast<const f16x8*>(At + oa + 32 * XROW * (I0 + i));
            xl[i] = *reinterpret_cast<const f16x8*>(At + (oa ^ 32) + 32 * XROW * (I0 + i));  // low parts: chunk + 2
        }
    };
    auto mm = [&](auto sw_c, auto i0_c, auto n_c, const f16x8* xh, const f16x8* xl, const f16x8 wh, const f16x8 wl) {
        constexpr bool SW = decltype(sw_c)::value;
        constexpr int I0 = decltype(i0_c)::value, N = decltype(n_c)::value;
#pragma unroll
        for (int i = 0; i < N; ++i) acc[I0 + i] = SW ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[i], acc[I0 + i], 0, 0, 0)
                                                     : __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[i], wh, acc[I0 + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[I0 + i] = SW ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[i], acc[I0 + i], 0, 0, 0)
                                                     : __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[i], wl, acc[I0 + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[I0 + i] = SW ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[i], acc[I0 + i], 0, 0, 0)
                                                     : __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[i], wh, acc[I0 + i], 0, 0, 0);
    };
    constexpr int WAIT_ALL = 0x0F70;                   // vmcnt(0): two stages, one tile in flight
    const int khalf = lane >> 5, l31 = lane & 31;
    const int hcol = (h0 + hh) * ATT_HD;                                         // first column of this wave's head inside a Q / K / V block
    const float sc = p.acc_scale;
    const int krows = EXACT ? L : G::KROWS;
    // A wave's whole program up to the attention core, once per PART (0 Q, 1 K, 2 V; compile time): K loop, finish, hand-over.  One
    // straight-line program per part instead of run-time choices inside a shared one -- those cost the register allocator dozens of
    // spilled accumulators -- and all three meet the same barriers (24 k tiles + 2).
    auto wave_program = [&](auto part_c) {
        using std::integral_constant;
        constexpr int PART = decltype(part_c)::value;
        constexpr std::integral_constant<bool, PART != 2> sw_c{};                // Q and K: W^T X^T (a lane holds four consecutive head dimensions of one row)
        constexpr integral_constant<int, 0> c0{}; constexpr integral_constant<int, 2> c2{}; constexpr integral_constant<int, 3> c3{};
        dma(0, 0);
        row_rstd();
        __builtin_amdgcn_s_waitcnt(WAIT_ALL);
        lds_barrier();
#ifdef HD_QA_STAMPS
        uint32_t ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // probe build: shader-clock stamps of k tiles 10 and 11 (start, [10: own DMA issued,] last MFMA issued, DMA landed) + start of tile 12
#endif
        // The loop carries two things over a tile's closing barrier: the tile's LAST MFMA batch (row tiles 3-4 of k step 1; its fragments are in
        // registers, the stage is free) and -- requested right behind the barrier, in front of that batch -- the next tile's first fragments.
        // The matrix pipe then has eighteen MFMAs per SIMD to run while the first LDS reads of the new tile are under way, instead of
        // draining at every barrier and idling for a read round trip behind it (tile stamps, NOTES.md D: ~600 of a tile's ~3 900 clocks).
        // Order of the products inside every accumulator: unchanged.
        if (HD_QABL(p) & 1) {                               // probe: no MFMAs (and no fragment reads)
            for (int kt = 0, st = 0; kt < nkt; ++kt, st ^= 1) {
                if (kt + 1 < nkt && !(HD_QABL(p) & 2)) dma(kt + 1, st ^ 1);
                __builtin_amdgcn_s_waitcnt(WAIT_ALL);
                lds_barrier();
            }
        } else {
            f16x8 a0h[3], a0l[3], b0h[2], b0l[2], a1h[3], a1l[3], b1h[2], b1l[2], w0h, w0l, w1h, w1l;
            {
                const char* At = qas + rg * TM * 32 * 128;
                const char* Wt = qas + A_BYTES + ct * 32 * 64;
                w0h = *reinterpret_cast<const f16x8*>(Wt + foff0); w0l = *reinterpret_cast<const f16x8*>(Wt + W_BYTES / 2 + foff0);
                load_x(At, aoff0, c0, c3, a0h, a0l);
            }
            for (int kt = 0, st = 0; kt < nkt; ++kt, st ^= 1) {
#ifdef HD_QA_STAMPS
                if (kt == 10) ts[0] = (uint32_t)__builtin_readcyclecounter();
                if (kt == 11) ts[4] = (uint32_t)__builtin_readcyclecounter();
                if (kt == 12) ts[7] = (uint32_t)__builtin_readcyclecounter();
#endif
                const char* At = qas + st * STAGE + rg * TM * 32 * 128;
                const char* Wt = qas + st * STAGE + A_BYTES + ct * 32 * 64;
                // the other stage was read in tile kt - 1 and every wave is past that tile's barrier
                if (kt + 1 < nkt && !(HD_QABL(p) & 2)) dma(kt + 1, st ^ 1);
#ifdef HD_QA_STAMPS
                if (kt == 10) ts[1] = (uint32_t)__builtin_readcyclecounter();
#endif
                load_x(At, aoff0, c3, c2, b0h, b0l);
                __builtin_amdgcn_sched_barrier(0);
                mm(sw_c, c0, c3, a0h, a0l, w0h, w0l);
                __builtin_amdgcn_sched_barrier(0);
                w1h = *reinterpret_cast<const f16x8*>(Wt + foff1); w1l = *reinterpret_cast<const f16x8*>(Wt + W_BYTES / 2 + foff1);
                load_x(At, aoff1, c0, c3, a1h, a1l);
                __builtin_amdgcn_sched_barrier(0);
                mm(sw_c, c3, c2, b0h, b0l, w0h, w0l);
                __builtin_amdgcn_sched_barrier(0);
                load_x(At, aoff1, c3, c2, b1h, b1l);
                __builtin_amdgcn_sched_barrier(0);
                mm(sw_c, c0, c3, a1h, a1l, w1h, w1l);
                __builtin_amdgcn_sched_barrier(0);
#ifdef HD_QA_STAMPS
                if (kt == 10) ts[2] = (uint32_t)__builtin_readcyclecounter();
                if (kt == 11) ts[5] = (uint32_t)__builtin_readcyclecounter();
#endif
                __builtin_amdgcn_s_waitcnt(WAIT_ALL);  // tile kt + 1 has landed ...
#ifdef HD_QA_STAMPS
                if (kt == 10) ts[3] = (uint32_t)__builtin_readcyclecounter();
                if (kt == 11) ts[6] = (uint32_t)__builtin_readcyclecounter();
#endif
                lds_barrier();                         // ... everybody's part of it; everybody has READ all of tile kt (the last fragments are in registers)
                {                                      // (behind the last tile: reads of the other stage nobody uses -- no branch in the loop body)
                    const char* An = qas + (st ^ 1) * STAGE + rg * TM * 32 * 128;
                    const char* Wn = qas + (st ^ 1) * STAGE + A_BYTES + ct * 32 * 64;
                    w0h = *reinterpret_cast<const f16x8*>(Wn + foff0); w0l = *reinterpret_cast<const f16x8*>(Wn + W_BYTES / 2 + foff0);
                    load_x(An, aoff0, c0, c3, a0h, a0l);
                }
                __builtin_amdgcn_sched_barrier(0);
                mm(sw_c, c3, c2, b1h, b1l, w1h, w1l);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(0);                                      // K loop done
#ifdef HD_QA_STAMPS
        if (h0 == 0 && lane == 0)
            for (int k = 1; k < 8; ++k) p.QKV[(long)(rA0 + p.sg.off[0] + 1) * p.ldq + 2 * p.att + wave * 8 + k] = (float)(ts[k] - ts[0]);
#endif
        // ---- hand-over, step 1: finish the values in place (scale, rstd of a folded LayerNorm, bias); Q rows go out -----------
        if constexpr (PART == 2) {                     // V: rows x d -- a lane holds rows 8 j + 4 khalf + (0 .. 3) of each tile: one 16-byte read of their rstd
            const float bv = bias_s[ct * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 rs4 = *reinterpret_cast<const f32x4*>(rstd_s + 32 * (rg * TM + i) + 8 * j + 4 * khalf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][4 * j + e] = acc[i][4 * j + e] * sc * rs4[e] + bv;
                }
                asm volatile("" : "+v"(acc[i]));       // (finished HERE: left alone the compiler sinks the arithmetic behind the barrier, into
                                                       // step 2, and carries all twenty rstd reads across it: 80 registers, spills)
            }
        } else {                                       // Q, K: d x rows
            f32x4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bias_s + ct * 32 + 8 * j + 4 * khalf);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float rs_k = rstd_s[32 * (rg * TM + i) + l31];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] * sc * rs_k + bv[r >> 2][r & 3];
            }
        }
        if constexpr (PART == 0) {
            // Q goes to the attention core through global memory (the planes fill the LDS), READY TO MULTIPLY (round 5, core stamps: the fp32
            // rows this used to store cost every query tile twelve scattered loads -- Q rows at a 6 KB stride, RoPE table rows -- before its
            // first MFMA: 4 200 .. 13 700 clocks per tile, the longest phase of the core).  Here the Q waves rotate, scale (log2(e) / 8) and
            // split their rows, and store them as the B-operand fragments of S^T = K Q^T in the order the core's lanes read them: the 4 KiB
            // block of query tile qt = [k step 0 hi | k step 0 lo | k step 1 hi | k step 1 lo] x [lane qi + 16 g] x 16 bytes, so that a tile
            // is four fully coalesced 1 KiB loads.  Block (head h, tile qt) lies in the Q | K thirds (4 KiB) of row h KT + qt of the
            // sequence's first-segment rows (the host checks nhead KT <= rows of the segment; the V third stays free for the probes' stamps).
            const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(p.QKV, 0, 0x7fffffff, 0x00020000);
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            constexpr float QS = 0.125f * 1.44269504088896340736f;
            float vmaxq = 0.f;                             // range guard of the Q split (the core used to raise it)
            f32x4 cs[2][4];
            auto load_cs = [&](int i, f32x4 (&dst)[4]) {
                const int key = 32 * (rg * TM + i) + l31;
                const int kc = key < L ? key : L - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dst[j] = *reinterpret_cast<const f32x4*>(p.rope_cs + (kc * 32 + ((32 * half + 8 * j + 4 * khalf) >> 1)) * 2);
            };
            load_cs(0, cs[0]);
            const uint32_t blk0 = (uint32_t)(rA0 + p.sg.off[0] + (h0 + hh) * KT) * (uint32_t)p.ldq * 4u;      // block of (this head, tile 0)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = 32 * (rg * TM + i) + l31;
                if (i + 1 < TM) load_cs(i + 1, cs[(i + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t tb = m < 16 * KT ? blk0 + (uint32_t)(m >> 4) * (uint32_t)p.ldq * 4u + (uint32_t)(2 * half) * 1024u + (uint32_t)(m & 15) * 16u + (uint32_t)khalf * 8u
                                                : BUF_OOB;                      // (32 RT rows are multiplied, 16 KT tiles exist)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 c4 = cs[i & 1][j];
                    const float q0 = acc[i][4 * j], q1 = acc[i][4 * j + 1], q2 = acc[i][4 * j + 2], q3 = acc[i][4 * j + 3];
                    f32x4 r;
                    r[0] = (q0 * c4[0] - q1 * c4[1]) * QS; r[1] = (q0 * c4[1] + q1 * c4[0]) * QS;
                    r[2] = (q2 * c4[2] - q3 * c4[3]) * QS; r[3] = (q2 * c4[3] + q3 * c4[2]) * QS;
                    f16x4 hv, lv;
                    split4(r, hv, lv);
                    if ((HD_GUARD_MASK & 16) && m < L) vmaxq = absmax4(vmaxq, r);
                    const uint32_t o = tb == BUF_OOB ? BUF_OOB : tb + (uint32_t)(16 * j) * 16u;       // lane qi + 16 g of the core, g = j
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, hv), q_rs, (int)o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, lv), q_rs, (int)(o == BUF_OOB ? BUF_OOB : o + 1024u), 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            raise_range_flag(p.rs, vmaxq);
        }
        stamp(1);                                      // step 1 done
        lds_barrier();                                 // every wave has left the K loop and read its rstd: the planes may overwrite both
                                                       // (LDS-only: the Q stores and the table loads stay in flight; the barrier behind step 2 drains them)

        // ---- hand-over, step 2: K (rotated, split) and V (split) into the LDS images attn_x3_tiles reads (AxGeom<KT>) ---------
        stamp(2);                                      // past barrier 1
        char* planes = qas + hh * Q::planes(L);
        char* Kh = planes;
        char* Kl = planes + krows * 128;
        char* Vh = planes + 2 * krows * 128;
        char* Vl = Vh + G::VPLANE;
        float vmax = 0.f;                              // range guard: K and V are split from fp32 values here (X16_LIMIT)
        if (HD_QABL(p) & 8) {
        } else if constexpr (PART == 1) {
            // cos, sin, cos, sin of the two RoPE pairs of head dimensions d0 .. d0 + 3 (one 16-byte load); the table rows of tile i + 1 are
            // requested before tile i is rotated (two tiles = 32 registers in flight: five serial round trips were most of this phase;
            // requesting tiles 0 and 1 in front of step 1 measured the same and spills)
            f32x4 cs[2][4];
            auto load_cs = [&](int i, f32x4 (&dst)[4]) {
                const int key = 32 * (rg * TM + i) + l31;
                const int kc = key < L ? key : L - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dst[j] = *reinterpret_cast<const f32x4*>(p.rope_cs + (kc * 32 + ((32 * half + 8 * j + 4 * khalf) >> 1)) * 2);
            };
            load_cs(0, cs[0]);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int key = 32 * (rg * TM + i) + l31;
                if (i + 1 < TM) load_cs(i + 1, cs[(i + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d0 = 32 * half + 8 * j + 4 * khalf;                // four consecutive head dimensions: two RoPE pairs
                    const f32x4 c4 = cs[i & 1][j];
                    const float k0 = acc[i][4 * j], k1 = acc[i][4 * j + 1], k2 = acc[i][4 * j + 2], k3 = acc[i][4 * j + 3];
                    f32x4 kr;
                    kr[0] = k0 * c4[0] - k1 * c4[1]; kr[1] = k0 * c4[1] + k1 * c4[0];
                    kr[2] = k2 * c4[2] - k3 * c4[3]; kr[3] = k2 * c4[3] + k3 * c4[2];
                    if (32 * (rg * TM + i) + 32 > L && key >= L) kr = f32x4{0.f, 0.f, 0.f, 0.f};      // padding rows (only the last tile has any; KT = 19: the planes hold 16 KT rows)
                    f16x4 hv, lv;
                    split4(kr, hv, lv);
                    if (HD_GUARD_MASK & 8) vmax = absmax4(vmax, kr);
                    if (key < krows) {
                        const int off = key * 128 + ((((d0 >> 3) ^ ((key >> 1) & 7))) << 4) + (d0 & 7) * 2;
                        *reinterpret_cast<f16x4*>(Kh + off) = hv;
                        *reinterpret_cast<f16x4*>(Kl + off) = lv;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (PART == 2) {
            const int d = 32 * half + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int t = rg * TM + i;                                       // 32-key block
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = 8 * j + 4 * khalf;                             // first of four consecutive keys inside the block
                    f32x4 x4 = {acc[i][4 * j], acc[i][4 * j + 1], acc[i][4 * j + 2], acc[i][4 * j + 3]};
                    if (32 * t + 32 > L) {                                       // (wave-uniform: only the last tile holds keys >= L)
#pragma unroll
                        for (int e = 0; e < 4; ++e) x4[e] = (32 * t + o + e < L) ? x4[e] : 0.f;
                    }
                    f16x4 hv, lv;
                    split4(x4, hv, lv);
                    if (HD_GUARD_MASK & 8) vmax = absmax4(vmax, x4);
                    // chunk 4 t + g holds keys {32 t + 4 g + r, 32 t + 16 + 4 g + r}: first / second half of the 16-byte chunk
                    const int g = (o & 15) >> 2, second = o >> 4;
                    const int off = d * (G::VKEYS * 2) + (G::vpos(4 * t + g, d) << 4) + 8 * second;
                    *reinterpret_cast<f16x4*>(Vh + off) = hv;
                    *reinterpret_cast<f16x4*>(Vl + off) = lv;
                }
            }
        }
        if constexpr (PART != 0) raise_range_flag(p.rs, vmax);
        stamp(3);                                      // step 2 done
        __syncthreads();                               // planes complete; the Q stores of this workgroup have drained (vmcnt(0) in front of the barrier)
        stamp(4);                                      // past barrier 2
    };
    if (part == 0) wave_program(std::integral_constant<int, 0>{});
    else if (part == 1) wave_program(std::integral_constant<int, 1>{});
    else wave_program(std::integral_constant<int, 2>{});

    // ---- attention core -------------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(p.O, 0, 0x7fffffff, 0x00020000);
    if (HD_QABL(p) & 4) return;
#pragma unroll
    for (int hx = 0; hx < NH; ++hx) {
        const char* pl = qas + hx * Q::planes(L);
#ifdef HD_QA_STAMPS
        float* cstamps = (h0 == 0 && hx == 0) ? p.QKV + (long)(rA0 + p.sg.off[0] + 2) * p.ldq + 2 * p.att : nullptr;      // third row of the sequence, V third
#else
        float* cstamps = nullptr;
#endif
        attn_x3_tiles<KT, QA_THREADS, true>(pl, pl + krows * 128, pl + 2 * krows * 128, pl + 2 * krows * 128 + G::VPLANE, p.QKV, p.ldq,
                                            (int)((uint32_t)(rA0 + p.sg.off[0] + (h0 + hx) * KT) * (uint32_t)p.ldq * 4u),      // (QL2: byte offset of the head's Q fragment blocks)
                                            p.rope_cos, p.rope_sin, o_rs, p.ldo, b, h0 + hx, p.sg, 1, p.rs, lane, wave, cstamps);
    }
    stamp(5);                                          // attention done
}

}  // namespace hd
