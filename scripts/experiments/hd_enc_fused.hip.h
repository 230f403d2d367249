// hd_enc_fused.hip.h -- the token-encoder stack of the denoiser (model/encoder/model.py:160-180: n_encoder_layers ByteNet blocks of
// width d = 256, hidden width 128, kernel 7, dilations 1 .. 32, per-chain weights) as ONE kernel on the split-precision route:
// one workgroup per (sequence, chain), the residual stream x [<= 160 rows, 256] register-resident for the whole stack.
//
// Why (VERDICT r3 "Next" #3 iii / #5): as separate launches the stack is 36 narrow GEMMs (N = 128 / 256) per denoiser step that
// neither fill the chip at small batches nor amortise their HBM round trips at large ones (8 % of a 256-row step, 20 % of a
// 8-row step).  Here nothing of the stack touches HBM but the token ids, the weight stream (L2 / Infinity-Cache resident: every
// workgroup of a chain streams the same 4.3 MB) and the final [rows, 256] output.
//
// Structure (4 waves, one per SIMD, up to 512 VGPRs each; 153 KB of LDS: one workgroup per CU):
//   x      two accumulator sets of 5 row tiles x 32 columns per wave: columns 32 w + 128 j + (lane & 31), j = 0, 1 -- the MFMA
//          accumulator layout, so that the last projection of a block accumulates straight onto x (residual add for free) and
//          the dropout runs elementwise on registers
//   ACT    the current 128-column GEMM operand, act(LN(.)) in split form: 160 rows (+ one zero row) of 512 B = per 32 columns
//          [32 high parts | 32 low parts], 16-byte chunks XOR-swizzled by (row & 15) over each half row (conflict-free ds_read_b128
//          A fragments; a padded row stride of 528 B instead -- no swizzle, immediate offsets -- gave wrong operands, cause not found).  The dilated taps read it at shifted rows; rows outside the chain read the zero row -- a workgroup holds
//          exactly one chain, so there is no neighbouring sequence to mask.
//   GEMMs  every product is [160 x 128] += ACT[160 x 128 (shifted)] W[128 x 128]: PFF1 in two K chunks (columns 0-127 / 128-255 of
//          act(LN1(x))), the 7 taps, PFF3 in two N halves.  W k tiles (16 KB: the tile images gemm_x3_k consumes) stream through
//          three LDS stages by DMA (buffer_load ... lds), two in flight.
//   LN     accumulators -> per-wave LDS transpose -> float4 rows: per-wave (mean, M2) of the wave's 32 (64) columns, merged across
//          the four waves (Chan et al.); a second transpose pass normalises, activates, splits and writes ACT.
// Arithmetic is the unfused route's (three fp16 MFMAs per product in the same order, fp32 accumulation, erfc GELU, the same
// counter-based dropout bits); only the association of the LayerNorm statistics and of the residual add differs (~1e-7).
#pragma once
#include "hd_kernels.hip.h"

namespace hd {

constexpr int ENC_MAX_LAYERS = 8;
struct EncLayerW {
    const uint16_t *w1, *wc, *w3;                 // this chain's split weight images (tiles [n tile][k tile] of hi[128][32] + lo[128][32])
    float s1, sc, s3;                             // 2^-shift of each (GemmP::acc_scale)
    int dil;
    const float *b1, *bc, *b3;                    // biases [128], [128], [256]
    const float *g1, *e1, *g2, *e2, *g3, *e3;     // LayerNorm gamma / beta: [256], [128], [128]
};
struct EncStackP {
    EncLayerW lw[2][ENC_MAX_LAYERS];              // [chain segment][layer]
    int nlayers, act;
    const int32_t* tokens;                        // [B, L]
    const float* emb;                             // [n_tokens, 256]
    const float* extra; int lde;                  // addend behind the last block's dropout (position + chain embedding), may be null
    float* out; int ldo;                          // [rows, ldo]: columns 0 .. 255 of the feature rows
    Segs sg;
    const RunState* rs;
    int drop_mode; uint32_t drop_thresh; float drop_scale;
    const uint8_t* drop_mask; long mask_layer_stride;    // DROP_INJECT: [layer][B, L, 256] keep-masks
    int abl;                                      // probes only (HUDIFF_ENC_ABL): 1 = no MFMAs, 2 = no weight DMA, 4 = no LayerNorm / activation passes
};

constexpr int ENC_ROWS = 160, ENC_RT = 5, ENC_ACT_ROW = 512, ENC_ZERO_ROW = ENC_ROWS;
constexpr int ENC_ACT_BYTES = (ENC_ROWS + 1) * ENC_ACT_ROW;
constexpr int ENC_NSTG = 3, ENC_WST_BYTES = ENC_NSTG * X3_TILE_BYTES;
constexpr int ENC_ES = 36, ENC_STG_BYTES = 4 * 32 * ENC_ES * 4;
constexpr int ENC_RED_BYTES = 4 * ENC_ROWS * 8, ENC_STAT_BYTES = ENC_ROWS * 8;
constexpr int ENC_LDS = ENC_ACT_BYTES + ENC_WST_BYTES + ENC_STG_BYTES + ENC_RED_BYTES + ENC_STAT_BYTES;
static_assert(lds_fill_ok(ENC_LDS, 256), "LDS co-residency rule");

template <int ACT_KIND>
__global__ void __launch_bounds__(256, 1) enc_stack_x3_k(const EncStackP p) {
    __shared__ __attribute__((aligned(1024))) char lds[ENC_LDS];
    char* WST = lds;                                      // first: the LDS-DMA destinations stay 1 KiB-aligned (a 16-byte-aligned stage base
    char* ACT = lds + ENC_WST_BYTES;                      // produced wrong tiles now and then)
    float* STG = reinterpret_cast<float*>(lds + ENC_ACT_BYTES + ENC_WST_BYTES);
    float2* RED = reinterpret_cast<float2*>(lds + ENC_ACT_BYTES + ENC_WST_BYTES + ENC_STG_BYTES);
    float2* STAT = RED + 4 * ENC_ROWS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, seg = blockIdx.y;
    const int R = p.sg.len[seg];                          // rows (slots) of this chain
    const long rbase = (long)p.sg.base[seg] + (long)b * R;  // first activation row of this (sequence, chain)
    const int slot0 = p.sg.off[seg];
    int khalf = lane >> 5, lcol = lane & 31;              // (not const: made opaque once per layer, see the layer loop)
    float* stage = STG + wave * (32 * ENC_ES);            // private to this wave
    // float4-row view of a staged 32 x 32 tile: 8 lanes per row, 8 rows per wave instruction
    int e_c4 = (lane & 7) * 4, e_r = lane >> 3;
    float vmax = 0.f;                                     // range guard of every split this block writes

    // the zero row of ACT
    for (int i = tid; i < ENC_ACT_ROW / 4; i += 256) reinterpret_cast<float*>(ACT + ENC_ZERO_ROW * ENC_ACT_ROW)[i] = 0.f;

    // ---- x = emb[token]: straight into the accumulator layout (row = 32 i + (r & 3) + 8 (r >> 2) + 4 khalf, column 32 w + 128 j + lcol)
    f32x16 x[2][ENC_RT];
    {
        // tokens of the chain -> LDS (RED is free here), then per row tile 16 token reads and 32 embedding loads in flight together
        // (buffer loads with 32-bit offsets: 160 flat loads with 64-bit addresses, all hoisted, were the register peak of the kernel;
        //  a token fetched from global right in front of each embedding load serialised 80 round trips)
        int* tok = reinterpret_cast<int*>(RED);
        for (int i = tid; i < ENC_ROWS; i += 256) tok[i] = i < R ? p.tokens[(long)b * p.sg.L + slot0 + i] : 0;
        __syncthreads();
        const __amdgpu_buffer_rsrc_t e_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.emb), 0, 0x7fffffff, 0x00020000);
        const int ecol = (32 * wave + lcol) * 4;
#pragma unroll
        for (int i = 0; i < ENC_RT; ++i) {
            int t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = tok[32 * i + (r & 3) + 8 * (r >> 2) + 4 * khalf];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * khalf;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float v = __builtin_bit_cast(float, (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(e_rs, t[r] * 1024 + ecol + 512 * j, 0, 0));
                    x[j][i][r] = row < R ? v : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- building blocks -------------------------------------------------------------------------------------------------
    // accumulator tile -> this wave's staging slice (row-major 32 x 32, row stride ENC_ES)
    auto stage_tile = [&](const f32x16& a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * khalf) * ENC_ES + lcol] = a[r];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    };
    auto staged_row4 = [&](int it) -> f32x4 { return *reinterpret_cast<const f32x4*>(stage + (it * 8 + e_r) * ENC_ES + e_c4); };
    // (mean, M2) of a float4's 32-column row slice across its 8 lanes
    auto slice_stat = [&](const f32x4 v, float& pm, float& pq) {
        float ps = (v[0] + v[1]) + (v[2] + v[3]);
        ps = group_sum<8>(ps);
        pm = ps * (1.0f / 32.0f);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float d = v[c] - pm; q += d * d; }
        pq = group_sum<8>(q);
    };
    // merge the four waves' slice partials (n columns each) of every row -> STAT (mean, rstd)
    auto merge_stats = [&](int n_per_wave) {
        __syncthreads();
        if (tid < ENC_ROWS) {
            float mean = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) mean += RED[w * ENC_ROWS + tid].x;
            mean *= 0.25f;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const float2 pr = RED[w * ENC_ROWS + tid]; const float d = pr.x - mean; m2 += pr.y + (float)n_per_wave * d * d; }
            STAT[tid] = make_float2(mean, 1.0f / sqrtf(m2 / (float)(4 * n_per_wave) + 1e-5f));
        }
        __syncthreads();
    };
    // act(LN(v)) of four columns c0 .. c0 + 3 (of a 128-column operand) of row `row` -> ACT in split form
    auto write_act4 = [&](int row, int c0, const f32x4 v, const float2 st, const f32x4 g, const f32x4 be) {
        f32x4 w;
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = act_f((v[c] - st.x) * st.y * g[c] + be[c], ACT_KIND);
        hd_f16x4 hh, ll;
        split4(w, hh, ll);
        vmax = absmax4(vmax, w);
        const int c32 = (c0 >> 5) * 8 + ((c0 & 31) >> 3);                     // chunk of the high parts; low parts: + 4
        char* rowp = ACT + row * ENC_ACT_ROW + (c0 & 7) * 2;
        *reinterpret_cast<hd_f16x4*>(rowp + ((c32 ^ (row & 15)) << 4)) = hh;
        *reinterpret_cast<hd_f16x4*>(rowp + (((c32 + 4) ^ (row & 15)) << 4)) = ll;
    };

    // [160 x 32-column slice] += ACT[rows (+ shift)][128] x W tiles: `ntiles` 16-KB weight tiles from `wimg`, tile t multiplies the
    // operand's columns 32 (t & 3) .. with the rows shifted by ((t >> 2) - half) * dil when `taps` (the dilated convolution)
    typedef __attribute__((address_space(3))) void* lds_vp;
    int fg = lane >> 5, fsw = (lane >> 2) & 3;
    auto gemm128 = [&](f32x16 (&acc)[ENC_RT], const uint16_t* wimg, int ntiles, bool taps, int dil) {
        const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wimg), 0, ntiles * X3_TILE_BYTES, 0x00020000);
        auto dma = [&](int t) {
            char* dst = WST + (t % ENC_NSTG) * X3_TILE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i)                   // 16 pieces of 1 KiB, four per wave: the image is copied linearly
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_vp)(dst + (4 * i + wave) * 1024), 16, (4 * i + wave) * 1024 + lane * 16,
                                                         t * X3_TILE_BYTES, 0, 0);
        };
        constexpr int KEEP = (ENC_NSTG - 2) * 4;
        constexpr int WAIT_STEADY = (KEEP & 0xF) | ((KEEP >> 4) << 14) | 0x0F70, WAIT_ALL = 0x0F70;
#pragma unroll
        for (int t = 0; t < ENC_NSTG - 1; ++t)
            if (t < ntiles) dma(t);
        if (ENC_NSTG - 1 <= ntiles) __builtin_amdgcn_s_waitcnt(WAIT_STEADY); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
        lds_barrier();
        int asw[ENC_RT];
        int arow[ENC_RT];                                 // byte offset of this lane's operand row per row tile (+ its k octet); an int, not a
                                                          // pointer: a pointer array decays to FLAT loads, which count on vmcnt beside the DMA
#pragma unroll 4
        for (int t = 0; t < ntiles; ++t) {                // (ntiles is a multiple of 4: the k group of a tile is a compile-time offset)
            const bool more = t + ENC_NSTG - 1 < ntiles;
            if (more && !(p.abl & 2)) dma(t + ENC_NSTG - 1);   // into the stage tile t - 1 was read from (everybody is past that barrier)
            if (!taps ? t == 0 : (t & 3) == 0) {          // a new tap: row shift, zero row outside the chain
                const int shift = taps ? ((t >> 2) - 3) * dil : 0;
#pragma unroll
                for (int i = 0; i < ENC_RT; ++i) {
                    const int rr = 32 * i + lcol + shift;
                    const int rc = (rr >= 0 && rr < R) ? rr : ENC_ZERO_ROW;
                    // chunk of a fragment = (k group, k step, hi / lo: compile-time bits 1 .. 4) | fg (bit 0), swizzled by row & 15:
                    // ((const | fg) ^ sw) << 4 == (const << 4) ^ ((fg ^ sw) << 4) -- one xor-add per read with the lane part hoisted
                    arow[i] = rc * ENC_ACT_ROW; asw[i] = (fg ^ (rc & 15)) << 4;
                }
            }
            if (p.abl & 1) { if (more) __builtin_amdgcn_s_waitcnt(WAIT_STEADY); else __builtin_amdgcn_s_waitcnt(WAIT_ALL); lds_barrier(); continue; }
            const char* Wt = WST + (t % ENC_NSTG) * X3_TILE_BYTES + (32 * wave + lcol) * 64;
            const int kg8 = (t & 3) * 8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(Wt + (((2 * ks + fg) ^ fsw) << 4));
                const f16x8 bl = *reinterpret_cast<const f16x8*>(Wt + X3_TILE_BYTES / 2 + (((2 * ks + fg) ^ fsw) << 4));
                f16x8 ah[ENC_RT], al[ENC_RT];
#pragma unroll
                for (int i = 0; i < ENC_RT; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(ACT + arow[i] + ((((kg8 + 2 * ks) << 4)) ^ asw[i]));
                    al[i] = *reinterpret_cast<const f16x8*>(ACT + arow[i] + ((((kg8 + 2 * ks + 4) << 4)) ^ asw[i]));
                }
#pragma unroll
                for (int i = 0; i < ENC_RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < ENC_RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < ENC_RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i], 0, 0, 0);
            }
            if (more) __builtin_amdgcn_s_waitcnt(WAIT_STEADY); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
            lds_barrier();
        }
    };

    // hidden activation h (accumulators of this wave's 32 columns, already scale * acc + bias) -> ACT = act(LN(h)) (128 columns)
    auto hidden_to_act = [&](f32x16 (&h)[ENC_RT], const float* gamma, const float* beta) {
        if (p.abl & 4) { __syncthreads(); return; }
#pragma unroll
        for (int i = 0; i < ENC_RT; ++i) {
            stage_tile(h[i]);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float pm, pq;
                slice_stat(staged_row4(it), pm, pq);
                if ((lane & 7) == 0) RED[wave * ENC_ROWS + 32 * i + it * 8 + e_r] = make_float2(pm, pq);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        merge_stats(32);
        const int c0 = 32 * wave + e_c4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
#pragma unroll
        for (int i = 0; i < ENC_RT; ++i) {
            stage_tile(h[i]);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = 32 * i + it * 8 + e_r;
                write_act4(row, c0, staged_row4(it), STAT[row], g, be);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    uint32_t k0 = 0, k1 = 0, grow = 0;
    for (int n = 0; n < p.nlayers; ++n) {
        const EncLayerW& lw = p.lw[seg][n];
        // Lane geometry made opaque once per layer: every LDS / staging address below is layer-invariant, and hipcc otherwise hoists
        // a few hundred of them out of this loop and parks them in scratch (168 spilled VGPRs); recomputing them costs a few adds.
        asm volatile("" : "+v"(khalf), "+v"(lcol), "+v"(e_c4), "+v"(e_r), "+v"(fg), "+v"(fsw));
        // ---- LayerNorm 1 statistics of x (256 columns: 64 per wave, two 32-column tiles merged in the lane) ---------------------
#pragma unroll
        for (int i = 0; i < ENC_RT; ++i) {
            float pm0[4], pq0[4];
            stage_tile(x[0][i]);
#pragma unroll
            for (int it = 0; it < 4; ++it) slice_stat(staged_row4(it), pm0[it], pq0[it]);
            __builtin_amdgcn_wave_barrier();
            stage_tile(x[1][i]);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float pm1, pq1;
                slice_stat(staged_row4(it), pm1, pq1);
                const float d = pm1 - pm0[it];
                if ((lane & 7) == 0) RED[wave * ENC_ROWS + 32 * i + it * 8 + e_r] = make_float2(0.5f * (pm0[it] + pm1), pq0[it] + pq1 + 16.0f * d * d);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        merge_stats(64);
        // ---- h1 = act(LN1(x)) W1 + b1, in two K chunks (operand columns 128 j ..) ---------------------------------------------
        f32x16 h[ENC_RT];
#pragma unroll
        for (int i = 0; i < ENC_RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c0 = 32 * wave + e_c4;                                   // column within the chunk
            const f32x4 g = *reinterpret_cast<const f32x4*>(lw.g1 + 128 * j + c0), be = *reinterpret_cast<const f32x4*>(lw.e1 + 128 * j + c0);
#pragma unroll
            for (int i = 0; i < ENC_RT; ++i) {
                stage_tile(x[j][i]);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = 32 * i + it * 8 + e_r;
                    write_act4(row, c0, staged_row4(it), STAT[row], g, be);
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            gemm128(h, lw.w1 + (long)(4 * j) * X3_TILE_HALFS, 4, false, 1);
        }
        {
            const float bv = lw.b1[32 * wave + lcol];
#pragma unroll
            for (int i = 0; i < ENC_RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[i][r] = h[i][r] * lw.s1 + bv;
        }
        hidden_to_act(h, lw.g2, lw.e2);
        // ---- h2 = conv7(act(LN2(h1))) + bc: 7 taps x 4 k tiles ------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < ENC_RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[i][r] = 0.f;
        gemm128(h, lw.wc, 28, true, lw.dil);
        {
            const float bv = lw.bc[32 * wave + lcol];
#pragma unroll
            for (int i = 0; i < ENC_RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[i][r] = h[i][r] * lw.sc + bv;
        }
        hidden_to_act(h, lw.g3, lw.e3);
        // ---- x = dropout(x + act(LN3(h2)) W3 + b3): the projection accumulates onto x (in the weights' power-of-two scale) -----
        if (p.drop_mode == DROP_GEN) {
            uint32_t o[4];
            philox4x32_10(0u, 0u, p.rs->step, (uint32_t)n, p.rs->seed_lo, p.rs->seed_hi, o);
            k0 = o[0]; k1 = o[1]; grow = p.rs->row0 + (uint32_t)b;
        }
        const float inv_s3 = 1.0f / lw.s3;                                     // exact: s3 is a power of two
        // (an opaque zero in the element indices below: they do not depend on the layer, and hipcc otherwise hoists all 160 hash
        //  inputs / mask addresses per lane out of the layer loop and parks them in scratch)
        int opq = 0;
        asm volatile("" : "+v"(opq));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < ENC_RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) x[j][i][r] *= inv_s3;
            gemm128(x[j], lw.w3 + (long)(4 * j) * X3_TILE_HALFS, 4, false, 1);   // n tile j of the [128 -> 256] image: 4 k tiles each
            const int col = 32 * wave + 128 * j + lcol;
            const float bv = lw.b3[col];
            const uint32_t rk = mix32(k0 ^ mix32(grow + k1));
#pragma unroll
            for (int i = 0; i < ENC_RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = x[j][i][r] * lw.s3 + bv;
                    const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * khalf + opq;
                    if (p.drop_mode == DROP_GEN) {
                        const uint32_t w = mix32(rk + (uint32_t)((slot0 + row) * 256 + col) * 0x9E3779B9U);
                        v = (w >= p.drop_thresh) ? v * p.drop_scale : 0.f;
                    } else if (p.drop_mode == DROP_INJECT) {
                        const bool keep = row < R && p.drop_mask[(long)n * p.mask_layer_stride + ((long)b * p.sg.L + slot0 + row) * 256 + col];
                        v = keep ? v * p.drop_scale : 0.f;
                    }
                    x[j][i][r] = row < R ? v : 0.f;
                }
        }
    }
    // ---- out[row, 0:256] = x (+ extra) --------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < ENC_RT; ++i) {
            stage_tile(x[j][i]);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = 32 * i + it * 8 + e_r;
                const int col = 32 * wave + 128 * j + e_c4;
                if (row < R) {
                    f32x4 v = staged_row4(it);
                    if (p.extra) v += *reinterpret_cast<const f32x4*>(p.extra + (rbase + row) * p.lde + col);
                    *reinterpret_cast<f32x4*>(p.out + (rbase + row) * p.ldo + col) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    raise_range_flag(p.rs, vmax);
}

}  // namespace hd
