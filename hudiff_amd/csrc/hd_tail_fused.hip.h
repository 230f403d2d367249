// hd_tail_fused.hip.h -- the pruned tail of a sampling step in fewer launches (round 4).
//
// After the last SelfAttBlock only the row each sequence visits at this step feeds the decoder (antibody_scripts/sample.py:508-513),
// so from "at = x + A1(x)" on the block is evaluated for that one row per sequence (hd_api.hip, pruned_tail): query projection,
// one-query attention against all keys, value side through the input rows, out-projection, feed-forward.  As separate launches that
// is twelve tiny kernels (gathers, row statistics, four 1-row GEMMs, attn_row_k, row_value_k, head_proj_k) of 5-70 us each: 0.25 ms of
// a single-sequence step, all of it launch and dependent-load latency.  fp32 throughout, every sum in a fixed order:
//   (one workgroup per sequence doing all of it, the draw included, was built in round 4, is correct and NOT faster -- 205 us at B = 8:
//    7.5 MB of weights and rows per sequence through one CU; the code lives in scripts/experiments/tail_fused_k.hip.h, outside the library)
//   tail_pw_k ... tail_ff2_k  the same phases as five launches sliced over heads / 128-column slices (below): 46 us at B = 8.  The default
//                           for lanes of at most 64 sequences (hd_api.hip, tail_form); larger lanes keep the separate launches.
//   reference: model/encoder/cross_attention.py:149-173 (AttLayer), :273-287 (SelfAttBlock), restricted to one query row.
#pragma once
#include "hd_kernels.hip.h"

namespace hd {

struct TailP {
    const float* AT; const float* Y; int D;              // at = x + A1(x) and the block input x, [rows, D]
    const float* QKV; int ldq; int A;                    // K of LN1(at) for every row at columns [A, 2A) (projected by the big GEMM)
    const float2* at_part; int at_pw; long at_rows;      // LayerNorm slice partials of `at` (rstd of every key row)
    const float* wqkv; const float* bqkv;                // second attention, LayerNorm 1 folded: [D, 3A] (q | k | v), [3A]
    const float* wo; const float* bo;                    // [A, D], [D]
    const float* wf1; const float* bf1; int Fd;          // LayerNorm 2 folded: [D, Fd], [Fd]
    const float* wf2; const float* bf2;                  // [Fd, D], [D]
    const float* rope_cos; const float* rope_sin;
    const int32_t* order; const int32_t* T; int Tmax;
    const RunState* rs;
    float* Xc;                                           // [B, D]: the block's output row of every sequence
    int nhead;
    Segs sg;
    // the draw of the step (sample_row, as sample_step_k)
    HeadW head; int32_t* tokens; const float* q_noise; int q_rows, q_off;
    // intermediates of the five-launch form (tail_pw_k ... tail_ff2_k)
    float* PW;                                           // [B, nhead, 320]: p_j rstd_j
    float* OP;                                           // [B, D / 128, A]: the value projection, partial over the column slices of `at`
    float* ATc;                                          // [B, D]: at + A2(LN1(at)) of the visited row
    float* F1c;                                          // [B, Fd]
};

constexpr int TAIL_MAXD = 768, TAIL_MAXA = 512, TAIL_MAXL = 320, TAIL_MAXF = 256;      // widths / lengths the sliced kernels are written for


// ------------------------------------------------------------------------------------------------------------------------------
// The same tail as FIVE launches whose grids are (sequences x slices): for a handful of sequences one workgroup per sequence leaves
// the chip idle and its 7.5 MB of weights and rows stream through ONE CU's L2 port (205 us measured, B = 8); sliced, every launch
// has 4-8 workgroups per sequence, every thread has all its loads in flight at once (one round trip per phase), and the K ranges of
// the vector-matrix products are split over the threads of a workgroup and combined in a fixed order.
//   tail_pw_k  (b, head)     q_h = LN1(at_c) Wq[:, h], RoPE, scores against the L keys, softmax -> PW[b, h, :] = p_j rstd_j
//   tail_val_k (b, slice s)  yv[h][c] = sum_j PW[b, h, j] at[row j][c] for the 128 columns c of slice s;  OP[b, s, :] = yv Wv[slice s, :]
//   tail_out_k (b, slice s)  o = sum_s OP[b, s, :] + t_v;  ATc[b, c] = at_c[c] + o Wo[:, c] + bo[c]     for c in slice s
//   tail_ff1_k (b, slice f)  F1c[b, n] = relu(LN2(ATc[b]) W1[:, n] + b1[n])                                for the 64 columns n of slice f
//   tail_ff2_k (b, slice s)  Xc[b, c] = x_c[c] + F1c[b] W2[:, c] + b2[c]
// then sample_step_k on Xc.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int TC_THREADS = 1024, TC_SLICE = 128;

// (mean, rstd) of v[0 .. C) (LDS), two-pass like row_stats_k; scratch: 64 floats of LDS
__device__ __forceinline__ float2 tail_row_stat(const float* v, int C, float* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (int c = tid; c < C; c += TC_THREADS) s += v[c];
    s = wave_sum(s);
    if (lane == 0) scratch[wave] = s;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < TC_THREADS / 64; ++w) tot += scratch[w];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.f;
    for (int c = tid; c < C; c += TC_THREADS) { const float d = v[c] - mean; q += d * d; }
    q = wave_sum(q);
    if (lane == 0) scratch[wave] = q;
    __syncthreads();
    float tq = 0.f;
    for (int w = 0; w < TC_THREADS / 64; ++w) tq += scratch[w];
    __syncthreads();
    return make_float2(mean, 1.0f / sqrtf(tq / (float)C + 1e-5f));
}

// out[n] = sum_k in[k] W[k * ldw + n], n < N (multiple of 4): thread = (column quad, K part), KN rows per thread, all KN loads in flight.
// K == (TC_THREADS / (N / 4)) * KN.  head_stride: the input vector of column n starts at in + (n / 64) * head_stride.
template <int KN>
__device__ __forceinline__ void tc_gemv(const float* in, int head_stride, const float* __restrict__ W, int ldw, int N, float* red, float* out) {
    const int tid = threadIdx.x, quads = N >> 2, quad = tid % quads, part = tid / quads, nparts = TC_THREADS / quads;
    const float* w = W + (long)part * KN * ldw + 4 * quad;
    const float* x = in + ((4 * quad) >> 6) * head_stride + part * KN;
    f32x4 wv[KN];
#pragma unroll
    for (int u = 0; u < KN; ++u) wv[u] = *reinterpret_cast<const f32x4*>(w + (long)u * ldw);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KN; u += 2) {
        a0 = __builtin_elementwise_fma(f32x4{x[u], x[u], x[u], x[u]}, wv[u], a0);
        a1 = __builtin_elementwise_fma(f32x4{x[u + 1], x[u + 1], x[u + 1], x[u + 1]}, wv[u + 1], a1);
    }
    *reinterpret_cast<f32x4*>(red + part * N + 4 * quad) = a0 + a1;
    __syncthreads();
    if (tid < N) {
        float s0 = 0.f, s1 = 0.f;
        for (int pp = 0; pp < nparts; pp += 2) { s0 += red[pp * N + tid]; s1 += red[(pp + 1) * N + tid]; }
        out[tid] = s0 + s1;
    }
    __syncthreads();
}

__device__ __forceinline__ bool tc_row(const TailP& p, int b, uint32_t& t, int& slot, long& row) {
    t = p.rs->step;
    if ((int)t >= p.T[b]) return false;                  // finished sequence: nothing reads its row (as sample_step_k)
    slot = p.order[(long)b * p.Tmax + t];
    row = p.sg.row(b, slot);
    return true;
}

template <int D>
__global__ void __launch_bounds__(TC_THREADS, 1) tail_pw_k(const TailP p) {
    static_assert(lds_fill_ok((D + 64 * 64 + 3 * 64) * 4, TC_THREADS), "LDS co-residency rule");
    __shared__ __attribute__((aligned(16))) float at_c[D];
    __shared__ __attribute__((aligned(16))) float red[64 * 64];
    __shared__ float qraw[64], qs[64], scratch[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, h = blockIdx.y, A = p.A, L = p.sg.L;
    uint32_t t; int slot; long row;
    if (!tc_row(p, b, t, slot, row)) return;
    // wave w < 5 owns the keys 64 w + lane: their rstd (slice partials of `at`) is asked for first
    const int key = 64 * wave + lane;
    const bool has_key = wave < 5 && key < L;
    float rstd_key = 0.f;
    const long krow = has_key ? p.sg.row(b, key) : 0;
    if (has_key) rstd_key = merge_row_stat(p.at_part, p.at_pw, p.at_rows, D, krow).y;
    if (tid < D) at_c[tid] = p.AT[row * D + tid];
    __syncthreads();
    const float2 st1 = tail_row_stat(at_c, D, scratch);
    tc_gemv<D / 64>(at_c, 0, p.wqkv + h * ATT_HD, 3 * A, ATT_HD, red, qraw);
    f32x4 kv[16];                                        // (held across the product they would not fit in 128 registers)
    if (has_key) {
        const float* kp = p.QKV + krow * p.ldq + A + h * ATT_HD;
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) kv[k4] = *reinterpret_cast<const f32x4*>(kp + 4 * k4);
    }
    if (tid < 32) {                                      // RoPE on the complex pair (2k, 2k + 1); log2 domain like attn_k
        const int n = h * ATT_HD + 2 * tid;
        const float xr = qraw[2 * tid] * st1.y + p.bqkv[n], xi = qraw[2 * tid + 1] * st1.y + p.bqkv[n + 1];
        const float c = p.rope_cos[slot * 32 + tid], s = p.rope_sin[slot * 32 + tid];
        constexpr float QS = 0.125f * 1.44269504088896340736f;
        qs[2 * tid] = (xr * c - xi * s) * QS;
        qs[2 * tid + 1] = (xr * s + xi * c) * QS;
    }
    __syncthreads();
    float sc = -INFINITY;
    if (has_key) {
        float a = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
            const float2 cs = *reinterpret_cast<const float2*>(p.rope_cos + key * 32 + 2 * k4);
            const float2 sn = *reinterpret_cast<const float2*>(p.rope_sin + key * 32 + 2 * k4);
            a += (kv[k4][0] * cs.x - kv[k4][1] * sn.x) * qs[4 * k4] + (kv[k4][0] * sn.x + kv[k4][1] * cs.x) * qs[4 * k4 + 1];
            a += (kv[k4][2] * cs.y - kv[k4][3] * sn.y) * qs[4 * k4 + 2] + (kv[k4][2] * sn.y + kv[k4][3] * cs.y) * qs[4 * k4 + 3];
        }
        sc = a;
    }
    const float wmx = wave_max(sc);
    if (lane == 0) scratch[wave] = wmx;
    __syncthreads();
    const float mx = fmaxf(fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3])), scratch[4]);
    const float e = has_key ? __builtin_amdgcn_exp2f(sc - mx) : 0.f;
    const float wsum = wave_sum(e);
    __syncthreads();
    if (lane == 0) scratch[wave] = wsum;
    __syncthreads();
    const float sum = (((scratch[0] + scratch[1]) + scratch[2]) + scratch[3]) + scratch[4];
    if (has_key) p.PW[((long)b * p.nhead + h) * 320 + key] = e * (1.0f / sum) * rstd_key;
}

template <int D>
__global__ void __launch_bounds__(TC_THREADS, 1) tail_val_k(const TailP p) {
    static_assert(lds_fill_ok((TAIL_MAXL * RV_MAX_HEADS + 16 * 4 * TC_SLICE + RV_MAX_HEADS * TC_SLICE) * 4, TC_THREADS), "LDS co-residency rule");
    __shared__ __attribute__((aligned(16))) float ps[TAIL_MAXL * RV_MAX_HEADS];             // [key][head]
    __shared__ __attribute__((aligned(16))) float part[16 * 4 * TC_SLICE];                  // [wave][4 heads][128]; later the GEMV's partial sums
    __shared__ __attribute__((aligned(16))) float yv[RV_MAX_HEADS * TC_SLICE];              // [head][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, s = blockIdx.y, A = p.A, L = p.sg.L, nhead = p.nhead;
    uint32_t t; int slot; long row;
    if (!tc_row(p, b, t, slot, row)) return;
    for (int i = tid; i < L * RV_MAX_HEADS; i += TC_THREADS) {
        const int key = i / RV_MAX_HEADS, h = i % RV_MAX_HEADS;
        ps[i] = h < nhead ? p.PW[((long)b * nhead + h) * 320 + key] : 0.f;
    }
    // thread = (column quad cq of the slice, row group g of 32): rows g, g + 32, ... of the sequence, at most ten, all in flight
    const int cq = lane & 31, g = 2 * wave + (lane >> 5);
    constexpr int RMAX = (TAIL_MAXL + 31) / 32;
    f32x4 xv[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int j = g + 32 * i;
        xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < L) xv[i] = *reinterpret_cast<const f32x4*>(p.AT + (long)p.sg.row(b, j) * D + s * TC_SLICE + 4 * cq);
    }
    __syncthreads();
    f32x4 acc[RV_MAX_HEADS];
#pragma unroll
    for (int h = 0; h < RV_MAX_HEADS; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int j = g + 32 * i;
        if (j < L) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(ps + j * RV_MAX_HEADS), p1 = *reinterpret_cast<const f32x4*>(ps + j * RV_MAX_HEADS + 4);
#pragma unroll
            for (int h = 0; h < RV_MAX_HEADS; ++h) {
                const float pw = h < 4 ? p0[h] : p1[h - 4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[h][c] = __builtin_fmaf(pw, xv[i][c], acc[h][c]);
            }
        }
    }
    // the 32 row groups: the two of a wave by a lane exchange, the 16 waves through LDS in wave order (four heads per round)
#pragma unroll
    for (int h = 0; h < RV_MAX_HEADS; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[h][c] += __shfl_xor(acc[h][c], 32);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (lane < 32) {
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) *reinterpret_cast<f32x4*>(part + (wave * 4 + hh) * TC_SLICE + 4 * cq) = acc[4 * r + hh];
        }
        __syncthreads();
        if (tid < 4 * TC_SLICE) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) a += part[w * 4 * TC_SLICE + tid];
            yv[4 * r * TC_SLICE + tid] = a;
        }
        __syncthreads();
    }
    // OP[b, s, n] = sum_{c in slice} yv[n / 64][c] Wv[s * 128 + c][n]
    tc_gemv<16>(yv, TC_SLICE, p.wqkv + 2 * A + (long)s * TC_SLICE * 3 * A, 3 * A, A, part, part + 8 * TAIL_MAXA);
    if (tid < A) p.OP[((long)b * (D / TC_SLICE) + s) * A + tid] = part[8 * TAIL_MAXA + tid];
}

template <int D>
__global__ void __launch_bounds__(TC_THREADS, 1) tail_out_k(const TailP p) {
    static_assert(lds_fill_ok((TAIL_MAXA + 33 * TC_SLICE) * 4, TC_THREADS), "LDS co-residency rule");
    __shared__ __attribute__((aligned(16))) float o[TAIL_MAXA];
    __shared__ __attribute__((aligned(16))) float red[32 * TC_SLICE];
    __shared__ float outc[TC_SLICE];
    const int tid = threadIdx.x, b = blockIdx.x, s = blockIdx.y, A = p.A;
    uint32_t t; int slot; long row;
    if (!tc_row(p, b, t, slot, row)) return;
    if (tid < A) {
        float a = 0.f;
#pragma unroll
        for (int ss = 0; ss < D / TC_SLICE; ++ss) a += p.OP[((long)b * (D / TC_SLICE) + ss) * A + tid];
        o[tid] = a + p.bqkv[2 * A + tid];
    }
    __syncthreads();
    tc_gemv<16>(o, 0, p.wo + s * TC_SLICE, D, TC_SLICE, red, outc);
    if (tid < TC_SLICE) {
        const int c = s * TC_SLICE + tid;
        p.ATc[(long)b * D + c] = p.AT[row * D + c] + outc[tid] + p.bo[c];
    }
}

template <int D>
__global__ void __launch_bounds__(TC_THREADS, 1) tail_ff1_k(const TailP p) {
    static_assert(lds_fill_ok((D + 64 * 64 + 2 * 64) * 4, TC_THREADS), "LDS co-residency rule");
    __shared__ __attribute__((aligned(16))) float at_c[D];
    __shared__ __attribute__((aligned(16))) float red[64 * 64];
    __shared__ float f1[64], scratch[64];
    const int tid = threadIdx.x, b = blockIdx.x, f = blockIdx.y;
    uint32_t t; int slot; long row;
    if (!tc_row(p, b, t, slot, row)) return;
    if (tid < D) at_c[tid] = p.ATc[(long)b * D + tid];
    __syncthreads();
    const float2 st2 = tail_row_stat(at_c, D, scratch);
    tc_gemv<D / 64>(at_c, 0, p.wf1 + f * 64, p.Fd, 64, red, f1);
    if (tid < 64) p.F1c[(long)b * p.Fd + f * 64 + tid] = fmaxf(f1[tid] * st2.y + p.bf1[f * 64 + tid], 0.f);
}

template <int D>
__global__ void __launch_bounds__(TC_THREADS, 1) tail_ff2_k(const TailP p) {
    static_assert(lds_fill_ok((TAIL_MAXF + 33 * TC_SLICE) * 4, TC_THREADS), "LDS co-residency rule");
    __shared__ __attribute__((aligned(16))) float f1[TAIL_MAXF];
    __shared__ __attribute__((aligned(16))) float red[32 * TC_SLICE];
    __shared__ float outc[TC_SLICE];
    const int tid = threadIdx.x, b = blockIdx.x, s = blockIdx.y;
    uint32_t t; int slot; long row;
    if (!tc_row(p, b, t, slot, row)) return;
    if (tid < p.Fd) f1[tid] = p.F1c[(long)b * p.Fd + tid];
    __syncthreads();
    tc_gemv<8>(f1, 0, p.wf2 + s * TC_SLICE, D, TC_SLICE, red, outc);       // Fd = 256: 32 K parts of 8
    if (tid < TC_SLICE) {
        const int c = s * TC_SLICE + tid;
        p.Xc[(long)b * D + c] = p.Y[row * D + c] + outc[tid] + p.bf2[c];
    }
}

}  // namespace hd
