// tail_fused_k -- EXPERIMENT, not part of libhudiff_hip.so (moved out of hudiff_amd/csrc in round 5).
// The pruned tail of a sampling step and the draw as ONE workgroup of 1024 threads per sequence, every intermediate in LDS.
// Correct (tokens identical to the separate launches on every sample compared in round 4) and NOT faster: 205 us at B = 8 against
// 246 us for the twelve separate launches and 46 us for the five sliced launches that ship (hd_tail_fused.hip.h); its five
// vector-matrix products pull 7.5 MB of weights and rows through ONE CU with too few bytes in flight.
// To rebuild it: include this file behind hd_tail_fused.hip.h in hd_api.hip and launch  tail_fused_k<<<B, TAIL_THREADS>>>(TailP).
#pragma once
#include "../../hudiff_amd/csrc/hd_tail_fused.hip.h"

namespace hd {

constexpr int TAIL_THREADS = 1024;
static_assert(TAIL_THREADS == TC_THREADS, "tail_row_stat (hd_tail_fused.hip.h) is written for TC_THREADS");
constexpr int TAIL_RED = 2048;                           // partial sums of a GEMV: (K parts) x N = 1024 threads x 2 columns
constexpr int TAIL_LDS_FLOATS = 2 * TAIL_MAXD /*at_c, x_c*/ + TAIL_MAXA /*q | o*/ + RV_MAX_HEADS * TAIL_MAXL /*p*/ + RV_MAX_HEADS * TAIL_MAXD /*yv*/ +
                                TAIL_MAXF /*f1*/ + TAIL_RED + 64 /*reductions*/;
static_assert(lds_fill_ok(TAIL_LDS_FLOATS * 4, TAIL_THREADS), "LDS co-residency rule");

// out[n] = sum_k in[k] W[k * ldw + n] (+ bias[n]) for n < N: column pairs x K parts over the block's threads; `in` and `out` in LDS.
// head_in: the input vector of column n is in + (n / 64) * head_stride (value projection per head), else 0.
__device__ __forceinline__ void tail_gemv(const float* in, int head_stride, const float* __restrict__ W, int ldw, int K, int N,
                                          float* red, float* out) {
    const int tid = threadIdx.x;
    const int pairs = N >> 1;
    const int kparts = TAIL_THREADS / pairs;             // 768 -> 2 (768 threads), 512 -> 4, 256 -> 8
    const int pair = tid % pairs, part = tid / pairs;
    if (part < kparts) {
        const int kq = (K + kparts - 1) / kparts, k0 = part * kq < K ? part * kq : K, kn = (k0 + kq < K ? k0 + kq : K) - k0;
        const float* x = in + ((2 * pair) >> 6) * head_stride + k0;
        const float* w = W + (long)k0 * ldw + 2 * pair;
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
        int k = 0;
        for (; k + 7 < kn; k += 8) {                     // eight weight rows in flight per step
            f32x2 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const f32x2*>(w + (long)(k + u) * ldw);
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                a0 = __builtin_elementwise_fma(f32x2{x[k + u], x[k + u]}, wv[u], a0);
                a1 = __builtin_elementwise_fma(f32x2{x[k + u + 1], x[k + u + 1]}, wv[u + 1], a1);
            }
        }
        for (; k < kn; ++k) a0 = __builtin_elementwise_fma(f32x2{x[k], x[k]}, *reinterpret_cast<const f32x2*>(w + (long)k * ldw), a0);
        *reinterpret_cast<f32x2*>(red + part * N + 2 * pair) = a0 + a1;
    }
    __syncthreads();
    for (int n = tid; n < N; n += TAIL_THREADS) {
        float s = 0.f;
        for (int pp = 0; pp < kparts; ++pp) s += red[pp * N + n];
        out[n] = s;
    }
    __syncthreads();
}


__global__ void __launch_bounds__(TAIL_THREADS, 1) tail_fused_k(const TailP p) {
    __shared__ __attribute__((aligned(16))) float lds[TAIL_LDS_FLOATS];
    float* at_c = lds;                                   // [D]
    float* x_c = at_c + TAIL_MAXD;                       // [D]
    float* qo = x_c + TAIL_MAXD;                         // [A]: the rotated query, later the attention output o
    float* ps = qo + TAIL_MAXA;                          // [320][8]: p_j rstd_j per head
    float* yv = ps + RV_MAX_HEADS * TAIL_MAXL;           // [nhead][D]: probability-weighted input rows
    float* f1 = yv + RV_MAX_HEADS * TAIL_MAXD;           // [Fd]
    float* red = f1 + TAIL_MAXF;                         // [K parts][N] partial sums of the GEMVs
    float* scratch = red + TAIL_RED;                     // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, D = p.D, A = p.A, L = p.sg.L, nhead = p.nhead;
    const uint32_t t = p.rs->step;
    if ((int)t >= p.T[b]) return;                        // finished sequence: nothing reads its row (as sample_step_k)
    const int slot = p.order[(long)b * p.Tmax + t];
    const long row = p.sg.row(b, slot);

    // ---- the visited rows of `at` and of the block input ----------------------------------------------------------------------
    for (int c = tid; c < D; c += TAIL_THREADS) { at_c[c] = p.AT[row * D + c]; x_c[c] = p.Y[row * D + c]; }
    __syncthreads();
    // ---- q = LN1(at_c) Wq + bq (LayerNorm folded into centred weight columns: rstd (at_c W'') + b), rotated and pre-scaled ------
    const float2 st1 = tail_row_stat(at_c, D, scratch);
    tail_gemv(at_c, 0, p.wqkv, 3 * A, D, A, red, qo);
    if (tid < A / 2) {                                   // RoPE on the complex pair (2k, 2k + 1) of its head; log2 domain like attn_k
        const int k = tid & 31;
        const float xr = qo[2 * tid] * st1.y + p.bqkv[2 * tid], xi = qo[2 * tid + 1] * st1.y + p.bqkv[2 * tid + 1];
        const float c = p.rope_cos[slot * 32 + k], s = p.rope_sin[slot * 32 + k];
        constexpr float QS = 0.125f * 1.44269504088896340736f;
        qo[2 * tid] = (xr * c - xi * s) * QS;
        qo[2 * tid + 1] = (xr * s + xi * c) * QS;
    }
    __syncthreads();
    // ---- one query per head against all L keys: two waves per head, lane = key; p_j rstd_j -> ps ------------------------------
    {
        const int h = wave >> 1, half = wave & 1;
        float sc[3];
        float mx = -INFINITY;
        if (h < nhead) {
            const float* qh = qo + h * ATT_HD;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int key = 64 * (2 * i + half) + lane;              // keys interleaved by 64 between the two waves: 0-63 | 64-127 | ...
                sc[i] = -INFINITY;
                if (key < L) {
                    const float* kp = p.QKV + (long)p.sg.row(b, key) * p.ldq + A + h * ATT_HD;
                    f32x4 kv[16];
#pragma unroll
                    for (int k4 = 0; k4 < 16; ++k4) kv[k4] = *reinterpret_cast<const f32x4*>(kp + 4 * k4);
                    float a = 0.f;
#pragma unroll
                    for (int k4 = 0; k4 < 16; ++k4) {
                        const float2 cs = *reinterpret_cast<const float2*>(p.rope_cos + key * 32 + 2 * k4);
                        const float2 sn = *reinterpret_cast<const float2*>(p.rope_sin + key * 32 + 2 * k4);
                        a += (kv[k4][0] * cs.x - kv[k4][1] * sn.x) * qh[4 * k4] + (kv[k4][0] * sn.x + kv[k4][1] * cs.x) * qh[4 * k4 + 1];
                        a += (kv[k4][2] * cs.y - kv[k4][3] * sn.y) * qh[4 * k4 + 2] + (kv[k4][2] * sn.y + kv[k4][3] * cs.y) * qh[4 * k4 + 3];
                    }
                    sc[i] = a;
                    mx = fmaxf(mx, a);
                }
            }
            mx = wave_max(mx);
            if (lane == 0) scratch[wave] = mx;
        }
        __syncthreads();
        float sum = 0.f;
        if (h < nhead) {
            mx = fmaxf(scratch[2 * h], scratch[2 * h + 1]);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int key = 64 * (2 * i + half) + lane;
                if (key < L) { sc[i] = __builtin_amdgcn_exp2f(sc[i] - mx); sum += sc[i]; }
            }
            sum = wave_sum(sum);
        }
        __syncthreads();
        if (h < nhead && lane == 0) scratch[wave] = sum;
        __syncthreads();
        if (h < nhead) {
            const float inv = 1.0f / (scratch[2 * h] + scratch[2 * h + 1]);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int key = 64 * (2 * i + half) + lane;
                if (key < L) ps[key * RV_MAX_HEADS + h] = sc[i] * inv * merge_row_stat(p.at_part, p.at_pw, p.at_rows, D, p.sg.row(b, key)).y;
            }
        } else if (h < RV_MAX_HEADS) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int key = 64 * (2 * i + half) + lane;
                if (key < L) ps[key * RV_MAX_HEADS + h] = 0.f;
            }
        }
        __syncthreads();
    }
    // ---- yv[h][c] = sum_j ps[h][j] at[row j][c]: float4 columns x 5 row groups, the groups added in a fixed order -----------------
    {
        const int c4n = D / 4;                                           // 192 float4 columns
        const int grp = tid / c4n, c4 = tid % c4n;
        const int NG = TAIL_THREADS / c4n;                               // 5 row groups at D = 768, 8 at 512
        f32x4 acc[RV_MAX_HEADS];
#pragma unroll
        for (int h = 0; h < RV_MAX_HEADS; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (grp < NG) {
            for (int s2 = 0; s2 < p.sg.nseg; ++s2) {
                const float* xr = p.AT + ((long)p.sg.base[s2] + (long)b * p.sg.len[s2]) * D + 4 * c4;
                const int j0 = p.sg.off[s2], n = p.sg.len[s2];
#pragma unroll 4
                for (int j = grp; j < n; j += NG) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + (long)j * D);
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(ps + (j0 + j) * RV_MAX_HEADS), p1 = *reinterpret_cast<const f32x4*>(ps + (j0 + j) * RV_MAX_HEADS + 4);
#pragma unroll
                    for (int h = 0; h < RV_MAX_HEADS; ++h) {
                        const float pw = h < 4 ? p0[h] : p1[h - 4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[h][c] = __builtin_fmaf(pw, xv[c], acc[h][c]);
                    }
                }
            }
        }
        for (int g = 0; g < NG; ++g) {
            if (grp == g) {
#pragma unroll
                for (int h = 0; h < RV_MAX_HEADS; ++h)
                    if (h < nhead) {
                        f32x4* dst = reinterpret_cast<f32x4*>(yv + h * TAIL_MAXD + 4 * c4);
                        *dst = g == 0 ? acc[h] : *dst + acc[h];
                    }
            }
            __syncthreads();
        }
    }
    // ---- o[h*64 + d] = sum_c yv[h][c] Wv[c][h*64 + d] + t_v;  at_c += o Wo + bo -------------------------------------------------
    tail_gemv(yv, TAIL_MAXD, p.wqkv + 2 * A, 3 * A, D, A, red, qo);
    for (int n = tid; n < A; n += TAIL_THREADS) qo[n] += p.bqkv[2 * A + n];
    __syncthreads();
    tail_gemv(qo, 0, p.wo, D, A, D, red, yv);                            // (yv is free: reuse its first D floats)
    for (int n = tid; n < D; n += TAIL_THREADS) at_c[n] += yv[n] + p.bo[n];
    __syncthreads();
    // ---- x_c += FF(LN2(at_c)): relu(rstd (at_c W1'') + b1) W2 + b2 ---------------------------------------------------------------
    const float2 st2 = tail_row_stat(at_c, D, scratch);
    tail_gemv(at_c, 0, p.wf1, p.Fd, D, p.Fd, red, f1);
    for (int n = tid; n < p.Fd; n += TAIL_THREADS) f1[n] = fmaxf(f1[n] * st2.y + p.bf1[n], 0.f);
    __syncthreads();
    tail_gemv(f1, 0, p.wf2, D, p.Fd, D, red, yv);
    for (int n = tid; n < D; n += TAIL_THREADS) { const float v = x_c[n] + yv[n] + p.bf2[n]; x_c[n] = v; p.Xc[(long)b * D + n] = v; }
    __syncthreads();
    // ---- last_norm, the 22 decoder rows, softmax and the draw (sample.py:508-513) --------------------------------------------------
    sample_row<TAIL_THREADS / 64>(x_c, D, p.head, p.tokens, b, slot, t, p.q_noise, p.q_rows, p.q_off, p.rs, L, scratch);
}
}  // namespace hd
