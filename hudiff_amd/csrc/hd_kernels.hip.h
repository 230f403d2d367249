// hd_kernels.hip.h -- device kernels of libhudiff_hip.so (gfx950 / CDNA4 only, wave = 64).
//
// Everything the denoiser forward needs is one of:
//   gemm_k        fp32 MFMA (v_mfma_f32_32x32x2_f32) GEMM  C = epi( pro(A) @ W + bias )
//                 pro : optional LayerNorm + ReLU/GELU of the A rows from precomputed or merged (mean, rstd); a LayerNorm
//                       with no activation behind it is folded into W instead (GemmP::ln_fold: no prologue)
//                 CONV: the k=7 dilated ByteNet convolution as 7 row-shifted tap GEMMs, zero outside
//                       the chain segment (model/encoder/model.py:170-180 via sequence_models MaskedConv1d)
//                 epi : [rstd scale of a folded LayerNorm] + bias, activation, + residual, functional dropout
//                       (generated or injected), + post-dropout addend, strided store, LayerNorm slice partials
//                 The BK = 16 instantiations keep their K loop free of vector-ALU instructions (buffer loads with
//                 SGPR descriptors, immediate-offset LDS reads): on gfx950 VALU and MFMA time-slice one issue port.
//   attn_k        RoPE + softmax(QK^T/8) V for one (row, head): K/V staged in LDS, S^T = K Q^T kept in
//                 MFMA accumulators (16x16x4 f32) so the softmax is lane-local and P feeds PV directly
//                 (model/encoder/cross_attention.py:149-173); small launches and the all-fp32 route
//   attn_x3_k     the same structure with every product as three fp16 MFMAs on fp16 (hi, lo) operand splits (fp32 accumulation and
//                 softmax, fp32 in / out): the attention core of the DEFAULT route for launches >= 8192 rows since round 3
//   gemm_x3_k     split-precision GEMM (HUDIFF_X3=1): operands as fp16 (hi, lo) planes by LDS DMA, three MFMAs per product; its
//                 ByteNet launches normalise their own output rows (ln_sync: the N tiles of an M tile meet at an L2-level counter)
//   guards        every producer of an fp16 split checks |x| < 65504 (RunState::pad[1] -> the host repeats the call on fp32 kernels);
//                 co-resident blocks never fill a CU's LDS to the last 2 KB (lds_fill_ok / lds_safe_request)
//   row_stats_k   per-row (mean, rstd) for LayerNorm, one wave per row, two-pass
//   small kernels token gather, region/position/side embedding, final LN+decoder+exponential-race
//                 sampling (antibody_scripts/sample.py:510-513), full decoder for hd_forward
//
// Activation rows live in "segment-major" order: all heavy-chain rows [B,152,C] first, then all
// light-chain rows [B,139,C]; ByteNet stacks have separate weights per chain, so a GEMM tile never
// mixes chains.  Segs::row(b, slot) is the only place that knows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "hd_kernels.hip.h is written for gfx950 (CDNA4) only: wave64, MFMA 32x32x16 f16, buffer_load ... lds, 160 KB LDS, s_barrier semantics of ended waves"
#endif

namespace hd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Probe code (ablation bits inside the kernels' loops) exists only in -DHD_PROBES builds (a separate .so, selected with HUDIFF_LIB);
// in the default build HD_ABL() / HD_QABL() are the constant 0 and every branch on them is compiled out.
#ifdef HD_PROBES
#define HD_ABL(p) ((p).x3_abl)
#define HD_QABL(p) ((p).abl)
#else
#define HD_ABL(p) 0
#define HD_QABL(p) 0
#endif

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };
enum { DROP_NONE = 0, DROP_GEN = 1, DROP_INJECT = 2 };
constexpr int PART_STRIDE = 32;   // max column slices of a GEMM output row (N <= 1024 with 32-wide slices)

struct Segs {
    int nseg;      // 1 (nanobody) or 2 (antibody: heavy, light)
    int B;         // rows (sequences) in the batch
    int L;         // slots per sequence
    int len[2];    // slots in segment
    int off[2];    // first slot of segment
    int base[2];   // first activation row of segment  (base[1] = B * len[0])
    __host__ __device__ inline int row(int b, int slot) const {
        int s = (nseg > 1 && slot >= off[1]) ? 1 : 0;
        return base[s] + b * len[s] + (slot - off[s]);
    }
    __host__ __device__ inline int rows() const { return B * L; }
};

// Device-resident per-run state; kernels read it so that a captured hipGraph can be replayed per step.
struct RunState {
    uint32_t step;
    uint32_t seed_lo, seed_hi;
    uint32_t row0;
    uint32_t pad[4];       // pad[0]: set by sample_step_k when a visited row had non-finite logits
                           // pad[1]: range guard of the split-precision kernels -- set by whoever writes an fp16 (hi, lo) split of
                           //         a value with |x| >= X16_LIMIT (hi would round to inf); the host re-runs on the fp32 kernels
    uint32_t done;         // workgroups of sample_step_k that have finished this step: the last one advances `step` (no launch for it)
};
// fp16 holds |x| < 65520 before rounding to infinity; operands of the split-precision kernels are not scaled (see gemm_x3_k),
// so every producer of a split checks its values against this and raises RunState::pad[1] (two v_max3_f32 per float4)
constexpr float X16_LIMIT = 65504.0f;
#ifndef HD_GUARD_MASK
#define HD_GUARD_MASK 31     // bisecting aid: 1 GEMM epilogue, 2 ln_apply_k, 4 attn_k, 8 attn_x3_k staging, 16 attn_x3_k Q
#endif
__device__ __forceinline__ float absmax4(float m, const float __attribute__((ext_vector_type(4))) v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
__device__ __forceinline__ void raise_range_flag(const RunState* rs, float vmax) {
    // NaN compares false: a NaN operand is not a range problem (it surfaces as HD_ERR_NUMERIC / in the logits either way)
    if (vmax >= X16_LIMIT) atomicOr(const_cast<uint32_t*>(&rs->pad[1]), 1u);
}

// ------------------------------------------------------------------------------------------------
// LDS co-residency rule.  A CU has 160 KB of LDS.  Round 2 met workgroups of ONE kernel that together filled it exactly
// (2 x 81 920 B) and then produced wrong rows now and then; 2 KB of slack cured it and the cause was never pinned down (DESIGN.md
// section 9).  The rule since: the blocks of a kernel that can be co-resident on a CU never sum to more than 160 KB - 2 KB.
//   n = min(160 KB / bytes, 32 waves / waves per block)  blocks fit; n * bytes <= LDS_CORESIDENT_MAX must hold.
// Kernels with static LDS assert it at compile time (lds_fill_ok); launches with dynamic LDS ask for lds_safe_request(bytes),
// which pads a request that would fill the CU until one block fewer fits (hd_api.hip).
// ------------------------------------------------------------------------------------------------
constexpr int LDS_PER_CU = 160 * 1024, LDS_CORESIDENT_MAX = LDS_PER_CU - 2048, WAVES_PER_CU = 32;
__host__ __device__ constexpr int lds_blocks_per_cu(int bytes, int threads) {
    const int by_lds = bytes > 0 ? LDS_PER_CU / bytes : WAVES_PER_CU, by_waves = WAVES_PER_CU / ((threads + 63) / 64);
    return by_lds < by_waves ? by_lds : by_waves;
}
__host__ __device__ constexpr bool lds_fill_ok(int bytes, int threads) {
    return (long)lds_blocks_per_cu(bytes, threads) * bytes <= LDS_CORESIDENT_MAX;
}
__host__ __device__ constexpr int lds_safe_request(int bytes, int threads) {
    // smallest request >= bytes that obeys the rule: if n blocks would overfill, ask for just too much for n blocks
    return lds_fill_ok(bytes, threads) ? bytes : LDS_PER_CU / lds_blocks_per_cu(bytes, threads) + 64;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53U, c0), lo0 = 0xD2511F53U * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57U, c2), lo1 = 0xCD9E8D57U * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9U; k1 += 0xBB67AE85U;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Exact (erf) GELU, x Phi(x) = 0.5 x erfc(-x / sqrt 2), with erfc(z >= 0) = t exp(-z^2 + P(t)), t = 1 / (1 + z / 2)
// (Numerical Recipes' erfcc: fractional error < 1.2e-7 everywhere).  Against a float64 GELU the float32 evaluation is
// within 3.8e-7 absolute -- the same as 0.5 x (1 + erff(x / sqrt 2)), which in addition loses all relative accuracy
// for x < -4 where 1 + erf cancels -- at about half the vector-ALU instructions of the libm erff path (one v_rcp_f32,
// one v_exp_f32, ten FMAs, no branches); those instructions are matrix time.
__device__ __forceinline__ float gelu_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.5f, z, 1.0f));
    float p = 0.17087277f;
    p = __builtin_fmaf(p, t, -0.82215223f);
    p = __builtin_fmaf(p, t, 1.48851587f);
    p = __builtin_fmaf(p, t, -1.13520398f);
    p = __builtin_fmaf(p, t, 0.27886807f);
    p = __builtin_fmaf(p, t, -0.18628806f);
    p = __builtin_fmaf(p, t, 0.09678418f);
    p = __builtin_fmaf(p, t, 0.37409196f);
    p = __builtin_fmaf(p, t, 1.00002368f);
    p = __builtin_fmaf(p, t, -1.26551223f);
    const float e = t * __builtin_amdgcn_exp2f(__builtin_fmaf(-z, z, p) * 1.44269504088896340736f);   // erfc(|x| / sqrt 2)
    return 0.5f * x * (x >= 0.0f ? 2.0f - e : e);
}
__device__ __forceinline__ float act_f(float x, int act) {
    if (act == ACT_RELU) return fmaxf(x, 0.0f);
    if (act == ACT_GELU) return gelu_f(x);
    return x;
}
// Sum over groups of N (8 or 16) consecutive lanes with DPP cross-lane moves (single VALU ops, no LDS crossbar):
// quad butterfly (xor 1, xor 2), then row_half_mirror (lane i <- 7 - i) and row_mirror (lane i <- 15 - i).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int N>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(N == 8 || N == 16 || N == 32, "group of 8, 16 or 32 lanes");
    v += dpp_f<0xB1>(v);       // quad_perm(1,0,3,2)
    v += dpp_f<0x4E>(v);       // quad_perm(2,3,0,1)
    v += dpp_f<0x141>(v);      // row_half_mirror
    if (N >= 16) v += dpp_f<0x140>(v);   // row_mirror
    if (N == 32) v += __shfl_xor(v, 16); // the neighbouring DPP row (wave tiles 128 columns wide)
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ------------------------------------------------------------------------------------------------
// GEMM
// ------------------------------------------------------------------------------------------------
struct GemmP {
    // operands
    const float* A; int lda;          // [rows, Kc] activation (segment-major rows)
    const float* W;                   // [taps*Kc, N] row-major, per segment at + seg * w_stride
    // split-precision variant (gemm_x3_k): W as pre-tiled (hi, lo) fp16 planes scaled by a power of two, per segment at
    // + seg * wx_stride halfs; acc_scale = 2^-(weight shift + activation shift) undoes the scaling in the epilogue
    const uint16_t* Wx; long wx_stride; float acc_scale;
    // Split activation format ("X16"): a row of K fp32 values is stored in the same 4 K bytes as, per group of 16 columns, 16 fp16
    // high parts followed by 16 fp16 low parts (x ~= hi + lo; x16_hi(); groups of 32 until round 5).  A (row, k tile of 32) is ONE 128-byte cache line:
    // rounds 2-3 kept all K high parts, then all K low parts, so that every DMA request of a k tile touched two half lines per row
    // and the vector L1 -- far smaller than the A panels in flight -- fetched each line twice.  gemm_x3_k reads its A operand in this
    // form (written by its producer: ln_apply_k, attn_k / attn_x3_k, or a GEMM epilogue with c_split / C2 / ln_sync), so its K loop
    // carries no conversion.
    int st_nt;                        // fp32 kernels: epilogue stores carry the non-temporal policy (gemm_x3_k always stores non-temporally:
                                      // streamed outputs then do not displace operand lines in L2)
#ifdef HD_PROBES
    int x3_abl;                       // probe builds only (-DHD_PROBES, HUDIFF_X3_ABL; scripts/r05/ab_env.sh): 1 = no MFMAs, 2 = no operand DMA after
                                      // the first tiles, 3 = neither (epilogue only), 4 = one LDS fragment read per k step, bit 3 = no epilogue,
                                      // bit 5 = no stores.  The default build has neither the field nor the branches.
#endif
    int dbg;                          // test aids (hd_debug_*): bit 0 = ln_sync meetings give up after one poll (hd_debug_fail_next_lnsync: exercises
                                      // the ln_sync guard), bit 1 = the N tiles of an M tile on DIFFERENT XCDs (hd_debug_scatter_lnsync),
                                      // bit 2 = every launch takes the all-features epilogue (HUDIFF_X3_ABL=64 at process start; tests)
    int c_split;                      // epilogue: C is written in split form (ldc == N), no fp32 copy
    float* C2;                        // epilogue: additional split copy of the output rows, row stride N (may be null)
    const float* bias;                // [N], per segment at + seg * n_stride (may be null)
    float* C; int ldc;                // [rows, N]
    int N, Kc, taps, dil;             // K = taps * Kc
    int ldw;                          // row stride of W (>= N; a column slice of a fused matrix has ldw > N)
    uint32_t a_bytes, w_bytes;        // extent of A (all rows) / of one segment's W, for the buffer descriptors (BK = 16 kernels)
    long w_stride; int n_stride; int k_stride;   // per-segment strides of W / bias / (gamma, beta)
    // prologue: LayerNorm over the Kc features of each A row, then activation
    const float2* stats;              // [rows] (mean, rstd) or null
    const float2* spart; int spw;     // alternative to `stats`: the producer's slice partials [PART_STRIDE][rows] and its
    long spart_rows;                  // slice width; merged on the fly in the prologue (non-CONV only)
    const float* gamma; const float* beta;
    int pro_act;
    // LayerNorm folded into the weights (no prologue at all): W holds diag(gamma) W with every COLUMN CENTRED (its mean
    // over k subtracted), bias holds beta W + b.  Since sum_k (x_k - mean) c = 0 for any constant c, centring the columns
    // makes  x W  equal  (x - mean) W : the row mean drops out inside the dot product and the epilogue only forms
    //   rstd_r * acc + bias[n]   from the row statistics (`stats` or `spart`, as for a prologue).
    // Exact algebra of LN(x) W + b with the round-off behaviour of the direct form (no mean * column-sum term that could
    // cancel); used where no activation follows the norm.
    int ln_fold;
    // epilogue
    int epi_act;
    const float* resid; int ldr;      // added before dropout (may alias C)
    const float* extra; int lde;      // added after dropout
    int drop_mode; uint32_t drop_thresh; float drop_scale; uint32_t drop_site;
    const uint8_t* drop_mask;         // [B, L, N] keep-mask (DROP_INJECT)
    const RunState* rs;
    float2* part; long part_rows;     // optional [PART_STRIDE][part_rows] LayerNorm (mean, M2) slice partials of the OUTPUT rows
    // ln_sync (gemm_x3_k only): LayerNorm + activation of the OUTPUT rows applied by this launch itself.  A row's statistics need
    // all of its N columns, i.e. the tiles_n blocks of its M tile; those run at the same time on one XCD (consecutive slots of
    // the XCD-aware tile order), so after leaving their slice partials in `part` they meet at a counter in `sync_ctr` (four ints
    // per M tile: arrivals, departures, XCC ids seen, spare; self-resetting), merge the partials and write   S = act(LN(row; gamma2, beta2))   in
    // split form from the values they still hold in registers -- instead of a separate ln_apply_k pass that reads the fp32
    // rows back from HBM and writes S.  C may be null (S is the only output).  A wait that exceeds its budget raises
    // RunState::pad[2] (the host repeats the call with ln_apply_k passes and the handle keeps those, hd_api.hip check_guards) -- never a hang.
    int ln_sync; int* sync_ctr; float* S; const float* gamma2; const float* beta2; int act2; int k2_stride;
    // geometry
    Segs sg;
    int tiles0;                       // number of M tiles of segment 0
    int tiles_m, tiles_n;             // total M tiles (both segments), N tiles
};

// PRO: A-operand prologue, fixed at compile time so the K loop is branch-free:
//   0 = none, 1 = LayerNorm, 2 = LayerNorm + ReLU, 3 = LayerNorm + GELU
template <int PRO>
__device__ __forceinline__ float pro_f(float x, float mean, float rstd, float g, float b) {
    if (PRO == 0) return x;
    float v = (x - mean) * rstd * g + b;
    if (PRO == 2) v = fmaxf(v, 0.0f);
    if (PRO == 3) v = gelu_f(v);
    return v;
}

// Two elements at once on the packed-fp32 pipe (v_pk_add / v_pk_mul / v_pk_fma: half the vector-ALU time of the scalar
// form, identical roundings: subtract, multiply, fused multiply-add).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PRO>
__device__ __forceinline__ f32x2 pro_f2(f32x2 x, float mean, float rstd, f32x2 g, f32x2 b) {
    if (PRO == 0) return x;
    const f32x2 m2 = {mean, mean}, r2 = {rstd, rstd};
    f32x2 v = __builtin_elementwise_fma((x - m2) * r2, g, b);
    if (PRO == 2) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); }
    if (PRO == 3) { v[0] = gelu_f(v[0]); v[1] = gelu_f(v[1]); }
    return v;
}

// Shared GEMM epilogue: each wave transposes its accumulators through its own slice of LDS so that every lane owns
// 4 consecutive columns of one row, then applies bias / activation / residual / dropout / addend on float4s, writes
// with 16-B stores and (optionally) leaves the LayerNorm slice partials of the rows it wrote.
// smem must hold NW * 32 * (BN/WN + 4) + NW * (BM/WM) * 2 floats (NW = WM * WN waves) and be free (all waves past their
// last LDS read).
// (hi, lo) fp16 split of four fp32 values, hi = fp16(x), lo = fp16(x - hi): two packed conversions for the high parts and one
// v_fma_mix{lo,hi}_f16 per low part (fp16 operand hi, fp32 operand x, fp16 result: x - hi is exact in fp32, so the single
// rounding is the one the conversion would commit) -- 6 instructions where convert / convert back / subtract / convert takes 14
// (10 as hipcc compiles the plain C form); bit-identical on 16 M random bit patterns incl. fp16 subnormals.
//
// The four v_fma_mix are ONE asm statement that carries its own wait states.  hipcc does not model the instructions inside an
// asm string, so it pads none of their hazards (guide section 5.7): a VGPR written by the string and then read as an MFMA
// operand needs 2 wait states, a half-register write (op_sel destination) 1 before any VALU reader.  Round 2's version issued
// four separate statements with no padding; it happened to work until an unrelated edit (the range guard in attn_x3_k) let the
// scheduler place an MFMA one instruction behind the last v_fma_mixhi: every row of HuDiff-Ab came out 2e-2 wrong, silently
// (found by bisecting with -DHD_SPLIT4_MODE, round 3).  Now: the two registers interleaved (lo, lo, hi, hi), early-clobber
// outputs, `s_nop 1` closing the string and `s_nop 0` opening it (an input that comes straight from a transcendental instruction
// -- v_exp for the softmax probabilities -- needs one wait state hipcc would not insert either).
// HD_SPLIT4_MODE=1 builds the compiler-visible form instead (A/B aid).
#ifndef HD_SPLIT4_MODE
#define HD_SPLIT4_MODE 0
#endif
typedef _Float16 hd_f16x4 __attribute__((ext_vector_type(4)));
// X16 row format (GemmP): half index of the HIGH part of column c; its low part sits X16_LO halfs further.  Columns c .. c + 3 of a
// float4 (c % 4 == 0) stay contiguous.  Round 5: groups of SIXTEEN columns -- 16 fp16 high parts, then 16 low parts = 64 bytes -- so that
// a k step of 16 columns of a row is one contiguous half cache line (rounds 2-4: groups of 32, hi and lo of a k step 64 bytes apart) and
// a k tile of 32 columns is still one 128-byte line: qkv_attn_x3_k stages k steps, gemm_x3_k whole k tiles.
__host__ __device__ __forceinline__ constexpr int x16_hi(int c) { return ((c >> 4) << 5) | (c & 15); }
constexpr int X16_LO = 16;
__device__ __forceinline__ void split4(const f32x4 v, hd_f16x4& hh, hd_f16x4& ll) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    hh = __builtin_convertvector(v, hd_f16x4);
#if HD_SPLIT4_MODE == 1
#pragma unroll
    for (int c = 0; c < 4; ++c) ll[c] = (_Float16)__builtin_fmaf((float)hh[c], -1.0f, v[c]);
#else
    const u32x2_t h = __builtin_bit_cast(u32x2_t, hh);
    unsigned int l0, l1;
    asm("s_nop 0\n\t"
        "v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(l0), "=&v"(l1)
        : "v"(h[0]), "v"(h[1]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    const u32x2_t l = {l0, l1};
    ll = __builtin_bit_cast(hd_f16x4, l);
#endif
}

// F: the epilogue features that MAY be present (each is still tested at run time); a caller that knows a launch uses only a few
// of them instantiates the epilogue with those bits, and the code of the others -- their descriptors, scalar loads and
// branches, 35 % of the vector-ALU instructions of a plain launch -- is not compiled in.  gemm_x3_k picks the smallest of a
// handful of masks that covers the launch (three fp16 MFMAs per product make the K loop 5x shorter than the fp32 one, so
// the epilogue is a third of the matrix time of the short-K projections there).
enum : int { EPI_FOLD = 1, EPI_ACT = 2, EPI_RESID = 4, EPI_DROP = 8, EPI_EXTRA = 16, EPI_PART = 32, EPI_CSPLIT = 64, EPI_C2 = 128,
             EPI_ALL = 255,
             EPI_X3 = 256,        // set by gemm_x3_k on every mask: acc_scale always applies, stores are non-temporal
             EPI_LNSYNC = 512 };  // GemmP::ln_sync: the blocks of an M tile exchange their LayerNorm partials and write the NEXT GEMM's
                                  // operand -- act(LN(output row)) in split form -- themselves (see gemm_epilogue)
__host__ __device__ __forceinline__ int epi_needs(const GemmP& p) {
    return (p.ln_fold ? EPI_FOLD : 0) | (p.epi_act ? EPI_ACT : 0) | (p.resid ? EPI_RESID : 0) | (p.drop_mode != DROP_NONE ? EPI_DROP : 0) |
           (p.extra ? EPI_EXTRA : 0) | (p.part ? EPI_PART : 0) | (p.c_split ? EPI_CSPLIT : 0) | (p.C2 ? EPI_C2 : 0) |
           (p.ln_sync ? EPI_LNSYNC : 0);
}
// What a small tile's epilogue would open with, requested BEFORE the K loop instead (gemm_x3_k, BM = 32): bias, the residual values of
// the wave's one 32-row pass, and gamma / beta of an ln_sync second pass.  A launch of a handful of sequences is a chain of memory round
// trips of ~1 us (NOTES.md D); these then travel with the first operand tiles.  (For the 128-row tiles the same was measured and dropped:
// see below; gemm_x3_k says why the 64-row tiles do not take it.)
template <int NIT>
struct EpiPre { f32x4 bv, gv, bv2, rres[NIT]; };
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epi_prefetch(const GemmP& p, EpiPre<32 / (64 / (BN / WN / 4))>& pre, int seg, int seg_rows, int rbase, int m0, int n0) {
    constexpr int WTM = BM / WM, WTN = BN / WN, LPR = WTN / 4, RPI = 64 / LPR;
    static_assert(WTM == 32, "one 32-row pass per wave");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int e_c4 = (lane % LPR) * 4, e_r = lane / LPR;
    const int col = n0 + wn * WTN + e_c4;
    const bool col_ok = col < p.N;
    const int colc = col_ok ? col : 0;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    pre.bv = z; pre.gv = z; pre.bv2 = z;
    if (p.bias && col_ok) pre.bv = *reinterpret_cast<const f32x4*>(p.bias + seg * p.n_stride + col);
    if (p.ln_sync && col_ok) {
        pre.gv = *reinterpret_cast<const f32x4*>(p.gamma2 + seg * p.k2_stride + col);
        pre.bv2 = *reinterpret_cast<const f32x4*>(p.beta2 + seg * p.k2_stride + col);
    }
    const int wrow0 = m0 + wm * WTM;
    const uint32_t r_vo = (uint32_t)((e_r * p.ldr + colc) * 4);
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
        pre.rres[it] = z;
        if (p.resid) {
            const int g0 = wrow0 + it * RPI;
            const bool ok = g0 + e_r < seg_rows && col_ok;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid) + (long)(rbase + g0) * p.ldr, 0, 0x7FFFFFFF, 0x00020000);
            pre.rres[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? r_vo : 0x80000000u), 0, 0));
        }
    }
}
// (Requesting a block's residual values BEFORE its K loop -- 64 registers in the 128 x 128 instantiation, so that their HBM round trips
//  travel under the loop instead of opening every 32-row pass -- was built and measured in round 5: 99.4 vs 99.9 sequences/s, removed.)
template <int BM, int BN, int WM, int WN, int F = EPI_ALL, bool PRE = false>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem,
                                              const float2* rowst, int seg, int seg_rows, int rbase, int Lc, int m0, int n0,
                                              int by, const EpiPre<32 / (64 / (BN / WN / 4))>* pre = nullptr) {
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int ES = WTN + 4;
    constexpr int EPI_FLOATS = WM * WN * 32 * ES;
    // the wave index is wave-uniform by construction; readfirstlane tells the compiler, so that everything derived
    // from it (row bases, buffer descriptors) lives in SGPRs and is advanced on the scalar unit
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int khalf = lane >> 5;
    const int N = p.N;
    const float* __restrict__ bias = p.bias ? p.bias + seg * p.n_stride : nullptr;
    uint32_t k0 = 0, k1 = 0, row0 = 0;
    const int drop_mode = (F & EPI_DROP) ? p.drop_mode : DROP_NONE;
    const int epi_act = (F & EPI_ACT) ? p.epi_act : 0;
    const bool has_resid = (F & EPI_RESID) && p.resid, has_extra = (F & EPI_EXTRA) && p.extra, has_part = (F & EPI_PART) && p.part;
    const bool ln_fold = (F & EPI_FOLD) && p.ln_fold, c_split = (F & EPI_CSPLIT) && p.c_split, has_c2 = (F & EPI_C2) && p.C2;
    const bool ln_sync = (F & EPI_LNSYNC) && p.ln_sync;
    const bool has_c = !(F & EPI_LNSYNC) || p.C != nullptr;      // an ln_sync launch may have S as its only output
    float acc_scale = p.acc_scale;
    asm volatile("" : "+v"(acc_scale));      // kept in a VGPR: the compiler otherwise re-loads it from the kernel arguments per row group
    if (drop_mode == DROP_GEN) {
        uint32_t o[4];
        philox4x32_10(0u, 0u, p.rs->step, p.drop_site, p.rs->seed_lo, p.rs->seed_hi, o);
        k0 = o[0]; k1 = o[1]; row0 = p.rs->row0;
    }
    float* stage = smem + wave * (32 * ES);           // private to this wave: no block barrier needed
    float2* wpart = reinterpret_cast<float2*>(smem + EPI_FLOATS) + wave * WTM;
    constexpr int LPR = WTN / 4;                      // lanes per staged row (float4 each)
    constexpr int RPI = 64 / LPR;                     // rows per wave instruction
    const int e_c4 = (lane % LPR) * 4, e_r = lane / LPR;
    const int col = n0 + wn * WTN + e_c4;
    const bool col_ok = col < N;
    const int colc = col_ok ? col : 0;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PRE) bv = pre->bv;
    else if (bias && col_ok) bv = *reinterpret_cast<const f32x4*>(bias + col);
    // Global accesses of the epilogue: descriptor whose base is the first row of the current 4-row group (moved with
    // scalar adds) + a lane offset fixed for the whole block; a lane switched off by BUF_OFF reads 0 / stores nothing
    // (offset beyond num_records), so there is no per-row 64-bit address arithmetic on the vector ALU.
    constexpr uint32_t BUF_OFF = 0x80000000u;
    constexpr int BUF_MAX = 0x7FFFFFFF;
    const uint32_t c_vo = (uint32_t)((e_r * p.ldc + colc) * 4);
    const uint32_t r_vo = (uint32_t)((e_r * p.ldr + colc) * 4);
    const uint32_t x_vo = (uint32_t)((e_r * p.lde + colc) * 4);
    const int wrow0 = m0 + wm * WTM;                  // first tile row of this wave within the segment (uniform)
    const int nv = min(WTN, N - (n0 + wn * WTN));     // valid columns of this wave's slice (uniform)
    const float inv_nv = 1.0f / (float)max(nv, 1);
    float vmax = 0.f;                                 // largest |value| this lane wrote in split form (range guard)
    f32x4 keep[(F & EPI_LNSYNC) ? TM : 1][(F & EPI_LNSYNC) ? 32 / RPI : 1];     // ln_sync: the finished output values, for the second pass
    // fp32 row and / or split copy of the finished values of one 4-row group
    auto store_out = [&](const f32x4 v, const bool valid, const int g0) {
        if (!c_split && has_c) {
                const __amdgpu_buffer_rsrc_t cs = __builtin_amdgcn_make_buffer_rsrc(
                    p.C + (long)(rbase + g0) * p.ldc, 0, BUF_MAX, 0x00020000);
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                if ((F & EPI_X3) || p.st_nt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), cs, (int)(valid ? c_vo : BUF_OFF), 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), cs, (int)(valid ? c_vo : BUF_OFF), 0, 0);
            }
            if (c_split || has_c2) {
                // split form of the row (hi plane, then lo plane, N halfs each) for a gemm_x3_k consumer
                typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                h16x4 hh, ll;
                split4(v, hh, ll);
                if ((HD_GUARD_MASK & 1) && valid) vmax = absmax4(vmax, v);
                float* base = c_split ? p.C : p.C2;
                const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc(base + (long)(rbase + g0) * N, 0, BUF_MAX, 0x00020000);
                const uint32_t s_vo = (uint32_t)(e_r * N * 4 + x16_hi(colc) * 2);
                if ((F & EPI_X3) || p.st_nt) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hh), ss, (int)(valid ? s_vo : BUF_OFF), 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ll), ss, (int)(valid ? s_vo + (uint32_t)X16_LO * 2 : BUF_OFF), 0, 2);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hh), ss, (int)(valid ? s_vo : BUF_OFF), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ll), ss, (int)(valid ? s_vo + (uint32_t)X16_LO * 2 : BUF_OFF), 0, 0);
                }
            }
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // residual values of this pass are requested up front (each lane reads exactly the elements it will
        // overwrite, so hoisting the loads above the stores is safe even when resid aliases C); they travel
        // while the accumulators are transposed through LDS instead of serialising load -> store per row
        f32x4 rres[32 / RPI];
        if constexpr (PRE) {
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) rres[it] = pre->rres[it];
        } else
        if (has_resid) {
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int g0 = wrow0 + 32 * i + it * RPI;                       // uniform
                const bool ok = g0 + e_r < seg_rows && col_ok;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.resid) + (long)(rbase + g0) * p.ldr, 0, BUF_MAX, 0x00020000);
                // (split route: the residual rows are read ONCE per launch -- 229 MB at 256 antibodies -- so they are asked for non-temporally and
                //  leave the XCD's L2 to the operand tiles its blocks share)
                rres[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? r_vo : BUF_OFF), 0, (F & EPI_X3) ? 2 : 0));
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[((r & 3) + 8 * (r >> 2) + 4 * khalf) * ES + 32 * j + (lane & 31)] = acc[i][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int rr = it * RPI + e_r;
            const int g0 = wrow0 + 32 * i + it * RPI;                           // uniform
            const int lrow = g0 + e_r;
            const bool valid = lrow < seg_rows && col_ok && !(HD_ABL(p) & 32);    // (probe bit 5: every store switched off)
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + rr * ES + e_c4);
            if (valid) {
                if ((F & EPI_X3) || p.Wx) v *= acc_scale;                               // split-precision operands were scaled by powers of two
                if (ln_fold) v *= rowst[wm * WTM + 32 * i + rr].y;      // folded LayerNorm: rstd * (x W''); beta W + b is in `bias`
                v += bv;
                if (epi_act == ACT_RELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.0f);
                } else if (epi_act == ACT_GELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = gelu_f(v[c]);
                }
                if (has_resid) v += rres[it];
                if (drop_mode != DROP_NONE) {
                    const int b = lrow / Lc;
                    const int slot = p.sg.off[seg] + (lrow - b * Lc);
                    if (drop_mode == DROP_GEN) {
                        const uint32_t rk = mix32(k0 ^ mix32(row0 + (uint32_t)b + k1));
                        // (slot * N + col + c) * GOLD = h0 + c * GOLD (mod 2^32): one quarter-rate multiply per float4
                        const uint32_t h0 = rk + (uint32_t)(slot * N + col) * 0x9E3779B9U;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t w = mix32(h0 + (uint32_t)c * 0x9E3779B9U);
                            v[c] = (w >= p.drop_thresh) ? v[c] * p.drop_scale : 0.f;
                        }
                    } else {
                        const uint8_t* mk = p.drop_mask + ((long)b * p.sg.L + slot) * N + col;
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = mk[c] ? v[c] * p.drop_scale : 0.f;
                    }
                }
            }
            if (has_extra) {
                const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.extra) + (long)(rbase + g0) * p.lde, 0, BUF_MAX, 0x00020000);
                v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, (int)(valid ? x_vo : BUF_OFF), 0, 0));
            }
            if constexpr ((F & EPI_LNSYNC) != 0) keep[i][it] = v;
            // (an ln_sync launch stores after it has published its partials: the stores then overlap the wait for the other tiles)
            if (!((F & EPI_LNSYNC) && ln_sync)) store_out(v, valid, g0);
            if (has_part) {
                // LayerNorm statistics of the row this GEMM just produced, for its consumer: every wave owns a
                // WTN-wide column slice of the row (LPR lanes x 4 columns); it reduces (mean, sum of squared
                // deviations) of its slice with DPP row reductions and the consumer merges the slices exactly
                // (Chan et al.), which spares a separate read pass over the activation.
                float ps = valid ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
                ps = group_sum<LPR>(ps);
                const float pm = ps * inv_nv;
                float pq = 0.f;
                if (valid) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const float d = v[c] - pm; pq += d * d; }
                }
                pq = group_sum<LPR>(pq);
                if ((lane % LPR) == 0) wpart[32 * i + rr] = make_float2(pm, pq);
            }
        }
        __builtin_amdgcn_wave_barrier();              // reads done before the next pass overwrites the slice
    }
    if (((F & EPI_CSPLIT) || (F & EPI_C2)) && (c_split || has_c2)) raise_range_flag(p.rs, vmax);
    if (has_part) {
        // slice-major [slice][row]: the wave's WTM row partials go out as one contiguous run
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int r = lane; r < WTM; r += 64) {
            const int lrow = m0 + wm * WTM + r;
            if (lrow < seg_rows && nv > 0) {
                float2* dst = p.part + (long)(by * WN + wn) * p.part_rows + rbase + lrow;
                if ((F & EPI_LNSYNC) && ln_sync) {
                    // read by OTHER workgroups of this launch (the meeting below): an agent-scope relaxed atomic store = a write-through
                    // (sc1) store, visible to agent-scope (sc1) loads once this wave's vmcnt has drained -- whatever XCD the reader is on
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), __builtin_bit_cast(unsigned long long, wpart[r]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *dst = wpart[r];
                }
            }
        }
    }
    if constexpr ((F & EPI_LNSYNC) != 0) {
        if (ln_sync) {
            float2* wst = reinterpret_cast<float2*>(stage);              // (mean, rstd) of this wave's WTM rows; the staging slice is free now
            if (p.tiles_n == 1) {
                // ---- one N tile holds whole rows: the WN slice partials of a row are all in this block's LDS ----------------
                __syncthreads();
                const float2* allp = reinterpret_cast<const float2*>(smem + EPI_FLOATS);
                for (int r = lane; r < WTM; r += 64) {
                    float mean = 0.f, m2 = 0.f;
#pragma unroll
                    for (int j = 0; j < WN; ++j) mean += allp[(wm * WN + j) * WTM + r].x * (float)min(WTN, N - j * WTN);
                    mean /= (float)N;
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const float2 pr = allp[(wm * WN + j) * WTM + r];
                        const float d = pr.x - mean;
                        m2 += pr.y + (float)min(WTN, N - j * WTN) * d * d;
                    }
                    wst[r] = make_float2(mean, 1.0f / sqrtf(m2 / (float)N + 1e-5f));
                }
                __syncthreads();
            } else {
            // ---- meet the other N tiles of this M tile ------------------------------------------------------------------
            // Hand-over of the slice partials between the tiles_n workgroups of an M tile, without a device-scope fence (on gfx950 a
            // release at agent scope is a write-back of the XCD's whole L2: measured at 2x the time of the entire step when every block
            // did it).  The recipe of the programming guide's Guideline 16 in its write-through form: the partials are stored with
            // agent-scope relaxed atomic stores (sc1: written through), every wave drains its vmcnt, the block's arrival is counted
            // with a relaxed agent-scope atomic, and the readers poll relaxed and load the partials with agent-scope relaxed atomic
            // loads (sc1: past the L1).  Correct wherever the workgroups run; FAST because the XCD-aware tile order puts all N tiles of
            // an M tile on one XCD's L2 at the same time (round 3 relied on that placement for correctness too and checked it; the
            // check stays as a counter of the premise, a mixed placement is no longer an error).  Progress needs the tiles_n blocks
            // co-resident: dispatch is in order per XCD and a group's missing members are the next to be dispatched; a meeting that
            // exceeds its polling budget raises RunState::pad[2] and the host repeats the call with ln_apply_k passes -- never a hang.
            // Nothing but the partials has been stored so far (loads and stores share one in-order counter on this ISA: bulk
            // stores ahead of the polls and of the partial loads would put their whole drain time into the meeting).
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);                               // vmcnt(0): this wave's partial stores have reached the L2
            __syncthreads();
            int* ctr = p.sync_ctr + 4 * (seg * 0x4000 + m0 / BM);       // (arrivals, departures, XCC-id mask, -) of this M tile
            if (tid == 0) {
                const int xcc = __builtin_amdgcn_s_getreg(6164) & 15;    // HW_REG_XCC_ID[3:0]
                __hip_atomic_fetch_or(ctr + 2, 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ~1 s of polling: far beyond any launch; then give up loudly (the host repeats the call with ln_apply_k passes).
                // GemmP::dbg bit 0 (hd_debug_fail_next_lnsync, tests only): a budget of one poll, so that the guard's path runs
                int budget = (p.dbg & 1) ? 1 : (1 << 20);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.tiles_n && --budget > 0)
                    __builtin_amdgcn_s_sleep(1);
                const int seen = __hip_atomic_load(ctr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (budget <= 0) atomicOr(const_cast<uint32_t*>(&p.rs->pad[2]), 1u);
                else if (seen & (seen - 1)) atomicOr(const_cast<uint32_t*>(&p.rs->pad[3]), 1u);      // placed on two XCDs: slower, not wrong (counted)
            }
            __syncthreads();
            // ---- (mean, rstd) of this wave's WTM rows from ALL slices of the row: lane r merges row r (Chan et al.) -----
            const int P = p.tiles_n * WN;
            constexpr int PMAX = WTN >= 64 ? 16 : 32;                    // N <= 1024 with 64-wide (32-wide: 32 x 128 tiles) slices
            for (int r = lane; r < WTM; r += 64) {
                const int lrow = m0 + wm * WTM + r;
                float2 st = make_float2(0.f, 0.f);
                if (lrow < seg_rows) {
                    const unsigned long long* pp = reinterpret_cast<const unsigned long long*>(p.part) + rbase + lrow;
                    unsigned long long u[PMAX];
#pragma unroll
                    for (int sl = 0; sl < PMAX; ++sl)                    // all loads in flight together (they bypass the L1)
                        if (sl < P) u[sl] = __hip_atomic_load(pp + (long)sl * p.part_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    float mean = 0.f;
#pragma unroll
                    for (int sl = 0; sl < PMAX; ++sl)
                        if (sl < P) mean += __uint_as_float((uint32_t)u[sl]) * (float)min(WTN, N - sl * WTN);
                    mean /= (float)N;
                    float m2 = 0.f;
#pragma unroll
                    for (int sl = 0; sl < PMAX; ++sl)
                        if (sl < P) {
                            const float d = __uint_as_float((uint32_t)u[sl]) - mean;
                            m2 += __uint_as_float((uint32_t)(u[sl] >> 32)) + (float)min(WTN, N - sl * WTN) * d * d;
                        }
                    st = make_float2(mean, 1.0f / sqrtf(m2 / (float)N + 1e-5f));
                }
                wst[r] = st;
            }
            __syncthreads();                                             // every wave has read the partials it needs ...
            if (tid == 0) {                                              // ... so this block may leave; the last one resets the counters
                if (__hip_atomic_fetch_add(ctr + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.tiles_n - 1) {
                    __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ctr + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ctr + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            }
            // ---- the launch's own outputs (fp32 rows, split copy) go out now, behind the meeting ------------------------------
            if (has_c || has_c2 || c_split) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int it = 0; it < 32 / RPI; ++it) {
                        const int g0 = wrow0 + 32 * i + it * RPI;
                        store_out(keep[i][it], g0 + e_r < seg_rows && col_ok, g0);
                    }
                if (c_split || has_c2) raise_range_flag(p.rs, vmax);
            }
            // ---- second pass: S = act(LN(row)) in split form, from the values kept in registers ------------------------
            const float* __restrict__ g2 = p.gamma2 + seg * p.k2_stride;
            const float* __restrict__ b2 = p.beta2 + seg * p.k2_stride;
            f32x4 gv = {0.f, 0.f, 0.f, 0.f}, bv2 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (PRE) { gv = pre->gv; bv2 = pre->bv2; }
            else if (col_ok) { gv = *reinterpret_cast<const f32x4*>(g2 + col); bv2 = *reinterpret_cast<const f32x4*>(b2 + col); }
            float smax = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int it = 0; it < 32 / RPI; ++it) {
                    const int rr = it * RPI + e_r;
                    const int g0 = wrow0 + 32 * i + it * RPI;                   // uniform
                    const bool valid = g0 + e_r < seg_rows && col_ok;
                    const float2 st = wst[32 * i + rr];
                    f32x4 w;
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[c] = act_f((keep[i][it][c] - st.x) * st.y * gv[c] + bv2[c], p.act2);
                    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    h16x4 hh, ll;
                    split4(w, hh, ll);
                    if (valid) smax = absmax4(smax, w);
                    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc(p.S + (long)(rbase + g0) * N, 0, BUF_MAX, 0x00020000);
                    const uint32_t s_vo = (uint32_t)(e_r * N * 4 + x16_hi(colc) * 2);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hh), ss, (int)(valid ? s_vo : BUF_OFF), 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ll), ss, (int)(valid ? s_vo + (uint32_t)X16_LO * 2 : BUF_OFF), 0, 2);
                }
            }
            raise_range_flag(p.rs, smax);
        }
    }
}

// (mean, rstd) of an A row: either ready-made or merged from the producing GEMM's (mean, M2) slice partials (Chan et al.)
__device__ __forceinline__ float2 merge_row_stat(const float2* __restrict__ spart, int spw, long spart_rows, int Kc, long grow) {
    const int P = (Kc + spw - 1) / spw;
    constexpr int PM = 24;                             // (768 columns in 32-wide slices -- the 32 x 128 tiles of small launches -- are 24)
    if (P <= PM) {
        // all slices of the row in flight at once, both passes from registers: ONE round trip (the partials were written by the
        // previous launch) instead of two; the same additions in the same order as the loops below
        float2 u[PM];
#pragma unroll
        for (int s = 0; s < PM; ++s)
            if (s < P) u[s] = spart[(long)s * spart_rows + grow];
        float mean = 0.f;
#pragma unroll
        for (int s = 0; s < PM; ++s)
            if (s < P) mean += u[s].x * (float)min(spw, Kc - s * spw);
        mean /= (float)Kc;
        float m2 = 0.f;
#pragma unroll
        for (int s = 0; s < PM; ++s)
            if (s < P) {
                const float d = u[s].x - mean;
                m2 += u[s].y + (float)min(spw, Kc - s * spw) * d * d;
            }
        return make_float2(mean, 1.0f / sqrtf(m2 / (float)Kc + 1e-5f));
    }
    float mean = 0.f;
    for (int s = 0; s < P; ++s) mean += spart[(long)s * spart_rows + grow].x * (float)min(spw, Kc - s * spw);
    mean /= (float)Kc;
    float m2 = 0.f;
    for (int s = 0; s < P; ++s) {
        const float2 pr = spart[(long)s * spart_rows + grow];
        const float d = pr.x - mean;
        m2 += pr.y + (float)min(spw, Kc - s * spw) * d * d;
    }
    return make_float2(mean, 1.0f / sqrtf(m2 / (float)Kc + 1e-5f));
}
__device__ __forceinline__ float2 gemm_row_stat(const GemmP& p, long grow) {
    if (!p.spart) return p.stats[grow];
    return merge_row_stat(p.spart, p.spw, p.spart_rows, p.Kc, grow);
}

// ABLATE (probe builds only, scripts/gemm_probe.hip): 1 = no global loads inside the K loop,
// 2 = additionally no LDS commit / barrier, 3 = MFMAs only.  0 in the product.
//
// Structure of one block (256 threads = 4 waves as WM x WN, wave tile (BM/WM) x (BN/WN) of 32x32 MFMA tiles):
//   K loop, BK = 32:  global -> registers for tile kt+1 is in flight while tile kt is multiplied from LDS
//     (As[k][m], Bs[k][n], k-major so that an MFMA operand read is one conflict-free ds_read per lane);
//     every A row is read as full 128-B lines; one LDS buffer, two barriers per k tile.
//   Epilogue: each wave transposes its accumulators through its own slice of the (now free) LDS so that
//     every lane owns 4 consecutive columns of one row: bias / activation / residual / dropout / addend are
//     applied on float4s and written with 16-B stores (4 rows x 256 B per wave instruction).
template <int BM, int BN, int WM, int WN, bool CONV, int PRO, int ABLATE = 0, int NBUF = 1, int BK = 32, int EPI = EPI_ALL>
__global__ void __launch_bounds__(256, BK == 16 ? 4 : (NBUF == 1 ? 3 : 2)) gemm_k(const GemmP p) {
    constexpr int KQ = BK / 4;                        // float4 per A row per k tile
    // BK == 16 kernels are launched only when Kc % 16 == 0 and every operand spans < 4 GiB (host-checked): their K
    // loop carries (almost) no vector-ALU work -- on gfx950 VALU and MFMA instructions time-slice one issue port
    // (scripts/hybrid_probe.hip), so every address add or select in the loop is MFMA time lost.  Operands are
    // addressed as wave-uniform base (SGPRs, advanced on the scalar unit) + a loop-invariant 32-bit lane offset.
    constexpr bool FAST = (BK == 16);
    constexpr int LDA = BM + 1, LDB = BN + 4;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int AIT = (BM * KQ + 255) / 256, BIT = (BN * KQ) / 256;
    constexpr bool A_FULL = (BM * KQ) % 256 == 0;     // every thread stages A
    constexpr int ES = WTN + 4;                       // epilogue staging row stride (floats)
    constexpr int BUF_FLOATS = BK * LDA + BK * LDB;
    constexpr int LOOP_FLOATS = NBUF * BUF_FLOATS, EPI_FLOATS = 4 * 32 * ES;
    constexpr int PART_FLOATS = 4 * WTM * 2;          // per-wave (mean, M2) of its WTM rows, written out coalesced
    constexpr int WORK_FLOATS = LOOP_FLOATS > EPI_FLOATS + PART_FLOATS ? LOOP_FLOATS : EPI_FLOATS + PART_FLOATS;
    constexpr int SM_FLOATS = WORK_FLOATS + 2 * BM;   // + (mean, rstd) of the block's rows for a folded LayerNorm (p.ln_fold)
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 MFMA tile");
    static_assert((BN * KQ) % 256 == 0, "every thread stages W");
    static_assert((BK * LDA) % 4 == 0, "Bs must stay 16-byte aligned");

    static_assert(lds_fill_ok(SM_FLOATS * 4, 256), "co-resident blocks of this kernel would fill the CU's LDS (see LDS co-residency rule)");
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
    float (*As)[LDA] = reinterpret_cast<float (*)[LDA]>(smem);
    float (*Bs)[LDB] = reinterpret_cast<float (*)[LDB]>(smem + BK * LDA);
    static_assert(BUF_FLOATS % 4 == 0, "second buffer must stay 16-byte aligned");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // Block -> (M tile, N tile).  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, an
    // observed property used for speed only), each with a private 4 MiB L2.  All N tiles of one M tile are
    // given to the SAME XCD on consecutive slots, so the fp32 A panel (BM x K x 4 B) is fetched from HBM once
    // and re-read from that L2 by the other N tiles instead of streaming from HBM N/BN times.
    int bx, by, seg = 0;
    {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        by = slot % p.tiles_n;
        bx = (slot / p.tiles_n) * 8 + xcd;
        if (bx >= p.tiles_m) return;
    }
    if (p.sg.nseg > 1 && bx >= p.tiles0) { seg = 1; bx -= p.tiles0; }
    const int Lc = p.sg.len[seg];
    const int seg_rows = p.sg.B * Lc;
    const int rbase = p.sg.base[seg];
    const int m0 = bx * BM, n0 = by * BN;
    const float* __restrict__ W = p.W + (long)seg * p.w_stride;
    const float* __restrict__ gamma = PRO ? p.gamma + seg * p.k_stride : nullptr;
    const float* __restrict__ beta = PRO ? p.beta + seg * p.k_stride : nullptr;
    const int N = p.N, Kc = p.Kc;
    const int nkt_tap = (Kc + BK - 1) / BK;
    const int nkt = nkt_tap * p.taps;
    const int half = (p.taps - 1) / 2;

    auto row_stat = [&](long grow) -> float2 { return gemm_row_stat(p, grow); };
    // per-thread staging coordinates (fixed over the K loop)
    const int a_kq = tid % KQ;                        // k quad within the k tile (same for every A row of a thread)
    int a_r[AIT], a_pos[AIT];
    bool a_ok[AIT];
    long a_row[AIT];
    float2 a_st[AIT];
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        const int idx = tid + 256 * i;
        a_r[i] = idx / KQ;
        const int lrow = m0 + a_r[i];
        a_ok[i] = (A_FULL || idx < BM * KQ) && (lrow < seg_rows);
        a_pos[i] = CONV ? (lrow % Lc) : 0;
        a_row[i] = a_ok[i] ? (long)rbase + lrow : 0;  // clamped: always a readable row
        a_st[i] = make_float2(0.f, 0.f);
        if (PRO && !CONV) a_st[i] = row_stat(a_row[i]);
    }
    if (PRO == 0 && p.ln_fold) {    // folded LayerNorm: the epilogue needs rstd of every row of the tile
        float2* rowst = reinterpret_cast<float2*>(smem + WORK_FLOATS);
        for (int r = tid; r < BM; r += 256) {
            const int lrow = m0 + r;
            rowst[r] = row_stat(lrow < seg_rows ? (long)rbase + lrow : (long)rbase);
        }
        // visible to every wave after the barriers of the K loop (each block runs at least one k tile)
    }
    constexpr int B_KSTEP = 256 / (BN / 4);
    const int b_kr = tid / (BN / 4), b_nq = tid % (BN / 4);
    const int b_col = n0 + 4 * b_nq;
    const int b_colc = b_col < N ? b_col : 0;

    f32x4 ra[AIT], rw[BIT], rg, rb;
    float2 rst[AIT];
    bool rav[AIT];

    // Loads are unconditional (addresses clamped to something readable) and invalid lanes are zeroed at
    // commit time: no divergent branch or select sits between a load and the MFMAs, so the whole k tile's
    // loads are issued back to back and stay in flight under the MFMAs of the current tile.
    // CONV: per-tap state (row shift, validity at the chain ends, statistics of the shifted row) changes only
    // every nkt_tap k tiles, so it is refreshed at tap boundaries instead of every k tile
    const float* t_ptr[AIT];
    bool t_ok[AIT];
    // FAST: byte offsets into the buffer descriptors.  Operands are < 2 GiB (host-checked), so BUF_OOB plus any k
    // advance stays beyond num_records without wrapping: such a lane reads zeros (rows past the segment end, conv
    // padding) and no select is needed between the load and the LDS write.
    constexpr uint32_t BUF_OOB = 0x80000000u;
    uint32_t t_boff[AIT], w_boff[BIT];
    __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, FAST ? (int)p.a_bytes : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, FAST ? (int)p.w_bytes : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gamma), 0, (FAST && PRO) ? Kc * 4 : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(beta), 0, (FAST && PRO) ? Kc * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        t_ptr[i] = p.A + a_row[i] * p.lda; t_ok[i] = a_ok[i];
        t_boff[i] = a_ok[i] ? (uint32_t)((a_row[i] * p.lda + 4 * a_kq) * 4) : BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) w_boff[i] = (uint32_t)(((long)(b_kr + B_KSTEP * i) * p.ldw + b_colc) * 4);
    auto fetch = [&](int kt) {
        const int tap = CONV ? kt / nkt_tap : 0;
        const int kk0 = (CONV ? kt - tap * nkt_tap : kt) * BK;
        if (FAST) {
            if (PRO) {
                rg = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rs, 16 * a_kq, kk0 * 4, 0));
                rb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, 16 * a_kq, kk0 * 4, 0));
            }
            if (CONV && kk0 == 0) {                    // wave-uniform: first k tile of a tap
                const int shift = (tap - half) * p.dil;
#pragma unroll
                for (int i = 0; i < AIT; ++i) {
                    const int sp = a_pos[i] + shift;
                    const bool v = a_ok[i] && sp >= 0 && sp < Lc;
                    const long srow = v ? a_row[i] + shift : 0;
                    // zero padding at the chain ends = an offset the descriptor's range check rejects (reads 0)
                    t_boff[i] = v ? (uint32_t)((srow * p.lda + 4 * a_kq) * 4) : BUF_OOB;
                    if (PRO) rst[i] = p.stats[srow];
                }
            }
            // buffer loads: descriptor in SGPRs, loop-invariant lane offset, k advance on the scalar unit
            const int a_so = kk0 * 4;
            const int w_so = (tap * Kc + kk0) * p.ldw * 4;
#pragma unroll
            for (int i = 0; i < AIT; ++i) {
                rav[i] = t_boff[i] != BUF_OOB;
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, (int)t_boff[i], a_so, 0));
            }
#pragma unroll
            for (int i = 0; i < BIT; ++i)
                rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)w_boff[i], w_so, 0));
            return;
        }
        const int col = kk0 + 4 * a_kq;
        const bool kv = col < Kc;
        const int colc = kv ? col : 0;
        if (PRO) {
            rg = *reinterpret_cast<const f32x4*>(gamma + colc);
            rb = *reinterpret_cast<const f32x4*>(beta + colc);
        }
        if (CONV && kk0 == 0) {                        // wave-uniform: first k tile of a tap
            const int shift = (tap - half) * p.dil;
#pragma unroll
            for (int i = 0; i < AIT; ++i) {
                const int sp = a_pos[i] + shift;
                const bool v = a_ok[i] && sp >= 0 && sp < Lc;
                const long srow = v ? a_row[i] + shift : 0;
                t_ok[i] = v;
                t_ptr[i] = p.A + srow * p.lda;
                if (PRO) rst[i] = p.stats[srow];
            }
        }
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            rav[i] = t_ok[i] && kv;
            ra[i] = *reinterpret_cast<const f32x4*>(t_ptr[i] + colc);
        }
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int kr = b_kr + B_KSTEP * i;
            const long wrow = (kk0 + kr < Kc) ? (long)(tap * Kc + kk0 + kr) : 0;
            // no zeroing: rows beyond Kc meet A values that commit() zeroes, columns beyond N are never
            // stored, and the clamped address always holds finite weights (0 * finite = 0)
            rw[i] = *reinterpret_cast<const f32x4*>(W + wrow * p.ldw + b_colc);
        }
    };
    auto commit = [&](int buf) {
        float (*Aw)[LDA] = reinterpret_cast<float (*)[LDA]>(smem + buf * BUF_FLOATS);
        float (*Bw)[LDB] = reinterpret_cast<float (*)[LDB]>(smem + buf * BUF_FLOATS + BK * LDA);
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            if (A_FULL || tid + 256 * i < BM * KQ) {
                const float2 st = CONV ? rst[i] : a_st[i];
#pragma unroll
                for (int c2 = 0; c2 < 4; c2 += 2) {
                    const f32x2 x2 = {ra[i][c2], ra[i][c2 + 1]}, g2 = {rg[c2], rg[c2 + 1]}, b2 = {rb[c2], rb[c2 + 1]};
                    const f32x2 v2 = pro_f2<PRO>(x2, st.x, st.y, g2, b2);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int c = c2 + e;
                        // FAST: invalid lanes already loaded zeros through the descriptor's range check.  Rows past the
                        // segment end are never stored, so only conv padding behind a prologue (must stay zero AFTER
                        // LayerNorm + activation) still needs the select
                        Aw[4 * a_kq + c][a_r[i]] = (FAST && !(CONV && PRO != 0)) ? v2[e] : (rav[i] ? v2[e] : 0.f);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BIT; ++i)
            *reinterpret_cast<f32x4*>(&Bw[b_kr + B_KSTEP * i][4 * b_nq]) = rw[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fetch(0);
    commit(0);
    __syncthreads();
    const int arow = wm * WTM + (lane & 31);
    const int bcol = wn * WTN + (lane & 31);
    const int khalf = lane >> 5;
    // MFMAs of k-steps [KS0, KS1) of the tile in LDS buffer `buf`; LDS -> register fragments are read LDS_AHEAD
    // k-steps ahead of the MFMAs that consume them
    auto mma = [&](int buf, auto ks0_c, auto ks1_c) {
        constexpr int KS0 = decltype(ks0_c)::value, KS1 = decltype(ks1_c)::value;
        constexpr int LDS_AHEAD = 2, RING = LDS_AHEAD + 1;
        float (*Ar)[LDA] = reinterpret_cast<float (*)[LDA]>(smem + buf * BUF_FLOATS);
        float (*Br)[LDB] = reinterpret_cast<float (*)[LDB]>(smem + buf * BUF_FLOATS + BK * LDA);
        float a[RING][TM], b[RING][TN];
        // FAST: one base address per operand, every fragment read is a ds_read_b32 with an immediate offset.  The reads
        // are volatile so that hipcc does not pair them into ds_read2_b32, whose 8-bit offsets would cost one
        // v_add_u32 per pair (vector-ALU time taken from the MFMAs); LDS instructions issue on their own port.
        typedef const volatile __attribute__((address_space(3))) float* lds_cvf;
        lds_cvf Av = (lds_cvf)&Ar[khalf][arow];
        lds_cvf Bv = (lds_cvf)&Br[khalf][bcol];
        auto lda_f = [&](int ks_, int i) { return FAST ? Av[2 * ks_ * LDA + 32 * i] : Ar[2 * ks_ + khalf][arow + 32 * i]; };
        auto ldb_f = [&](int ks_, int j) { return FAST ? Bv[2 * ks_ * LDB + 32 * j] : Br[2 * ks_ + khalf][bcol + 32 * j]; };
#pragma unroll
        for (int d = 0; d < LDS_AHEAD; ++d) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[d][i] = lda_f(KS0 + d, i);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[d][j] = ldb_f(KS0 + d, j);
        }
#pragma unroll
        for (int ks = KS0; ks < KS1; ++ks) {
            const int c = (ks - KS0) % RING, nx = (ks - KS0 + LDS_AHEAD) % RING;
            if (ABLATE == 3) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nx][i] = a[c][i];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nx][j] = b[c][j];
            } else if (ks + LDS_AHEAD < KS1) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nx][i] = lda_f(ks + LDS_AHEAD, i);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nx][j] = ldb_f(ks + LDS_AHEAD, j);
            }
            if (FAST) __builtin_amdgcn_sched_barrier(0);     // the reads above stay above the MFMAs below
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            // pin the order "reads of a later k-step, then the MFMAs of k-step ks" (hipcc otherwise sinks the
            // reads next to their use and every k-step pays the LDS latency)
            if (FAST) {
                __builtin_amdgcn_sched_barrier(0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            }
        }
    };
    using std::integral_constant;
    for (int kt = 0; kt < nkt; ++kt) {
        if (ABLATE == 0 && kt + 1 < nkt) fetch(kt + 1);
        if (NBUF == 1) {
            mma(0, integral_constant<int, 0>{}, integral_constant<int, BK / 2>{});
            if (ABLATE < 2) {
                __syncthreads();                      // every wave is done reading this tile
                if (kt + 1 < nkt) commit(0);
                __syncthreads();
            }
        } else {
            // two LDS buffers: the next tile is written to the other buffer half-way through this tile's
            // MFMAs (its loads have had half a tile to land), one barrier per k tile
            const int cur = kt & 1;
            mma(cur, integral_constant<int, 0>{}, integral_constant<int, BK / 4>{});
            if (ABLATE < 2 && kt + 1 < nkt) commit(cur ^ 1);
            mma(cur, integral_constant<int, BK / 4>{}, integral_constant<int, BK / 2>{});
            if (ABLATE < 2) __syncthreads();
        }
    }
    if (NBUF == 2) __syncthreads();

    // EPI: the epilogue features this instantiation carries (chosen by the host, launch_gemm_t: one epilogue per kernel -- four
    // copies behind an in-kernel branch, as gemm_x3_k has them, push this 128-VGPR kernel into scratch: 600 spilled registers)
    gemm_epilogue<BM, BN, WM, WN, EPI>(p, acc, smem, reinterpret_cast<const float2*>(smem + WORK_FLOATS), seg, seg_rows, rbase, Lc, m0, n0, by);
}


// ------------------------------------------------------------------------------------------------
// Split-precision GEMM (behind HUDIFF_X3=1; the fp32 gemm_k stays the reference product path).
//
// gfx950 multiplies fp32 operands on the matrix cores at 1/16 of the fp16 rate and has no TF32-like mode.  Here every
// fp32 operand is written as hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significand bits; x - hi is exact in
// fp32) and   a w  ~=  a_hi w_hi + a_hi w_lo + a_lo w_hi   runs as three v_mfma_f32_32x32x16_f16 with fp32 accumulation:
// every fp16 x fp16 product is exact in fp32, the dropped a_lo w_lo term and the two roundings of the lo parts are each
// <= 2^-22 relative -- the size of the rounding an fp32 accumulation makes at EVERY one of its K steps.  Three
// 32-cycle MFMAs cover K = 16 for which the fp32 form needs eight 64-cycle ones.
//   Weights are split once at hd_finalize (power-of-two scaled so that the lo parts stay normal fp16 numbers) and laid
//   out as the 128 x 32 LDS tile images the blocks consume.  Activations arrive ALREADY split ("X16" rows, see GemmP):
//   whoever produces an operand of this kernel -- ln_apply_k, attn_k, a GEMM epilogue -- converts each element once,
//   instead of every consumer block converting it again for each of its N tiles.  Activations are not scaled: fp16 holds
//   |x| < 65504 (far beyond any LayerNorm-ed network's residual stream) and below |x| = 2^-3 the low part is exact to
//   2^-25 ABSOLUTE, half an ulp of an fp32 number near 1.  The epilogue multiplies the accumulators by GemmP::acc_scale
//   and is otherwise the fp32 kernel's own (bias, activation, residual, dropout, addend, LayerNorm partials, folded
//   LayerNorm, fp32 and / or split output).
// Shapes: Kc % 32 == 0, N % 128 == 0, no A prologue (LayerNorm folded or applied by ln_apply_k), operands < 2 GiB.
//
// K loop (k tile = 32): both operand tiles of k tile kt+1 travel global -> LDS by DMA (buffer_load ... lds: no registers,
// no conversion, no ds_write) into the other of two LDS stages while tile kt is multiplied; one barrier per tile.  Tile
// rows are 64 B (32 halfs) with the 16-byte chunks XOR-swizzled by (row >> 2) & 3 -- baked into the weight images at
// hd_finalize, applied to the A rows by the lane -> chunk assignment of the DMA -- so that every ds_read_b128 of a
// fragment is bank-conflict free without padding.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 32;
constexpr int X3_TILE_HALFS = 2 * X3_BN * X3_BK;   // one operand tile image: hi[128][32] then lo[128][32]
constexpr int X3_TILE_BYTES = X3_TILE_HALFS * 2;   // 16 KiB
// Block barrier that waits for this wave's LDS instructions only (lgkmcnt(0)): __syncthreads() -- and any fence the
// compiler can see -- would drain every outstanding vector-memory instruction as well.  The operand DMA (vector-memory
// instructions that write LDS) is waited for explicitly with s_waitcnt vmcnt; the empty asm statements keep the compiler
// from moving LDS accesses across the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Tile shapes.  The loop is bound by the rate at which a CU can pull operand tiles through its vector L1 into LDS
// (~15 B / clk / CU measured: 128 x 128 tiles load 32 KiB per 1.05 MFLOP and stop at ~260 TFLOP/s-equivalent whatever the
// pipelining), so the big launches use 256 x 256 tiles (8 waves, wave tile 128 x 64: twice the flops per byte) where N
// allows, 256 x 128 where it does not, and 128 x 128 only for launches too small to fill the chip with larger tiles.
// NS LDS stages: NS - 1 operand tiles are in flight while one is multiplied.  A tile's DMA round trip is ~1.5 us (3 000 cycles)
// under load -- four times the MFMA time of a 128 x 128 x 32 tile -- so with two stages the loop is bound by that latency
// (DMA-only ablation: 24 round trips per block); the big launches use three stages of 256 x 128 tiles (144 KB, one block of
// eight waves per CU).
// LW > 0: LW extra waves per block that only issue the operand DMA (waves NW .. NW + LW - 1), while the NW MFMA waves issue none: in the
// small launches a wave's five DMA instructions per k tile (~0.2 us of issue) and its chain of six dependent MFMAs (~0.1 us) are otherwise
// serial in the same wave.  LW must equal NW (the piece -> wave mapping is the MFMA waves' own).
// EPISET: which epilogues the instantiation carries -- 0 all of them (the launch picks by its features), 1 every one but the ln_sync
// meeting, 2 the ln_sync ones only.  The 128 x 128 tile exists as 1 and 2 (launch_gemm picks by GemmP::ln_sync): the meeting epilogue
// keeps a whole output tile in registers and would otherwise set the register count of every 128 x 128 launch.
template <int BM, int BN, int WM, int WN, bool CONV, int NS = 2, int LW = 0, int EPISET = 0>
__global__ void __launch_bounds__(64 * (WM * WN + LW), (WM * WN == 4 && NS <= 3 && LW == 0) ? 2 : 1) gemm_x3_k(const GemmP p) {
    static_assert(LW == 0 || LW == WM * WN, "loader waves mirror the MFMA waves");
    constexpr int BK = X3_BK, NW = WM * WN, NT = 64 * (NW + LW);
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int ES = WTN + 4;
    constexpr int EPI_FLOATS = NW * 32 * ES, PART_FLOATS = NW * WTM * 2;
    constexpr int A_BYTES = 2 * BM * 64, W_BYTES = 2 * BN * 64;         // (hi, lo) images of BM / BN rows x 32 halfs
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int LOOP_FLOATS = NS * STAGE_BYTES / 4;
    constexpr int WORK_FLOATS = LOOP_FLOATS > EPI_FLOATS + PART_FLOATS ? LOOP_FLOATS : EPI_FLOATS + PART_FLOATS;
    constexpr int SM_FLOATS = WORK_FLOATS + 2 * BM;
    constexpr int A_PIECES = A_BYTES / 1024 / NW, W_PIECES = W_BYTES / 1024 / NW;   // 1 KiB DMA pieces per wave and tile
    // LDS image of a stage: A as BM rows of 128 B = [hi 32 halfs | lo 32 halfs] (the X16 row format: one cache line per row and
    // k tile), its eight 16-byte chunks XOR-swizzled by (row >> 1) & 7; W as two planes (hi, lo) of BN rows x 64 B, chunks
    // swizzled by (row >> 2) & 3 (baked into the weight images)
    static_assert(A_BYTES / 1024 % NW == 0 && W_BYTES / 1024 % NW == 0 && BN % X3_BN == 0, "tile / wave split");
    static_assert(lds_fill_ok(SM_FLOATS * 4, NT), "co-resident blocks of this kernel would fill the CU's LDS (see LDS co-residency rule)");
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
    char* St = reinterpret_cast<char*>(smem);          // stage s at St + s * STAGE_BYTES: A hi, A lo, W hi, W lo

    const int tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LW > 0 && wave_all >= NW;      // (wave-uniform)
    const int wave = loader ? wave_all - NW : wave_all;
    const int wm = wave / WN, wn = wave % WN;
    int bx, by, seg = 0;
    {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;        // all N tiles of an M tile on one XCD (see gemm_k)
        by = slot % p.tiles_n;
        bx = (slot / p.tiles_n) * 8 + xcd;
        // test aid (hd_debug_scatter_lnsync, GemmP::dbg bit 1): consecutive workgroups -- which go to DIFFERENT XCDs -- are the N tiles of an
        // M tile, so that the ln_sync meeting runs its cross-XCD path (same results, tests/test_gpu_x3.py)
        if (p.dbg & 2) { by = b % p.tiles_n; bx = b / p.tiles_n; }
        if (bx >= p.tiles_m) return;
    }
    if (p.sg.nseg > 1 && bx >= p.tiles0) { seg = 1; bx -= p.tiles0; }
    const int abl_mode = HD_ABL(p) & 7;               // (0 unless -DHD_PROBES)
    const int Lc = p.sg.len[seg];
    const int seg_rows = p.sg.B * Lc;
    const int rbase = p.sg.base[seg];
    const int m0 = bx * BM, n0 = by * BN;
    const int Kc = p.Kc;
    const int nkt_tap = Kc / BK;
    const int nkt = nkt_tap * p.taps;
    const int half = (p.taps - 1) / 2;
    // weight images are packed per 128-column tile: this block's BN / 128 tiles are nkt * 16 KiB apart
    const uint16_t* __restrict__ Wx = p.Wx + (long)seg * p.wx_stride + (long)by * (BN / X3_BN) * nkt * X3_TILE_HALFS;

    // A pieces of 1 KiB = 8 rows x 128 B: lane l lands at piece base + 16 l = row l >> 3, slot l & 7, and fetches chunk
    // slot ^ swizzle(row) of the row's 128-byte (hi | lo) group of this k tile -- one full cache line per row.  W pieces of 1 KiB =
    // 16 rows x 64 B of one plane, copied linearly.  Rows past the segment end and conv padding get an offset the descriptor's
    // range check rejects: the DMA writes zeros.
    constexpr uint32_t BUF_OOB = 0x80000000u;
    // a_base: byte offset of the piece's row (no shift) + swizzled chunk; a_pos: slot of that row in its chain (conv), a value no shift
    // brings into [0, Lc) for a row past the segment end
    int a_pos[A_PIECES];
    uint32_t a_base[A_PIECES], a_vo[A_PIECES], w_vo[W_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const int piece = NW * i + wave;                                 // rows 8 piece ..
        const int r = 8 * piece + (lane >> 3);
        const int lrow = m0 + r;
        const bool ok = lrow < seg_rows;
        a_pos[i] = !ok ? -(1 << 24) : CONV ? (lrow % Lc) : 0;
        const uint32_t a_in = (uint32_t)((((lane & 7) ^ ((r >> 1) & 7))) << 4);      // swizzled chunk of the 128-byte group
        a_base[i] = ok ? (uint32_t)(((long)rbase + lrow) * p.lda * 4) + a_in : 0u;
        a_vo[i] = ok ? a_base[i] : BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
        const int piece = NW * i + wave;                                 // LDS image: hi plane of all BN rows, then lo plane
        const int plane = piece / (BN / 16), rg = piece % (BN / 16);   // row group of 16 rows
        const int t128 = rg / 8, rg8 = rg % 8;                           // which 128-column tile, row group within it
        w_vo[i] = (uint32_t)(t128 * nkt * X3_TILE_BYTES + plane * (X3_TILE_BYTES / 2) + rg8 * 1024 + lane * 16);
    }
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Wx), 0, (BN / X3_BN) * nkt * X3_TILE_BYTES, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_vp;
    // (Round 5, measured and not kept -- NOTES.md D: the taps as the INNER loop index and an XCD walking through neighbouring M tiles cut what
    //  the tap GEMM pulls through the fabric from 7.7 x to 2.3 x its A operand (L2 hit rate 0.80 -> 0.92) and made the launch 10 % slower.)
    int d_tap = 0, d_kk = 0;                           // (tap, k tile inside the tap) of the NEXT dma() call: calls come in loop order
    // dma() = dma_a() + dma_w(): the A pieces, then the W pieces of the next tile in loop order (the pipelined loop below issues the two
    // halves in the issue slots between dependent MFMA groups)
    auto dma_a = [&](int kt, int st) {
        if (CONV && d_kk == 0) {                       // first k tile of a tap: row shift + zero padding at the chain ends
            const int shift = (d_tap - half) * p.dil;
            const uint32_t soff = (uint32_t)(shift * p.lda * 4);
#pragma unroll
            for (int i = 0; i < A_PIECES; ++i)
                a_vo[i] = (unsigned)(a_pos[i] + shift) < (unsigned)Lc ? a_base[i] + soff : BUF_OOB;
        }
        char* dst = St + st * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_vp)(dst + (NW * i + wave) * 1024), 16, (int)a_vo[i], (CONV ? d_kk : kt) * BK * 4, 0, 0);
    };
    auto dma_w = [&](int kt, int st) {
        char* dst = St + st * STAGE_BYTES;
        const int wt = CONV ? d_tap * nkt_tap + d_kk : kt;        // weight images are stored tap-major
#pragma unroll
        for (int i = 0; i < W_PIECES; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_vp)(dst + A_BYTES + (NW * i + wave) * 1024), 16, (int)w_vo[i],
                                                     wt * X3_TILE_BYTES, 0, 0);
        if (CONV && ++d_kk == nkt_tap) { d_kk = 0; ++d_tap; }
    };
    auto dma = [&](int kt, int st) { dma_a(kt, st); dma_w(kt, st); };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // W fragment of row r = (wave part) + 32 t + (lane & 31), k step ks, k octet g = lane >> 5: chunk 2 ks + g sits at slot
    // chunk ^ ((r >> 2) & 3) of its 64-byte plane row; A fragment (X16 groups of 16 columns: hi k 0-7, hi k 8-15, lo k 0-7, lo k 8-15 per
    // k step): high-part chunk 4 ks + g (low part: + 2) at slot chunk ^ ((r >> 1) & 7) of the 128-byte row.  The wave part and 32 t do
    // not touch bits 1..3 of r.
    const int fsw = (lane >> 2) & 3, fg = lane >> 5, asw = (lane >> 1) & 7;
    const int foff0 = (lane & 31) * 64 + (((0 + fg) ^ fsw) << 4), foff1 = (lane & 31) * 64 + (((2 + fg) ^ fsw) << 4);
    const int aoff0 = (lane & 31) * 128 + (((0 + fg) ^ asw) << 4), aoff1 = (lane & 31) * 128 + (((4 + fg) ^ asw) << 4);
    auto mma = [&](int st) {
        const char* At = St + st * STAGE_BYTES + wm * WTM * 128;
        const char* Wt = St + st * STAGE_BYTES + A_BYTES + wn * WTN * 64;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int o = ks ? foff1 : foff0, oa = ks ? aoff1 : aoff0;
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
            if (abl_mode == 4) {                       // probe: one fragment read per k step instead of 2 (TM + TN)
                const f16x8 f = *reinterpret_cast<const f16x8*>(At + oa);
#pragma unroll
                for (int i = 0; i < TM; ++i) { ah[i] = f; al[i] = f; }
#pragma unroll
                for (int j = 0; j < TN; ++j) { bh[j] = f; bl[j] = f; }
            } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(At + oa + 32 * 128 * i);
                al[i] = *reinterpret_cast<const f16x8*>(At + (oa ^ 32) + 32 * 128 * i);      // low parts: chunk + 2
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(Wt + o + 32 * 64 * j);
                bl[j] = *reinterpret_cast<const f16x8*>(Wt + W_BYTES / 2 + o + 32 * 64 * j);
            }
            }
            // the two cross terms first, the leading term last: TM x TN independent accumulators per term
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
        }
    };
    // s_waitcnt vmcnt(n): this wave's DMA instructions except the n most recent have landed.  In the steady state the tiles
    // issued after tile kt+1 are kt+2 .. kt+NS-1, i.e. (NS - 2) * (A_PIECES + W_PIECES) instructions.
    constexpr int PER_TILE = A_PIECES + W_PIECES, KEEP = (NS - 2) * PER_TILE;
    static_assert(KEEP < 64, "vmcnt is a 6-bit counter");
    constexpr int WAIT_STEADY = (KEEP & 0xF) | ((KEEP >> 4) << 14) | 0x0F70, WAIT_ALL = 0x0F70;
    const bool does_dma = LW == 0 || loader, does_mma = LW == 0 || !loader;
    // 32-row tiles (launches of a handful of sequences): the epilogue's opening loads go out in front of the K loop (EpiPre).  Not the
    // 64-row tiles: the 44 registers this holds across the loop would take them from three blocks per CU to two.
    constexpr bool PRE = BM == 32;
    EpiPre<32 / (64 / (WTN / 4))> pre;
    if constexpr (PRE) { if (does_mma) epi_prefetch<BM, BN, WM, WN>(p, pre, seg, seg_rows, rbase, m0, n0); }
    // PIPE: the K loop of the 4-wave, 64- / 128-row tiles, software-pipelined by hand (round 5; same products in the same order).  The
    // compiler's loop (below; kept for the small tiles, the loader-wave form and the probes) reads a k step's fragments right in front of
    // its MFMAs and drains the matrix pipe at every barrier: isolated, its MFMA-only time was 1.35 x the MFMA floor.  Here the fragments
    // of k step s + 1 are requested BEFORE the MFMAs of step s are issued, a tile's second MFMA batch is carried over the tile's closing
    // barrier (its fragments are in registers, so the stage is free), and the next tile's DMA instructions go out in the issue slots
    // between the three dependent MFMA groups of that batch -- what qkv_attn_x3_k does (hd_attn_fused.hip.h).
    constexpr bool PIPE = LW == 0 && NW == 4 && BM >= 64;
    const bool pipe = PIPE && abl_mode == 0;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nkt && does_dma) dma(t, t);
    if (pipe && NS - 1 < nkt) dma(NS - 1, NS - 1);     // (the pipelined loop keeps NS - 1 tiles in flight BEHIND the one it multiplies)
    if (p.ln_fold) {                                   // folded LayerNorm: the epilogue needs rstd of every row of the tile
        // (requested behind the first tiles' DMA, so that this round trip -- one: merge_row_stat -- travels with theirs; a small launch
        //  is a chain of such round trips)
        float2* rowst_w = reinterpret_cast<float2*>(smem + WORK_FLOATS);
        for (int r = tid; r < BM; r += NT) {
            const int lrow = m0 + r;
            rowst_w[r] = gemm_row_stat(p, lrow < seg_rows ? (long)rbase + lrow : (long)rbase);
        }
        // visible to every wave after the barriers of the K loop (each block runs at least one k tile)
    }
    if constexpr (PIPE) {
        if (pipe) {
            constexpr int KEEP0 = (NS - 1) * PER_TILE;
            static_assert(KEEP0 < 64, "vmcnt is a 6-bit counter");
            constexpr int WAIT_FIRST = (KEEP0 & 0xF) | ((KEEP0 >> 4) << 14) | 0x0F70;
            if (NS <= nkt) __builtin_amdgcn_s_waitcnt(WAIT_FIRST); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);   // tile 0 has landed
            lds_barrier();
            f16x8 ah0[TM], al0[TM], bh0[TN], bl0[TN], ah1[TM], al1[TM], bh1[TN], bl1[TN];
            auto frag = [&](int st, int o, int oa, f16x8* ah, f16x8* al, f16x8* bh, f16x8* bl) {
                const char* At = St + st * STAGE_BYTES + wm * WTM * 128;
                const char* Wt = St + st * STAGE_BYTES + A_BYTES + wn * WTN * 64;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(At + oa + 32 * 128 * i);
                    al[i] = *reinterpret_cast<const f16x8*>(At + (oa ^ 32) + 32 * 128 * i);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const f16x8*>(Wt + o + 32 * 64 * j);
                    bl[j] = *reinterpret_cast<const f16x8*>(Wt + W_BYTES / 2 + o + 32 * 64 * j);
                }
            };
            // the three dependent groups of a k step: cross terms first, the leading term last (mma()'s order)
            auto g1 = [&](const f16x8* al, const f16x8* bh) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            };
            frag(0, foff0, aoff0, ah0, al0, bh0, bl0);
            int st = 0;
            for (int kt = 0; kt < nkt; ++kt) {
                const int st_next = st + 1 == NS ? 0 : st + 1;
                frag(st, foff1, aoff1, ah1, al1, bh1, bl1);
                __builtin_amdgcn_sched_barrier(0);
                g1(al0, bh0); g1(ah0, bl0); g1(ah0, bh0);
                __builtin_amdgcn_sched_barrier(0);
                // tile kt + 1 has landed (tiles kt + 2 .. kt + NS - 1 may stay in flight) ...
                if (kt + NS - 1 < nkt) __builtin_amdgcn_s_waitcnt(WAIT_STEADY); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
                lds_barrier();                         // ... everybody's part of it; everybody has READ all of tile kt (k step 1 is in registers)
                frag(st_next, foff0, aoff0, ah0, al0, bh0, bl0);      // (behind the last tile: a read nobody uses -- no branch in the loop body)
                __builtin_amdgcn_sched_barrier(0);
                g1(al1, bh1);
                __builtin_amdgcn_sched_barrier(0);
                if (kt + NS < nkt) dma_a(kt + NS, st);                 // stage st is free: tile kt + NS
                __builtin_amdgcn_sched_barrier(0);
                g1(ah1, bl1);
                __builtin_amdgcn_sched_barrier(0);
                if (kt + NS < nkt) dma_w(kt + NS, st);
                __builtin_amdgcn_sched_barrier(0);
                g1(ah1, bh1);
                __builtin_amdgcn_sched_barrier(0);
                st = st_next;
            }
        }
    }
    if (!pipe) {
    if (NS - 1 <= nkt) __builtin_amdgcn_s_waitcnt(WAIT_STEADY); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);   // tile 0 has landed
    lds_barrier();
    int st = 0, st_in = NS - 1;                        // stage of tile kt / stage the next DMA fills (= the one tile kt-1 used)
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + NS - 1 < nkt;
        // stage st_in was read in tile kt-1; every wave is past the barrier that ended that tile
        if (more && does_dma && abl_mode != 2 && abl_mode != 3) dma(kt + NS - 1, st_in);
        if (does_mma && abl_mode != 1 && abl_mode != 3) mma(st);
        // tile kt+1 must have landed before the next iteration reads it ...
        if (more) __builtin_amdgcn_s_waitcnt(WAIT_STEADY); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
        lds_barrier();                                 // ... everybody's part of it; everybody is done reading tile kt
        st_in = st;
        st = st + 1 == NS ? 0 : st + 1;
    }
    }
    // The loader waves end here, ahead of the epilogue's barriers (ln_sync meeting).  gfx9-family hardware (gfx950 included: ISA
    // "s_barrier": a wave that has terminated no longer takes part) completes a barrier when every wave of the workgroup that has
    // NOT ended arrived, so the four MFMA waves synchronise among themselves; the static_assert at the top of this file pins the
    // target this relies on.
    if (LW > 0 && loader) return;
    if (HD_ABL(p) & 8) {                               // probe: no epilogue (the accumulators stay live through a never-true store)
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
    // the smallest feature mask that covers this launch (uniform): PFF1 / tap GEMM, Q|K|V, FF1, out-projection / FF2, the rest
    const float2* rowst = reinterpret_cast<const float2*>(smem + WORK_FLOATS);
    const int need = (p.dbg & 4) ? (EPI_ALL | (epi_needs(p) & EPI_LNSYNC)) : epi_needs(p);      // (dbg bit 2: test_x3_feature_masked_epilogues_change_nothing)
#define HD_EPI(F) gemm_epilogue<BM, BN, WM, WN, (F) | EPI_X3, PRE>(p, acc, smem, rowst, seg, seg_rows, rbase, Lc, m0, n0, by, &pre)
    if (need & EPI_LNSYNC) {
        // (only the 4-wave 128 x 128 instantiation is ever launched with ln_sync; the others keep the code out)
        if constexpr (NW == 4 && EPISET != 1) {
            if (!(need & ~(EPI_PART | EPI_LNSYNC))) HD_EPI(EPI_PART | EPI_LNSYNC);
            else HD_EPI(EPI_ALL | EPI_LNSYNC);
        }
    }
    else if constexpr (EPISET != 2) {
        if (!(need & ~EPI_PART)) HD_EPI(EPI_PART);
        else if (!(need & ~EPI_FOLD)) HD_EPI(EPI_FOLD);
        else if (!(need & ~(EPI_FOLD | EPI_ACT | EPI_CSPLIT))) HD_EPI(EPI_FOLD | EPI_ACT | EPI_CSPLIT);
        else if (!(need & ~(EPI_RESID | EPI_PART | EPI_C2))) HD_EPI(EPI_RESID | EPI_PART | EPI_C2);
        else if (!(need & ~(EPI_RESID | EPI_PART | EPI_CSPLIT))) HD_EPI(EPI_RESID | EPI_PART | EPI_CSPLIT);
        else HD_EPI(EPI_ALL);
    }
#undef HD_EPI
}

// ------------------------------------------------------------------------------------------------
// LayerNorm + activation applied IN PLACE to a GEMM output whose epilogue left (mean, M2) slice partials:
// used in front of the dilated convolution, which would otherwise redo the normalisation and the (erf) GELU
// for each of its 7 taps and each of its N tiles.  One wave per row; gamma / beta are per chain segment.
// ------------------------------------------------------------------------------------------------
// `stats` (ready-made (mean, rstd) per row) may replace the partials; Y / ldy: output rows (== X, ldx for in place).
// split_out: the output row is written in split form (C fp16 high parts, then C fp16 low parts; ldy == C) for gemm_x3_k.
// In place this overwrites bytes other lanes still have to read, so the whole row is read before anything is written.
__global__ void __launch_bounds__(256) ln_apply_k(const float2* __restrict__ part, int pw, int C, int rows,
                                                   const float* X, int ldx, float* Y, int ldy, const float2* __restrict__ stats,
                                                   const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, int k_stride, int seg1_row0,
                                                   int act, int split_out, const RunState* __restrict__ rs) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    // the row itself is requested before the statistics are merged (the pass is bound by load round trips, not bandwidth:
    // one row per wave, two dependent trips otherwise)
    f32x4 v[4];
    if (split_out) {
        const float* x0 = X + (long)row * ldx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < C) v[j] = *reinterpret_cast<const f32x4*>(x0 + c);
        }
    }
    float mean, rstd;
    if (stats) {
        const float2 st = stats[row];
        mean = st.x; rstd = st.y;
    } else if (part) {
        const float2 st = merge_row_stat(part, pw, rows, C, row);
        mean = st.x; rstd = st.y;
    } else {
        // neither: the statistics of the row this wave already holds (split_out only), with row_stats_k's own arithmetic -- the
        // same sums in the same order -- so that a separate statistics pass over the tensor is not needed
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane * 4 + 256 * j < C) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane * 4 + 256 * j < C) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float d = v[j][k] - mean; q += d * d; }
            }
        rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
    }
    const int seg = row >= seg1_row0 ? 1 : 0;
    const float* g = gamma + seg * k_stride;
    const float* b = beta + seg * k_stride;
    const float* x = X + (long)row * ldx;
    float* y = Y + (long)row * ldy;
    if (split_out) {                                   // C <= 1024
        _Float16* yh = reinterpret_cast<_Float16*>(y);
        float vmax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < C) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(g + c), bv = *reinterpret_cast<const f32x4*>(b + c);
                f32x4 w;
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = act_f((v[j][k] - mean) * rstd * gv[k] + bv[k], act);
                f16x4 hh, ll;
                split4(w, hh, ll);
                if (HD_GUARD_MASK & 2) vmax = absmax4(vmax, w);
                *reinterpret_cast<f16x4*>(yh + x16_hi(c)) = hh;
                *reinterpret_cast<f16x4*>(yh + x16_hi(c) + X16_LO) = ll;
            }
        }
        raise_range_flag(rs, vmax);
        return;
    }
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + c), bv = *reinterpret_cast<const f32x4*>(b + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act_f((v[k] - mean) * rstd * gv[k] + bv[k], act);
        *reinterpret_cast<f32x4*>(y + c) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm statistics: one wave per row, two-pass (mean, then centred variance), eps = 1e-5
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) row_stats_k(const float* __restrict__ X, int ldx, int C, int rows,
                                                    float2* __restrict__ stats) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = X + (long)row * ldx;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) { float d = v[k] - mean; q += d * d; }
    }
    const float var = wave_sum(q) / (float)C;
    if (lane == 0) stats[row] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
}

// ------------------------------------------------------------------------------------------------
// Token embedding gather: X[row(b,l), :] = emb[tokens[b,l], :]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) embed_tokens_k(const int32_t* __restrict__ tokens,
                                                       const float* __restrict__ emb,
                                                       const float2* __restrict__ emb_stats, int d,
                                                       float* __restrict__ X, float2* __restrict__ stats, Segs sg) {
    const int bl = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (bl >= sg.B * sg.L) return;
    const int b = bl / sg.L, l = bl - b * sg.L;
    const float* e = emb + (long)tokens[bl] * d;
    float* x = X + (long)sg.row(b, l) * d;
    if (lane == 0) stats[sg.row(b, l)] = emb_stats[tokens[bl]];    // LayerNorm statistics of an embedding row
    for (int c = lane * 4; c < d; c += 256)
        *reinterpret_cast<f32x4*>(x + c) = *reinterpret_cast<const f32x4*>(e + c);
}

// ------------------------------------------------------------------------------------------------
// Region branch up to "+ sinusoid PE" (RegionEmbedder model.py:222-230 + PositionalEncoding :80-87):
//   x = ReLU(LN_d(W(ReLU(LN_re(emb[region]))) + b)) + pe[l]
// one wave per (b, l); d <= 512, r_embedding <= 8.
// ------------------------------------------------------------------------------------------------
struct RegionW {
    const float* emb;                       // [n_region, re]
    const float* ln0_g; const float* ln0_b; // [re]
    const float* w; const float* b;         // [re, d] (transposed), [d]
    const float* ln1_g; const float* ln1_b; // [d]
    const float* pe;                        // [L, d]
};
__global__ void __launch_bounds__(256) region_embed_k(const int32_t* __restrict__ region, RegionW w, int re,
                                                       int d, float* __restrict__ X, Segs sg) {
    const int bl = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (bl >= sg.B * sg.L) return;
    const int b = bl / sg.L, l = bl - b * sg.L;
    float e[8];
    float m = 0.f;
    const float* er = w.emb + (long)region[bl] * re;
    for (int k = 0; k < re; ++k) { e[k] = er[k]; m += e[k]; }
    m /= (float)re;
    float var = 0.f;
    for (int k = 0; k < re; ++k) { float t = e[k] - m; var += t * t; }
    const float rstd = 1.0f / sqrtf(var / (float)re + 1e-5f);
    for (int k = 0; k < re; ++k) e[k] = fmaxf((e[k] - m) * rstd * w.ln0_g[k] + w.ln0_b[k], 0.f);
    float y[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = lane + 64 * i;
        y[i] = 0.f;
        if (c < d) {
            float a = w.b[c];
            for (int k = 0; k < re; ++k) a += e[k] * w.w[k * d + c];
            y[i] = a; s += a;
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { int c = lane + 64 * i; if (c < d) { float t = y[i] - mean; q += t * t; } }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    float* x = X + (long)sg.row(b, l) * d;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = lane + 64 * i;
        if (c < d) x[c] = fmaxf((y[i] - mean) * rs * w.ln1_g[c] + w.ln1_b[c], 0.f) + w.pe[(long)l * d + c];
    }
}

// ------------------------------------------------------------------------------------------------
// Side branch (SideEmbedder model.py:197-205) for each of the n_side chain types:
//   vec[c] = W2 ReLU(LN(W1 emb[c] + b1)) + b2      one block per chain type, d <= 1024
// ------------------------------------------------------------------------------------------------
struct SideW {
    const float* emb;                      // [n_side, se]
    const float* w1; const float* b1;      // [se, d], [d]
    const float* ln_g; const float* ln_b;  // [d]
    const float* w2; const float* b2;      // [d, d] (in, out), [d]
};
__global__ void __launch_bounds__(256) side_vec_k(SideW w, int se, int d, float* __restrict__ vec) {
    __shared__ float h[1024];
    __shared__ float red[8];
    const int c = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int j = tid; j < d; j += 256) {
        float a = w.b1[j];
        for (int k = 0; k < se; ++k) a += w.emb[c * se + k] * w.w1[k * d + j];
        h[j] = a; s += a;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)d;
    __syncthreads();
    float q = 0.f;
    for (int j = tid; j < d; j += 256) { float t = h[j] - mean; q += t * t; }
    q = wave_sum(q);
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    const float rs = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)d + 1e-5f);
    for (int j = tid; j < d; j += 256) h[j] = fmaxf((h[j] - mean) * rs * w.ln_g[j] + w.ln_b[j], 0.f);
    __syncthreads();
    for (int j = tid; j < d; j += 256) {
        float a = w.b2[j];
        for (int k = 0; k < d; ++k) a += h[k] * w.w2[(long)k * d + j];
        vec[(long)c * d + j] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// Static (token-independent) part of the feature (AntiTFNet._encoder model.py:351-359):
//   extra[row] = pos[row] (+ chn)      -> added to the token-encoder output every step
//   feat[row, d:2d] = pos ; feat[row, 2d:3d] = chn (antibody)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) static_feature_k(const float* __restrict__ pos,
                                                         const float* __restrict__ side_vec,
                                                         const int32_t* __restrict__ chain, int d, int D,
                                                         float* __restrict__ extra, float* __restrict__ feat,
                                                         Segs sg) {
    const int bl = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (bl >= sg.B * sg.L) return;
    const int b = bl / sg.L, l = bl - b * sg.L;
    const long row = sg.row(b, l);
    const float* sv = nullptr;
    if (side_vec) {
        int seg = (sg.nseg > 1 && l >= sg.off[1]) ? 1 : 0;
        sv = side_vec + (long)chain[seg * sg.B + b] * d;
    }
    for (int c = lane; c < d; c += 64) {
        float pv = pos[row * d + c];
        float cv = sv ? sv[c] : 0.f;
        extra[row * d + c] = pv + cv;
        feat[row * D + d + c] = pv;
        if (sv) feat[row * D + 2 * d + c] = cv;
    }
}

// ------------------------------------------------------------------------------------------------
// Attention for one (sequence, head): head_dim = 64, all L keys, no mask (cross_attention.py:149-173).
// K (RoPE applied) and V live in LDS; per 16-query tile the wave keeps S^T[key, q] = K Q^T in 16x16x4
// MFMA accumulators: lane (q = lane & 15, g = lane >> 4) owns keys 16*kt + 4*g + r, so max / sum over
// keys is in-lane plus two xor-shuffles and exp(S) is already the B operand of O^T = V^T P^T.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_HD = 64;
constexpr int ATT_KS = 68;   // K row stride in LDS (floats): 16-B aligned, spreads ds_read_b128 banks
constexpr int ATT_VS = 68;   // V row stride: rows 4 apart land 16 banks apart, so the 4 key rows a ds_read_b32 touches do not collide
// Short sequences (L <= 160, the nanobody model): with a 65-float V stride K + V take 80.9 KB, so TWO blocks share a CU's
// 160 KB (and the kernel is held to 128 VGPRs): one block's staging and softmax run under the other's MFMAs.
__host__ __device__ constexpr int att_vs(int nkt) { return nkt <= 10 ? 65 : ATT_VS; }
constexpr int ATT_MAX_KT = 19;   // ceil(291 / 16)

constexpr int ATT_THREADS = 512;   // 8 waves, two per SIMD: one wave's softmax / LDS latency hides under the other's MFMAs

// ABL (scripts/attn_probe.hip only): 1 = no softmax arithmetic, 2 = no S MFMAs, 3 = no PV MFMAs, 4 = no K / V staging
template <int NKT, int ABL = 0>
__global__ void __launch_bounds__(ATT_THREADS, NKT <= 10 ? 4 : 2) attn_k(const float* __restrict__ QKV, int ldq, int att,
                                                          const float* __restrict__ rope_cos,
                                                          const float* __restrict__ rope_sin,
                                                          float* __restrict__ O, int ldo, int nhead, Segs sg, int o_split,
                                                          const RunState* __restrict__ rs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int VS = att_vs(NKT);
    float vmax = 0.f;                                  // range guard of the split output (o_split)
    const int L = sg.L;
    float* Ks = smem;
    float* Vs = smem + (size_t)L * ATT_KS;
    const int b = blockIdx.x / nhead, h = blockIdx.x % nhead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qoff = h * ATT_HD, koff = att + h * ATT_HD, voff = 2 * att + h * ATT_HD;

    // ---- stage K (rotated) and V ------------------------------------------------------------
    // Every request of the thread's share (K, V, cos, sin: 4 x NST loads) is issued before the first one is consumed.
    // (With a run-time trip count hipcc leaves a rolled loop whose every iteration waits for its own loads: ten exposed
    // HBM latencies per block, 20 % of the kernel's time with one 150 KB-LDS block per CU and nothing to overlap it.)
    {
        constexpr int NST = (NKT * 16 * 16 + ATT_THREADS - 1) / ATT_THREADS;
        f32x4 kb[NST], vb[NST];
        float2 cb[NST], sb[NST];
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int idx = tid + ATT_THREADS * k;
            const int key = min(idx >> 4, L - 1), c4 = (idx & 15) * 4;
            const long row = sg.row(b, key);
            if (ABL != 4) {
                kb[k] = *reinterpret_cast<const f32x4*>(QKV + row * ldq + koff + c4);
                vb[k] = *reinterpret_cast<const f32x4*>(QKV + row * ldq + voff + c4);
            }
            cb[k] = *reinterpret_cast<const float2*>(rope_cos + key * 32 + (c4 >> 1));
            sb[k] = *reinterpret_cast<const float2*>(rope_sin + key * 32 + (c4 >> 1));
        }
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int idx = tid + ATT_THREADS * k;
            if (ABL != 4 && idx < L * 16) {
                const int key = idx >> 4, c4 = (idx & 15) * 4;
                const f32x4 kv = kb[k];
                f32x4 kr;
                kr[0] = kv[0] * cb[k].x - kv[1] * sb[k].x; kr[1] = kv[0] * sb[k].x + kv[1] * cb[k].x;
                kr[2] = kv[2] * cb[k].y - kv[3] * sb[k].y; kr[3] = kv[2] * sb[k].y + kv[3] * cb[k].y;
                *reinterpret_cast<f32x4*>(Ks + key * ATT_KS + c4) = kr;
                if (VS % 4 == 0) {
                    *reinterpret_cast<f32x4*>(Vs + key * VS + c4) = vb[k];
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) Vs[key * VS + c4 + c] = vb[k][c];
                }
            }
        }
    }
    __syncthreads();

    const int qi = lane & 15, g = lane >> 4;
    const int nqt = (L + 15) / 16;
    // raw Q rows of the NEXT tile are requested one tile ahead so that their L2 / HBM latency hides under the
    // current tile's MFMAs
    f32x4 qraw[4];
    auto load_q = [&](int qt) {
        const int qq = qt * 16 + qi;
        const int qc = qq < L ? qq : L - 1;
        const long qrow = sg.row(b, qc);
#pragma unroll
        for (int s = 0; s < 4; ++s) qraw[s] = *reinterpret_cast<const f32x4*>(QKV + qrow * ldq + qoff + 16 * s + 4 * g);
    };
    if (wave < nqt) load_q(wave);
    for (int qt = wave; qt < nqt; qt += ATT_THREADS / 64) {
        const int q = qt * 16 + qi;
        const int qc = q < L ? q : L - 1;
        const long qrow = sg.row(b, qc);
        // Q fragment (B operand): lane holds Q[q][16 s + 4 g + c], rotated and pre-scaled by 1/8
        f32x4 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c4 = 16 * s + 4 * g;
            const f32x4 v = qraw[s];
            const float2 cs = *reinterpret_cast<const float2*>(rope_cos + qc * 32 + (c4 >> 1));
            const float2 sn = *reinterpret_cast<const float2*>(rope_sin + qc * 32 + (c4 >> 1));
            // 1/sqrt(64) and log2(e) in one factor: the scores live in the log2 domain, exp() below is one v_exp_f32
            constexpr float QS = 0.125f * 1.44269504088896340736f;
            qf[s][0] = (v[0] * cs.x - v[1] * sn.x) * QS; qf[s][1] = (v[0] * sn.x + v[1] * cs.x) * QS;
            qf[s][2] = (v[2] * cs.y - v[3] * sn.y) * QS; qf[s][3] = (v[2] * sn.y + v[3] * cs.y) * QS;
        }
        if (qt + ATT_THREADS / 64 < nqt) load_q(qt + ATT_THREADS / 64);
        // S^T tiles, two key tiles per pass so that two independent accumulator chains are in flight
        f32x4 st[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; kt += 2) {
            constexpr int dummy = 0; (void)dummy;
            const bool two = kt + 1 < NKT;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            // A operand row index i = lane & 15.  Only the last key tile can run past L (NKT = ceil(L / 16)): every
            // other tile's addresses are a loop-invariant base + a compile-time offset (no vector-ALU work)
            int key0 = kt * 16 + qi, key1 = (kt + 1) * 16 + qi;
            if (kt == NKT - 1) key0 = key0 < L ? key0 : L - 1;
            if (kt + 1 == NKT - 1) key1 = key1 < L ? key1 : L - 1;
            const float* kp0 = Ks + key0 * ATT_KS + 4 * g;
            const float* kp1 = Ks + key1 * ATT_KS + 4 * g;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 kf0 = *reinterpret_cast<const f32x4*>(kp0 + 16 * s);
                f32x4 kf1 = kf0;
                if (two) kf1 = *reinterpret_cast<const f32x4*>(kp1 + 16 * s);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (ABL == 2) { acc0[c] += kf0[c] * qf[s][c]; if (two) acc1[c] += kf1[c]; continue; }
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf0[c], qf[s][c], acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf1[c], qf[s][c], acc1, 0, 0, 0);
                }
            }
            st[kt] = acc0;
            if (two) st[kt + 1] = acc1;
            __builtin_amdgcn_sched_barrier(0);   // keep later tiles' LDS reads from being hoisted (VGPR pressure)
        }
        // softmax over keys (rows of S^T); this lane owns keys 16 kt + 4 g + r
        float mx = -INFINITY;
        float sum = 0.f;
        if (ABL != 1) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kt == NKT - 1 && kt * 16 + 4 * g + r >= L) st[kt][r] = -INFINITY;
                    mx = fmaxf(mx, st[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { float e = __builtin_amdgcn_exp2f(st[kt][r] - mx); st[kt][r] = e; sum += e; }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
        } else {
            sum = 1.f;
        }
        const float inv = 1.0f / sum;
        // O^T[d, q] = sum_key V[key, d] P[key, q]; the V fragments of key tile kt+1 are read while the MFMAs of
        // tile kt run (register double buffer, order pinned with sched_group_barrier)
        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; oacc[dt] = z; }
        float vf[2][16];
        auto load_v = [&](int kt, float (&dst)[16]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int key = kt * 16 + 4 * g + r;
                if (kt == NKT - 1) key = key < L ? key : L - 1;            // P is exactly 0 there
                const float* vp = Vs + key * VS + qi;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dst[4 * r + dt] = vp[16 * dt];
            }
        };
        load_v(0, vf[0]);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int c = kt & 1;
            if (kt + 1 < NKT) load_v(kt + 1, vf[c ^ 1]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if (ABL == 3) { oacc[dt][r] += vf[c][4 * r + dt] * st[kt][r]; continue; }
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[c][4 * r + dt], st[kt][r], oacc[dt], 0, 0, 0);
                }
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        if (q < L) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o = oacc[dt];
                o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
                const int col = h * ATT_HD + 16 * dt + 4 * g;
                if (o_split) {                         // split rows for the out-projection's gemm_x3_k (ldo halfs hi, then lo)
                    f16x4 hh, ll;
                    split4(o, hh, ll);
                    if (HD_GUARD_MASK & 4) vmax = absmax4(vmax, o);
                    _Float16* orow = reinterpret_cast<_Float16*>(O + qrow * ldo);
                    // x16_hi(64 h + 16 dt + 4 g) with dt a compile-time constant: lane part 128 h + 4 g, the rest an immediate (the
                    // generic shift-and-mask form cost the 128-VGPR kernels six to eight spilled registers)
                    const int oc = 128 * h + 4 * g + 32 * dt;
                    *reinterpret_cast<f16x4*>(orow + oc) = hh;
                    *reinterpret_cast<f16x4*>(orow + oc + X16_LO) = ll;
                } else {
                    *reinterpret_cast<f32x4*>(O + qrow * ldo + col) = o;
                }
            }
        }
    }
    if (o_split) raise_range_flag(rs, vmax);
}

// ------------------------------------------------------------------------------------------------
// Split-precision attention (HUDIFF_X3=1, L <= 304, head dim 64): the structure of attn_k -- one block per (sequence,
// head), K and V resident in LDS, S^T = K Q^T in accumulators so that the softmax is lane-local and exp(S) is directly
// the B operand of O^T = V^T P^T -- with every product as three fp16 MFMAs (v_mfma_f32_16x16x32_f16, fp32 accumulate):
//   S  ~= K_hi Q_hi + K_hi Q_lo + K_lo Q_hi        O ~= V_hi P_hi + V_hi P_lo + V_lo P_hi
// K (rotated) and V are split once while they are staged (the same bytes in LDS as fp32: hi + lo halfs); Q per query
// tile and P per 32-key step are split in registers.  Each 16x16x32 MFMA covers K = 32 in 16 cycles where the fp32 form
// needs eight 32-cycle 16x16x4 ones.
//   K planes  [304 keys][64 halfs]: 128-byte rows, 16-byte chunks XOR-swizzled by (key >> 1) & 7 (conflict-free fragments)
//   V^T planes [64 d][320 keys]   : key order inside every 32-key block permuted to the order the S^T accumulators hold
//                                   P in -- chunk 4 t + g = keys {32 t + 4 g + r, 32 t + 16 + 4 g + r}, r = 0..3 -- so P
//                                   goes from the softmax registers into the MFMA without a shuffle; same swizzle by d.
// ------------------------------------------------------------------------------------------------
// KT = 16-key tiles: 19 (L <= 304, the antibody model: 159 744 B of LDS, one block per CU) or 10 (L <= 160, the nanobody model:
// 81 920 B, two blocks per CU).  The V^T rows are 64 KT' bytes (KT' = KT rounded up to even), i.e. 32 or 16 banks apart modulo the
// 64 banks: the swizzle that spreads the 16 rows a fragment read touches is (d >> 1) & 7 over groups of 8 chunks, resp.
// (d >> 2) & 3 over groups of 4.
template <int KT> struct AxGeom {
    static constexpr int KROWS = 16 * KT, VKEYS = 32 * ((KT + 1) / 2);
    static constexpr int KPLANE = KROWS * 128, VPLANE = 64 * VKEYS * 2;             // bytes
    static constexpr int SMEM = 2 * KPLANE + 2 * VPLANE;
    static constexpr bool WIDE = (VKEYS * 2) % 256 == 128;                          // 640-byte rows (KT = 19); else 320 (KT = 10)
    static_assert(WIDE || (VKEYS * 2) % 256 == 64, "V^T row stride must be 32 or 16 banks modulo 64");
    __device__ static constexpr int vpos(int c, int d) {                            // chunk c of row d -> swizzled chunk position
        return WIDE ? ((c & ~7) | ((c & 7) ^ ((d >> 1) & 7))) : ((c & ~3) | ((c & 3) ^ ((d >> 2) & 3)));
    }
};
constexpr int AX_KROWS = AxGeom<19>::KROWS;        // longest sequence the kernel family covers
constexpr int AX19_THREADS = 768;                  // block size of attn_x3_k<19> (HUDIFF_ATTN_WAVES=8 launches the 512-thread instantiation)

// The query-tile loop of the split-precision attention core: K (rotated) and V^T planes are in LDS (Kh / Kl / Vh / Vl, the layouts
// above); every wave walks over 16-query tiles -- Q from global (fp32 rows, rotated, pre-scaled and split here), S^T = K Q^T, softmax,
// O^T = V^T P^T, O rows out.  Shared by attn_x3_k (K / V staged from the projection's fp32 rows) and qkv_attn_x3_k
// (hd_attn_fused.hip.h: K / V written straight from the projection's accumulators).  QL2 (qkv_attn_x3_k): Q arrives as ready-made MFMA
// fragments that other waves of THIS workgroup stored a moment ago (rotated, scaled, split; 4 KiB per query tile at byte offset
// `qoff` + tile x ldq x 4 of QKV): four coalesced loads per tile, past the L1 (sc1), where a line of the previous launch could linger.
template <int KT, int NTH, bool QL2>
__device__ __forceinline__ void attn_x3_tiles(const char* Kh, const char* Kl, const char* Vh, const char* Vl, const float* __restrict__ QKV, int ldq,
                                              int qoff, const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                              const __amdgpu_buffer_rsrc_t o_rs, int ldo, int b, int h, const Segs& sg, int o_split,
                                              const RunState* __restrict__ rs, int lane, int wave, float* core_stamps = nullptr) {
    typedef AxGeom<KT> G;
    // probe builds (-DHD_QA_STAMPS, scripts/r05/core_stamps.py): shader-clock stamps of this wave's first two query tiles -- tile start, Q fragment
    // ready, S^T done, softmax done, O^T done, rows stored -- 16 floats per wave at core_stamps (nullptr in the product: folded away)
    const unsigned long long cs_t0 = core_stamps ? __builtin_readcyclecounter() : 0ull;
    int cs_tile = 0;
    auto cstamp = [&](int k) {
        if (core_stamps && cs_tile < 2 && lane == 0) core_stamps[wave * 16 + cs_tile * 6 + k] = (float)(__builtin_readcyclecounter() - cs_t0);
    };
    constexpr int AX_KT = KT, AX_VKEYS = G::VKEYS, AX_VPLANE = G::VPLANE;
    constexpr bool EXACT = KT <= 10;
    const int L = sg.L;
    const int qi = lane & 15, g = lane >> 4;
    const int sw = (qi >> 1) & 7;                      // swizzle of this lane's K row (rows 16 kt + qi)
    const int sw_last = ((L - 1) >> 1) & 7;            // ... and of row L - 1, which stands in for keys >= L
    const int nqt = (L + 15) / 16;
    // V^T fragment of step t (keys 32 t ..), d tile dt: chunk 4 t + g of row 16 dt + qi.  Its swizzled position is
    // 8 (t >> 1) + a lane term that depends on t only through its parity, so two lane-constant bases per plane and
    // compile-time offsets (128 (t >> 1) + dt x 16 rows) address every read: no vector-ALU address arithmetic in the PV loop.
    const char* vph[2];
    const char* vpl[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        vph[par] = Vh + qi * (AX_VKEYS * 2) + (G::vpos(4 * par + g, qi) << 4);
        vpl[par] = vph[par] + AX_VPLANE;
    }
    typedef unsigned int u32x4_q __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t qf_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(QKV), 0, 0x7fffffff, 0x00020000);
    f16x8 qh[2], ql[2];                                // Q fragment (B operand) of the current tile; QL2: loop-carried (the next tile's loads land here)
    auto load_qf = [&](int qt) {
        const uint32_t qb = (uint32_t)qoff + (uint32_t)qt * (uint32_t)ldq * 4u + (uint32_t)lane * 16u;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qh[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(qf_rs, (int)(qb + (uint32_t)(2 * ks) * 1024u), 0, 16));
            ql[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(qf_rs, (int)(qb + (uint32_t)(2 * ks + 1) * 1024u), 0, 16));
        }
    };
    if constexpr (QL2) { if (wave + (NTH / 64) * (int)blockIdx.y < nqt) load_qf(wave + (NTH / 64) * (int)blockIdx.y); }
    // (gridDim.y > 1: a few sequences -- the query tiles of a (sequence, head) are shared out over gridDim.y workgroups, each staging K and V)
    for (int qt = wave + (NTH / 64) * blockIdx.y; qt < nqt; qt += (NTH / 64) * gridDim.y) {
        const int q = qt * 16 + qi;
        const int qc = q < L ? q : L - 1;
        const long qrow = sg.row(b, qc);
        cstamp(0);
        // Q fragment (B operand): lane holds Q[q][32 ks + 8 g + 0..7], rotated, pre-scaled by log2(e) / 8, split
        float vmax = 0.f;                              // range guard of this tile's Q split
        // QL2: qkv_attn_x3_k stored the fragments themselves (its hand-over): four coalesced 1 KiB loads per tile, `qoff` = byte offset of the
        // head's blocks in QKV, one row of ldq floats per tile; requested in front of the loop / behind the previous tile's S^T (load_qf)
        if constexpr (!QL2)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int c4 = 32 * ks + 8 * g + 4 * hf;
                const f32x4 v = *reinterpret_cast<const f32x4*>(QKV + qrow * ldq + qoff + c4);
                const float2 cs = *reinterpret_cast<const float2*>(rope_cos + qc * 32 + (c4 >> 1));
                const float2 sn = *reinterpret_cast<const float2*>(rope_sin + qc * 32 + (c4 >> 1));
                constexpr float QS = 0.125f * 1.44269504088896340736f;
                f32x4 r;
                r[0] = (v[0] * cs.x - v[1] * sn.x) * QS; r[1] = (v[0] * sn.x + v[1] * cs.x) * QS;
                r[2] = (v[2] * cs.y - v[3] * sn.y) * QS; r[3] = (v[2] * sn.y + v[3] * cs.y) * QS;
                f16x4 hh, ll;
                split4(r, hh, ll);
                if (HD_GUARD_MASK & 16) vmax = absmax4(vmax, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) { qh[ks][4 * hf + e] = hh[e]; ql[ks][4 * hf + e] = ll[e]; }
            }
        }
        raise_range_flag(rs, vmax);
        if (core_stamps) { asm volatile("" :: "v"(qh[0]), "v"(ql[1])); cstamp(1); }
        // S^T tiles (keys x queries), two key tiles per pass: two independent accumulator chains
        f32x4 st[AX_KT + 1];
#pragma unroll
        for (int kt = 0; kt < AX_KT; kt += 2) {
            const bool two = kt + 1 < AX_KT;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            // keys >= L (last tile only) read row L - 1: their scores are masked to -inf below
            const int r0 = (EXACT ? min(kt * 16 + qi, L - 1) : kt * 16 + qi) * 128;
            const int r1 = (EXACT ? min((kt + 1) * 16 + qi, L - 1) : (kt + 1) * 16 + qi) * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int co0 = ((4 * ks + g) ^ (!EXACT || kt * 16 + qi < L ? sw : sw_last)) << 4;
                const int co1 = ((4 * ks + g) ^ (!EXACT || (kt + 1) * 16 + qi < L ? sw : sw_last)) << 4;
                const f16x8 kh0 = *reinterpret_cast<const f16x8*>(Kh + r0 + co0), kl0 = *reinterpret_cast<const f16x8*>(Kl + r0 + co0);
                f16x8 kh1 = kh0, kl1 = kl0;
                if (two) { kh1 = *reinterpret_cast<const f16x8*>(Kh + r1 + co1); kl1 = *reinterpret_cast<const f16x8*>(Kl + r1 + co1); }
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl0, qh[ks], acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl1, qh[ks], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, ql[ks], acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, ql[ks], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh0, qh[ks], acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh1, qh[ks], acc1, 0, 0, 0);
            }
            st[kt] = acc0;
            if (two) st[kt + 1] = acc1;
            __builtin_amdgcn_sched_barrier(0);   // keep later tiles' LDS reads from being hoisted (VGPR pressure)
        }
        st[AX_KT] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (QL2) {                           // S^T is done with this tile's fragments: the next tile's travel under softmax and O^T
            const int qn = qt + (NTH / 64) * (int)gridDim.y;
            if (qn < nqt) load_qf(qn);
        }
        if (core_stamps) { asm volatile("" :: "v"(st[AX_KT - 1])); cstamp(2); }
        // softmax over keys (rows of S^T); this lane owns keys 16 kt + 4 g + r.  Scores live in the log2 domain.
        float mx = -INFINITY, sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < AX_KT; ++kt)
            {
                // the launcher guarantees 16 (KT - 1) < L <= 16 KT: keys >= L exist only in the last tile, and only there are the
                // scores compared / selected (with L unconstrained the compiler parks 76 lane masks in spilled SGPRs: 240 of the
                // 870 vector instructions per query tile were v_readlane / v_cndmask)
                if (kt == AX_KT - 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt * 16 + 4 * g + r >= L) st[kt][r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
        for (int kt = 0; kt < AX_KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(st[kt][r] - mx); st[kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        if (core_stamps) { asm volatile("" :: "v"(inv)); cstamp(3); }
        // O^T[d, q] = sum_key V^T[d, key] P^T[key, q], 32 keys per step: P of two S^T tiles is this lane's B operand as it is
        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < (AX_KT + 1) / 2; ++t) {
            f16x8 ph, pl;
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                f16x4 h4, l4;
                split4(st[2 * t + q4], h4, l4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { ph[4 * q4 + e] = h4[e]; pl[4 * q4 + e] = l4[e]; }
            }
            // two d tiles at a time, term by term: the three products of an accumulator are dependent MFMAs, and back to back they
            // leave the matrix pipe idle for their latency (round-5 core stamps: 5 300 clocks for 120 MFMAs of 16).  Not in the 128-register
            // instantiation attn_x3_k<10> (two more fragments: 7 spilled registers)
            if constexpr (KT <= 10 && !QL2) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int ro = 128 * (t >> 1) + 16 * dt * (AX_VKEYS * 2);          // compile-time (t, dt unrolled)
                    const f16x8 vh = *reinterpret_cast<const f16x8*>(vph[t & 1] + ro), vl = *reinterpret_cast<const f16x8*>(vpl[t & 1] + ro);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, oacc[dt], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int dp = 0; dp < 4; dp += 2) {
                const int ro0 = 128 * (t >> 1) + 16 * dp * (AX_VKEYS * 2), ro1 = ro0 + 16 * (AX_VKEYS * 2);     // compile-time (t, dp unrolled)
                const f16x8 vh0 = *reinterpret_cast<const f16x8*>(vph[t & 1] + ro0), vl0 = *reinterpret_cast<const f16x8*>(vpl[t & 1] + ro0);
                const f16x8 vh1 = *reinterpret_cast<const f16x8*>(vph[t & 1] + ro1), vl1 = *reinterpret_cast<const f16x8*>(vpl[t & 1] + ro1);
                oacc[dp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl0, ph, oacc[dp], 0, 0, 0);
                oacc[dp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl1, ph, oacc[dp + 1], 0, 0, 0);
                oacc[dp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh0, pl, oacc[dp], 0, 0, 0);
                oacc[dp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh1, pl, oacc[dp + 1], 0, 0, 0);
                oacc[dp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh0, ph, oacc[dp], 0, 0, 0);
                oacc[dp + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh1, ph, oacc[dp + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (core_stamps) { asm volatile("" :: "v"(oacc[3])); cstamp(4); }
        if (q < L) {
            // O rows through a buffer descriptor with 32-bit byte offsets (the launcher checks rows x ldq x 4 < 2 GiB, and ldo < ldq):
            // the 64-bit column offsets of plain pointers were loop invariants the 128-register instantiation had to spill
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            const uint32_t ob = (uint32_t)qrow * (uint32_t)ldo * 4u;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o = oacc[dt];
                o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
                if (o_split) {
                    f16x4 hh, ll;
                    split4(o, hh, ll);
                    // x16_hi(64 h + 16 dt + 4 g) with dt a compile-time constant: lane part 128 h + 4 g, the rest an immediate
                    const uint32_t oc = ob + (uint32_t)(128 * h + 4 * g) * 2u;
                    const uint32_t imm = (uint32_t)(32 * dt) * 2u;                                   // (dt is unrolled: an immediate)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, hh), o_rs, (int)(oc + imm), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, ll), o_rs, (int)(oc + imm + (uint32_t)X16_LO * 2u), 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), o_rs,
                                                           (int)(ob + (uint32_t)(h * ATT_HD + 4 * g) * 4u + (uint32_t)(16 * dt) * 4u), 0, 0);
                }
            }
        }
        cstamp(5);
        ++cs_tile;
    }
}

// NTH: threads per block.  KT = 19 runs twelve waves (three per SIMD, 160 VGPRs): the tile loop's MFMA, LDS and vector-ALU phases of
// three waves interleave better than those of two (227.9 -> 217.7 us per launch); KT = 10 keeps eight (two blocks per CU).
template <int KT, int NTH = ATT_THREADS>
__global__ void __launch_bounds__(NTH, KT <= 10 ? 4 : (NTH > 512 ? 3 : 1)) attn_x3_k(const float* __restrict__ QKV, int ldq, int att,
                                                            const float* __restrict__ rope_cos,
                                                            const float* __restrict__ rope_sin,
                                                            float* __restrict__ O, int ldo, int nhead, Segs sg, int o_split,
                                                            const RunState* __restrict__ rs) {
    typedef AxGeom<KT> G;
    constexpr int AX_KT = KT, AX_KROWS = G::KROWS, AX_VKEYS = G::VKEYS, AX_KPLANE = G::KPLANE, AX_VPLANE = G::VPLANE;
    // KT <= 10 (EXACT): the K planes hold exactly L rows (the launch sizes the dynamic LDS as 2 * 128 L + 2 * VPLANE): for the
    // nanobody model that is 79 872 B -- two blocks per CU with 4 KB to spare.  (With 160 zero-padded rows the block took 81 920 B,
    // two blocks filled the CU's 163 840 B exactly, and blocks that started beside a running one produced wrong rows now and then.)
    // Reads of keys >= L are then redirected to row L - 1 (their scores are masked anyway).  KT = 19 runs one block per CU: its
    // planes keep 16 KT zero-padded rows and the K loop carries no clamps (they cost 317 vs 272 us per launch).
    constexpr bool EXACT = KT <= 10;
    extern __shared__ __attribute__((aligned(16))) char axs[];
    const int L = sg.L;
    const int krows = EXACT ? L : AX_KROWS;
    char* Kh = axs;
    char* Kl = axs + krows * 128;
    char* Vh = axs + 2 * krows * 128;
    char* Vl = Vh + AX_VPLANE;
    const int b = blockIdx.x / nhead, h = blockIdx.x % nhead;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qoff = h * ATT_HD, koff = att + h * ATT_HD, voff = 2 * att + h * ATT_HD;
    // activation row of key slot s of this sequence = s + (s >= off[1] ? A1 : A0); byte offsets inside QKV are 32-bit (the launcher
    // checks rows x ldq x 4 < 2 GiB), loads go through buffer descriptors: ~6 vector instructions of addressing per K load and
    // none per V load (its row is wave-uniform -> scalar offset) instead of the 14 of 64-bit pointer arithmetic.
    const int rA0 = sg.base[0] + b * sg.len[0] - sg.off[0];
    const int rA1 = sg.nseg > 1 ? sg.base[1] + b * sg.len[1] - sg.off[1] : rA0;
    const int roff1 = sg.nseg > 1 ? sg.off[1] : 0x7fffffff;
    const uint32_t rowb = (uint32_t)ldq * 4u;
    const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(QKV), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rope_cos), 0, L * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t s_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rope_sin), 0, L * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(O, 0, 0x7fffffff, 0x00020000);

    // ---- stage K (rotate, split, swizzled 8-byte writes; keys >= L are zero rows) and V^T (a wave takes 8-key chunks,
    //      lane = d: 8 row loads of 256 B each, two 16-byte writes).  Every global load of both is issued before the first
    //      value is consumed: one exposed round trip per block instead of two.
    {
        float vmax = 0.f;                              // range guard: K and V are split from fp32 values here (X16_LIMIT)
        constexpr int NST = (AX_KROWS * 16 + NTH - 1) / NTH;
        constexpr int NCHUNK = AX_VKEYS / 8;                          // 40 / 20 chunks of 8 keys
        constexpr int NCH = (NCHUNK + NTH / 64 - 1) / (NTH / 64);
        f32x4 kb[NST];
        float2 cb[NST], sb[NST];
        float vv[NCH][8];
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        const int kcol = (koff + (tid & 15) * 4) * 4, rcol = (tid & 15) * 8;              // byte offsets inside a row
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int key = min((tid >> 4) + (NTH / 16) * k, L - 1);
            const int row = key + (key >= roff1 ? rA1 : rA0);
            kb[k] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(q_rs, (int)((uint32_t)row * rowb + (uint32_t)kcol), 0, 0));
            cb[k] = __builtin_bit_cast(float2, (u32x2_t)__builtin_amdgcn_raw_buffer_load_b64(c_rs, key * 128 + rcol, 0, 0));
            sb[k] = __builtin_bit_cast(float2, (u32x2_t)__builtin_amdgcn_raw_buffer_load_b64(s_rs, key * 128 + rcol, 0, 0));
        }
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
            const int c = min(wave + (NTH / 64) * cc, NCHUNK - 1);            // chunk 4 t + g (wave-uniform)
            const int t = c >> 2, g = c & 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = min(32 * t + 16 * (j >> 2) + 4 * g + (j & 3), L - 1);
                const int row = key + (key >= roff1 ? rA1 : rA0);                       // scalar
                vv[cc][j] = __builtin_bit_cast(float, (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(q_rs, (voff + lane) * 4, (int)((uint32_t)row * rowb), 0));
            }
        }
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int idx = tid + NTH * k;
            if (idx < krows * 16) {
                const int key = idx >> 4, c4 = (idx & 15) * 4;
                const f32x4 kv = kb[k];
                f32x4 kr;
                kr[0] = kv[0] * cb[k].x - kv[1] * sb[k].x; kr[1] = kv[0] * sb[k].x + kv[1] * cb[k].x;
                kr[2] = kv[2] * cb[k].y - kv[3] * sb[k].y; kr[3] = kv[2] * sb[k].y + kv[3] * cb[k].y;
                if (!EXACT && key >= L) kr = f32x4{0.f, 0.f, 0.f, 0.f};       // padding rows
                f16x4 hh, ll;
                split4(kr, hh, ll);
                if (HD_GUARD_MASK & 8) vmax = absmax4(vmax, kr);
                const int off = key * 128 + ((((c4 >> 3) ^ ((key >> 1) & 7))) << 4) + (c4 & 7) * 2;
                *reinterpret_cast<f16x4*>(Kh + off) = hh;
                *reinterpret_cast<f16x4*>(Kl + off) = ll;
            }
        }
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
            const int c = wave + (NTH / 64) * cc;
            if (c >= NCHUNK) continue;
            const int t = c >> 2, g = c & 3;
            f16x8 hh, ll;
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                f32x4 x4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * q4 + e;
                    const int key = 32 * t + 16 * (j >> 2) + 4 * g + (j & 3);
                    x4[e] = key < L ? vv[cc][j] : 0.f;
                }
                f16x4 h4, l4;
                split4(x4, h4, l4);
                if (HD_GUARD_MASK & 8) vmax = absmax4(vmax, x4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { hh[4 * q4 + e] = h4[e]; ll[4 * q4 + e] = l4[e]; }
            }
            const int off = lane * (AX_VKEYS * 2) + (G::vpos(c, lane) << 4);
            *reinterpret_cast<f16x8*>(Vh + off) = hh;
            *reinterpret_cast<f16x8*>(Vl + off) = ll;
        }
        raise_range_flag(rs, vmax);    // (O is a convex combination of the V rows: covered by the check on V)
    }
    __syncthreads();

    attn_x3_tiles<KT, NTH, false>(Kh, Kl, Vh, Vl, QKV, ldq, qoff, rope_cos, rope_sin, o_rs, ldo, b, h, sg, o_split, rs, lane, wave);
}

// ------------------------------------------------------------------------------------------------
// Pruned last attention block (sampling only).  After the last SelfAttBlock only the row of the slot visited
// at this step feeds the decoder (sample.py:508-513), so for that block the second attention's query side,
// the out-projection and the feed-forward are evaluated for ONE row per sequence:
//   gather_rows_k   compacts rows (b, slot_b) of a [rows, C] buffer into [B, C]
//   attn_row_k      one query per (sequence, head) against all L keys / values (RoPE on the fly)
// The K / V projections and the first attention of the block still cover every row.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int step_slot(const int32_t* __restrict__ order, const int32_t* __restrict__ T,
                                         int Tmax, int b, uint32_t t) {
    return ((int)t < T[b]) ? order[(long)b * Tmax + t] : 0;      // finished rows: any valid slot, never written
}

__global__ void __launch_bounds__(256) gather_rows_k(const float* __restrict__ src, int C, float* __restrict__ dst,
                                                      const int32_t* __restrict__ order, const int32_t* __restrict__ T,
                                                      int Tmax, const RunState* __restrict__ rs, Segs sg) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= sg.B) return;
    const int slot = step_slot(order, T, Tmax, b, rs->step);
    const float* s = src + (long)sg.row(b, slot) * C;
    float* d = dst + (long)b * C;
    for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<f32x4*>(d + c) = *reinterpret_cast<const f32x4*>(s + c);
}

// One wave per (sequence, head).  Qc: [B, att] compact query projection; QKV: full-row buffer holding K at column
// `att + h*64` and V at `2*att + h*64`.  Oc: [B, att].
// With Pw != nullptr the wave stops after the softmax and leaves  Pw[(b * nhead + h) * 320 + key] = p_key * rstd_key
// (rstd of the key's input row, merged from the producer's LayerNorm slice partials `spart`): the value side is then
// evaluated through the INPUT rows (row_value_k + head_proj_k below) and the V projection of all rows is never formed.
__global__ void __launch_bounds__(256) attn_row_k(const float* __restrict__ Qc, const float* __restrict__ QKV, int ldq,
                                                   int att, const float* __restrict__ rope_cos,
                                                   const float* __restrict__ rope_sin, float* __restrict__ Oc,
                                                   int nhead, const int32_t* __restrict__ order,
                                                   const int32_t* __restrict__ T, int Tmax,
                                                   const RunState* __restrict__ rs, Segs sg, float* __restrict__ Pw,
                                                   const float2* __restrict__ spart, int spw, long spart_rows, int Kc) {
    static_assert(lds_fill_ok((4 * ATT_HD + 4 * 320) * 4, 256), "LDS co-residency rule");
    __shared__ float qs[4][ATT_HD];
    __shared__ float ps[4][320];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + w;
    if (bh >= sg.B * nhead) return;
    const int b = bh / nhead, h = bh % nhead, L = sg.L;
    const int slot = step_slot(order, T, Tmax, b, rs->step);
    {   // rotated, pre-scaled query: lane handles the complex pair (2k, 2k+1), k = lane & 31
        const int k = lane & 31;
        if (lane < 32) {
            const float xr = Qc[(long)b * att + h * ATT_HD + 2 * k], xi = Qc[(long)b * att + h * ATT_HD + 2 * k + 1];
            const float c = rope_cos[slot * 32 + k], s = rope_sin[slot * 32 + k];
            constexpr float QS = 0.125f * 1.44269504088896340736f;       // log2 domain, as attn_k
            qs[w][2 * k] = (xr * c - xi * s) * QS;
            qs[w][2 * k + 1] = (xr * s + xi * c) * QS;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float sc[5];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int key = lane + 64 * i;
        sc[i] = -INFINITY;
        if (key < L) {
            const float* kp = QKV + (long)sg.row(b, key) * ldq + att + h * ATT_HD;
            float a = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + 4 * k4);
                const float2 cs = *reinterpret_cast<const float2*>(rope_cos + key * 32 + 2 * k4);
                const float2 sn = *reinterpret_cast<const float2*>(rope_sin + key * 32 + 2 * k4);
                a += (kv[0] * cs.x - kv[1] * sn.x) * qs[w][4 * k4] + (kv[0] * sn.x + kv[1] * cs.x) * qs[w][4 * k4 + 1];
                a += (kv[2] * cs.y - kv[3] * sn.y) * qs[w][4 * k4 + 2] + (kv[2] * sn.y + kv[3] * cs.y) * qs[w][4 * k4 + 3];
            }
            sc[i] = a;
            mx = fmaxf(mx, a);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int key = lane + 64 * i;
        const float e = (key < L) ? __builtin_amdgcn_exp2f(sc[i] - mx) : 0.f;
        if (key < 320) ps[w][key] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (Pw) {
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int key = lane + 64 * i;
            if (key < L) {
                const long row = sg.row(b, key);
                // Chan merge of the row's (mean, M2) slice partials -> rstd (merge_row_stat: all slices in flight at once)
                const float rstd = merge_row_stat(spart, spw, spart_rows, Kc, row).y;
                const float e = __builtin_amdgcn_exp2f(sc[i] - mx);
                Pw[(long)bh * 320 + key] = e * inv * rstd;
            }
        }
        return;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // O[d] = sum_key p[key] V[key, d]; lane = d
    float o = 0.f;
    for (int key = 0; key < L; ++key)
        o += ps[w][key] * QKV[(long)sg.row(b, key) * ldq + 2 * att + h * ATT_HD + lane];
    Oc[(long)b * att + h * ATT_HD + lane] = o / sum;
}

// ------------------------------------------------------------------------------------------------
// Value side of the pruned last attention, through the input rows.  With the LayerNorm folded into column-centred
// weights, v_j = rstd_j (W"v^T x_j) + t_v, hence for the one query of a sequence
//   o = sum_j p_j v_j = W"v^T ( sum_j p_j rstd_j x_j ) + t_v          (sum_j p_j = 1):
// a probability-weighted sum of the INPUT rows per head (row_value_k) followed by one D x 64 projection per head
// (head_proj_k) replaces the V projection of all L rows (2 L D A flops per sequence) -- exact algebra, no RoPE on V.
//   row_value_k : block = sequence.  Y[b, h, c] = sum_j Pw[b, h, j] X[row(b, j), c]        (Pw already holds p_j rstd_j)
//   head_proj_k : block = sequence.  Oc[b, h*64 + d] = sum_c Y[b, h, c] Wv[c, h*64 + d] + tv[h*64 + d]
// ------------------------------------------------------------------------------------------------
constexpr int RV_MAX_HEADS = 8;
// grid = (sequences, ceil(D / 256)): a thread owns one column for all heads; the row loop is unrolled so that eight row
// loads are in flight per thread (the kernel is bound by reading X once: L D floats per sequence)
__global__ void __launch_bounds__(256) row_value_k(const float* __restrict__ X, int D, const float* __restrict__ Pw,
                                                    float* __restrict__ Y, int nhead, Segs sg) {
    static_assert(lds_fill_ok(320 * RV_MAX_HEADS * 4, 256), "LDS co-residency rule");
    __shared__ float ps[320][RV_MAX_HEADS];
    const int b = blockIdx.x, L = sg.L, tid = threadIdx.x;
    for (int i = tid; i < RV_MAX_HEADS * 320; i += 256) {
        const int key = i / RV_MAX_HEADS, h = i % RV_MAX_HEADS;
        ps[key][h] = (key < L && h < nhead) ? Pw[((long)b * nhead + h) * 320 + key] : 0.f;
    }
    __syncthreads();
    const int c = blockIdx.y * 256 + tid;
    if (c >= D) return;
    float acc[RV_MAX_HEADS];
#pragma unroll
    for (int h = 0; h < RV_MAX_HEADS; ++h) acc[h] = 0.f;
    // rows of a sequence are contiguous within a chain segment: walk each segment with a constant stride
    for (int s2 = 0; s2 < sg.nseg; ++s2) {
        const float* xr = X + ((long)sg.base[s2] + (long)b * sg.len[s2]) * D + c;
        const int j0 = sg.off[s2], n = sg.len[s2];
#pragma unroll 8
        for (int j = 0; j < n; ++j) {
            const float xv = xr[(long)j * D];
#pragma unroll
            for (int h = 0; h < RV_MAX_HEADS; ++h) acc[h] = __builtin_fmaf(ps[j0 + j][h], xv, acc[h]);
        }
    }
#pragma unroll
    for (int h = 0; h < RV_MAX_HEADS; ++h)
        if (h < nhead) Y[((long)b * nhead + h) * D + c] = acc[h];
}

// grid = (ceil(B / 4), nhead): a block projects one head of four sequences, so that the head's D x 64 weight slice is read
// once per four sequences; thread = (sequence of the group, output feature d)
__global__ void __launch_bounds__(256) head_proj_k(const float* __restrict__ Y, int D, const float* __restrict__ Wv, int ldw,
                                                    const float* __restrict__ tv, float* __restrict__ Oc, int att, int nhead,
                                                    int B) {
    extern __shared__ __attribute__((aligned(16))) float ys[];        // [4, D]
    const int h = blockIdx.y, b0 = blockIdx.x * 4, tid = threadIdx.x;
    for (int i = tid; i < 4 * D; i += 256) {
        const int bb = i / D, c = i % D;
        ys[i] = b0 + bb < B ? Y[((long)(b0 + bb) * nhead + h) * D + c] : 0.f;
    }
    __syncthreads();
    const int bb = tid >> 6, d = tid & 63;
    const float* yh = ys + bb * D;
    const float* wp = Wv + h * ATT_HD + d;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int c = 0;
    // sixteen weight loads in flight per step (the loop is bound by their round trips); every accumulator still receives its
    // terms c = k (mod 4) in increasing c, i.e. the additions and their order are those of the plain loop below
    for (; c + 15 < D; c += 16) {
        float wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = wp[(long)(c + u) * ldw];
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            a0 = __builtin_fmaf(yh[c + u], wv[u], a0);
            a1 = __builtin_fmaf(yh[c + u + 1], wv[u + 1], a1);
            a2 = __builtin_fmaf(yh[c + u + 2], wv[u + 2], a2);
            a3 = __builtin_fmaf(yh[c + u + 3], wv[u + 3], a3);
        }
    }
    for (; c + 3 < D; c += 4) {
        a0 = __builtin_fmaf(yh[c], wp[(long)c * ldw], a0);
        a1 = __builtin_fmaf(yh[c + 1], wp[(long)(c + 1) * ldw], a1);
        a2 = __builtin_fmaf(yh[c + 2], wp[(long)(c + 2) * ldw], a2);
        a3 = __builtin_fmaf(yh[c + 3], wp[(long)(c + 3) * ldw], a3);
    }
    for (; c < D; ++c) a0 = __builtin_fmaf(yh[c], wp[(long)c * ldw], a0);
    if (b0 + bb < B) Oc[(long)(b0 + bb) * att + h * ATT_HD + d] = ((a0 + a1) + (a2 + a3)) + tv[h * ATT_HD + d];
}

// ------------------------------------------------------------------------------------------------
// Final stage of a sampling step (sample.py:508-513): for row b at its slot of this step,
//   logits = Linear(LN(h[row]))[0:22]; p = softmax(logits); s = argmax p / q; tokens[b, slot] = s
// one wave per sequence.  D <= 1024.
// ------------------------------------------------------------------------------------------------
struct HeadW {
    const float* ln_g; const float* ln_b;   // last_norm [D]
    const float* w; const float* b;         // decoder.weight [n_tokens, D] (torch layout), bias [n_tokens]
};
__device__ __forceinline__ void ln_row_regs(const float* __restrict__ x, int D, int lane, const HeadW& w,
                                            float (&y)[16]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { int c = lane + 64 * i; y[i] = (c < D) ? x[c] : 0.f; s += y[i]; }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { int c = lane + 64 * i; if (c < D) { float t = y[i] - mean; q += t * t; } }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int c = lane + 64 * i;
        y[i] = (c < D) ? (y[i] - mean) * rs * w.ln_g[c] + w.ln_b[c] : 0.f;
    }
}

// NW waves per sequence: each normalises the row (the same arithmetic, so the same values) and takes the decoder rows
// j = wave, wave + NW, ...; wave 0 then runs the softmax and the draw on the 22 logits (one wave per sequence with 22 serial
// dot products was 103 us of latency per step).  x: the row (global or LDS); lg: 32 floats of LDS.  Called by every wave of the block.
template <int NW>
__device__ __forceinline__ void sample_row(const float* x, int D, const HeadW& w, int32_t* __restrict__ tokens, int b, int slot, uint32_t t,
                                           const float* __restrict__ q_noise, int q_rows, int q_off, const RunState* __restrict__ rs, int L,
                                           float* lg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float y[16];
    ln_row_regs(x, D, lane, w, y);
#pragma unroll
    for (int jj = 0; jj < (22 + NW - 1) / NW; ++jj) {
        const int j = wave + NW * jj;
        if (j < 22) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { int c = lane + 64 * i; if (c < D) a += y[i] * w.w[(long)j * D + c]; }
            a = wave_sum(a) + w.b[j];
            if (lane == 0) lg[j] = a;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    const float mylogit = lane < 22 ? lg[lane] : -INFINITY;
    const float mx = wave_max(mylogit);
    const float e = (lane < 22) ? expf(mylogit - mx) : 0.f;
    const float esum = wave_sum(e);
    // a NaN / infinite logit makes the sum NaN (the reference's torch.multinomial raises on such a row, sample.py:512); the
    // step still writes a token, hd_sample_end reports the flag
    if (lane == 0 && !(esum > 0.f && esum < INFINITY)) atomicOr(const_cast<uint32_t*>(&rs->pad[0]), 1u);
    const float p = e / esum;
    float q = 1.f;
    if (lane < 22) {
        if (q_noise) {
            q = q_noise[((long)t * q_rows + q_off + b) * 22 + lane];     // [Tmax, rows of the whole batch, 22]
        } else {
            uint32_t o[4];
            philox4x32_10((uint32_t)lane >> 2, rs->row0 + (uint32_t)b, t, 0xFFFFFFFFU, rs->seed_lo, rs->seed_hi, o);
            const float u = ((float)(o[lane & 3] >> 8) + 0.5f) * 5.9604644775390625e-08f;
            q = -logf(u);
        }
    }
    float ratio = (lane < 22) ? p / q : -INFINITY;
    int best = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float r2 = __shfl_xor(ratio, o);
        int b2 = __shfl_xor(best, o);
        if (r2 > ratio || (r2 == ratio && b2 < best)) { ratio = r2; best = b2; }
    }
    if (lane == 0) tokens[(long)b * L + slot] = best;
}

// advance != 0: the last workgroup to finish moves rs->step on (every thread of the grid has read it by then), which saves the
// one-thread launch per step that advance_step_k was.
constexpr int SS_WAVES = 16;
__global__ void __launch_bounds__(64 * SS_WAVES) sample_step_k(const float* __restrict__ Hm, int D, HeadW w,
                                                     int32_t* __restrict__ tokens,
                                                     const int32_t* __restrict__ order,
                                                     const int32_t* __restrict__ T, int Tmax,
                                                     const float* __restrict__ q_noise, int q_rows, int q_off,
                                                     RunState* __restrict__ rs, Segs sg, int compact, int advance) {
    __shared__ float lg[32];
    const int b = blockIdx.x;
    const uint32_t t = __hip_atomic_load(&rs->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)t < T[b]) {
        const int slot = order[(long)b * Tmax + t];
        // compact: Hm is [B, D] holding only the visited row of each sequence (pruned last block)
        sample_row<SS_WAVES>(Hm + (compact ? (long)b : (long)sg.row(b, slot)) * D, D, w, tokens, b, slot, t, q_noise, q_rows, q_off, rs, sg.L, lg);
    } else {
        __syncthreads();                                 // (sample_row has one: every wave has read the step before thread 0 goes on)
    }
    if (advance && threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&rs->done, 1u) == gridDim.x - 1) {
            __hip_atomic_store(&rs->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&rs->step, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Full decoder for hd_forward: logits[b, l, :] = Linear(LN(h[row(b,l)])) , one wave per (b, l).
__global__ void __launch_bounds__(256) decode_all_k(const float* __restrict__ Hm, int D, HeadW w, int n_tokens,
                                                     float* __restrict__ logits, Segs sg) {
    const int bl = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (bl >= sg.B * sg.L) return;
    const int b = bl / sg.L, l = bl - b * sg.L;
    float y[16];
    ln_row_regs(Hm + (long)sg.row(b, l) * D, D, lane, w, y);
    for (int j = 0; j < n_tokens; ++j) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { int c = lane + 64 * i; if (c < D) a += y[i] * w.w[(long)j * D + c]; }
        a = wave_sum(a) + w.b[j];
        if (lane == 0) logits[(long)bl * n_tokens + j] = a;
    }
}

__global__ void set_step_k(RunState* rs, uint32_t step) { rs->step = step; rs->done = 0; }
__global__ void advance_step_k(RunState* rs) { rs->step += 1; }

}  // namespace hd
