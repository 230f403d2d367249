/*
 * hudiff_hip.h -- C ABI of libhudiff_hip.so: HuDiff's denoiser forward + T-step sampling loop on
 * MI355X (gfx950), hand-written HIP.  This is the drop-in boundary for ONE path of the reference:
 *
 *   reference interface replaced                          (file:line under /root/reference)
 *   ------------------------------------------------------------------------------------------------
 *   model_selected(config)  -> nn.Module                   utils/train_utils.py:43-55
 *   model.load_state_dict(ckpt['model'])                   antibody_scripts/sample.py:456-458
 *                                                          nanobody_scripts/nanosample.py:252-288
 *   model(H_L_seq, H_L_region_type, H_L_chn_type)          model/encoder/model.py:366-384
 *        -> logits[B, L, n_tokens]                         model/nanoencoder/model.py:325-343
 *   the sampling loop (forward, softmax[:, i, :22],        antibody_scripts/sample.py:499-513
 *        torch.multinomial, tokens[:, i] = s)              nanobody_scripts/nanosample.py:316-329
 *
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.  Every function returns an
 * HdStatus (0 = ok); hd_last_error() gives the message of the last failure on the calling thread.
 * A handle is bound to one device and one HIP stream and is NOT re-entrant; use one handle per GPU.
 * All tensors are caller-owned host memory unless a name ends in _dev; nothing is retained after return
 * (except by hd_load_tensor, which copies).
 *
 * Row / slot conventions: L = cfg.max_len IMGT slots per row (291 = 152 heavy + 139 light for the
 * antibody model, 152 for the nanobody model); tokens in [0,22] (utils/tokenizer.py:55-62: 20 residues,
 * X=20, '-'=21, <msk>=22); region in [0, n_region); chain type in {0 (H), 1 (L), 2 (K)} laid out as
 * chain[0:B] = heavy rows, chain[B:2B] = light rows (sample.py:172-176).
 *
 * Noise ("what makes sampled ids bit-exact under a fixed seed"):
 *   sampling  s = argmax_j softmax(logits[slot, 0:22])_j / q_j   (first maximum wins), q ~ Exp(1).
 *             q is either injected (q_noise) or generated:  Philox4x32-10, key = seed,
 *             counter = (j >> 2, global_row, step, 0xFFFFFFFF), word j & 3,
 *             u = ((w >> 8) + 0.5) * 2^-24,  q = -log(u).
 *   dropout   (the reference runs F.dropout with training=True at inference whenever cfg.dropout > 0,
 *             model/encoder/model.py:176-178, 295-303): keep-mask either injected or generated:
 *             (k0, k1) = Philox4x32-10(counter = (0, 0, step, site), key = seed)[0:2],
 *             rk = mix32(k0 ^ mix32(global_row + k1)),  w = mix32(rk + (slot * width + feature) * 0x9E3779B9),
 *             keep <=> w >= floor(p * 2^32);  kept values are scaled by 1/(1-p).
 *             site = layer for the token encoder (p = cfg.dropout), 64 + layer for Dual/NanoConv (p = 0.5).
 *             mix32(x): x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16.
 *   global_row = row0 + b, so results do not depend on how rows are sharded over GPUs.
 */
#ifndef HUDIFF_HIP_H
#define HUDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HD_ABI_VERSION 1      /* layout of HdConfig; rounds 4-5 added entry points only (hd_set_precision, hd_precision_report, hd_precision_reset,
                                 hd_set_option, hd_get_option, hd_debug_scatter_lnsync) */

typedef enum HdStatus {
    HD_OK = 0,
    HD_ERR_INVALID = 1,      /* bad argument / value out of range (token, region, chain, L, ...) */
    HD_ERR_UNSUPPORTED = 2,  /* configuration outside what the kernels implement                 */
    HD_ERR_STATE = 3,        /* call order (e.g. forward before finalize, missing tensors)       */
    HD_ERR_HIP = 4,          /* a HIP runtime call failed; message has the hipError string       */
    HD_ERR_NO_DEVICE = 5,    /* no usable gfx950 device: the product path never falls back to CPU */
    HD_ERR_NUMERIC = 6       /* hd_sample / hd_sample_end: a visited row had NaN / infinite logits at some step (the
                                reference's torch.multinomial raises there, sample.py:512); tokens are still returned */
} HdStatus;

enum { HD_KIND_ANTIBODY = 0, HD_KIND_NANOBODY = 1 };
enum { HD_ACT_RELU = 1, HD_ACT_GELU = 2 };

/* flags for hd_forward / hd_sample */
enum {
    HD_DROPOUT_FAITHFUL = 0u,  /* default: dropout active iff cfg.dropout > 0, generated masks      */
    HD_DROPOUT_OFF      = 1u,  /* sites are identity                                                */
    HD_DROPOUT_INJECT   = 2u,  /* caller supplies keep-masks (parity tests)                         */
    HD_DROPOUT_MASK     = 3u,
    HD_NO_GRAPH         = 4u,  /* launch kernels eagerly instead of replaying the captured hipGraph */
    HD_NO_PRUNE         = 8u,  /* hd_sample: evaluate the last attention block for every row (as hd_forward
                                  does) instead of only for the row each sequence visits at that step      */
    HD_ONE_LANE         = 16u, /* hd_sample: keep the batch on one stream (default: batches >= 16 rows are
                                  split into two halves that run concurrently on two streams)              */
    HD_LOOP_GRAPH       = 32u  /* hd_sample / hd_sample_run: the whole T-step loop of a lane is ONE hipGraph (a chain
                                  of T child-graph nodes of the captured step) launched once, instead of T replays
                                  of the step graph (same results; measured 2.4 % slower on MI355X) */
};

/* Hyper-parameters: the `model:` section of configs/antibody_train.yml:3-24 / heavy_train.yml:3-21,
 * i.e. what sample.py reads from ckpt['config'|'pretrain_config'].model. */
typedef struct HdConfig {
    int32_t abi_version;       /* HD_ABI_VERSION */
    int32_t kind;              /* HD_KIND_* */
    int32_t n_tokens;          /* 23 */
    int32_t max_len;           /* 291 | 152 */
    int32_t h_len;             /* heavy slots: 152 (antibody: light = max_len - h_len) ; nanobody: max_len */
    int32_t d_model;           /* 256  (= d_embedding = s_model = r_model = n_pos_model) */
    int32_t sum_d_model;       /* 768 | 512 */
    int32_t n_encoder_layers;  /* 6 */
    int32_t dual_layers;       /* 6 */
    int32_t kernel_size;       /* 7 */
    int32_t r;                 /* 128: dilation of layer n = 2^(n mod (log2(r)+1)) */
    int32_t att_model;         /* 512 */
    int32_t nhead;             /* 8  (att_model / nhead must be 64) */
    int32_t dim_feedforward;   /* 256 */
    int32_t cs_layers;         /* 5 */
    int32_t n_region;          /* 7 */
    int32_t r_embedding;       /* 4 */
    int32_t n_side;            /* 3 (antibody only) */
    int32_t s_embedding;       /* 4 (antibody only) */
    int32_t enc_act;           /* HD_ACT_*: config.activation (token encoder)                        */
    int32_t conv_act;          /* HD_ACT_*: relu for DualConv (model.py:345), gelu for NanoConv       */
    float   dropout;           /* config.dropout */
} HdConfig;

typedef struct HdModel HdModel;

/* ---- construction (replaces model_selected + load_state_dict) ---------------------------------- */
int hd_device_count(void);
HdStatus hd_create(const HdConfig* cfg, int device, HdModel** out);
/* One call per state_dict entry, `key` = the reference's own key, data = float32 in
 * the torch layout ([out,in] Linear, [out,in,k] Conv1d).  The checkpoint's buffers are used when given:
 * 'pos_encoder.pos_embedding.pe' [L,1,d] and '...rope' (complex64 [L,32] passed as float32 [L,32,2],
 * identical in every layer); when absent they are recomputed.  Unknown keys -> HD_ERR_INVALID (strict). */
HdStatus hd_load_tensor(HdModel* m, const char* key, const float* data, const int64_t* shape, int32_t ndim);
/* Checks every required key was loaded with the right shape, re-lays weights for the kernels and uploads. */
HdStatus hd_finalize(HdModel* m);
void hd_destroy(HdModel* m);
const char* hd_last_error(void);

/* ---- one denoiser forward (replaces model(tokens, region, chain)) -------------------------------
 * logits: [B, L, 23] float32.  chain: [2B] (antibody) or NULL (nanobody).
 * enc_masks [n_encoder_layers, B, L, d_model], conv_masks [dual_layers, B, L, sum_d_model] uint8
 * (1 = keep) are read only with HD_DROPOUT_INJECT.  seed/row0/step key generated masks. */
HdStatus hd_forward(HdModel* m, const int32_t* tokens, const int32_t* region, const int32_t* chain,
                    int32_t B, uint32_t flags, uint64_t seed, uint64_t row0, uint32_t step,
                    const uint8_t* enc_masks, const uint8_t* conv_masks, float* logits);

/* ---- the whole T-step sampling loop (replaces sample.py:499-513) --------------------------------
 * tokens [B, L] in/out.  order [B, Tmax]: slot visited by row b at step t; T [B]: steps of row b
 * (rows with t >= T[b] are left untouched at step t; T[b] == 0 is allowed).
 * q_noise [Tmax, B, 22] or NULL.  With HD_DROPOUT_INJECT: enc_masks [Tmax, n_enc, B, L, d],
 * conv_masks [Tmax, n_conv, B, L, D]. */
HdStatus hd_sample(HdModel* m, int32_t* tokens, const int32_t* region, const int32_t* chain,
                   const int32_t* order, const int32_t* T, int32_t B, int32_t Tmax, uint32_t flags,
                   uint64_t seed, uint64_t row0, const float* q_noise,
                   const uint8_t* enc_masks, const uint8_t* conv_masks);

/* Split form of hd_sample so that a caller can time the device-resident part:
 *   begin : validates, uploads inputs, computes the token-independent embedding branch
 *   run   : enqueues steps [t0, t1) on the handle's stream and returns without synchronising
 *   end   : synchronises and copies the tokens back                                                  */
HdStatus hd_sample_begin(HdModel* m, const int32_t* tokens, const int32_t* region, const int32_t* chain,
                         const int32_t* order, const int32_t* T, int32_t B, int32_t Tmax, uint32_t flags,
                         uint64_t seed, uint64_t row0, const float* q_noise,
                         const uint8_t* enc_masks, const uint8_t* conv_masks);
HdStatus hd_sample_run(HdModel* m, int32_t t0, int32_t t1);
/* Restores the tokens to their state at hd_sample_begin (device-to-device) and re-keys the noise with a new
 * seed, so that the same resident batch can be sampled again (independent replicas; bench.py). */
HdStatus hd_sample_restart(HdModel* m, uint64_t seed);
HdStatus hd_sample_end(HdModel* m, int32_t* tokens);
HdStatus hd_sync(HdModel* m);
/* Copies the tokens of the open session as they stand after the steps enqueued so far (synchronises; the session stays open):
 * lets a caller that restarts one resident batch many times keep every sample's result (bench.py: token agreement between
 * precision routes over all timed samples).  A guard that fired (see "precision routes") makes this fail with HD_ERR_STATE --
 * only hd_sample_end repeats a sample. */
HdStatus hd_sample_tokens(HdModel* m, int32_t* tokens);

/* ---- measurement helpers ------------------------------------------------------------------------
 * hd_sample_run brackets the steps it enqueues with HIP events on the handle's stream;
 * hd_last_run_ms returns the elapsed device time of the last completed run (after hd_sync/hd_sample_end). */
HdStatus hd_last_run_ms(HdModel* m, float* ms, int32_t* steps);
/* Algorithmic FLOPs of one forward of one row (SURVEY.md §8d formula). */
double hd_flops_per_row_forward(const HdConfig* cfg);
/* FLOPs one hd_sample step actually executes per row (last attention block pruned to the visited row, its value side
 * taken through the input rows instead of a V projection of every row). */
double hd_flops_per_row_sample_step(const HdConfig* cfg);
/* ---- precision routes --------------------------------------------------------------------------------
 * The reference computes in fp32 (PyTorch CPU / CUDA defaults, model/encoder/model.py:366-384).  gfx950 multiplies fp32 operands
 * on the matrix cores at 1/16 of the fp16 rate and has no TF32-like mode, so the library has three routes through the same
 * kernels' interfaces; all three hold the 1e-4 logit bound and reproduce the reference's recorded sampling traces bit for bit
 * (tests/test_prod_trace.py, tests/test_gpu_adversarial.py):
 *   HD_PRECISION_SPLIT     every large GEMM and the attention core as THREE fp16 MFMAs per fp32 product: each operand is hi + lo with
 *                          hi = fp16(x), lo = fp16(x - hi) (22 significand bits; x - hi is exact), a w ~= a_hi w_hi + a_hi w_lo +
 *                          a_lo w_hi with fp32 accumulation -- as close to the exact dot product as the fp32 kernels.  Launches of
 *                          fewer than 128 activation rows (none of the two models' single sequences), the pruned tail's compact GEMMs and the
 *                          static branch run the fp32 kernels.
 *   HD_PRECISION_F32_GEMM  GEMMs on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32); only the attention core (QK^T, PV) of launches
 *                          >= 8192 rows as three fp16 MFMAs per product (the round-3 default)
 *   HD_PRECISION_F32_ALL   every product on the fp32 MFMA pipe (rounds 1-2)
 *   HD_PRECISION_DEFAULT   what a handle starts with: the library default, HD_PRECISION_SPLIT since round 4 -- unless the environment
 *                          of hd_finalize overrides the DEFAULT (only the default; an explicit hd_set_precision wins):
 *                          HUDIFF_PRECISION=split|f32_gemm|f32_all, or the older switches HUDIFF_X3=0|1 (GEMMs) and
 *                          HUDIFF_ATTN_X3=0|1 (attention core).
 * hd_set_precision must be called before hd_finalize (the split weight images are built there); afterwards HD_ERR_STATE.
 *
 * Guards of the split kernels -- never an error, never a silently wrong row; the CALL is repeated inside the library and the event
 * is counted:
 *   range guard   split operands are not scaled, so the split is valid for |x| < 65504 only.  Every producer of a split checks
 *                 its values; when one is out of range, hd_forward / hd_sample[_end] repeats the whole call on the fp32 kernels (same
 *                 resident inputs, same noise key, same steps) and the handle stays on them (weights whose stream leaves the
 *                 range do so at every step) until hd_precision_reset.  hd_sample_restart and hd_sync notice the flag as well.
 *   ln_sync guard the ByteNet GEMMs of the split route normalise their own output rows: the N tiles of an M tile exchange LayerNorm
 *                 partials at a counter (write-through stores, agent-scope loads: correct wherever the blocks run), which presumes
 *                 they are co-resident.  A meeting that times out raises a flag; the call is repeated with separate LayerNorm passes
 *                 (ln_apply_k) and the handle keeps those.                                                                          */
enum { HD_PRECISION_DEFAULT = 0, HD_PRECISION_F32_GEMM = 1, HD_PRECISION_F32_ALL = 2, HD_PRECISION_SPLIT = 3 };
HdStatus hd_set_precision(HdModel* m, int32_t precision);
typedef struct HdPrecisionInfo {
    int32_t precision;          /* resolved route of the handle: HD_PRECISION_SPLIT / F32_GEMM / F32_ALL (never DEFAULT after hd_finalize) */
    int32_t split_built;        /* bit 0: split-precision weight images exist (GEMMs), bit 1: split-precision attention core     */
    int32_t split_in_use;       /* 1 while eligible launches take the split kernels, 0 after the range guard switched them off   */
    int32_t lnsync_in_use;      /* 1 while the split ByteNet GEMMs normalise their own outputs, 0 = separate ln_apply_k passes   */
    int64_t range_fallbacks;    /* calls repeated on the fp32 kernels by the range guard                                          */
    int64_t lnsync_fallbacks;   /* calls repeated with ln_apply_k passes because an ln_sync meeting failed                        */
    int32_t last_call_repeated; /* 1 if the last hd_forward / hd_sample_end repeated its call (hd_last_run_ms then times the repeat) */
    int32_t lnsync_cross_xcd;   /* 1 once an ln_sync meeting saw its blocks on two XCDs: still correct (write-through hand-over), slower */
} HdPrecisionInfo;
/* size = sizeof(HdPrecisionInfo) of the caller (fields beyond it are not written) */
HdStatus hd_precision_report(HdModel* m, HdPrecisionInfo* out, size_t size);
/* The three original fields of the report (kept for round-3 callers). */
HdStatus hd_precision_info(HdModel* m, int32_t* split_built, int32_t* split_in_use, int64_t* range_fallbacks);
/* Puts a handle whose guards switched kernels off back on its configured route (between calls; HD_ERR_STATE inside a session). */
HdStatus hd_precision_reset(HdModel* m);
/* ---- tuning options ----------------------------------------------------------------------------------
 * The knobs that SELECT KERNELS or change how a batch is scheduled are part of the ABI (round 5; rounds 1-4 read them from the
 * environment only).  Every option has a library default (the value the published numbers use).  Precedence, as for the precision
 * route: an explicit hd_set_option wins; an option nobody set takes the environment variable named beside it if that is exported
 * when hd_create runs, else the default.  hd_set_option is legal between calls (HD_ERR_STATE inside a sampling session; options
 * marked [create] only before hd_finalize); it drops the handle's captured graphs.  Values outside [lo, hi] -> HD_ERR_INVALID.
 * None of them changes results beyond the last-ulp reassociation documented in INTEGRATION.md (tile shapes share one K order).     */
typedef enum HdOption {
    HD_OPT_LANES = 0,               /* HUDIFF_LANES            2     [1, 4]   lanes (stream + workspace + graph) a sampling batch is split into          */
    HD_OPT_LANE_MIN_ROWS = 1,       /* HUDIFF_LANE_MIN_B       16    [2, ..]  fewest rows of a batch that is split into lanes                           */
    HD_OPT_SPLIT_MIN_ROWS = 2,      /* HUDIFF_X3_ROWS          128   [1, ..]  fewest activation rows of a launch that takes the split-precision kernels   */
    HD_OPT_BIG_MIN_ROWS = 3,        /* HUDIFF_BIG_ROWS         8192  [1, ..]  fewest activation rows of a launch that takes the fp32 128-row-tile kernels */
    HD_OPT_LNSYNC_LEVEL = 4,        /* HUDIFF_X3_LNSYNC        2     [0, 2]   split ByteNet GEMMs normalise their own output: 0 no (ln_apply_k passes),
                                                                             1 the two inner GEMMs of a block, 2 the block's last GEMM as well           */
    HD_OPT_TAIL_FORM = 5,           /* HUDIFF_TAIL             2     {0, 2}   pruned tail of a sampling step: 0 twelve separate launches, 2 five sliced ones */
    HD_OPT_TAIL_MAX_ROWS = 6,       /* HUDIFF_TAIL_MAX_B       64    [0, ..]  largest lane that takes the sliced tail                                    */
    HD_OPT_SMALL_GRID = 7,          /* HUDIFF_X3_SMALL_GRID    320   [0, ..]  largest 128 x 128 grid of a split GEMM that takes 64 x 128 tiles instead    */
    HD_OPT_TINY_GRID = 8,           /* HUDIFF_X3_TINY_GRID     150   [0, ..]  largest 64 x 128 grid that takes 32 x 128 tiles instead                    */
    HD_OPT_LOADER_WAVES = 9,        /* HUDIFF_X3_LOADERS       1     [0, 1]   32 x 128 blocks carry four DMA-issuing waves while they fit one per CU      */
    HD_OPT_TINY_STAGES = 10,        /* HUDIFF_X3_TINY_NS       3     [2, 3]   LDS stages of the 32 x 128 split tiles                                     */
    HD_OPT_SMALL_STAGES = 11,       /* HUDIFF_X3_SMALL_NS      0     {0,2,3}  LDS stages of the 64 x 128 split tiles; 0 = three up to the grid below      */
    HD_OPT_SMALL_STAGES3_MAX_GRID = 12, /* HUDIFF_X3_SMALL_NS3_MAX 256 [0, ..]                                                                           */
    HD_OPT_ATTN_QSPLIT_MAX_GRID = 13,   /* HUDIFF_ATTN_QSPLIT_MAX 128 [0, ..] largest (sequence, head) grid whose query tiles are shared by two workgroups */
    HD_OPT_ATTN_WAVES = 14,         /* HUDIFF_ATTN_WAVES       12    {8, 12}  waves of the split attention core for L in (288, 304]                       */
    HD_OPT_LOOP_GRAPH = 15,         /* HUDIFF_LOOP_GRAPH       0     [0, 1]   1 = every hd_sample_run behaves as if HD_LOOP_GRAPH were set                 */
    HD_OPT_PRUNE_VALUE_VIA_ROWS = 16,   /* HUDIFF_PRUNE_V      1     [0, 1]   pruned last block: value side through the input rows (no V projection)      */
    HD_OPT_SPLIT_TILE = 17,         /* HUDIFF_X3_TILE          0     {0,128,256,512} force the big split GEMM tile: 128 x 128 / 256 x 128 / 256 x 256; 0 = by shape */
    HD_OPT_GEMM_SMALL_TILES = 18,   /* HUDIFF_GEMM_SMALL       1536  [0, ..]  fp32 launches with fewer 128-row tiles take 64 x 128 tiles                  */
    HD_OPT_STORE_NT = 19,           /* HUDIFF_ST_NT            0     [0, 1]   non-temporal epilogue stores in the fp32 GEMMs (the split GEMMs always use them) */
    HD_OPT_SPLIT_LAYER_MASK = 20,   /* HUDIFF_X3_MASK          3     [0, 3]   [create] bit 0 ByteNet blocks, bit 1 attention blocks get split weight images */
    HD_OPT_SPLIT_ATTN = 21,         /* HUDIFF_X3_ATTN          1     [0, 1]   split route: 0 keeps the fp32 attention core                                */
    HD_OPT_FUSED_ATTN = 22,         /* HUDIFF_FUSED_ATTN       1     [0, 1]   split route: the Q|K|V projection runs inside the attention kernel (one workgroup per
                                                                             (sequence, head group), K / V never leave the CU); 0 = projection GEMM + attention core    */
    HD_OPT_FUSED_ATTN_MIN_GRID = 23,/* HUDIFF_FUSED_ATTN_MIN_GRID 128 [0, ..] fewest (sequence, head group) workgroups -- over all lanes -- for which the fused
                                                                             form is taken (below, the two-launch form's finer tiles fill the chip better)        */
    HD_OPT_BN_CHAIN = 24,           /* HUDIFF_BN_CHAIN         0     [0, 3]   [create] split route: ByteNet stacks on the row-owner chain kernel (a wave owns whole rows
                                                                           through conv -> LN -> PFF3 -> residual -> LN -> PFF1 -> LN; n + 1 launches per stack): bit 0
                                                                           Dual / NanoConv, bit 1 token encoder; 0 (default: measured no faster, NOTES.md E) = three
                                                                           gemm_x3_k launches per block                                                                        */
    HD_OPT_BN_CHAIN_MIN_TILES = 25, /* HUDIFF_BN_CHAIN_MIN_TILES 128 [0, ..]  fewest workgroup tiles (128 rows; token encoder 256), over all lanes, that take the chain kernel */
    HD_OPT_COUNT = 26
} HdOption;
HdStatus hd_set_option(HdModel* m, int32_t option, int64_t value);
HdStatus hd_get_option(HdModel* m, int32_t option, int64_t* value);

/* Device facts for the bench JSON. */
HdStatus hd_device_info(int device, char* name, size_t name_len, int32_t* cu_count, int64_t* hbm_bytes);

/* ---- debugging aids (tests only) -----------------------------------------------------------------
 * hd_debug_stop_after: make hd_forward return after a stage (0 = off, 1 = token encoder, 2+n = before
 * attention block n, 100+n = right behind the first attention of block n); hd_debug_read copies an activation
 * buffer ("X","FEAT","Y","POS","EXTRA","AT","O","QKV"; split route also "ATX","YX") of the last call back as fp32
 * [B, L, width].  "AT" is x + A1(x) behind a block's first attention and at + A2(LN1(at)) behind its second on every
 * route: the split route keeps the second sum in split form only and the call decodes it (hi + lo); "O", "ATX",
 * "YX" are decoded the same way when the split kernels wrote them. */
HdStatus hd_debug_stop_after(HdModel* m, int32_t stage);
/* Makes the ln_sync meetings of the next call give up after one poll, so that the ln_sync guard (see "precision routes") fires and
 * its repeat path can be tested; cleared when the guard has fired. */
HdStatus hd_debug_fail_next_lnsync(HdModel* m);
HdStatus hd_debug_read(HdModel* m, const char* name, int32_t B, float* out, int64_t n_floats);
/* on != 0: the ln_sync launches deal the N tiles of an M tile to DIFFERENT XCDs (consecutive workgroup ids) instead of one XCD's
 * consecutive slots, so that the cross-XCD path of the meeting (write-through partials, agent-scope loads) is the one that runs;
 * results must be bit-identical to the normal placement and hd_precision_report then says lnsync_cross_xcd = 1. */
HdStatus hd_debug_scatter_lnsync(HdModel* m, int32_t on);

#ifdef __cplusplus
}
#endif
#endif /* HUDIFF_HIP_H */
