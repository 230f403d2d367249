// hd_api.hip -- host side of libhudiff_hip.so: the C ABI declared in include/hudiff_hip.h.
//
// Replaces, for the sampling path only, the reference's
//   model_selected + load_state_dict   (utils/train_utils.py:43-55, antibody_scripts/sample.py:456-458)
//   AntiTFNet.forward / NanoAntiTFNet.forward   (model/encoder/model.py:366-384, model/nanoencoder/model.py:325-343)
//   the T-step loop                    (antibody_scripts/sample.py:499-513, nanobody_scripts/nanosample.py:316-329)
// There is no CPU fallback: without a gfx950 device every entry point fails with HD_ERR_NO_DEVICE.
#include "../../include/hudiff_hip.h"
#include "hd_kernels.hip.h"
#include "hd_tail_fused.hip.h"
#include "hd_attn_fused.hip.h"
#include "hd_chain.hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace hd;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static thread_local std::string g_launch_err;       // set by a launch helper that refused to launch (checked by forward_body)

static HdStatus fail(HdStatus st, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(HD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define HD_TRY(expr)                                                                               \
    do {                                                                                           \
        HdStatus s_ = (expr);                                                                      \
        if (s_ != HD_OK) return s_;                                                                \
    } while (0)

extern "C" const char* hd_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

// A weight matrix in the split-precision form gemm_x3_k consumes (HUDIFF_X3=1): pre-tiled (hi, lo) fp16 planes,
// per-segment stride in halfs, and the power-of-two factor that undoes weight and activation scaling.
struct X3W { const uint16_t* w = nullptr; long seg_stride = 0; float acc_scale = 1.f; long ntile_stride = 0; };
struct ByteNetW {   // one ByteNet block, both segments packed back to back
    const float *ln1_g, *ln1_b, *w1, *b1, *ln2_g, *ln2_b, *wc, *bc, *ln3_g, *ln3_b, *w3, *b3;
    int dil;
    X3W wcx, w1x, w3x;
    X3W w3px;           // w3 with the k order of hd_chain.hip.h's phase B (permuted inside every group of 16)
};
struct AttLayerW { const float *wqkv, *bqkv, *wo, *bo; X3W wqkvx, wox; };
struct AttBlockW {
    AttLayerW a1, a2;
    const float *n1_g, *n1_b, *n2_g, *n2_b, *wf1, *bf1, *wf2, *bf2;     // a2.wqkv / bqkv and wf1 / bf1 hold norm_hl1 / norm_hl2 folded in
    X3W wf1x, wf2x;
};

struct Workspace {
    int capB = 0;
    float *X = nullptr, *H1 = nullptr, *H2 = nullptr;        // token encoder: [M,d], [M,dh] x2
    float *FEAT = nullptr, *Y = nullptr, *G1 = nullptr, *G2 = nullptr;  // conv stage: [M,D] x2, [M,Dh] x2
    float *QKV = nullptr, *O = nullptr, *AT = nullptr, *F1 = nullptr;   // attention stage
    float *EXTRA = nullptr, *POS = nullptr, *PH = nullptr;   // static branch: [M,d], [M,d], [M,2d]
    float *S1 = nullptr, *YX = nullptr, *ATX = nullptr;      // split-precision route: act(LN(x)) of a ByteNet block's input; split copies of Y / AT, [M,D]
    bool at_in_atx = false, o_is_split = false;              // what the launches issued last left (hd_debug_read): the second attention's sum only in ws.ATX; ws.O as X16 rows
    float2* ST = nullptr;
    int* SYNC = nullptr;                                     // ln_sync meeting counters, 4 ints per (segment, M tile): arrivals, departures, XCC-id mask, spare; zero between launches
    float2* PART[2] = {nullptr, nullptr};                    // ping-pong [PART_STRIDE][M] LayerNorm partials from GEMM epilogues
    int part_next = 0;                                       // buffer the next producing GEMM writes
    const float2* part_last = nullptr; int part_last_pw = 0; // what the last producing GEMM wrote (for its consumer)
    float* LOGITS = nullptr;                                 // [B, L, n_tokens] (hd_forward)
    float *ATc = nullptr, *Xc = nullptr, *Qc = nullptr, *Oc = nullptr, *F1c = nullptr;   // pruned last block: [B, *]
    float2* STc = nullptr;
    float *PW = nullptr, *YV = nullptr;                      // pruned last block: p_j rstd_j [B, nhead, 320], weighted input rows [B, nhead, D]
    int32_t *tokens = nullptr, *tokens0 = nullptr, *region = nullptr, *chain = nullptr, *order = nullptr, *T = nullptr;
    int capT = 0;
    uint8_t *enc_masks = nullptr, *conv_masks = nullptr; size_t enc_cap = 0, conv_cap = 0;
    std::vector<void*> owned;
};

constexpr int HD_MAX_LANES = 4;

// ---- tuning options (include/hudiff_hip.h "tuning options") -----------------------------------------------------------------------
// One table: id, the environment variable that overrides the DEFAULT of an option nobody set, default, legal range (`set` = the
// legal values when they are not a range), `create_only` = fixed at hd_finalize.  hd_create fills HdModel::opt from it.
struct OptDef { int id; const char* env; int64_t def, lo, hi; int64_t set[4]; int nset; bool create_only; };
constexpr int64_t OPT_BIG = (int64_t)1 << 40;
static const OptDef OPTS[HD_OPT_COUNT] = {
    {HD_OPT_LANES, "HUDIFF_LANES", 2, 1, HD_MAX_LANES, {}, 0, false},
    {HD_OPT_LANE_MIN_ROWS, "HUDIFF_LANE_MIN_B", 16, 2, OPT_BIG, {}, 0, false},      // two lanes pay from 16 rows on (round 4 sweeps: B = 16 50.8 -> 52.9 sequences/s; B = 8 loses 8 %)
    {HD_OPT_SPLIT_MIN_ROWS, "HUDIFF_X3_ROWS", 128, 1, OPT_BIG, {}, 0, false},       // the split kernels pay from a single sequence on (tiles sized to the grid, launch_gemm)
    {HD_OPT_BIG_MIN_ROWS, "HUDIFF_BIG_ROWS", 8192, 1, OPT_BIG, {}, 0, false},       // the fp32 big-launch kernels gain nothing below 8192 rows
    {HD_OPT_LNSYNC_LEVEL, "HUDIFF_X3_LNSYNC", 2, 0, 2, {}, 0, false},
    {HD_OPT_TAIL_FORM, "HUDIFF_TAIL", 2, 0, 2, {0, 2}, 2, false},
    {HD_OPT_TAIL_MAX_ROWS, "HUDIFF_TAIL_MAX_B", 64, 0, OPT_BIG, {}, 0, false},
    {HD_OPT_SMALL_GRID, "HUDIFF_X3_SMALL_GRID", 320, 0, OPT_BIG, {}, 0, false},     // B = 8 antibodies 20.0 -> 25.1 sequences/s, B = 16 36.5 -> 42.4, B = 48 73.3 -> 78.4; larger limits lose again
                                                                                    // (round 5: 300 -> 320 puts the 304-block launches of 256 nanobodies on 64 x 128 tiles: 446 -> 455 sequences/s)
    {HD_OPT_TINY_GRID, "HUDIFF_X3_TINY_GRID", 150, 0, OPT_BIG, {}, 0, false},
    {HD_OPT_LOADER_WAVES, "HUDIFF_X3_LOADERS", 1, 0, 1, {}, 0, false},
    {HD_OPT_TINY_STAGES, "HUDIFF_X3_TINY_NS", 3, 2, 3, {}, 0, false},
    {HD_OPT_SMALL_STAGES, "HUDIFF_X3_SMALL_NS", 0, 0, 3, {0, 2, 3}, 3, false},
    {HD_OPT_SMALL_STAGES3_MAX_GRID, "HUDIFF_X3_SMALL_NS3_MAX", 256, 0, OPT_BIG, {}, 0, false},
    {HD_OPT_ATTN_QSPLIT_MAX_GRID, "HUDIFF_ATTN_QSPLIT_MAX", 128, 0, OPT_BIG, {}, 0, false},
    {HD_OPT_ATTN_WAVES, "HUDIFF_ATTN_WAVES", 12, 8, 12, {8, 12}, 2, false},
    {HD_OPT_LOOP_GRAPH, "HUDIFF_LOOP_GRAPH", 0, 0, 1, {}, 0, false},
    {HD_OPT_PRUNE_VALUE_VIA_ROWS, "HUDIFF_PRUNE_V", 1, 0, 1, {}, 0, false},
    {HD_OPT_SPLIT_TILE, "HUDIFF_X3_TILE", 0, 0, 512, {0, 128, 256, 512}, 4, false},
    {HD_OPT_GEMM_SMALL_TILES, "HUDIFF_GEMM_SMALL", 1536, 0, OPT_BIG, {}, 0, false},
    {HD_OPT_STORE_NT, "HUDIFF_ST_NT", 0, 0, 1, {}, 0, false},
    {HD_OPT_SPLIT_LAYER_MASK, "HUDIFF_X3_MASK", 3, 0, 3, {}, 0, true},
    {HD_OPT_SPLIT_ATTN, "HUDIFF_X3_ATTN", 1, 0, 1, {}, 0, false},
    {HD_OPT_FUSED_ATTN, "HUDIFF_FUSED_ATTN", 1, 0, 1, {}, 0, false},
    // one workgroup per (sequence, head group): with fewer than ~half the CUs busy the two-launch form's finer tiles win (round 5 sweep,
    // profiles/r05: antibodies B = 8 30.9 vs 34.1 sequences/s, B = 16 52.1 vs 51.4, B = 64 89.0 vs 85.1; nanobodies B = 16 129 vs 138, B = 64 322 vs 293)
    {HD_OPT_FUSED_ATTN_MIN_GRID, "HUDIFF_FUSED_ATTN_MIN_GRID", 128, 0, OPT_BIG, {}, 0, false},
    // round 6: the row-owner chain kernel (hd_chain.hip.h) for ByteNet stacks of the split route: bit 0 Dual / NanoConv, bit 1 token encoder.
    // OFF by default: correct (tests/test_gpu_chain.py) and n + 1 launches per stack instead of 3 n + 1, but measured no faster than the
    // three gemm_x3_k launches per block at 256 antibodies (dual stack 1 058 vs 920 us per block, sample 104.5 vs 108.0 sequences/s) nor
    // at 256 nanobodies (488 vs 486): one wave per SIMD has nobody to cover its DMA issue and LDS latency (NOTES.md E, profiles/r06)
    {HD_OPT_BN_CHAIN, "HUDIFF_BN_CHAIN", 0, 0, 3, {}, 0, true},
    // fewest workgroup tiles (128 rows; 256 for the token encoder) of a launch, over all lanes, for which the chain kernel is taken: one
    // workgroup owns a CU, so a launch that leaves most CUs empty keeps the gemm_x3_k tiles
    {HD_OPT_BN_CHAIN_MIN_TILES, "HUDIFF_BN_CHAIN_MIN_TILES", 128, 0, OPT_BIG, {}, 0, false},
};
static bool opt_legal(const OptDef& d, int64_t v) {
    if (v < d.lo || v > d.hi) return false;
    if (d.nset == 0) return true;
    for (int i = 0; i < d.nset; ++i) if (d.set[i] == v) return true;
    return false;
}

struct HdModel {
    HdConfig cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool finalized = false;
    std::map<std::string, HostTensor> host;
    // derived
    int nseg = 1, d = 0, dh = 0, D = 0, Dh = 0, A = 0, Fd = 0, L = 0;
    float p_enc = 0.f, p_conv = 0.f;
    // device weights
    float* blob = nullptr;
    uint16_t* blobx = nullptr;        // split-precision copies of the GEMM weights (route HD_PRECISION_SPLIT)
    // Precision route (include/hudiff_hip.h "precision routes"): what the caller asked for (hd_set_precision; DEFAULT = the library
    // default unless the environment of hd_finalize overrides it) and what hd_finalize resolved it to.
    int precision_req = HD_PRECISION_DEFAULT, precision = HD_PRECISION_SPLIT;
    int64_t opt[HD_OPT_COUNT] = {};                  // tuning options (OPTS; hd_set_option / hd_get_option)
    bool debug_lnsync_scatter = false;               // hd_debug_scatter_lnsync: the N tiles of an ln_sync M tile go to different XCDs
    bool x3 = false;                  // split-precision GEMMs (weight images built)
    // Range guard of the split-precision kernels: their fp16 (hi, lo) operands are not scaled, so a producer that meets
    // |x| >= 65504 raises RunState::pad[1]; the forward / sample is then re-run on the fp32 kernels and the model stays on
    // them (weights whose residual stream leaves the fp16 range do so at every step): x3_suspended, counted in range_fallbacks.
    bool x3_suspended = false;
    int64_t range_fallbacks = 0;
    bool attn_x3 = true;                             // split-precision attention core (attn_x3_k): routes SPLIT and F32_GEMM
    // ln_sync guard: a failed meeting of a GEMM's N tiles (RunState::pad[2]) repeats the call with ln_apply_k passes and the handle
    // keeps those (lnsync_level 0) until hd_precision_reset; counted in lnsync_fallbacks.
    int lnsync_level_cfg = 2, lnsync_level = 2;      // 0 = ln_apply_k passes, 1 = the two inner GEMMs of a ByteNet block normalise their own output, 2 = the last GEMM as well
    int64_t lnsync_fallbacks = 0;
    bool last_call_repeated = false;                 // the last hd_forward / hd_sample_end repeated its call (either guard)
    bool lnsync_cross_xcd = false;                   // some ln_sync meeting saw its blocks on two XCDs (the placement premise of its speed did not hold)
    const float* emb = nullptr;
    std::vector<ByteNetW> enc, conv;
    std::vector<AttBlockW> att;
    RegionW regw{};
    SideW sidew{};
    const float *pos_w1 = nullptr, *pos_b1 = nullptr, *pos_w2 = nullptr, *pos_b2 = nullptr;
    HeadW head{};
    const float *rope_cos = nullptr, *rope_sin = nullptr, *rope_cs = nullptr;      // [L, 32] cos, sin; the same interleaved [L, 32, 2]
    float* side_vec = nullptr;
    float2* emb_stats = nullptr;      // [n_tokens] LayerNorm (mean, rstd) of each embedding row
    // injected Exp(1) noise of the open session, [Tmax, B, 22] for the WHOLE batch (every lane indexes it by its row
    // offset).  Owned by the model, not by a lane's workspace: a captured graph holds this pointer, so it is part of
    // every lane's graph key and outlives any lane's workspace regrowth.
    float* qnoise = nullptr; size_t qnoise_cap = 0;
    // Two independent "lanes" (stream + workspace + graph): hd_sample splits a batch into two halves that run
    // concurrently on their own streams.  Rows are independent, so there is no cross-lane dependency; one lane's
    // kernel tails / attention staging / launch gaps are filled by the other lane's kernels.
    struct Lane {
        hipStream_t stream = nullptr;
        RunState* rs = nullptr;
        Workspace ws;
        int B = 0, row_off = 0;                      // rows of the open session handled by this lane
        hipGraph_t graph = nullptr;
        hipGraphExec_t graph_exec = nullptr;
        int graph_B = -1; uint32_t graph_flags = 0; int graph_drop = -1; bool graph_q = false; int graph_Tmax = -1;
        int graph_qB = -1, graph_qoff = -1;
        int graph_x3 = -1;                           // kernel_set() when the graph was captured (split kernels in use, ln_sync level)
        const float* graph_qptr = nullptr;           // the injected-noise buffer the captured sample_step_k reads
        // the T-step loop as ONE graph: `loop_steps` child-graph nodes of `graph` in a chain (hd_sample_run)
        hipGraph_t loop_graph = nullptr;
        hipGraphExec_t loop_exec = nullptr;
        int loop_steps = 0;
        void drop_graphs() {
            if (loop_exec) { hipGraphExecDestroy(loop_exec); loop_exec = nullptr; }
            if (loop_graph) { hipGraphDestroy(loop_graph); loop_graph = nullptr; }
            loop_steps = 0;
            if (graph_exec) { hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
            if (graph) { hipGraphDestroy(graph); graph = nullptr; }
            graph_B = -1;
        }
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
    } lane[HD_MAX_LANES];
    int cl = 0;                                      // lane the helper functions currently address
    int nlanes = 1;
    // sampling session
    bool in_session = false;
    int sB = 0, sTmax = 0;
    uint64_t s_row0 = 0;
    uint64_t s_seed = 0;                             // seed of the last hd_sample_begin / hd_sample_restart (range-guard re-run)
    int s_steps = 0;                                 // steps enqueued since then (largest t1 of hd_sample_run)
    uint32_t sflags = 0;
    bool s_has_q = false;
    bool s_dirty = false;                            // a guard fired in the steps run since the last begin / restart: their tokens are invalid
    int last_steps = 0; bool timed = false;
    int debug_stop_after = 0;     // 0 = run everything (hd_debug_stop_after)
    bool debug_lnsync_fail = false;   // hd_debug_fail_next_lnsync: ln_sync meetings of the next call give up after one poll
};

// the lane (stream + workspace + graph) the helper functions currently address
static inline HdModel::Lane& cur(HdModel* m) { return m->lane[m->cl]; }
static inline const HdModel::Lane& cur(const HdModel* m) { return m->lane[m->cl]; }

static int dilation_of(const HdConfig& c, int n) {
    int log2r = 0;
    while ((1 << (log2r + 1)) <= c.r) ++log2r;
    return 1 << (n % (log2r + 1));
}

extern "C" int hd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" double hd_flops_per_row_forward(const HdConfig* c) {
    const double L = c->max_len, d = c->d_model, dh = c->d_model / 2, D = c->sum_d_model, Dh = c->sum_d_model / 2;
    const double A = c->att_model, Fd = c->dim_feedforward, k = c->kernel_size;
    return L * (c->n_encoder_layers * (4 * d * dh + 2 * k * dh * dh) + c->dual_layers * (4 * D * Dh + 2 * k * Dh * Dh) +
                2.0 * c->cs_layers * (8 * D * A + 4 * L * A) + c->cs_layers * 4 * D * Fd + 2 * D * c->n_tokens);
}

extern "C" double hd_flops_per_row_sample_step(const HdConfig* c) {
    // what one step of hd_sample executes per row: the canonical forward minus the parts of the last attention
    // block (2nd attention's query / core / out projection, feed-forward) and of the decoder that are evaluated
    // for the visited row only
    const double L = c->max_len, D = c->sum_d_model, A = c->att_model, Fd = c->dim_feedforward;
    // ... and the V projection of the last attention is replaced by a weighted sum of the input rows per head
    // (2 nhead L D) and one D x A projection for the visited row
    return hd_flops_per_row_forward(c) - (L - 1) * (4 * D * A + 4 * L * A + 4 * D * Fd + 2 * D * c->n_tokens)
           - (L - 1) * 2 * D * A + 2.0 * c->nhead * L * D;
}

extern "C" HdStatus hd_device_info(int device, char* name, size_t name_len, int32_t* cu_count, int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return HD_OK;
}

extern "C" HdStatus hd_create(const HdConfig* cfg, int device, HdModel** out) {
    if (!cfg || !out) return fail(HD_ERR_INVALID, "hd_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != HD_ABI_VERSION)
        return fail(HD_ERR_INVALID, "hd_create: abi_version %d != %d", cfg->abi_version, HD_ABI_VERSION);
    const HdConfig& c = *cfg;
    if (c.kind != HD_KIND_ANTIBODY && c.kind != HD_KIND_NANOBODY) return fail(HD_ERR_INVALID, "hd_create: kind %d", c.kind);
    if (c.att_model != c.nhead * ATT_HD)
        return fail(HD_ERR_UNSUPPORTED, "hd_create: head dim %d/%d unsupported (kernels implement 64)", c.att_model, c.nhead);
    const int nkt = (c.max_len + 15) / 16;
    if (nkt != 19 && nkt != 10)
        return fail(HD_ERR_UNSUPPORTED, "hd_create: max_len %d unsupported (291 or 152)", c.max_len);
    if (c.d_model % 8 || c.sum_d_model % 8 || c.dim_feedforward % 4 || c.d_model > 512 || c.sum_d_model > 1024)
        return fail(HD_ERR_UNSUPPORTED, "hd_create: widths must be multiples of 8 (d_model<=512, sum_d_model<=1024)");
    if (c.r_embedding > 8 || c.s_embedding > 8 || c.n_tokens < 23 || c.n_tokens > 64)
        return fail(HD_ERR_UNSUPPORTED, "hd_create: r_embedding/s_embedding <= 8, n_tokens in [23,64]");
    if (c.kind == HD_KIND_ANTIBODY) {
        if (c.h_len <= 0 || c.h_len >= c.max_len) return fail(HD_ERR_INVALID, "hd_create: h_len %d", c.h_len);
        if (c.sum_d_model != 3 * c.d_model) return fail(HD_ERR_INVALID, "hd_create: sum_d_model != 3*d_model");
    } else {
        if (c.h_len != c.max_len) return fail(HD_ERR_INVALID, "hd_create: nanobody h_len must equal max_len");
        if (c.sum_d_model != 2 * c.d_model) return fail(HD_ERR_INVALID, "hd_create: sum_d_model != 2*d_model");
    }
    if (c.kernel_size != 7 && c.kernel_size != 5 && c.kernel_size != 3)
        return fail(HD_ERR_UNSUPPORTED, "hd_create: kernel_size %d", c.kernel_size);
    if (!(c.dropout >= 0.f && c.dropout < 1.f)) return fail(HD_ERR_INVALID, "hd_create: dropout %f", c.dropout);
    for (int a : {c.enc_act, c.conv_act})
        if (a != HD_ACT_RELU && a != HD_ACT_GELU) return fail(HD_ERR_INVALID, "hd_create: activation %d", a);

    // the configuration is validated first so that a bad one is reported as such even on a machine without a GPU
    int ndev = hd_device_count();
    if (ndev <= 0) return fail(HD_ERR_NO_DEVICE, "hd_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(HD_ERR_INVALID, "hd_create: device %d out of range [0,%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HD_ERR_NO_DEVICE, "hd_create: device %d is %s, kernels are built for gfx950 only", device, prop.gcnArchName);
    HdModel* m = new HdModel();
    m->cfg = c;
    for (const OptDef& d : OPTS) {                   // library default, or the environment's override of the default (if legal)
        m->opt[d.id] = d.def;
        if (const char* e = getenv(d.env)) {
            const int64_t v = atoll(e);
            if (opt_legal(d, v)) m->opt[d.id] = v;
            else {
                if (d.nset == 0) m->opt[d.id] = v < d.lo ? d.lo : d.hi;        // out of range: clamped (as rounds 1-4 did)
                // (an old A/B script that exports a value this build no longer accepts must not measure the default against itself in silence)
                if (!getenv("HUDIFF_QUIET"))
                    fprintf(stderr, "[hudiff_hip] %s=%s is not a legal value of that option: %s %lld\n", d.env, e,
                            d.nset == 0 ? "clamped to" : "ignored, the default stays", (long long)m->opt[d.id]);
            }
        }
    }
    if (!getenv("HUDIFF_QUIET")) {                   // variables earlier rounds read and this build does not (INTEGRATION.md "Removed variables")
        static const char* const removed[] = {"HUDIFF_GEMM_NBUF", "HUDIFF_PFF3_APPLY", "HUDIFF_ENC_FUSED", "HUDIFF_ENC_ABL", "HUDIFF_X3_PERSIST"};
        for (const char* r : removed)
            if (getenv(r)) fprintf(stderr, "[hudiff_hip] %s is set, but this build has no such switch any more: ignored\n", r);
    }
    m->device = device;
    m->nseg = c.kind == HD_KIND_ANTIBODY ? 2 : 1;
    m->d = c.d_model; m->dh = c.d_model / 2; m->D = c.sum_d_model; m->Dh = c.sum_d_model / 2;
    m->A = c.att_model; m->Fd = c.dim_feedforward; m->L = c.max_len;
    m->p_enc = c.dropout;
    m->p_conv = c.dropout > 0.f ? 0.5f : 0.f;   // F.dropout(x) default p, gated by cfg.dropout > 0 (model.py:295-303)
    for (auto& ln : m->lane) {
        hipError_t e = hipStreamCreate(&ln.stream);
        if (e != hipSuccess) { delete m; return fail(HD_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        hipEventCreate(&ln.ev0);
        hipEventCreate(&ln.ev1);
    }
    *out = m;
    return HD_OK;
}

static void free_ws(Workspace& ws) {
    for (void* p : ws.owned) hipFree(p);
    ws = Workspace();
}

extern "C" void hd_destroy(HdModel* m) {
    if (!m) return;
    hipSetDevice(m->device);
    for (auto& ln : m->lane) {
        if (ln.stream) hipStreamSynchronize(ln.stream);
        ln.drop_graphs();
        free_ws(ln.ws);
        if (ln.rs) hipFree(ln.rs);
        if (ln.ev0) hipEventDestroy(ln.ev0);
        if (ln.ev1) hipEventDestroy(ln.ev1);
        if (ln.stream) hipStreamDestroy(ln.stream);
    }
    if (m->blob) hipFree(m->blob);
    if (m->blobx) hipFree(m->blobx);
    if (m->side_vec) hipFree(m->side_vec);
    if (m->emb_stats) hipFree(m->emb_stats);
    if (m->qnoise) hipFree(m->qnoise);
    delete m;
}

static bool ends_with(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

extern "C" HdStatus hd_load_tensor(HdModel* m, const char* key, const float* data, const int64_t* shape, int32_t ndim) {
    if (!m || !key || !shape || ndim < 0 || ndim > 4) return fail(HD_ERR_INVALID, "hd_load_tensor: bad argument");
    if (m->finalized) return fail(HD_ERR_STATE, "hd_load_tensor: model already finalized");
    std::string k(key);
    if (k.rfind("module.", 0) == 0) k = k.substr(7);          // antibody_train.py:23-30 strips this prefix
    if (!data) return fail(HD_ERR_INVALID, "hd_load_tensor: null data for %s", key);
    // Buffers of the checkpoint: kept so the kernels use the reference's own float32
    // tables bit for bit.  '...rope' is complex64 [L, hd/2] viewed as float32 [L, hd/2, 2]; every layer
    // holds the same table (cross_attention.py:145-146), so one copy is stored.
    if (ends_with(k, ".rope")) k = "rope";
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] <= 0) return fail(HD_ERR_INVALID, "hd_load_tensor: %s bad shape", key); t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(data, data + n);
    if (k == "rope" && m->host.count(k)) {
        const HostTensor& prev = m->host[k];
        if (prev.shape != t.shape || memcmp(prev.data.data(), t.data.data(), n * sizeof(float)) != 0)
            return fail(HD_ERR_UNSUPPORTED, "hd_load_tensor: %s differs from the other layers' RoPE table", key);
        return HD_OK;
    }
    m->host[k] = std::move(t);
    return HD_OK;
}

// ---- weight packing ---------------------------------------------------------------------------
struct Packer {
    std::vector<float> buf;
    size_t add(const std::vector<float>& v) {
        size_t off = (buf.size() + 63) / 64 * 64;
        buf.resize(off);
        buf.insert(buf.end(), v.begin(), v.end());
        return off;
    }
};

struct Loader {
    HdModel* m;
    std::map<std::string, bool> used;
    std::string err;
    const HostTensor* get(const std::string& key, std::initializer_list<int64_t> shape) {
        auto it = m->host.find(key);
        if (it == m->host.end()) { if (err.empty()) err = "missing tensor '" + key + "'"; return nullptr; }
        std::vector<int64_t> want(shape);
        if (it->second.shape != want) {
            if (err.empty()) {
                err = "tensor '" + key + "' has shape [";
                for (auto s : it->second.shape) err += std::to_string(s) + ",";
                err += "] expected [";
                for (auto s : want) err += std::to_string(s) + ",";
                err += "]";
            }
            return nullptr;
        }
        used[key] = true;
        return &it->second;
    }
    // vector [n]
    std::vector<float> vec(const std::string& key, int64_t n) {
        auto t = get(key, {n});
        return t ? t->data : std::vector<float>((size_t)n, 0.f);
    }
    // torch Linear weight [out, in] -> [in, out]
    std::vector<float> lin_t(const std::string& key, int64_t out, int64_t in, bool conv1 = false) {
        const HostTensor* t = conv1 ? get(key, {out, in, 1}) : get(key, {out, in});
        std::vector<float> r((size_t)(in * out), 0.f);
        if (t) for (int64_t o = 0; o < out; ++o) for (int64_t i = 0; i < in; ++i) r[(size_t)(i * out + o)] = t->data[(size_t)(o * in + i)];
        return r;
    }
    // Conv1d weight [out, in, k] -> [k][in][out]
    std::vector<float> conv_t(const std::string& key, int64_t out, int64_t in, int64_t k) {
        auto t = get(key, {out, in, k});
        std::vector<float> r((size_t)(k * in * out), 0.f);
        if (t) for (int64_t o = 0; o < out; ++o) for (int64_t i = 0; i < in; ++i) for (int64_t j = 0; j < k; ++j)
            r[(size_t)((j * in + i) * out + o)] = t->data[(size_t)((o * in + i) * k + j)];
        return r;
    }
};

static void append(std::vector<float>& a, const std::vector<float>& b) { a.insert(a.end(), b.begin(), b.end()); }

struct X3Off;
struct X3Packer;
struct ByteNetOff { size_t ln1_g, ln1_b, w1, b1, ln2_g, ln2_b, wc, bc, ln3_g, ln3_b, w3, b3; std::vector<float> wc_copy, w1_copy, w3_copy; };

static ByteNetOff pack_bytenet(Loader& ld, Packer& pk, const std::vector<std::string>& prefixes, int din, int dh, int ks) {
    std::vector<float> ln1_g, ln1_b, w1, b1, ln2_g, ln2_b, wc, bc, ln3_g, ln3_b, w3, b3;
    for (const auto& p : prefixes) {
        append(ln1_g, ld.vec(p + "sequence1.0.weight", din)); append(ln1_b, ld.vec(p + "sequence1.0.bias", din));
        append(w1, ld.lin_t(p + "sequence1.2.conv.weight", dh, din, true)); append(b1, ld.vec(p + "sequence1.2.conv.bias", dh));
        append(ln2_g, ld.vec(p + "sequence1.3.weight", dh)); append(ln2_b, ld.vec(p + "sequence1.3.bias", dh));
        append(wc, ld.conv_t(p + "conv.weight", dh, dh, ks)); append(bc, ld.vec(p + "conv.bias", dh));
        append(ln3_g, ld.vec(p + "sequence2.0.weight", dh)); append(ln3_b, ld.vec(p + "sequence2.0.bias", dh));
        append(w3, ld.lin_t(p + "sequence2.2.conv.weight", din, dh, true)); append(b3, ld.vec(p + "sequence2.2.conv.bias", din));
    }
    ByteNetOff o;
    o.ln1_g = pk.add(ln1_g); o.ln1_b = pk.add(ln1_b); o.w1 = pk.add(w1); o.b1 = pk.add(b1);
    o.ln2_g = pk.add(ln2_g); o.ln2_b = pk.add(ln2_b); o.wc = pk.add(wc); o.bc = pk.add(bc);
    o.ln3_g = pk.add(ln3_g); o.ln3_b = pk.add(ln3_b); o.w3 = pk.add(w3); o.b3 = pk.add(b3);
    o.wc_copy = wc; o.w1_copy = w1; o.w3_copy = w3;      // kept for the split-precision copies (hd_finalize)
    return o;
}

// Folds y = LN(x; g, beta) W + b into the weights: W <- diag(g) W with every column centred (its mean over k subtracted),
// b <- beta W + b.  With centred columns x W equals (x - mean) W, so y = rstd (x W) + b needs neither a prologue nor a
// mean-times-column-sum correction (gemm_epilogue, p.ln_fold).  Sums in double.
static void fold_layernorm(std::vector<float>& w, std::vector<float>& b, const std::vector<float>& g,
                           const std::vector<float>& beta, int K, int N) {
    std::vector<double> s(N, 0.0), t(N, 0.0);
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) {
            const double w0 = w[(size_t)k * N + n];
            t[n] += (double)beta[k] * w0;
            s[n] += (double)g[k] * w0;
        }
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n)
            w[(size_t)k * N + n] = (float)((double)g[k] * (double)w[(size_t)k * N + n] - s[n] / K);
    for (int n = 0; n < N; ++n) b[n] = (float)((double)b[n] + t[n]);
}

// ---- split-precision weight packing (gemm_x3_k) ----------------------------------------------------
// w: nseg matrices [Ktot, N] (row-major, back to back).  Output per segment: tiles [N/128][Ktot/32] of hi[128 n][32 k] then
// lo[128][32] fp16, values scaled by 2^shift with shift chosen so that the largest |w| lands in [2^13, 2^14): the lo parts
// (2^-11 of the value) then stay normal fp16 numbers for everything within 2^-12 of the largest weight.
struct X3Off { size_t off = 0; long seg_stride = 0; float acc_scale = 1.f; long ntile_stride = 0; bool ok = false; };
struct X3Packer {
    std::vector<uint16_t> buf;
    static uint16_t h16(float x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
    static float f16(float x) { return (float)(_Float16)x; }
    // kperm: inside every group of 16 k the image holds k in the order {0-3, 8-11, 4-7, 12-15} -- the order in which a lane of hd_chain.hip.h's
    // phase B holds the channels of its accumulators (MFMA output rows 8 q + 4 (lane >> 5) + e), which it feeds back as the B operand
    X3Off add(const std::vector<float>& w, int nseg, int Ktot, int N, bool kperm = false) {
        X3Off o;
        if (Ktot % X3_BK || N % X3_BN || w.size() != (size_t)nseg * Ktot * N) return o;
        float mx = 0.f;
        for (float v : w) mx = fmaxf(mx, fabsf(v));
        int e = 0;
        if (mx > 0.f) frexpf(mx, &e);                   // mx = f * 2^e, f in [0.5, 1)
        const int shift = 14 - e;                       // mx * 2^shift in [2^13, 2^14)
        const float sc = ldexpf(1.f, shift);
        o.off = (buf.size() + 63) / 64 * 64;
        const int nt = N / X3_BN, kt = Ktot / X3_BK;
        o.seg_stride = (long)nt * kt * X3_TILE_HALFS;
        o.ntile_stride = (long)kt * X3_TILE_HALFS;
        o.acc_scale = ldexpf(1.f, -shift);
        buf.resize(o.off + (size_t)nseg * o.seg_stride);
        for (int sgi = 0; sgi < nseg; ++sgi) {
            const float* ws = w.data() + (size_t)sgi * Ktot * N;
            uint16_t* dst = buf.data() + o.off + (size_t)sgi * o.seg_stride;
            for (int a = 0; a < nt; ++a)
                for (int b = 0; b < kt; ++b) {
                    // the tile is the LDS image gemm_x3_k copies with linear 16-byte DMA pieces: row n = 32 halfs (64 B), its
                    // four 8-half chunks XOR-swizzled by (n >> 2) & 3 (conflict-free ds_read_b128 without padding)
                    uint16_t* t = dst + ((size_t)a * kt + b) * X3_TILE_HALFS;
                    for (int n = 0; n < X3_BN; ++n)
                        for (int k = 0; k < X3_BK; ++k) {
                            static const int src16[16] = {0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};
                            const int ks = kperm ? ((k & ~15) | src16[k & 15]) : k;          // image position k holds source row ks
                            const float v = ws[(size_t)(b * X3_BK + ks) * N + a * X3_BN + n] * sc;
                            const float hi = f16(v);
                            const int pos = n * X3_BK + ((((k >> 3) ^ ((n >> 2) & 3)) << 3) | (k & 7));
                            t[pos] = h16(hi);
                            t[X3_BN * X3_BK + pos] = h16(v - hi);
                        }
                }
        }
        o.ok = true;
        return o;
    }
};

struct AttLayerOff { size_t wqkv, bqkv, wo, bo; X3Off wqkvx, wox; };
static AttLayerOff pack_attlayer(Loader& ld, Packer& pk, X3Packer* xp, const std::string& p, int D, int A,
                                 const std::vector<float>* ln_g = nullptr, const std::vector<float>* ln_b = nullptr) {
    // fused [D, 3A] = [query | key | value]
    auto q = ld.lin_t(p + "query.weight", A, D), k = ld.lin_t(p + "key.weight", A, D), v = ld.lin_t(p + "value.weight", A, D);
    std::vector<float> w((size_t)D * 3 * A);
    for (int i = 0; i < D; ++i) {
        memcpy(&w[(size_t)i * 3 * A], &q[(size_t)i * A], sizeof(float) * A);
        memcpy(&w[(size_t)i * 3 * A + A], &k[(size_t)i * A], sizeof(float) * A);
        memcpy(&w[(size_t)i * 3 * A + 2 * A], &v[(size_t)i * A], sizeof(float) * A);
    }
    std::vector<float> b;
    append(b, ld.vec(p + "query.bias", A)); append(b, ld.vec(p + "key.bias", A)); append(b, ld.vec(p + "value.bias", A));
    AttLayerOff o;
    if (ln_g) fold_layernorm(w, b, *ln_g, *ln_b, D, 3 * A);
    o.wqkv = pk.add(w); o.bqkv = pk.add(b);
    const std::vector<float> wo = ld.lin_t(p + "out_put.weight", D, A);
    o.wo = pk.add(wo); o.bo = pk.add(ld.vec(p + "out_put.bias", D));
    if (xp) { o.wqkvx = xp->add(w, 1, D, 3 * A); o.wox = xp->add(wo, 1, A, D); }
    return o;
}

extern "C" HdStatus hd_set_precision(HdModel* m, int32_t precision) {
    if (!m) return fail(HD_ERR_INVALID, "hd_set_precision: null model");
    if (m->finalized) return fail(HD_ERR_STATE, "hd_set_precision: call it before hd_finalize (the split weight images are built there)");
    if (precision != HD_PRECISION_DEFAULT && precision != HD_PRECISION_F32_GEMM && precision != HD_PRECISION_F32_ALL && precision != HD_PRECISION_SPLIT)
        return fail(HD_ERR_INVALID, "hd_set_precision: unknown route %d", precision);
    m->precision_req = precision;
    return HD_OK;
}

// The route an unspecified (HD_PRECISION_DEFAULT) handle takes: HD_PRECISION_SPLIT, unless the environment says otherwise --
// HUDIFF_PRECISION=split|f32_gemm|f32_all, or the round-2/3 switches HUDIFF_X3 (GEMMs) and HUDIFF_ATTN_X3 (attention core).
static int default_route_from_env(std::string* err) {
    if (const char* e = getenv("HUDIFF_PRECISION")) {
        const std::string v(e);
        if (v == "split" || v == "default" || v.empty()) return HD_PRECISION_SPLIT;
        if (v == "f32_gemm") return HD_PRECISION_F32_GEMM;
        if (v == "f32_all") return HD_PRECISION_F32_ALL;
        *err = "HUDIFF_PRECISION=" + v + " (expected split, f32_gemm or f32_all)";
        return HD_PRECISION_SPLIT;
    }
    // HUDIFF_X3=0 -> fp32 GEMMs (round-3 default route), + HUDIFF_ATTN_X3=0 -> every kernel fp32 (HUDIFF_ATTN_X3=0 alone means
    // that too, as it did when HUDIFF_X3 defaulted to 0); HUDIFF_X3=1 -> split precision
    const char *ex = getenv("HUDIFF_X3"), *ea = getenv("HUDIFF_ATTN_X3");
    const bool ax = ea ? atoi(ea) != 0 : true;
    const bool x3 = ex ? atoi(ex) != 0 : ax;
    if (x3) return HD_PRECISION_SPLIT;
    return ax ? HD_PRECISION_F32_GEMM : HD_PRECISION_F32_ALL;
}

extern "C" HdStatus hd_finalize(HdModel* m) {
    if (!m) return fail(HD_ERR_INVALID, "hd_finalize: null model");
    if (m->finalized) return fail(HD_ERR_STATE, "hd_finalize: already finalized");
    HIP_TRY(hipSetDevice(m->device));
    const HdConfig& c = m->cfg;
    const int d = m->d, dh = m->dh, D = m->D, Dh = m->Dh, A = m->A, Fd = m->Fd, L = m->L, ks = c.kernel_size;
    Loader ld{m};
    Packer pk;
    X3Packer xpk;
    // Precision route (include/hudiff_hip.h): an explicit hd_set_precision wins; HD_PRECISION_DEFAULT is the library default
    // (split precision since round 4: VERDICT r3 "Next" #1) unless the environment overrides the default.
    {
        std::string perr;
        m->precision = m->precision_req != HD_PRECISION_DEFAULT ? m->precision_req : default_route_from_env(&perr);
        if (!perr.empty()) return fail(HD_ERR_INVALID, "hd_finalize: %s", perr.c_str());
        m->x3 = m->precision == HD_PRECISION_SPLIT;
        // attention core of launches >= 8192 activation rows: attn_x3_k (S = K Q^T and O = V^T P^T as three fp16 MFMAs per product
        // on fp16 (hi, lo) splits of the fp32 Q, K, V, P; fp32 accumulation and softmax) on every route but F32_ALL
        m->attn_x3 = m->precision != HD_PRECISION_F32_ALL;
        m->lnsync_level_cfg = m->lnsync_level = (int)m->opt[HD_OPT_LNSYNC_LEVEL];
    }
    X3Packer* xp = m->x3 ? &xpk : nullptr;
    // HD_OPT_SPLIT_LAYER_MASK (ablation aid): 1 = ByteNet blocks, 2 = attention blocks take the split-precision kernels
    const int x3_mask = (int)m->opt[HD_OPT_SPLIT_LAYER_MASK];
    X3Packer* xp_bn = (x3_mask & 1) ? xp : nullptr;
    X3Packer* xp_at = (x3_mask & 2) ? xp : nullptr;
    const bool ab = c.kind == HD_KIND_ANTIBODY;
    const std::vector<std::string> segn = ab ? std::vector<std::string>{"h_layers", "l_layers"} : std::vector<std::string>{"layers"};
    const std::string convp = ab ? "dual_conv_block." : "nano_conv_block.";

    // embedder [n_tokens, d]
    std::vector<float> embv;
    { auto t = ld.get("aa_encoder.embedder.weight", {c.n_tokens, d}); embv = t ? t->data : std::vector<float>((size_t)c.n_tokens * d); }
    const size_t o_emb = pk.add(embv);
    std::vector<ByteNetOff> enc_off, conv_off;
    for (int n = 0; n < c.n_encoder_layers; ++n) {
        std::vector<std::string> pf;
        for (auto& s : segn) pf.push_back("aa_encoder." + s + "." + std::to_string(n) + ".");
        enc_off.push_back(pack_bytenet(ld, pk, pf, d, dh, ks));
    }
    for (int n = 0; n < c.dual_layers; ++n) {
        std::vector<std::string> pf;
        for (auto& s : segn) pf.push_back(convp + s + "." + std::to_string(n) + ".");
        conv_off.push_back(pack_bytenet(ld, pk, pf, D, Dh, ks));
    }
    struct BnX { X3Off wc, w1, w3, w3p; };
    std::vector<BnX> enc_x, conv_x;
    auto bnx = [&](ByteNetOff& o, int din, int dhh) {
        BnX x;
        if (xp_bn) { x.wc = xp_bn->add(o.wc_copy, m->nseg, ks * dhh, dhh); x.w1 = xp_bn->add(o.w1_copy, m->nseg, din, dhh); x.w3 = xp_bn->add(o.w3_copy, m->nseg, dhh, din);
                     if (m->opt[HD_OPT_BN_CHAIN]) x.w3p = xp_bn->add(o.w3_copy, m->nseg, dhh, din, /*kperm=*/true); }
        for (auto* v : {&o.wc_copy, &o.w1_copy, &o.w3_copy}) { v->clear(); v->shrink_to_fit(); }
        return x;
    };
    for (auto& o : enc_off) enc_x.push_back(bnx(o, d, dh));
    for (auto& o : conv_off) conv_x.push_back(bnx(o, D, Dh));
    struct AttOff { AttLayerOff a1, a2; size_t n1_g, n1_b, n2_g, n2_b, wf1, bf1, wf2, bf2; X3Off wf1x, wf2x; };
    std::vector<AttOff> att_off;
    for (int n = 0; n < c.cs_layers; ++n) {
        std::string p = "self_at.layers." + std::to_string(n) + ".";
        AttOff o;
        // norm_hl1 is folded into the second attention's Q|K|V projection, norm_hl2 into the first FF layer
        const std::vector<float> n1g = ld.vec(p + "norm_hl1.weight", D), n1b = ld.vec(p + "norm_hl1.bias", D);
        const std::vector<float> n2g = ld.vec(p + "norm_hl2.weight", D), n2b = ld.vec(p + "norm_hl2.bias", D);
        o.a1 = pack_attlayer(ld, pk, xp_at, p + "attn_hl.", D, A);
        o.a2 = pack_attlayer(ld, pk, xp_at, p + "attn_hl_c.", D, A, &n1g, &n1b);
        o.n1_g = pk.add(n1g); o.n1_b = pk.add(n1b);
        o.n2_g = pk.add(n2g); o.n2_b = pk.add(n2b);
        {
            std::vector<float> wf1 = ld.lin_t(p + "ff_hl.0.weight", Fd, D), bf1 = ld.vec(p + "ff_hl.0.bias", Fd);
            fold_layernorm(wf1, bf1, n2g, n2b, D, Fd);
            o.wf1 = pk.add(wf1); o.bf1 = pk.add(bf1);
            if (xp_at) o.wf1x = xp_at->add(wf1, 1, D, Fd);
        }
        {
            const std::vector<float> wf2 = ld.lin_t(p + "ff_hl.2.weight", D, Fd);
            o.wf2 = pk.add(wf2); o.bf2 = pk.add(ld.vec(p + "ff_hl.2.bias", D));
            if (xp_at) o.wf2x = xp_at->add(wf2, 1, Fd, D);
        }
        att_off.push_back(o);
    }
    // region / position branch
    const int re = c.r_embedding;
    size_t r_emb, r_l0g, r_l0b, r_w, r_b, r_l1g, r_l1b, r_pe;
    { auto t = ld.get("region_encoder.region_embedding.weight", {c.n_region, re}); r_emb = pk.add(t ? t->data : std::vector<float>((size_t)c.n_region * re)); }
    r_l0g = pk.add(ld.vec("region_encoder.region_layer1.0.weight", re)); r_l0b = pk.add(ld.vec("region_encoder.region_layer1.0.bias", re));
    r_w = pk.add(ld.lin_t("region_encoder.region_layer1.2.conv.weight", d, re, true)); r_b = pk.add(ld.vec("region_encoder.region_layer1.2.conv.bias", d));
    r_l1g = pk.add(ld.vec("region_encoder.region_layer1.3.weight", d)); r_l1b = pk.add(ld.vec("region_encoder.region_layer1.3.bias", d));
    if (m->host.count("pos_encoder.pos_embedding.pe")) {
        auto t = ld.get("pos_encoder.pos_embedding.pe", {L, 1, d});
        r_pe = pk.add(t ? t->data : std::vector<float>((size_t)L * d));
    } else {   // sinusoid table (model/encoder/model.py:70-78), evaluated in double and rounded once
        std::vector<float> pe((size_t)L * d);
        for (int i = 0; i < d; i += 2) {
            double div = exp((double)i * (-log(10000.0) / d));
            for (int l = 0; l < L; ++l) {
                double ang = (double)l * div;
                pe[(size_t)l * d + i] = (float)sin(ang);
                pe[(size_t)l * d + i + 1] = (float)cos(ang);
            }
        }
        r_pe = pk.add(pe);
    }
    size_t p_w1 = pk.add(ld.lin_t("pos_encoder.pos_lin.ln1.weight", 2 * d, d)), p_b1 = pk.add(ld.vec("pos_encoder.pos_lin.ln1.bias", 2 * d));
    size_t p_w2 = pk.add(ld.lin_t("pos_encoder.pos_lin.ln2.weight", d, 2 * d)), p_b2 = pk.add(ld.vec("pos_encoder.pos_lin.ln2.bias", d));
    size_t s_emb = 0, s_w1 = 0, s_b1 = 0, s_lg = 0, s_lb = 0, s_w2 = 0, s_b2 = 0;
    const int se = c.s_embedding;
    if (ab) {
        { auto t = ld.get("side_encoder.side_embeddinng.weight", {c.n_side, se}); s_emb = pk.add(t ? t->data : std::vector<float>((size_t)c.n_side * se)); }
        s_w1 = pk.add(ld.lin_t("side_encoder.side_mlp.0.weight", d, se)); s_b1 = pk.add(ld.vec("side_encoder.side_mlp.0.bias", d));
        s_lg = pk.add(ld.vec("side_encoder.side_mlp.1.weight", d)); s_lb = pk.add(ld.vec("side_encoder.side_mlp.1.bias", d));
        s_w2 = pk.add(ld.lin_t("side_encoder.side_mlp.3.weight", d, d)); s_b2 = pk.add(ld.vec("side_encoder.side_mlp.3.bias", d));
    }
    size_t h_g = pk.add(ld.vec("last_norm.weight", D)), h_b = pk.add(ld.vec("last_norm.bias", D));
    size_t h_w, h_bias;
    { auto t = ld.get("decoder.weight", {c.n_tokens, D}); h_w = pk.add(t ? t->data : std::vector<float>((size_t)c.n_tokens * D)); }
    h_bias = pk.add(ld.vec("decoder.bias", c.n_tokens));
    // RoPE table (cross_attention.py:35-56): angle = t * theta^(-2k/hd), float32 like torch
    size_t o_cos, o_sin, o_cs;
    {
        std::vector<float> cs((size_t)L * 32), sn((size_t)L * 32);
        if (m->host.count("rope")) {
            auto t = ld.get("rope", {L, ATT_HD / 2, 2});
            if (t) for (size_t i = 0; i < (size_t)L * 32; ++i) { cs[i] = t->data[2 * i]; sn[i] = t->data[2 * i + 1]; }
        } else {
            for (int k = 0; k < 32; ++k) {
                double freq = 1.0 / pow(10000.0, (double)(2 * k) / (double)ATT_HD);
                for (int t = 0; t < L; ++t) {
                    double ang = (double)t * freq;
                    cs[(size_t)t * 32 + k] = (float)cos(ang);
                    sn[(size_t)t * 32 + k] = (float)sin(ang);
                }
            }
        }
        o_cos = pk.add(cs); o_sin = pk.add(sn);
        std::vector<float> csi((size_t)L * 64);
        for (size_t i = 0; i < (size_t)L * 32; ++i) { csi[2 * i] = cs[i]; csi[2 * i + 1] = sn[i]; }
        o_cs = pk.add(csi);
    }
    if (!ld.err.empty()) return fail(HD_ERR_STATE, "hd_finalize: %s", ld.err.c_str());
    for (auto& kv : m->host)
        if (!ld.used.count(kv.first)) return fail(HD_ERR_INVALID, "hd_finalize: unexpected tensor '%s' (strict load)", kv.first.c_str());

    HIP_TRY(hipMalloc(&m->blob, pk.buf.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(m->blob, pk.buf.data(), pk.buf.size() * sizeof(float), hipMemcpyHostToDevice));
    const float* B0 = m->blob;
    if (m->x3 && !xpk.buf.empty()) {
        HIP_TRY(hipMalloc(&m->blobx, xpk.buf.size() * sizeof(uint16_t)));
        HIP_TRY(hipMemcpy(m->blobx, xpk.buf.data(), xpk.buf.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    auto mkx = [&](const X3Off& o) {
        X3W x;
        if (o.ok && m->blobx) { x.w = m->blobx + o.off; x.seg_stride = o.seg_stride; x.acc_scale = o.acc_scale; x.ntile_stride = o.ntile_stride; }
        return x;
    };
    m->emb = B0 + o_emb;
    auto mk = [&](const ByteNetOff& o, int n) {
        ByteNetW w{B0 + o.ln1_g, B0 + o.ln1_b, B0 + o.w1, B0 + o.b1, B0 + o.ln2_g, B0 + o.ln2_b, B0 + o.wc, B0 + o.bc,
                   B0 + o.ln3_g, B0 + o.ln3_b, B0 + o.w3, B0 + o.b3, dilation_of(c, n)};
        return w;
    };
    for (int n = 0; n < c.n_encoder_layers; ++n) {
        m->enc.push_back(mk(enc_off[n], n));
        m->enc.back().wcx = mkx(enc_x[n].wc); m->enc.back().w1x = mkx(enc_x[n].w1); m->enc.back().w3x = mkx(enc_x[n].w3); m->enc.back().w3px = mkx(enc_x[n].w3p);
    }
    for (int n = 0; n < c.dual_layers; ++n) {
        m->conv.push_back(mk(conv_off[n], n));
        m->conv.back().wcx = mkx(conv_x[n].wc); m->conv.back().w1x = mkx(conv_x[n].w1); m->conv.back().w3x = mkx(conv_x[n].w3); m->conv.back().w3px = mkx(conv_x[n].w3p);
    }
    for (auto& o : att_off) {
        AttBlockW w;
        w.a1 = {B0 + o.a1.wqkv, B0 + o.a1.bqkv, B0 + o.a1.wo, B0 + o.a1.bo, mkx(o.a1.wqkvx), mkx(o.a1.wox)};
        w.a2 = {B0 + o.a2.wqkv, B0 + o.a2.bqkv, B0 + o.a2.wo, B0 + o.a2.bo, mkx(o.a2.wqkvx), mkx(o.a2.wox)};
        w.wf1x = mkx(o.wf1x); w.wf2x = mkx(o.wf2x);
        w.n1_g = B0 + o.n1_g; w.n1_b = B0 + o.n1_b; w.n2_g = B0 + o.n2_g; w.n2_b = B0 + o.n2_b;
        w.wf1 = B0 + o.wf1; w.bf1 = B0 + o.bf1; w.wf2 = B0 + o.wf2; w.bf2 = B0 + o.bf2;
        m->att.push_back(w);
    }
    m->regw = {B0 + r_emb, B0 + r_l0g, B0 + r_l0b, B0 + r_w, B0 + r_b, B0 + r_l1g, B0 + r_l1b, B0 + r_pe};
    m->pos_w1 = B0 + p_w1; m->pos_b1 = B0 + p_b1; m->pos_w2 = B0 + p_w2; m->pos_b2 = B0 + p_b2;
    if (ab) m->sidew = {B0 + s_emb, B0 + s_w1, B0 + s_b1, B0 + s_lg, B0 + s_lb, B0 + s_w2, B0 + s_b2};
    m->head = {B0 + h_g, B0 + h_b, B0 + h_w, B0 + h_bias};
    m->rope_cos = B0 + o_cos; m->rope_sin = B0 + o_sin; m->rope_cs = B0 + o_cs;
    for (auto& ln : m->lane) {
        HIP_TRY(hipMalloc(&ln.rs, sizeof(RunState)));
        HIP_TRY(hipMemset(ln.rs, 0, sizeof(RunState)));
    }
    m->cl = 0;
    HIP_TRY(hipMalloc(&m->emb_stats, sizeof(float2) * c.n_tokens));
    hipLaunchKernelGGL(row_stats_k, dim3((c.n_tokens + 3) / 4), dim3(256), 0, cur(m).stream, m->emb, d, d, c.n_tokens, m->emb_stats);
    if (ab) {
        HIP_TRY(hipMalloc(&m->side_vec, sizeof(float) * c.n_side * d));
        hipLaunchKernelGGL(side_vec_k, dim3(c.n_side), dim3(256), 0, cur(m).stream, m->sidew, se, d, m->side_vec);
        HIP_TRY(hipGetLastError());
    }
    // upper limits of the dynamic LDS requests (the launches ask for lds_safe_request(...) <= these)
    const int att_cap = LDS_PER_CU;
    // (a split-precision model still runs the fp32 attention kernel for launches too small for the split kernels)
    if (L > 160) HIP_TRY(hipFuncSetAttribute((const void*)attn_k<19>, hipFuncAttributeMaxDynamicSharedMemorySize, att_cap));
    else HIP_TRY(hipFuncSetAttribute((const void*)attn_k<10>, hipFuncAttributeMaxDynamicSharedMemorySize, att_cap));
    HIP_TRY(hipFuncSetAttribute((const void*)attn_x3_k<19>, hipFuncAttributeMaxDynamicSharedMemorySize, att_cap));
    HIP_TRY(hipFuncSetAttribute((const void*)attn_x3_k<19, AX19_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, att_cap));
    HIP_TRY(hipFuncSetAttribute((const void*)attn_x3_k<10>, hipFuncAttributeMaxDynamicSharedMemorySize, att_cap));
    HIP_TRY(hipFuncSetAttribute((const void*)qkv_attn_x3_k<19, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, att_cap));
    HIP_TRY(hipFuncSetAttribute((const void*)qkv_attn_x3_k<10, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, att_cap));
    if (m->x3 && m->opt[HD_OPT_BN_CHAIN]) HIP_TRY(bn_chain_prepare());
    static_assert(lds_safe_request(AxGeom<19>::SMEM, ATT_THREADS) <= LDS_PER_CU && lds_safe_request(AxGeom<10>::SMEM, ATT_THREADS) <= LDS_PER_CU, "LDS co-residency rule");
    HIP_TRY(hipStreamSynchronize(cur(m).stream));
    m->host.clear();
    m->finalized = true;
    return HD_OK;
}

// ---- workspace ----------------------------------------------------------------------------------
template <typename T>
static HdStatus dalloc(Workspace& ws, T** p, size_t n) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, n * sizeof(T) + 256));
    ws.owned.push_back(q);
    *p = (T*)q;
    return HD_OK;
}

static HdStatus ensure_ws(HdModel* m, int B) {
    Workspace& ws = cur(m).ws;
    if (B <= ws.capB) return HD_OK;
    HIP_TRY(hipStreamSynchronize(cur(m).stream));
    cur(m).drop_graphs();
    free_ws(ws);
    const size_t M = (size_t)B * m->L;
    const int d = m->d, dh = m->dh, D = m->D, Dh = m->Dh, A = m->A, Fd = m->Fd;
    HD_TRY(dalloc(ws, &ws.X, M * d)); HD_TRY(dalloc(ws, &ws.H1, M * dh)); HD_TRY(dalloc(ws, &ws.H2, M * dh));
    HD_TRY(dalloc(ws, &ws.FEAT, M * D)); HD_TRY(dalloc(ws, &ws.Y, M * D));
    HD_TRY(dalloc(ws, &ws.G1, M * Dh)); HD_TRY(dalloc(ws, &ws.G2, M * Dh));
    HD_TRY(dalloc(ws, &ws.QKV, M * 3 * A)); HD_TRY(dalloc(ws, &ws.O, M * A)); HD_TRY(dalloc(ws, &ws.AT, M * D));
    HD_TRY(dalloc(ws, &ws.F1, M * Fd));
    HD_TRY(dalloc(ws, &ws.EXTRA, M * d)); HD_TRY(dalloc(ws, &ws.POS, M * d)); HD_TRY(dalloc(ws, &ws.PH, M * 2 * d));
    HD_TRY(dalloc(ws, &ws.ST, M)); HD_TRY(dalloc(ws, &ws.PART[0], M * PART_STRIDE)); HD_TRY(dalloc(ws, &ws.PART[1], M * PART_STRIDE));
    HD_TRY(dalloc(ws, &ws.LOGITS, M * m->cfg.n_tokens));
    if (m->x3) {
        HD_TRY(dalloc(ws, &ws.S1, M * D)); HD_TRY(dalloc(ws, &ws.YX, M * D)); HD_TRY(dalloc(ws, &ws.ATX, M * D));
        HD_TRY(dalloc(ws, &ws.SYNC, (size_t)8 * 0x4000));
        HIP_TRY(hipMemsetAsync(ws.SYNC, 0, (size_t)8 * 0x4000 * sizeof(int), cur(m).stream));
    }
    HD_TRY(dalloc(ws, &ws.ATc, (size_t)B * D)); HD_TRY(dalloc(ws, &ws.Xc, (size_t)B * D)); HD_TRY(dalloc(ws, &ws.Qc, (size_t)B * A));
    HD_TRY(dalloc(ws, &ws.Oc, (size_t)B * A)); HD_TRY(dalloc(ws, &ws.F1c, (size_t)B * Fd)); HD_TRY(dalloc(ws, &ws.STc, (size_t)B));
    HD_TRY(dalloc(ws, &ws.PW, (size_t)B * m->cfg.nhead * 320)); HD_TRY(dalloc(ws, &ws.YV, (size_t)B * m->cfg.nhead * m->D));
    HD_TRY(dalloc(ws, &ws.tokens, M)); HD_TRY(dalloc(ws, &ws.tokens0, M)); HD_TRY(dalloc(ws, &ws.region, M)); HD_TRY(dalloc(ws, &ws.chain, (size_t)2 * B));
    HD_TRY(dalloc(ws, &ws.T, (size_t)B));
    ws.capB = B;
    return HD_OK;
}

static Segs make_segs(const HdModel* m, int B) {
    Segs s{};
    s.nseg = m->nseg; s.B = B; s.L = m->L;
    s.len[0] = m->cfg.h_len; s.off[0] = 0; s.base[0] = 0;
    s.len[1] = m->L - m->cfg.h_len; s.off[1] = m->cfg.h_len; s.base[1] = B * m->cfg.h_len;
    return s;
}

// ---- kernel launch helpers ----------------------------------------------------------------------
struct Drop { int mode = DROP_NONE; float p = 0.f; uint32_t site = 0; const uint8_t* mask = nullptr; };

static GemmP base_gemm(const HdModel* m, const Segs& sg) {
    GemmP p{};
    p.sg = sg; p.taps = 1; p.dil = 1; p.rs = cur(m).rs;
    return p;
}

// Split-precision launches (gemm_x3_k): the A operand must already be in split form, so the decision is taken by the
// caller BEFORE it asks the producer for split rows -- x3_use() holds every condition launch_gemm checks again.
static bool x3_use(const HdModel* m, const Segs& sg, const X3W& x) {
    const long rows = (long)sg.B * sg.L, widest = 3L * m->A > m->D ? 3L * m->A : m->D;       // 32-bit byte offsets in every operand
    return m->x3 && !m->x3_suspended && x.w && rows >= m->opt[HD_OPT_SPLIT_MIN_ROWS] && rows * widest * 4 < (1L << 31);
}
static void use_x3(GemmP& p, const X3W& x, int ntile0 = 0) {
    p.Wx = x.w + (long)ntile0 * x.ntile_stride; p.wx_stride = x.seg_stride; p.acc_scale = x.acc_scale;
}

// k-tile depth of the GEMM K loop: 16 by default, HUDIFF_GEMM_BK=32 selects the 32-deep variant (tuning aid)
static int gemm_bk() {
    static int bk = [] { const char* e = getenv("HUDIFF_GEMM_BK"); return (e && atoi(e) == 32) ? 32 : 16; }();
    return bk;
}


template <int BM, int BN, int WM, int WN, int BKT, int NB = 1>
static void launch_gemm_t(GemmP& p, bool conv, bool per_seg, hipStream_t st) {
    Segs run = p.sg;
    if (!per_seg) {           // weights shared by all rows: treat the whole batch as one segment
        run.nseg = 1; run.len[0] = p.sg.L; run.off[0] = 0; run.base[0] = 0;
    }
    GemmP q = p;
    q.sg = run;
    if (q.ldw == 0) q.ldw = q.N;
    const int rows0 = run.B * run.len[0];
    const int rows1 = run.nseg > 1 ? run.B * run.len[1] : 0;
    q.tiles0 = (rows0 + BM - 1) / BM;
    q.tiles_m = q.tiles0 + (rows1 + BM - 1) / BM;
    q.tiles_n = (q.N + BN - 1) / BN;
    dim3 grid(((q.tiles_m + 7) / 8) * 8 * q.tiles_n), blk(256);
    // 0 none (also a LayerNorm folded into the weights, p.ln_fold), 1 LN, 2 LN+ReLU, 3 LN+GELU
    const int pro = (!q.ln_fold && (q.stats || q.spart)) ? 1 + q.pro_act : 0;
    // epilogue feature mask (hd_kernels.hip.h, gemm_epilogue): the hot shapes exist in four instantiations -- tap GEMM / PFF1
    // (LayerNorm partials only), Q|K|V and FF1 (folded LayerNorm, activation), out-projection / FF2 (residual, partials), everything --
    // and a launch takes the smallest that covers it; the other shapes carry the full epilogue
    constexpr bool HOT = BKT == 16 && NB == 1 && BM >= 64;
    const int need = epi_needs(q);
    const int mask = !HOT ? 3 : !(need & ~EPI_PART) ? 0 : !(need & ~(EPI_FOLD | EPI_ACT)) ? 1 : !(need & ~(EPI_RESID | EPI_PART)) ? 2 : 3;
#define HD_LAUNCH_E(CONV, PRO, E) hipLaunchKernelGGL((gemm_k<BM, BN, WM, WN, CONV, PRO, 0, NB, BKT, E>), grid, blk, 0, st, q)
#define HD_LAUNCH(CONV, PRO)                                                                       \
    do {                                                                                           \
        if constexpr (HOT) {                                                                       \
            if (mask == 0) HD_LAUNCH_E(CONV, PRO, EPI_PART);                                       \
            else if (mask == 1) HD_LAUNCH_E(CONV, PRO, EPI_FOLD | EPI_ACT);                        \
            else if (mask == 2) HD_LAUNCH_E(CONV, PRO, EPI_RESID | EPI_PART);                      \
            else HD_LAUNCH_E(CONV, PRO, EPI_ALL);                                                  \
        } else {                                                                                   \
            HD_LAUNCH_E(CONV, PRO, EPI_ALL);                                                       \
        }                                                                                          \
    } while (0)
    if (!conv) {
        switch (pro) {
            case 0: HD_LAUNCH(false, 0); break;
            case 1: HD_LAUNCH(false, 1); break;
            case 2: HD_LAUNCH(false, 2); break;
            default: HD_LAUNCH(false, 3); break;
        }
    } else {
        switch (pro) {
            case 0: HD_LAUNCH(true, 0); break;       // input already normalised + activated (ln_apply_k)
            case 2: HD_LAUNCH(true, 2); break;
            default: HD_LAUNCH(true, 3); break;
        }
    }
}
#undef HD_LAUNCH
#undef HD_LAUNCH_E

// p.part != nullptr: the epilogue leaves LayerNorm partials of the output rows and they are merged into `stats_out`.
struct LnApply { const float* gamma = nullptr; const float* beta = nullptr; int k_stride = 0; int act = 0; int split = 0; };
enum StatsOut { STATS_NONE = 0, STATS_PARTIALS = 1 };

// Output LayerNorm statistics: STATS_PARTIALS leaves the epilogue's slice partials for a consumer GEMM that merges
// them in its prologue (use_partials()); `apply` normalises + activates the output in place instead (ln_apply_k).
static void launch_gemm(HdModel* m, GemmP& p, bool conv, bool per_seg, int stats_out = STATS_NONE, const LnApply* apply = nullptr) {
    Workspace& ws = cur(m).ws;
    hipStream_t st = cur(m).stream;
    const long rows = (long)p.sg.B * p.sg.L;
    const bool big = rows >= m->opt[HD_OPT_BIG_MIN_ROWS] || p.Wx != nullptr;      // (split-precision launches: x3_use() decided)
    int pw = big ? 64 : 32;                              // column-slice width of the LayerNorm partials this launch leaves (its waves' WTN)
    p.part_rows = rows;
    // non-temporal epilogue stores: +0.9 % on the split-precision sample (3 x 3 interleaved runs; gemm_x3_k always uses them),
    // nothing on the fp32 one (HD_OPT_STORE_NT turns them on there)
    p.st_nt = big ? (int)m->opt[HD_OPT_STORE_NT] : 0;
    if (stats_out != STATS_NONE || apply || p.ln_sync) p.part = ws.PART[ws.part_next];
    if (p.ln_sync) p.sync_ctr = ws.SYNC;
    const long small_tiles = m->opt[HD_OPT_GEMM_SMALL_TILES];
    const long tiles128 = ((rows + 127) / 128) * ((p.N + 127) / 128);
    // the BK = 16 kernels assume whole k tiles and 32-bit byte offsets inside every operand (gemm_k, FAST)
    const long lda = p.lda, ldw = p.ldw ? p.ldw : p.N;
    const bool fast_ok = p.Kc % 16 == 0 && rows * lda * 4 < (1L << 31) && (long)p.taps * p.Kc * ldw * 4 < (1L << 31);
    p.a_bytes = fast_ok ? (uint32_t)(rows * lda * 4) : 0;
    p.w_bytes = fast_ok ? (uint32_t)((long)p.taps * p.Kc * ldw * 4) : 0;
    // split-precision variant (p.Wx set by the caller after x3_use(); its A operand is in split form)
    if (p.Wx) {
        const int xpro = (!p.ln_fold && (p.stats || p.spart)) ? 1 : 0;
        if (!(big && fast_ok && !xpro && p.Kc % X3_BK == 0 && p.N % X3_BN == 0 && (long)p.taps * p.Kc * X3_BN * 4 < (1L << 31))) {
            // cannot happen for shapes x3_use() admits; never run an operand that is not fp32 through an fp32 kernel:
            // nothing is launched and the forward reports the error (forward_body)
            char buf[160];
            snprintf(buf, sizeof(buf), "split-precision launch with an ineligible shape (rows %ld Kc %d N %d taps %d)", rows, p.Kc, p.N, p.taps);
            g_launch_err = buf;
            return;
        }
        Segs run = p.sg;
        if (!per_seg) { run.nseg = 1; run.len[0] = p.sg.L; run.off[0] = 0; run.base[0] = 0; }
        GemmP q = p;
        q.sg = run;
#ifdef HD_PROBES
        static const int abl = [] { const char* e = getenv("HUDIFF_X3_ABL"); return e ? atoi(e) : 0; }();
        q.x3_abl = abl;
#endif
        static const int full_epi = [] { const char* e = getenv("HUDIFF_X3_ABL"); return e && (atoi(e) & 64) ? 4 : 0; }();      // (tests: the all-features epilogue everywhere)
        q.dbg = ((m->debug_lnsync_fail && q.ln_sync) ? 1 : 0) | ((m->debug_lnsync_scatter && q.ln_sync) ? 2 : 0) | full_epi;
        const int rows0 = run.B * run.len[0], rows1 = run.nseg > 1 ? run.B * run.len[1] : 0;
        // tile shape / pipeline depth, by measurement: 256 x 256 tiles (two stages, one 8-wave block per CU)
        // for the widest output (Q|K|V, N = 1536: 621 vs 650 us), two stages of 128 x 128 tiles (two blocks per CU) elsewhere; three
        // stages of 256 x 128 tiles were no faster anywhere (640 us).  HD_OPT_SPLIT_TILE forces 128 / 256 (x 128, three stages) / 512 (= 256 x 256)
        const int force = (int)m->opt[HD_OPT_SPLIT_TILE];
        const long t256 = (rows0 + 255) / 256 + (rows1 + 255) / 256;
        int shape = (q.N % 256 == 0 && q.N >= 1024 && t256 * (q.N / 256) >= 384) ? 512 : 128;
        if (force == 128 || force == 256 || (force == 512 && q.N % 256 == 0)) shape = force;
        if (q.ln_sync) shape = 128;                     // (never the 8-wave tiles: every 4-wave instantiation -- 128, 64 and 32 rows, EPISET 0 / 2 -- carries the meeting epilogue,
                                                        //  so the small_grid / tiny_grid downgrades below keep an ln_sync launch legal)
        // under-filled grids (mid-size batches): 64 x 128 tiles double the blocks of a launch whose 128 x 128 grid leaves CUs idle or
        // with one latency-bound block each.  HD_OPT_SMALL_GRID = largest 128 x 128 grid that takes them (320: B = 8 antibodies
        // 20.0 -> 25.1 sequences/s, B = 16 36.5 -> 42.4, B = 48 73.3 -> 78.4; larger limits lose again).
        const long small_grid = m->opt[HD_OPT_SMALL_GRID];
        if (shape == 128 && ((rows0 + 127) / 128 + (rows1 + 127) / 128) * (long)(q.N / 128) <= small_grid) shape = 64;
        // ... and 32 x 128 tiles (four waves side by side, 32 x 32 each) when even those leave most CUs empty (a handful of sequences)
        const long tiny_grid = m->opt[HD_OPT_TINY_GRID];
        if (shape == 64 && ((rows0 + 63) / 64 + (rows1 + 63) / 64) * (long)(q.N / 128) <= tiny_grid) { shape = 32; pw = 32; }
        const int bm = shape <= 64 ? shape : (shape == 128 ? 128 : 256), bn = shape == 512 ? 256 : 128;
        q.tiles0 = (rows0 + bm - 1) / bm;
        q.tiles_m = q.tiles0 + (rows1 + bm - 1) / bm;
        q.tiles_n = q.N / bn;
        dim3 grid(((q.tiles_m + 7) / 8) * 8 * q.tiles_n);
        if (shape == 512) {
            if (conv) hipLaunchKernelGGL((gemm_x3_k<256, 256, 2, 4, true, 2>), grid, dim3(512), 0, st, q);
            else hipLaunchKernelGGL((gemm_x3_k<256, 256, 2, 4, false, 2>), grid, dim3(512), 0, st, q);
        } else if (shape == 256) {
            if (conv) hipLaunchKernelGGL((gemm_x3_k<256, 128, 4, 2, true, 3>), grid, dim3(512), 0, st, q);
            else hipLaunchKernelGGL((gemm_x3_k<256, 128, 4, 2, false, 3>), grid, dim3(512), 0, st, q);
        } else if (shape == 32) {
            // few blocks per CU: the K loop is a chain of DMA round trips (~0.8 us per k tile, whatever the grid), so a third LDS stage --
            // two tiles in flight, 60 KB, still two blocks per CU -- pays: B = 1 3.88 -> 4.60 sequences/s, B = 8 29.8 -> 32.8, B = 16
            // 45.7 -> 46.8 (Nb: 8.47 -> 9.51, 64.7 -> 71.1, 120.6 -> 130.4).  Four stages (80 KB, one block per CU) gain less and lose
            // from B = 16 on (instantiation removed).  HD_OPT_TINY_STAGES = 2 restores two stages.
            const int ns = (int)m->opt[HD_OPT_TINY_STAGES];
            // ... with four extra waves per block that only issue the operand DMA (gemm_x3_k, LW): the five DMA instructions per k tile and
            // the chain of six dependent MFMAs then run in different waves (B = 1 4.61 -> 4.92 sequences/s, B = 8 33.3 -> 34.4, Nb B = 8
            // 72.4 -> 77.0).  Such a block takes a CU alone (512 threads, 146 registers), so only while all lanes' blocks of the launch
            // fit one per CU (B = 16 on two lanes: 52.3 -> 49.5 otherwise).  HD_OPT_LOADER_WAVES = 0 switches them off.
            const int loaders = (int)m->opt[HD_OPT_LOADER_WAVES];
            if (loaders && ns == 3 && (long)q.tiles_m * q.tiles_n * (m->in_session ? m->nlanes : 1) <= 256) {
                if (conv) hipLaunchKernelGGL((gemm_x3_k<32, 128, 1, 4, true, 3, 4>), grid, dim3(512), 0, st, q);
                else hipLaunchKernelGGL((gemm_x3_k<32, 128, 1, 4, false, 3, 4>), grid, dim3(512), 0, st, q);
            } else
            if (ns == 3) {
                if (conv) hipLaunchKernelGGL((gemm_x3_k<32, 128, 1, 4, true, 3>), grid, dim3(256), 0, st, q);
                else hipLaunchKernelGGL((gemm_x3_k<32, 128, 1, 4, false, 3>), grid, dim3(256), 0, st, q);
            } else {
                if (conv) hipLaunchKernelGGL((gemm_x3_k<32, 128, 1, 4, true, 2>), grid, dim3(256), 0, st, q);
                else hipLaunchKernelGGL((gemm_x3_k<32, 128, 1, 4, false, 2>), grid, dim3(256), 0, st, q);
            }
        } else if (shape == 64) {
            // three stages (72 KB, still two blocks per CU) while the launch has at most one block per CU: B = 16 48.3 -> 51.4 sequences/s,
            // Nb B = 32 200.4 -> 205.9; with 257 .. 384 blocks it gains in places (Nb B = 64 301 -> 317) and loses in others (B = 48
            // 82.2 -> 78.1), so larger grids keep two stages.  HD_OPT_SMALL_STAGES = 2 / 3 forces, HD_OPT_SMALL_STAGES3_MAX_GRID moves the limit.
            const int ns64 = (int)m->opt[HD_OPT_SMALL_STAGES];
            const long ns3_max = m->opt[HD_OPT_SMALL_STAGES3_MAX_GRID];
            if (ns64 == 3 || (ns64 == 0 && (long)q.tiles_m * q.tiles_n <= ns3_max)) {
                if (conv) hipLaunchKernelGGL((gemm_x3_k<64, 128, 2, 2, true, 3>), grid, dim3(256), 0, st, q);
                else hipLaunchKernelGGL((gemm_x3_k<64, 128, 2, 2, false, 3>), grid, dim3(256), 0, st, q);
            } else
            if (conv) hipLaunchKernelGGL((gemm_x3_k<64, 128, 2, 2, true, 2>), grid, dim3(256), 0, st, q);
            else hipLaunchKernelGGL((gemm_x3_k<64, 128, 2, 2, false, 2>), grid, dim3(256), 0, st, q);
        } else if (q.ln_sync) {      // the 128 x 128 tile exists per epilogue set (gemm_x3_k, EPISET): the meeting epilogue has its own registers
            if (conv) hipLaunchKernelGGL((gemm_x3_k<128, 128, 2, 2, true, 2, 0, 2>), grid, dim3(256), 0, st, q);
            else hipLaunchKernelGGL((gemm_x3_k<128, 128, 2, 2, false, 2, 0, 2>), grid, dim3(256), 0, st, q);
        } else {
            if (conv) hipLaunchKernelGGL((gemm_x3_k<128, 128, 2, 2, true, 2, 0, 1>), grid, dim3(256), 0, st, q);
            else hipLaunchKernelGGL((gemm_x3_k<128, 128, 2, 2, false, 2, 0, 1>), grid, dim3(256), 0, st, q);
        }
    } else
    if (big && fast_ok && tiles128 < small_tiles) {
        // few 128-row tiles (narrow outputs of the token encoder): 64-row tiles balance the 256 CUs better
        launch_gemm_t<64, 128, 2, 2, 16>(p, conv, per_seg, st);
    } else if (big) {
        if (gemm_bk() == 32 || !fast_ok) launch_gemm_t<128, 128, 2, 2, 32>(p, conv, per_seg, st);
        else launch_gemm_t<128, 128, 2, 2, 16>(p, conv, per_seg, st);
    } else {
        launch_gemm_t<32, 128, 1, 4, 32>(p, conv, per_seg, st);
    }
    if (!p.part) return;
    ws.part_last = p.part; ws.part_last_pw = pw; ws.part_next ^= 1;
    if (apply) {      // normalise + activate the output in place (consumer: the tap GEMM, which then needs no prologue)
        const int seg1 = p.sg.nseg > 1 ? p.sg.base[1] : (int)rows;
        hipLaunchKernelGGL(ln_apply_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p.part, pw, p.N, (int)rows,
                           (const float*)p.C, p.ldc, p.C, p.ldc, (const float2*)nullptr, apply->gamma, apply->beta, apply->k_stride, seg1,
                           apply->act, apply->split, (const RunState*)cur(m).rs);
    }
}

// prologue statistics of `p` = the slice partials the previous producing GEMM left (merged in the kernel)
static void use_partials(HdModel* m, GemmP& p) {
    const Workspace& ws = cur(m).ws;
    p.stats = nullptr; p.spart = ws.part_last; p.spw = ws.part_last_pw; p.spart_rows = (long)p.sg.B * p.sg.L;
}

static void launch_stats(const HdModel* m, const float* X, int ldx, int C, int rows, hipStream_t st) {
    hipLaunchKernelGGL(row_stats_k, dim3((rows + 3) / 4), dim3(256), 0, st, X, ldx, C, rows, cur(m).ws.ST);
}

static void set_drop(GemmP& p, const Drop& dr) {
    p.drop_mode = dr.mode;
    if (dr.mode == DROP_NONE) return;
    double t = floor((double)dr.p * 4294967296.0);
    p.drop_thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    p.drop_scale = (float)(1.0 / (1.0 - (double)dr.p));
    p.drop_site = dr.site;
    p.drop_mask = dr.mask;
}

// Where the LayerNorm statistics of a GEMM's input rows come from
enum XStats { X_FINAL = 0,      // ws.ST already holds (mean, rstd)
              X_PARTIALS = 1,   // the previous GEMM left slice partials (merged in the prologue)
              X_NONE = 2,       // nothing yet: run row_stats_k
              X_S1 = 3 };       // split-precision route: the previous block already wrote act(LN1(x)) in split form into ws.S1

// One ByteNet block:  out = dropout(x + PFF2(act(LN(conv(act(LN(PFF1(act(LN(x))))))))))   [+ extra]
// LayerNorm statistics travel with the data: every GEMM epilogue leaves (mean, M2) slice partials of the rows it
// wrote and the next GEMM merges them in its prologue; in front of the tap GEMM, LayerNorm + activation are applied
// once in place (ln_apply_k).  want_out_stats: leave partials of `out` for the next block's first GEMM.
// ln_sync on / off (HUDIFF_X3_LNSYNC, default on): the split-precision ByteNet GEMMs normalise + activate + split their own
// output rows (GemmP::ln_sync) instead of leaving that to a separate ln_apply_k pass over HBM
// level (HdModel::lnsync_level; HUDIFF_X3_LNSYNC at hd_finalize, 0 after a failed meeting): 0 off, 1 the two inner GEMMs of a block
// (h1, h2), 2 also the block's last GEMM (writes the NEXT block's first operand)
static int x3_lnsync_level(const HdModel* m) { return m->lnsync_level; }
static bool x3_lnsync(const HdModel* m) { return m->lnsync_level > 0; }

// `next`: the block that follows in the same stack (same widths) when it will also take the split-precision route and reads
// exactly `out` (ldo == din, no columns beside it): this block's last GEMM then writes act(LN1_next(out)) in split form into ws.S1
// itself, and the next call passes x_stats = X_S1 (its first operand is ready).
static void bytenet_block(HdModel* m, const Segs& sg, const ByteNetW& w, int din, int dh, int act,
                          const float* x, int ldx, float* h1, float* h2, float* out, int ldo,
                          const Drop& dr, const float* extra, int lde, XStats x_stats, bool want_out_stats,
                          float* out_split = nullptr, const ByteNetW* next = nullptr) {
    const int rows = sg.rows();
    const int ks = m->cfg.kernel_size;
    const bool x3_route = x3_use(m, sg, w.w1x) && x3_use(m, sg, w.wcx) && x3_use(m, sg, w.w3x) && cur(m).ws.S1;
    // (the split-precision route's ln_apply_k computes the statistics of a row it holds anyway: no separate pass)
    if (x_stats == X_NONE && !(x3_route && din <= 1024)) launch_stats(m, x, ldx, din, rows, cur(m).stream);
    if (x3_route) {
        // Split-precision route.  Every GEMM operand is act(LN(.)) of the previous result in split (hi, lo) form and the three
        // projections run on gemm_x3_k without a prologue.  (The fp32 route recomputes LayerNorm + activation in every N tile's
        // prologue, which costs more than the MFMAs once those are three fp16 instructions.)  Who writes the operand:
        //   ln_sync on  : the producing GEMM itself -- the blocks of an M tile exchange their LayerNorm partials and normalise the
        //                 rows they still hold in registers (GemmP::ln_sync); h1 / h2 never exist as fp32 rows.  Only the first block
        //                 of a stack still needs one ln_apply_k pass for x (its producer is not a GEMM of this kind).
        //   ln_sync off : ln_apply_k, one HBM round trip per operand (out of place for x, in place for h1 / h2).
        Workspace& ws = cur(m).ws;
        hipStream_t st = cur(m).stream;
        const int seg1 = sg.nseg > 1 ? sg.base[1] : rows;
        const bool sync = x3_lnsync(m) && ws.SYNC;
        if (x_stats != X_S1) {
            const bool from_part = x_stats == X_PARTIALS, self = x_stats == X_NONE && din <= 1024;
            hipLaunchKernelGGL(ln_apply_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, from_part ? ws.part_last : (const float2*)nullptr,
                               ws.part_last_pw, din, rows, x, ldx, ws.S1, din, (from_part || self) ? (const float2*)nullptr : (const float2*)ws.ST,
                               w.ln1_g, w.ln1_b, din, seg1, act, 1, (const RunState*)cur(m).rs);
        }
        GemmP p = base_gemm(m, sg);
        p.A = ws.S1; p.lda = din; p.W = w.w1; p.bias = w.b1; p.C = h1; p.ldc = dh; p.N = dh; p.Kc = din;
        p.w_stride = (long)din * dh; p.n_stride = dh; p.k_stride = din;
        const LnApply ap2{w.ln2_g, w.ln2_b, dh, act, 1};
        use_x3(p, w.w1x);
        if (sync) { p.ln_sync = 1; p.C = nullptr; p.S = h1; p.gamma2 = w.ln2_g; p.beta2 = w.ln2_b; p.act2 = act; p.k2_stride = dh; }
        launch_gemm(m, p, false, true, STATS_NONE, sync ? nullptr : &ap2);

        p = base_gemm(m, sg);
        p.A = h1; p.lda = dh; p.W = w.wc; p.bias = w.bc; p.C = h2; p.ldc = dh; p.N = dh; p.Kc = dh; p.taps = ks; p.dil = w.dil;
        p.w_stride = (long)ks * dh * dh; p.n_stride = dh; p.k_stride = dh;
        const LnApply ap3{w.ln3_g, w.ln3_b, dh, act, 1};
        use_x3(p, w.wcx);
        if (sync) { p.ln_sync = 1; p.C = nullptr; p.S = h2; p.gamma2 = w.ln3_g; p.beta2 = w.ln3_b; p.act2 = act; p.k2_stride = dh; }
        launch_gemm(m, p, true, true, STATS_NONE, sync ? nullptr : &ap3);

        p = base_gemm(m, sg);
        p.A = h2; p.lda = dh; p.W = w.w3; p.bias = w.b3; p.C = out; p.ldc = ldo; p.N = din; p.Kc = dh;
        p.w_stride = (long)dh * din; p.n_stride = din; p.k_stride = dh;
        p.resid = x; p.ldr = ldx; p.extra = extra; p.lde = lde;
        p.C2 = out_split;
        use_x3(p, w.w3x);
        set_drop(p, dr);
        if (sync && next) { p.ln_sync = 1; p.S = ws.S1; p.gamma2 = next->ln1_g; p.beta2 = next->ln1_b; p.act2 = act; p.k2_stride = din; }
        launch_gemm(m, p, false, true, want_out_stats ? STATS_PARTIALS : STATS_NONE);
        return;
    }
    GemmP p = base_gemm(m, sg);
    p.A = x; p.lda = ldx; p.W = w.w1; p.bias = w.b1; p.C = h1; p.ldc = dh; p.N = dh; p.Kc = din;
    p.w_stride = (long)din * dh; p.n_stride = dh; p.k_stride = din;
    p.stats = cur(m).ws.ST; p.gamma = w.ln1_g; p.beta = w.ln1_b; p.pro_act = act;
    if (x_stats == X_PARTIALS) use_partials(m, p);
    // h1 <- act(LN(h1)) in place, once, instead of in the tap GEMM's prologue (7 taps x N tiles times per element)
    const LnApply ap{w.ln2_g, w.ln2_b, dh, act};
    launch_gemm(m, p, false, true, STATS_NONE, &ap);

    // (normalising h2 in place by ln_apply_k as well, so that the last projection runs without its prologue, was measured in rounds 2
    // and 3: the extra HBM pass costs more than the prologue recomputed per N tile -- NOTES.md)
    p = base_gemm(m, sg);
    p.A = h1; p.lda = dh; p.W = w.wc; p.bias = w.bc; p.C = h2; p.ldc = dh; p.N = dh; p.Kc = dh; p.taps = ks; p.dil = w.dil;
    p.w_stride = (long)ks * dh * dh; p.n_stride = dh; p.k_stride = dh;
    launch_gemm(m, p, true, true, STATS_PARTIALS);

    p = base_gemm(m, sg);
    p.A = h2; p.lda = dh; p.W = w.w3; p.bias = w.b3; p.C = out; p.ldc = ldo; p.N = din; p.Kc = dh;
    p.w_stride = (long)dh * din; p.n_stride = din; p.k_stride = dh;
    p.gamma = w.ln3_g; p.beta = w.ln3_b; p.pro_act = act;
    use_partials(m, p);
    p.resid = x; p.ldr = ldx; p.extra = extra; p.lde = lde;
    p.C2 = out_split;
    set_drop(p, dr);
    launch_gemm(m, p, false, true, want_out_stats ? STATS_PARTIALS : STATS_NONE);
}

// out = resid + Attn(x) (AttLayer, cross_attention.py:149-173).  ln: x is LayerNorm'ed in front of the fused Q|K|V
// projection; the norm is folded into that projection's weights, its row statistics are the partials the previous GEMM left.
// x3: the whole attention stage runs on the split-precision kernels.  x_split = split copy of x (written by x's producer),
// out_split = where to leave the split copy of `out` for the next consumer (may be null).
static bool att_x3(const HdModel* m, const Segs& sg) {
    if (m->att.empty()) return false;
    const AttBlockW& w = m->att[0];
    return x3_use(m, sg, w.a1.wqkvx) && x3_use(m, sg, w.a1.wox) && x3_use(m, sg, w.a2.wqkvx) && x3_use(m, sg, w.a2.wox) &&
           x3_use(m, sg, w.wf1x) && x3_use(m, sg, w.wf2x) && cur(m).ws.YX && cur(m).ws.ATX;
}
// HUDIFF_LDS_NO_PAD=1 (diagnostic, scripts/lds_fill_probe.py): dynamic LDS requests are NOT padded by the co-residency rule, so
// that the exact-fill geometry of round 2 (two 81 920-byte blocks on a CU) can be reproduced on demand
static size_t lds_request(int bytes, int threads) {
    static const bool no_pad = [] { const char* e = getenv("HUDIFF_LDS_NO_PAD"); return e && atoi(e) == 1; }();
    return (size_t)(no_pad ? bytes : lds_safe_request(bytes, threads));
}

// split_only (x3, out_split set): nobody reads `out` as fp32 rows -- the out-projection writes the split copy alone (GemmP::c_split).
static void attention_layer(HdModel* m, const Segs& sg, const AttLayerW& w, const float* x, bool ln,
                            const float* resid, float* out, bool want_out_stats, bool x3 = false,
                            const float* x_split = nullptr, float* out_split = nullptr, bool split_only = false) {
    hipStream_t st = cur(m).stream;
    const int D = m->D, A = m->A;
    GemmP p = base_gemm(m, sg);
    p.A = x; p.lda = D; p.W = w.wqkv; p.bias = w.bqkv; p.C = cur(m).ws.QKV; p.ldc = 3 * A; p.N = 3 * A; p.Kc = D;
    if (ln) { p.ln_fold = 1; use_partials(m, p); }      // LayerNorm folded into wqkv / bqkv (hd_finalize)
    // The projection fused into the attention core (hd_attn_fused.hip.h: one workgroup per (sequence, head group) computes its heads' Q | K | V
    // from the split input rows, leaves K / V in LDS and runs the attention on them): split route, the two shipped lengths, 64-wide heads.
    // (two heads of the short model share a workgroup: their planes must leave the co-residency slack, i.e. L <= 156 -- a 160-slot model,
    //  whose planes would fill the CU's 160 KB to the last byte, keeps the two-launch form)
    const bool fuse19 = m->L > 16 * 18 && m->L <= 16 * 19,
               fuse10 = m->L > 16 * 9 && m->L <= 16 * 10 && m->cfg.nhead % 2 == 0 && lds_fill_ok(QaGeom<10, 2>::smem(m->L), QA_THREADS);
    const long fused_blocks = (long)sg.B * (m->cfg.nhead / (fuse19 ? 1 : 2)) * (m->in_session ? m->nlanes : 1);      // workgroups of all lanes' launches
    if (x3 && x_split && w.wqkvx.w && m->opt[HD_OPT_FUSED_ATTN] && fused_blocks >= m->opt[HD_OPT_FUSED_ATTN_MIN_GRID] && m->opt[HD_OPT_SPLIT_ATTN] && (fuse19 || fuse10) && D % X3_BK == 0 && A % X3_BN == 0 &&
        (long)sg.rows() * 3 * A * 4 < (1L << 31) && (long)sg.rows() * D * 4 < (1L << 31) &&
        // (the Q fragments of head h, query tile qt travel through the Q | K thirds of row h KT + qt of the sequence's first-segment rows: 4 KiB each)
        m->cfg.nhead * (fuse19 ? 19 : 10) <= sg.len[0] && 2 * A * 4 >= 4096) {
        QkvAttnP q{};
        q.X = x_split; q.ldx = D; q.x_bytes = (uint32_t)((long)sg.rows() * D * 4);
        q.Wx = w.wqkvx.w; q.acc_scale = w.wqkvx.acc_scale; q.bias = w.bqkv;
        q.ln_fold = p.ln_fold; q.stats = p.stats; q.spart = p.spart; q.spw = p.spw; q.spart_rows = p.spart_rows;
        q.QKV = cur(m).ws.QKV; q.ldq = 3 * A; q.att = A; q.rope_cos = m->rope_cos; q.rope_sin = m->rope_sin; q.rope_cs = m->rope_cs;
        q.O = cur(m).ws.O; q.ldo = A; q.nhead = m->cfg.nhead; q.sg = sg; q.rs = cur(m).rs;
#ifdef HD_PROBES
        static const int qa_abl = [] { const char* e = getenv("HUDIFF_QA_ABL"); return e ? atoi(e) : 0; }();
        q.abl = qa_abl;
#endif
        const int NH = fuse19 ? 1 : 2;
        dim3 fgrid((unsigned)(((sg.B + 7) / 8) * 8 * (m->cfg.nhead / NH)));
        if (fuse19) hipLaunchKernelGGL((qkv_attn_x3_k<19, 1>), fgrid, dim3(QA_THREADS), lds_request(QaGeom<19, 1>::smem(m->L), QA_THREADS), st, q);
        else hipLaunchKernelGGL((qkv_attn_x3_k<10, 2>), fgrid, dim3(QA_THREADS), lds_request(QaGeom<10, 2>::smem(m->L), QA_THREADS), st, q);
        p = base_gemm(m, sg);
        p.A = cur(m).ws.O; p.lda = A; p.W = w.wo; p.bias = w.bo; p.C = out; p.ldc = D; p.N = D; p.Kc = A;
        p.resid = resid; p.ldr = D;
        use_x3(p, w.wox); p.C2 = out_split;
        if (split_only && out_split) { p.C = out_split; p.c_split = 1; p.C2 = nullptr; }
        launch_gemm(m, p, false, false, want_out_stats ? STATS_PARTIALS : STATS_NONE);
        return;
    }
    if (x3) { p.A = x_split; use_x3(p, w.wqkvx); }
    launch_gemm(m, p, false, false);
    // dynamic LDS requests obey the co-residency rule (hd_kernels.hip.h): a request whose co-resident blocks would fill the
    // CU's 160 KB is padded until one block fewer fits
    const size_t smem = lds_request((int)((size_t)m->L * (ATT_KS + att_vs(m->L > 160 ? 19 : 10)) * sizeof(float)), ATT_THREADS);
    dim3 grid(sg.B * m->cfg.nhead);
    // x3: the out-projection reads O in split form; L in (160, 304] has a split-precision attention kernel as well
    const bool ax_on = m->opt[HD_OPT_SPLIT_ATTN] != 0;
    // attn_x3_k<KT> masks only its last key tile: 16 (KT - 1) < L <= 16 KT (291 and 152 qualify); other lengths keep attn_k
    // ... and address QKV with 32-bit byte offsets
    // m->attn_x3 (default; HUDIFF_ATTN_X3=0 at hd_finalize turns it off): the split-precision attention kernel inside the fp32 path as
    // well (fp32 Q|K|V in, fp32 O out)
    const bool ax_ok = ((x3 && ax_on) || (m->attn_x3 && !m->x3_suspended && sg.rows() >= m->opt[HD_OPT_BIG_MIN_ROWS])) && (long)sg.rows() * 3 * A * 4 < (1L << 31);
    const RunState* rsp = cur(m).rs;
    const int osp = x3 ? 1 : 0;
    // few sequences: two workgroups per (sequence, head), each with half of the query tiles (one round of the tile loop instead of two)
    // (only the attn_x3_k branches below read blockIdx.y: a length outside their windows falls through to attn_k with grid.y = 1)
    const bool ax_kernel = ax_ok && ((m->L > 16 * 18 && m->L <= 16 * 19) || (m->L > 16 * 9 && m->L <= 16 * 10));
    if (ax_kernel && (long)grid.x <= m->opt[HD_OPT_ATTN_QSPLIT_MAX_GRID]) grid.y = 2;
    const bool ax_w8 = m->opt[HD_OPT_ATTN_WAVES] == 8;
    if (ax_ok && m->L > 16 * 18 && m->L <= 16 * 19 && !ax_w8)
        hipLaunchKernelGGL((attn_x3_k<19, AX19_THREADS>), grid, dim3(AX19_THREADS), lds_request(AxGeom<19>::SMEM, AX19_THREADS), st, cur(m).ws.QKV, 3 * A, A, m->rope_cos, m->rope_sin, cur(m).ws.O, A, m->cfg.nhead, sg, osp, rsp);
    else if (ax_ok && m->L > 16 * 18 && m->L <= 16 * 19)
        hipLaunchKernelGGL(attn_x3_k<19>, grid, dim3(ATT_THREADS), lds_request(AxGeom<19>::SMEM, ATT_THREADS), st, cur(m).ws.QKV, 3 * A, A, m->rope_cos, m->rope_sin, cur(m).ws.O, A, m->cfg.nhead, sg, osp, rsp);
    else if (ax_ok && m->L > 16 * 9 && m->L <= 16 * 10) {
        hipLaunchKernelGGL(attn_x3_k<10>, grid, dim3(ATT_THREADS), lds_request(2 * 128 * m->L + 2 * AxGeom<10>::VPLANE, ATT_THREADS), st, cur(m).ws.QKV, 3 * A, A, m->rope_cos, m->rope_sin, cur(m).ws.O, A, m->cfg.nhead, sg, osp, rsp);
    } else if (m->L > 160)
        hipLaunchKernelGGL(attn_k<19>, grid, dim3(ATT_THREADS), smem, st, cur(m).ws.QKV, 3 * A, A, m->rope_cos, m->rope_sin, cur(m).ws.O, A, m->cfg.nhead, sg, x3 ? 1 : 0, rsp);
    else
        hipLaunchKernelGGL(attn_k<10>, grid, dim3(ATT_THREADS), smem, st, cur(m).ws.QKV, 3 * A, A, m->rope_cos, m->rope_sin, cur(m).ws.O, A, m->cfg.nhead, sg, x3 ? 1 : 0, rsp);
    p = base_gemm(m, sg);
    p.A = cur(m).ws.O; p.lda = A; p.W = w.wo; p.bias = w.bo; p.C = out; p.ldc = D; p.N = D; p.Kc = A;
    p.resid = resid; p.ldr = D;
    if (x3) { use_x3(p, w.wox); p.C2 = out_split; }
    if (x3 && split_only && out_split) { p.C = out_split; p.c_split = 1; p.C2 = nullptr; }
    launch_gemm(m, p, false, false, want_out_stats ? STATS_PARTIALS : STATS_NONE);
}

// The token-independent branch (RegionEmbedder, PosEmbedder, SideEmbedder): once per batch.
static HdStatus static_branch(HdModel* m, const Segs& sg) {
    hipStream_t st = cur(m).stream;
    Workspace& ws = cur(m).ws;
    const int d = m->d, D = m->D, rows = sg.rows();
    hipLaunchKernelGGL(region_embed_k, dim3((rows + 3) / 4), dim3(256), 0, st, ws.region, m->regw, m->cfg.r_embedding, d, ws.POS, sg);
    // pos = x + W2 gelu(W1 x + b1) + b2      (MLP model.py:28-33; nn.Dropout is inactive in eval mode)
    GemmP p = base_gemm(m, sg);
    p.A = ws.POS; p.lda = d; p.W = m->pos_w1; p.bias = m->pos_b1; p.C = ws.PH; p.ldc = 2 * d; p.N = 2 * d; p.Kc = d; p.epi_act = ACT_GELU;
    launch_gemm(m, p, false, false);
    p = base_gemm(m, sg);
    p.A = ws.PH; p.lda = 2 * d; p.W = m->pos_w2; p.bias = m->pos_b2; p.C = ws.POS; p.ldc = d; p.N = d; p.Kc = 2 * d;
    p.resid = ws.POS; p.ldr = d;
    launch_gemm(m, p, false, false);
    hipLaunchKernelGGL(static_feature_k, dim3((rows + 3) / 4), dim3(256), 0, st, ws.POS,
                       m->nseg > 1 ? m->side_vec : nullptr, ws.chain, d, D, ws.EXTRA, ws.FEAT, sg);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

// Forms of the pruned tail (hd_tail_fused.hip.h).  HD_OPT_TAIL_FORM / HUDIFF_TAIL: 0 = the separate launches below, 2 = five sliced
// launches + sample_step_k: the default for lanes of at most HD_OPT_TAIL_MAX_B (64) sequences, where the tail is a visible part of a
// step; above, the separate launches are as fast or faster (Nb, 512 sequences: 465 against 455 sequences/s).  (1 was round 4's
// one-workgroup-per-sequence kernel: not faster, moved to scripts/experiments/tail_fused_k.hip.h.)
enum { TAIL_LAUNCHES = 0, TAIL_SLICED = 2 };
static int tail_form(const HdModel* m, int B) {
    const int want = (int)m->opt[HD_OPT_TAIL_FORM];
    const bool value_via_rows = m->opt[HD_OPT_PRUNE_VALUE_VIA_ROWS] != 0;
    const int D = m->D, A = m->A, Fd = m->Fd;
    const bool one_ok = value_via_rows && m->cfg.nhead <= RV_MAX_HEADS && m->cfg.nhead * ATT_HD == A && m->L <= TAIL_MAXL && D <= TAIL_MAXD && D % 8 == 0 &&
                        A <= TAIL_MAXA && A % 64 == 0 && Fd <= TAIL_MAXF && Fd % 2 == 0 && D / 4 <= TC_THREADS;
    const bool sliced_ok = one_ok && (D == 768 || D == 512) && A == 512 && Fd == 256 && m->cfg.nhead == 8;      // the shipped widths
    if (want == TAIL_SLICED) return sliced_ok && B <= m->opt[HD_OPT_TAIL_MAX_ROWS] ? TAIL_SLICED : TAIL_LAUNCHES;
    return TAIL_LAUNCHES;
}
template <int D>
static void launch_tail_sliced(const TailP& t, int B, int nhead, int Fd, hipStream_t st) {
    hipLaunchKernelGGL(tail_pw_k<D>, dim3(B, nhead), dim3(TC_THREADS), 0, st, t);
    hipLaunchKernelGGL(tail_val_k<D>, dim3(B, D / TC_SLICE), dim3(TC_THREADS), 0, st, t);
    hipLaunchKernelGGL(tail_out_k<D>, dim3(B, D / TC_SLICE), dim3(TC_THREADS), 0, st, t);
    hipLaunchKernelGGL(tail_ff1_k<D>, dim3(B, Fd / 64), dim3(TC_THREADS), 0, st, t);
    hipLaunchKernelGGL(tail_ff2_k<D>, dim3(B, D / TC_SLICE), dim3(TC_THREADS), 0, st, t);
}

// Last SelfAttBlock of a sampling step, from "at = x + A1(x)" (in ws.AT, statistics in ws.ST) on, evaluated only
// for the row each sequence visits at this step (see gather_rows_k).  Result: ws.Xc [B, D] = block output rows.
static void pruned_tail(HdModel* m, const Segs& sg, const AttBlockW& w) {
    hipStream_t st = cur(m).stream;
    Workspace& ws = cur(m).ws;
    const int D = m->D, A = m->A, Fd = m->Fd, B = sg.B;
    Segs cs{};                      // compact [B, *] matrices: one "sequence" of B single-slot rows
    cs.nseg = 1; cs.B = B; cs.L = 1; cs.len[0] = 1;
    // K projection of LN1(at) for every row (columns [A, 2A) of the fused weight).  V is never projected for all rows:
    // the one query of each sequence takes its value side through the input rows (row_value_k / head_proj_k).
    const bool via_rows = m->opt[HD_OPT_PRUNE_VALUE_VIA_ROWS] != 0 && m->cfg.nhead <= RV_MAX_HEADS && m->L <= 320;
    GemmP p = base_gemm(m, sg);
    p.A = ws.AT; p.lda = D; p.W = w.a2.wqkv + A; p.ldw = 3 * A; p.bias = w.a2.bqkv + A; p.C = ws.QKV + A; p.ldc = 3 * A;
    p.N = via_rows ? A : 2 * A; p.Kc = D; p.ln_fold = 1;
    use_partials(m, p);             // statistics of `at`: partials left by the first attention's out-projection
    const float2* at_part = p.spart; const int at_pw = p.spw; const long at_rows = p.spart_rows;
    if (att_x3(m, sg)) { p.A = ws.ATX; use_x3(p, w.a2.wqkvx, A / X3_BN); }     // column slice [A, ...) = n tiles from A / 128 on
    launch_gemm(m, p, false, false);
    if (tail_form(m, B) == TAIL_SLICED) {          // everything behind the K projection in five launches (hd_tail_fused.hip.h)
        TailP t{};
        t.AT = ws.AT; t.Y = ws.Y; t.D = D; t.QKV = ws.QKV; t.ldq = 3 * A; t.A = A;
        t.at_part = at_part; t.at_pw = at_pw; t.at_rows = at_rows;
        t.wqkv = w.a2.wqkv; t.bqkv = w.a2.bqkv; t.wo = w.a2.wo; t.bo = w.a2.bo;
        t.wf1 = w.wf1; t.bf1 = w.bf1; t.Fd = Fd; t.wf2 = w.wf2; t.bf2 = w.bf2;
        t.rope_cos = m->rope_cos; t.rope_sin = m->rope_sin; t.order = ws.order; t.T = ws.T; t.Tmax = m->sTmax; t.rs = cur(m).rs;
        t.Xc = ws.Xc; t.nhead = m->cfg.nhead; t.sg = sg;
        t.head = m->head; t.tokens = ws.tokens; t.q_noise = m->s_has_q ? m->qnoise : nullptr; t.q_rows = m->sB; t.q_off = cur(m).row_off;
        t.PW = ws.PW; t.OP = ws.YV; t.ATc = ws.ATc; t.F1c = ws.F1c;
        if (D == 768) launch_tail_sliced<768>(t, B, m->cfg.nhead, Fd, st);
        else launch_tail_sliced<512>(t, B, m->cfg.nhead, Fd, st);
        return;
    }
    // visited rows of `at` and of the block input x
    hipLaunchKernelGGL(gather_rows_k, dim3((B + 3) / 4), dim3(256), 0, st, ws.AT, D, ws.ATc, ws.order, ws.T, m->sTmax, cur(m).rs, sg);
    hipLaunchKernelGGL(gather_rows_k, dim3((B + 3) / 4), dim3(256), 0, st, ws.Y, D, ws.Xc, ws.order, ws.T, m->sTmax, cur(m).rs, sg);
    hipLaunchKernelGGL(row_stats_k, dim3((B + 3) / 4), dim3(256), 0, st, ws.ATc, D, D, B, ws.STc);
    // q = LN1(at_c) Wq + bq
    p = base_gemm(m, cs);
    p.A = ws.ATc; p.lda = D; p.W = w.a2.wqkv; p.ldw = 3 * A; p.bias = w.a2.bqkv; p.C = ws.Qc; p.ldc = A; p.N = A; p.Kc = D;
    p.stats = ws.STc; p.ln_fold = 1;
    launch_gemm(m, p, false, false);
    hipLaunchKernelGGL(attn_row_k, dim3((B * m->cfg.nhead + 3) / 4), dim3(256), 0, st, ws.Qc, ws.QKV, 3 * A, A, m->rope_cos,
                       m->rope_sin, ws.Oc, m->cfg.nhead, ws.order, ws.T, m->sTmax, cur(m).rs, sg, via_rows ? ws.PW : nullptr,
                       at_part, at_pw, at_rows, D);
    if (via_rows) {
        hipLaunchKernelGGL(row_value_k, dim3(B, (D + 255) / 256), dim3(256), 0, st, ws.AT, D, ws.PW, ws.YV, m->cfg.nhead, sg);
        hipLaunchKernelGGL(head_proj_k, dim3((B + 3) / 4, m->cfg.nhead), dim3(256), (size_t)lds_safe_request(4 * D * (int)sizeof(float), 256), st, ws.YV, D,
                           w.a2.wqkv + 2 * A, 3 * A, w.a2.bqkv + 2 * A, ws.Oc, A, m->cfg.nhead, B);
    }
    // at_c = at_c + o Wo + bo
    p = base_gemm(m, cs);
    p.A = ws.Oc; p.lda = A; p.W = w.a2.wo; p.bias = w.a2.bo; p.C = ws.ATc; p.ldc = D; p.N = D; p.Kc = A; p.resid = ws.ATc; p.ldr = D;
    launch_gemm(m, p, false, false);
    // x_c = FF(LN2(at_c)) + x_c
    hipLaunchKernelGGL(row_stats_k, dim3((B + 3) / 4), dim3(256), 0, st, ws.ATc, D, D, B, ws.STc);
    p = base_gemm(m, cs);
    p.A = ws.ATc; p.lda = D; p.W = w.wf1; p.bias = w.bf1; p.C = ws.F1c; p.ldc = Fd; p.N = Fd; p.Kc = D;
    p.stats = ws.STc; p.ln_fold = 1; p.epi_act = ACT_RELU;
    launch_gemm(m, p, false, false);
    p = base_gemm(m, cs);
    p.A = ws.F1c; p.lda = Fd; p.W = w.wf2; p.bias = w.bf2; p.C = ws.Xc; p.ldc = D; p.N = D; p.Kc = Fd; p.resid = ws.Xc; p.ldr = D;
    launch_gemm(m, p, false, false);
}

// ---- ByteNet stack on the row-owner chain kernel (hd_chain.hip.h; round 6) -----------------------------------------------------------
// A stack of n blocks as n + 1 launches:  C(0) | A B(0) C(1) | ... | A B(n - 1).  x0 [rows, ld0] fp32 with its (mean, rstd) in ws.ST;
// block k reads its input from x0 (k = 0) or `out`, writes `out` (the last block: `last_out`, row stride `last_ld`, + `extra`).  h1 of
// consecutive blocks alternates between ha / hb (phase A of a launch reads the buffer phase C of the previous launch wrote, neighbours'
// rows included, while this launch's phase C writes the other one).  last_split: X16 copy of the last block's output (attention operand).
static bool bn_chain_use(const HdModel* m, const Segs& sg, const std::vector<ByteNetW>& blocks, int din, int dh, int act, int stack_bit, int drop_mode, bool last_operand) {
    // (injected keep-masks -- parity tests -- and a last block without its optional operand keep the gemm_x3_k path: no instantiation for them)
    if (drop_mode == DROP_INJECT || !last_operand) return false;
    if (!(m->opt[HD_OPT_BN_CHAIN] & stack_bit) || blocks.empty() || !bn_chain_supported(dh, din, act) || m->cfg.kernel_size != 7) return false;
    for (const ByteNetW& w : blocks)
        if (!(x3_use(m, sg, w.w1x) && x3_use(m, sg, w.wcx) && w.w3px.w)) return false;
    const int tr = bn_chain_tile_rows(dh);
    long tiles = ((long)sg.B * sg.len[0] + tr - 1) / tr + (sg.nseg > 1 ? ((long)sg.B * sg.len[1] + tr - 1) / tr : 0);
    return tiles * (m->in_session ? m->nlanes : 1) >= m->opt[HD_OPT_BN_CHAIN_MIN_TILES];
}
static void bytenet_stack_chain(HdModel* m, const Segs& sg, const std::vector<ByteNetW>& blocks, int din, int dh, int act,
                                const float* x0, int ld0, float* out, float* last_out, int last_ld, float* ha, float* hb,
                                int drop_mode, float drop_p, uint32_t site0, const uint8_t* masks, size_t mask_stride,
                                const float* extra, int lde, float* last_split) {
    hipStream_t st = cur(m).stream;
    Workspace& ws = cur(m).ws;
    const long rows = sg.rows();
    const int n = (int)blocks.size();
    auto base = [&]() {
        ChainP p{};
        p.sg = sg; p.rs = cur(m).rs; p.act = act;
        return p;
    };
    auto set_c = [&](ChainP& p, const ByteNetW& w, float* h1out) {      // phase C: open block `w`
        p.W1 = w.w1x.w; p.w1_seg = w.w1x.seg_stride; p.sc_1 = w.w1x.acc_scale;
        p.b1 = w.b1; p.g1 = w.ln1_g; p.be1 = w.ln1_b; p.g2 = w.ln2_g; p.be2 = w.ln2_b; p.H1out = h1out;
    };
    {   // C(0)
        ChainP p = base();
        p.phases = 2;
        p.Yin = x0; p.ldyin = ld0; p.yin_bytes = (uint32_t)(rows * ld0 * 4); p.STin = ws.ST;
        set_c(p, blocks[0], ha);
        launch_bn_chain(p, dh, din, st);
    }
#ifdef HD_CHAIN_EXPERIMENT
    static const int chain_exp = [] { const char* e = getenv("HUDIFF_CHAIN_EXP"); return e ? atoi(e) : 0; }();      // 1: phase A alone per block (bn_chain_k PH = 4), 2: bn_pair_a_k -- TIMING ONLY
    if (chain_exp && dh >= 256) {
        for (int k = 0; k < n; ++k) {
            const ByteNetW& w = blocks[k];
            ChainP p = base();
            p.phases = 4;
            p.H1 = ha; p.h1_bytes = (uint32_t)(rows * dh * 4); p.taps = m->cfg.kernel_size; p.dil = w.dil;
            p.Wc = w.wcx.w; p.wc_seg = w.wcx.seg_stride; p.sc_c = w.wcx.acc_scale; p.bc = w.bc; p.g3 = w.ln3_g; p.be3 = w.ln3_b;
            p.H2dbg = hb;
            if (chain_exp == 2) launch_bn_pair_a(p, dh, st); else launch_bn_chain(p, dh, din, st);
        }
        return;
    }
#endif
    for (int k = 0; k < n; ++k) {
        const ByteNetW& w = blocks[k];
        const bool last = k + 1 == n;
        ChainP p = base();
        p.phases = last ? 1 : 3;
        p.H1 = (k & 1) ? hb : ha; p.h1_bytes = (uint32_t)(rows * dh * 4); p.taps = m->cfg.kernel_size; p.dil = w.dil;
        p.Wc = w.wcx.w; p.wc_seg = w.wcx.seg_stride; p.sc_c = w.wcx.acc_scale; p.bc = w.bc; p.g3 = w.ln3_g; p.be3 = w.ln3_b;
        p.W3 = w.w3px.w; p.w3_seg = w.w3px.seg_stride; p.sc_3 = w.w3px.acc_scale; p.b3 = w.b3;
        p.X = k == 0 ? x0 : out; p.ldx = k == 0 ? ld0 : din; p.x_bytes = (uint32_t)(rows * p.ldx * 4);
        p.Y = last ? last_out : out; p.ldy = last ? last_ld : din;
        p.YX = last ? last_split : nullptr;
        p.extra = last ? extra : nullptr; p.lde = lde;
        p.drop_mode = DROP_NONE;
        if (drop_mode != DROP_NONE && drop_p > 0.f) {
            const double t = floor((double)drop_p * 4294967296.0);
            p.drop_mode = drop_mode;
            p.drop_thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
            p.drop_scale = (float)(1.0 / (1.0 - (double)drop_p));
            p.drop_site = site0 + (uint32_t)k;
            p.drop_mask = masks ? masks + (size_t)k * mask_stride : nullptr;
        }
        if (!last) set_c(p, blocks[k + 1], (k & 1) ? ha : hb);
        launch_bn_chain(p, dh, din, st);
    }
}

// One denoiser forward up to the last attention block; result rows in ws.Y.
static HdStatus forward_body(HdModel* m, const Segs& sg, int drop_mode, const uint8_t* enc_masks, const uint8_t* conv_masks,
                             bool prune_last = false) {
    hipStream_t st = cur(m).stream;
    Workspace& ws = cur(m).ws;
    const HdConfig& c = m->cfg;
    const int d = m->d, dh = m->dh, D = m->D, Dh = m->Dh, rows = sg.rows();
    const size_t enc_stride = (size_t)sg.B * m->L * d, conv_stride = (size_t)sg.B * m->L * D;
    hipLaunchKernelGGL(embed_tokens_k, dim3((rows + 3) / 4), dim3(256), 0, st, ws.tokens, m->emb, m->emb_stats, d, ws.X, ws.ST, sg);
    const bool enc_chain = bn_chain_use(m, sg, m->enc, d, dh, c.enc_act, 2, drop_mode, ws.EXTRA != nullptr) && ws.S1;
    if (enc_chain)      // (ws.ST: the embedding rows' statistics, written by embed_tokens_k)
        bytenet_stack_chain(m, sg, m->enc, d, dh, c.enc_act, ws.X, d, ws.X, ws.FEAT, D, ws.H1, ws.H2, m->p_enc > 0.f ? drop_mode : DROP_NONE, m->p_enc, 0u,
                            enc_masks, enc_stride, ws.EXTRA, d, nullptr);
    for (int n = 0; n < c.n_encoder_layers && !enc_chain; ++n) {
        Drop dr;
        if (drop_mode != DROP_NONE && m->p_enc > 0.f) { dr.mode = drop_mode; dr.p = m->p_enc; dr.site = (uint32_t)n; dr.mask = enc_masks ? enc_masks + n * enc_stride : nullptr; }
        const bool last = n == c.n_encoder_layers - 1;
        // split-precision route with ln_sync: block n's last GEMM writes block n + 1's first operand (x_stats = X_S1 then)
        const bool chain = x3_lnsync_level(m) > 1 && ws.SYNC && x3_use(m, sg, m->enc[n].w1x) && x3_use(m, sg, m->enc[n].wcx) && x3_use(m, sg, m->enc[n].w3x);
        bytenet_block(m, sg, m->enc[n], d, dh, c.enc_act, ws.X, d, ws.H1, ws.H2, last ? ws.FEAT : ws.X, last ? D : d, dr,
                      last ? ws.EXTRA : nullptr, d, n == 0 ? X_FINAL : (chain ? X_S1 : X_PARTIALS), /*want_out_stats=*/!last,
                      nullptr, (chain && !last) ? &m->enc[n + 1] : nullptr);
    }
    if (m->debug_stop_after == 1) return HD_OK;
    const bool ax3 = att_x3(m, sg);
    const bool conv_chain = bn_chain_use(m, sg, m->conv, D, Dh, c.conv_act, 1, drop_mode, ax3 && ws.YX) && ws.S1;
    if (conv_chain) {
        launch_stats(m, ws.FEAT, D, D, rows, st);       // FEAT's static two thirds were not written by a GEMM: one statistics pass
        bytenet_stack_chain(m, sg, m->conv, D, Dh, c.conv_act, ws.FEAT, D, ws.Y, ws.Y, D, ws.G1, ws.G2, m->p_conv > 0.f ? drop_mode : DROP_NONE, m->p_conv, 64u,
                            conv_masks, conv_stride, nullptr, 0, ax3 ? ws.YX : nullptr);
    }
    for (int n = 0; n < c.dual_layers && !conv_chain; ++n) {
        Drop dr;
        if (drop_mode != DROP_NONE && m->p_conv > 0.f) { dr.mode = drop_mode; dr.p = m->p_conv; dr.site = 64u + (uint32_t)n; dr.mask = conv_masks ? conv_masks + n * conv_stride : nullptr; }
        // block 0 reads FEAT, whose static two thirds were not written by a GEMM: one explicit statistics pass
        // x3: the last block also leaves Y in split form for the first attention's Q|K|V projection
        const bool chain = x3_lnsync_level(m) > 1 && ws.SYNC && x3_use(m, sg, m->conv[n].w1x) && x3_use(m, sg, m->conv[n].wcx) && x3_use(m, sg, m->conv[n].w3x);
        bytenet_block(m, sg, m->conv[n], D, Dh, c.conv_act, n == 0 ? ws.FEAT : ws.Y, D, ws.G1, ws.G2, ws.Y, D, dr, nullptr, 0,
                      n == 0 ? X_NONE : (chain ? X_S1 : X_PARTIALS), /*want_out_stats=*/n + 1 < c.dual_layers,
                      (ax3 && n + 1 == c.dual_layers) ? ws.YX : nullptr, (chain && n + 1 < c.dual_layers) ? &m->conv[n + 1] : nullptr);
    }
    for (int n = 0; n < c.cs_layers; ++n) {
        if (m->debug_stop_after == 2 + n) return HD_OK;
        const AttBlockW& w = m->att[n];
        // at = x + A1(x)
        attention_layer(m, sg, w.a1, ws.Y, false, ws.Y, ws.AT, /*want_out_stats=*/true, ax3, ws.YX, ws.ATX);
        ws.at_in_atx = false; ws.o_is_split = ax3;
        if (m->debug_stop_after == 100 + n) return HD_OK;        // (tests: right behind the first attention of block n)
        // at = at + A2(LN1(at))      (statistics of `at` come from the out-projection's epilogue)
        if (prune_last && n == c.cs_layers - 1) { pruned_tail(m, sg, w); break; }
        // (split route: FF1 is the only reader of this sum -- the block's last residual comes from the block INPUT -- and reads it in split
        //  form, so the fp32 rows are not written: 229 MB per launch at 256 antibodies)
        attention_layer(m, sg, w.a2, ws.AT, true, ws.AT, ws.AT, /*want_out_stats=*/true, ax3, ws.ATX, ws.ATX, /*split_only=*/ax3);
        ws.at_in_atx = ax3;
        // x = FF(LN2(at)) + x       (residual from the block INPUT, cross_attention.py:282-286)
        GemmP p = base_gemm(m, sg);
        p.A = ws.AT; p.lda = D; p.W = w.wf1; p.bias = w.bf1; p.C = ws.F1; p.ldc = m->Fd; p.N = m->Fd; p.Kc = D;
        p.ln_fold = 1; p.epi_act = ACT_RELU;       // LN2 folded into wf1 / bf1
        use_partials(m, p);         // statistics of `at`: partials left by the second attention's out-projection
        if (ax3) { p.A = ws.ATX; use_x3(p, w.wf1x); p.c_split = 1; }       // F1 is only read by FF2: split form only
        launch_gemm(m, p, false, false);
        p = base_gemm(m, sg);
        p.A = ws.F1; p.lda = m->Fd; p.W = w.wf2; p.bias = w.bf2; p.C = ws.Y; p.ldc = D; p.N = D; p.Kc = m->Fd;
        p.resid = ws.Y; p.ldr = D;
        if (ax3) { use_x3(p, w.wf2x); p.C2 = ws.YX; }                      // split copy of Y for the next block's Q|K|V
        launch_gemm(m, p, false, false);
    }
    HIP_TRY(hipGetLastError());
    if (!g_launch_err.empty()) { const std::string e = g_launch_err; g_launch_err.clear(); return fail(HD_ERR_STATE, "%s", e.c_str()); }
    return HD_OK;
}

static HdStatus validate_inputs(const HdModel* m, const int32_t* tokens, const int32_t* region, const int32_t* chain, int B) {
    const int L = m->L;
    for (long i = 0; i < (long)B * L; ++i) {
        if (tokens[i] < 0 || tokens[i] >= m->cfg.n_tokens) return fail(HD_ERR_INVALID, "token %d at row %ld slot %ld out of [0,%d)", tokens[i], i / L, i % L, m->cfg.n_tokens);
        if (region[i] < 0 || region[i] >= m->cfg.n_region) return fail(HD_ERR_INVALID, "region %d at row %ld slot %ld out of [0,%d)", region[i], i / L, i % L, m->cfg.n_region);
    }
    if (m->nseg > 1) {
        if (!chain) return fail(HD_ERR_INVALID, "antibody model needs chain types [2B]");
        for (int b = 0; b < B; ++b) {
            if (chain[b] != 0) return fail(HD_ERR_INVALID, "chain[%d] = %d: heavy rows must be 0 (H)", b, chain[b]);
            if (chain[B + b] < 1 || chain[B + b] >= m->cfg.n_side) return fail(HD_ERR_INVALID, "chain[%d] = %d: light rows must be in [1,%d)", B + b, chain[B + b], m->cfg.n_side);
        }
    }
    return HD_OK;
}

static HdStatus upload_common(HdModel* m, const int32_t* tokens, const int32_t* region, const int32_t* chain, int B) {
    Workspace& ws = cur(m).ws;
    const size_t M = (size_t)B * m->L;
    HIP_TRY(hipMemcpyAsync(ws.tokens, tokens, M * sizeof(int32_t), hipMemcpyHostToDevice, cur(m).stream));
    HIP_TRY(hipMemcpyAsync(ws.region, region, M * sizeof(int32_t), hipMemcpyHostToDevice, cur(m).stream));
    if (m->nseg > 1) HIP_TRY(hipMemcpyAsync(ws.chain, chain, (size_t)2 * B * sizeof(int32_t), hipMemcpyHostToDevice, cur(m).stream));
    return HD_OK;
}

static HdStatus set_run_state(HdModel* m, uint64_t seed, uint64_t row0, uint32_t step) {
    RunState h{};
    h.step = step; h.seed_lo = (uint32_t)(seed & 0xFFFFFFFFu); h.seed_hi = (uint32_t)(seed >> 32); h.row0 = (uint32_t)row0;
    HIP_TRY(hipMemcpyAsync(cur(m).rs, &h, sizeof(h), hipMemcpyHostToDevice, cur(m).stream));
    HIP_TRY(hipStreamSynchronize(cur(m).stream));   // h is a stack object
    return HD_OK;
}

static int drop_mode_of(const HdModel* m, uint32_t flags) {
    const uint32_t f = flags & HD_DROPOUT_MASK;
    if (f == HD_DROPOUT_OFF || m->cfg.dropout <= 0.f) return DROP_NONE;
    if (f == HD_DROPOUT_INJECT) return DROP_INJECT;
    return DROP_GEN;
}

// split-precision kernels (whole path or attention only) currently in use / switched off for good by the range guard
static bool split_active(const HdModel* m) { return (m->x3 || m->attn_x3) && !m->x3_suspended; }
// which kernels a captured step graph holds: split kernels in use, ln_sync level
static int kernel_set(const HdModel* m) { return (split_active(m) ? 1 : 0) | ((m->lnsync_level & 3) << 1) | (m->nlanes << 4); }       // (the lane count picks kernels too: launch_gemm)
static void suspend_split(HdModel* m) {
    // said once per handle (stderr; HUDIFF_QUIET=1 silences it): from here on the handle runs the all-fp32 kernels, at their speed
    static const bool quiet = [] { const char* e = getenv("HUDIFF_QUIET"); return e && atoi(e) == 1; }();
    if (!m->x3_suspended && !quiet)
        fprintf(stderr, "[hudiff_hip] an activation left the fp16 range (|x| >= 65504): this call is repeated on the fp32 kernels and the handle "
                        "stays on them (hd_precision_info.range_fallbacks)\n");
    m->x3_suspended = true;
    m->range_fallbacks += 1;
    for (auto& ln : m->lane) ln.drop_graphs();      // captured with the split kernels
}

static void suspend_lnsync(HdModel* m) {
    static const bool quiet = [] { const char* e = getenv("HUDIFF_QUIET"); return e && atoi(e) == 1; }();
    if (!quiet)
        fprintf(stderr, "[hudiff_hip] an ln_sync meeting timed out (the blocks of an M tile did not run together): this call is repeated with "
                        "separate LayerNorm passes and the handle keeps them (hd_precision_report.lnsync_fallbacks)\n");
    m->lnsync_level = 0;
    m->lnsync_fallbacks += 1;
    m->debug_lnsync_fail = false;
    for (auto& ln : m->lane) ln.drop_graphs();      // captured with the meeting epilogues
}

// Guard flags of the lanes of the current call / session (RunState::pad: [0] non-finite logits, [1] range guard, [2] ln_sync guard).
// Synchronises the lanes.  A guard that fired switches the failing kernels off (suspend_*) and marks the steps run since the last
// begin / restart invalid (m->s_dirty): the caller repeats them.  One guard per round: a failed meeting leaves garbage that can trip
// the range guard too, so the range flag of such a run is not believed -- the repeat raises it again if it is real.
static HdStatus check_guards(HdModel* m, int nlanes, bool* numeric) {
    uint32_t pad[4] = {0, 0, 0, 0};
    for (int l = 0; l < nlanes; ++l) HIP_TRY(hipStreamSynchronize(m->lane[l].stream));
    for (int l = 0; l < nlanes; ++l) {
        RunState h{};
        HIP_TRY(hipMemcpy(&h, m->lane[l].rs, sizeof(h), hipMemcpyDeviceToHost));
        for (int i = 0; i < 4; ++i) pad[i] |= h.pad[i];
    }
    if (numeric) *numeric = pad[0] != 0;
    if (pad[3]) m->lnsync_cross_xcd = true;         // an ln_sync meeting spanned two XCDs: correct (write-through hand-over), slower
    if (m->s_dirty) return HD_OK;                   // already known; nothing more is read out of an invalid run
    if (pad[2] && m->x3 && !m->x3_suspended && m->lnsync_level > 0) { suspend_lnsync(m); m->s_dirty = true; }
    else if (pad[1] && split_active(m)) { suspend_split(m); m->s_dirty = true; }
    return HD_OK;
}

template <typename T>
static HdStatus ensure_buf(HdModel* m, T** p, size_t* cap, size_t n) {
    if (n <= *cap) return HD_OK;
    HD_TRY(dalloc(cur(m).ws, p, n));
    *cap = n;
    return HD_OK;
}

extern "C" HdStatus hd_forward(HdModel* m, const int32_t* tokens, const int32_t* region, const int32_t* chain,
                               int32_t B, uint32_t flags, uint64_t seed, uint64_t row0, uint32_t step,
                               const uint8_t* enc_masks, const uint8_t* conv_masks, float* logits) {
    if (!m || !tokens || !region || !logits) return fail(HD_ERR_INVALID, "hd_forward: null argument");
    if (!m->finalized) return fail(HD_ERR_STATE, "hd_forward: call hd_finalize first");
    if (m->in_session) return fail(HD_ERR_STATE, "hd_forward: a sampling session is open (hd_sample_end it first)");
    if (B < 0) return fail(HD_ERR_INVALID, "hd_forward: B = %d", B);
    if (B == 0) return HD_OK;
    HIP_TRY(hipSetDevice(m->device));
    m->cl = 0;
    HD_TRY(validate_inputs(m, tokens, region, chain, B));
    const int dm = drop_mode_of(m, flags);
    if (dm == DROP_INJECT && (!enc_masks || !conv_masks)) return fail(HD_ERR_INVALID, "hd_forward: HD_DROPOUT_INJECT needs enc_masks and conv_masks");
    HD_TRY(ensure_ws(m, B));
    Workspace& ws = cur(m).ws;
    const Segs sg = make_segs(m, B);
    HD_TRY(upload_common(m, tokens, region, chain, B));
    HD_TRY(set_run_state(m, seed, row0, step));
    const uint8_t *dem = nullptr, *dcm = nullptr;
    if (dm == DROP_INJECT) {
        const size_t ne = (size_t)m->cfg.n_encoder_layers * B * m->L * m->d, nc = (size_t)m->cfg.dual_layers * B * m->L * m->D;
        HD_TRY(ensure_buf(m, &ws.enc_masks, &ws.enc_cap, ne));
        HD_TRY(ensure_buf(m, &ws.conv_masks, &ws.conv_cap, nc));
        HIP_TRY(hipMemcpyAsync(ws.enc_masks, enc_masks, ne, hipMemcpyHostToDevice, cur(m).stream));
        HIP_TRY(hipMemcpyAsync(ws.conv_masks, conv_masks, nc, hipMemcpyHostToDevice, cur(m).stream));
        dem = ws.enc_masks; dcm = ws.conv_masks;
    }
    HD_TRY(static_branch(m, sg));
    HD_TRY(forward_body(m, sg, dm, dem, dcm));
    const int rows = sg.rows();
    hipLaunchKernelGGL(decode_all_k, dim3((rows + 3) / 4), dim3(256), 0, cur(m).stream, ws.Y, m->D, m->head, m->cfg.n_tokens, ws.LOGITS, sg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(logits, ws.LOGITS, (size_t)rows * m->cfg.n_tokens * sizeof(float), hipMemcpyDeviceToHost, cur(m).stream));
    HIP_TRY(hipStreamSynchronize(cur(m).stream));
    // guards of the split-precision kernels (range, ln_sync): the call is repeated on the kernels that do not need them
    m->s_dirty = false;
    HD_TRY(check_guards(m, 1, nullptr));
    if (m->s_dirty) {
        m->s_dirty = false;
        const HdStatus s = hd_forward(m, tokens, region, chain, B, flags, seed, row0, step, enc_masks, conv_masks, logits);
        m->last_call_repeated = true;
        return s;
    }
    m->last_call_repeated = false;
    return HD_OK;
}

// ---- sampling session ---------------------------------------------------------------------------
static HdStatus one_step(HdModel* m, const Segs& sg, int dm, const uint8_t* em, const uint8_t* cm) {
    HdModel::Lane& ln = cur(m);
    const bool prune = !(m->sflags & HD_NO_PRUNE);
    HD_TRY(forward_body(m, sg, dm, em, cm, prune));
    Workspace& ws = ln.ws;
    // the injected Exp(1) noise lives once, for the whole batch, in the model (m->qnoise)
    // (the last workgroup of sample_step_k advances the step)
    hipLaunchKernelGGL(sample_step_k, dim3(sg.B), dim3(64 * SS_WAVES), 0, ln.stream, prune ? ws.Xc : ws.Y, m->D, m->head, ws.tokens, ws.order,
                       ws.T, m->sTmax, m->s_has_q ? m->qnoise : nullptr, m->sB, ln.row_off, ln.rs, sg, prune ? 1 : 0, 1);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

static HdStatus sample_begin_impl(HdModel* m, const int32_t* tokens, const int32_t* region, const int32_t* chain,
                                  const int32_t* order, const int32_t* T, int32_t B, int32_t Tmax, uint32_t flags,
                                  uint64_t seed, uint64_t row0, const float* q_noise,
                                  const uint8_t* enc_masks, const uint8_t* conv_masks) {
    if (!m || !tokens || !region || !T || (Tmax > 0 && !order)) return fail(HD_ERR_INVALID, "hd_sample_begin: null argument");
    if (!m->finalized) return fail(HD_ERR_STATE, "hd_sample_begin: call hd_finalize first");
    if (m->in_session) return fail(HD_ERR_STATE, "hd_sample_begin: session already open");
    if (B < 0 || Tmax < 0) return fail(HD_ERR_INVALID, "hd_sample_begin: B = %d, Tmax = %d", B, Tmax);
    HIP_TRY(hipSetDevice(m->device));
    m->sB = B; m->sTmax = Tmax; m->sflags = flags; m->s_has_q = q_noise != nullptr; m->timed = false; m->last_steps = 0;
    m->s_seed = seed; m->s_steps = 0; m->s_dirty = false;
    m->nlanes = 1; m->cl = 0;
    if (B == 0) { m->in_session = true; return HD_OK; }
    HD_TRY(validate_inputs(m, tokens, region, chain, B));
    for (int b = 0; b < B; ++b) {
        if (T[b] < 0 || T[b] > Tmax) return fail(HD_ERR_INVALID, "T[%d] = %d out of [0,%d]", b, T[b], Tmax);
        for (int t = 0; t < T[b]; ++t) {
            int s = order[(size_t)b * Tmax + t];
            if (s < 0 || s >= m->L) return fail(HD_ERR_INVALID, "order[%d,%d] = %d out of [0,%d)", b, t, s, m->L);
        }
    }
    const int dm = drop_mode_of(m, flags);
    if (dm == DROP_INJECT && (!enc_masks || !conv_masks)) return fail(HD_ERR_INVALID, "hd_sample_begin: HD_DROPOUT_INJECT needs masks");
    // two concurrent half-batches unless the batch is small, masks are injected (their layout is per full batch)
    // or the caller asked for one lane
    const long lane_min_b = m->opt[HD_OPT_LANE_MIN_ROWS];
    m->nlanes = (B >= lane_min_b && dm != DROP_INJECT && !(flags & HD_ONE_LANE)) ? (int)m->opt[HD_OPT_LANES] : 1;
    if (m->nlanes > B) m->nlanes = B;
    for (int l = 0, off = 0; l < m->nlanes; ++l) {            // balanced contiguous row blocks
        const int Bl = B / m->nlanes + (l < B % m->nlanes ? 1 : 0);
        m->lane[l].B = Bl; m->lane[l].row_off = off;
        off += Bl;
    }
    std::vector<int32_t> chain_l;
    for (int l = 0; l < m->nlanes; ++l) {
        m->cl = l;
        HdModel::Lane& ln = m->lane[l];
        const int Bl = ln.B, off = ln.row_off;
        HD_TRY(ensure_ws(m, Bl));
        Workspace& ws = ln.ws;
        const Segs sg = make_segs(m, Bl);
        const int32_t* ch = nullptr;
        if (m->nseg > 1) {            // chain[0:B] heavy ids, chain[B:2B] light ids -> this lane's [heavy | light]
            chain_l.assign((size_t)2 * Bl, 0);
            for (int b = 0; b < Bl; ++b) { chain_l[b] = chain[off + b]; chain_l[Bl + b] = chain[B + off + b]; }
            ch = chain_l.data();
        }
        HD_TRY(upload_common(m, tokens + (size_t)off * m->L, region + (size_t)off * m->L, ch, Bl));
        HIP_TRY(hipStreamSynchronize(ln.stream));          // chain_l is reused by the next lane
        {
            const size_t need = (size_t)Bl * (Tmax > 0 ? Tmax : 1);
            if (need > (size_t)ws.capT) {
                if (ws.order) {                 // regrown: release the old buffer now, not at the next free_ws
                    for (auto it = ws.owned.begin(); it != ws.owned.end(); ++it)
                        if (*it == (void*)ws.order) { ws.owned.erase(it); break; }
                    HIP_TRY(hipStreamSynchronize(ln.stream));
                    hipFree(ws.order);
                    ws.order = nullptr; ws.capT = 0;
                    // a captured graph holds the old pointer
                    ln.drop_graphs();
                }
                HD_TRY(dalloc(ws, &ws.order, need));
                ws.capT = (int)need;
            }
        }
        if (Tmax > 0) HIP_TRY(hipMemcpyAsync(ws.order, order + (size_t)off * Tmax, (size_t)Bl * Tmax * sizeof(int32_t), hipMemcpyHostToDevice, ln.stream));
        HIP_TRY(hipMemcpyAsync(ws.T, T + off, (size_t)Bl * sizeof(int32_t), hipMemcpyHostToDevice, ln.stream));
        if (l == 0 && q_noise && Tmax > 0) {
            const size_t n = (size_t)Tmax * B * 22;
            if (n > m->qnoise_cap) {
                for (auto& o : m->lane) if (o.stream) HIP_TRY(hipStreamSynchronize(o.stream));
                if (m->qnoise) { hipFree(m->qnoise); m->qnoise = nullptr; m->qnoise_cap = 0; }
                HIP_TRY(hipMalloc(&m->qnoise, n * sizeof(float)));
                m->qnoise_cap = n;
            }
            HIP_TRY(hipMemcpyAsync(m->qnoise, q_noise, n * sizeof(float), hipMemcpyHostToDevice, ln.stream));
        }
        if (dm == DROP_INJECT && Tmax > 0) {
            const size_t ne = (size_t)Tmax * m->cfg.n_encoder_layers * B * m->L * m->d, nc = (size_t)Tmax * m->cfg.dual_layers * B * m->L * m->D;
            HD_TRY(ensure_buf(m, &ws.enc_masks, &ws.enc_cap, ne));
            HD_TRY(ensure_buf(m, &ws.conv_masks, &ws.conv_cap, nc));
            HIP_TRY(hipMemcpyAsync(ws.enc_masks, enc_masks, ne, hipMemcpyHostToDevice, ln.stream));
            HIP_TRY(hipMemcpyAsync(ws.conv_masks, conv_masks, nc, hipMemcpyHostToDevice, ln.stream));
        }
        HIP_TRY(hipMemcpyAsync(ws.tokens0, ws.tokens, (size_t)Bl * m->L * sizeof(int32_t), hipMemcpyDeviceToDevice, ln.stream));
        HD_TRY(set_run_state(m, seed, row0 + (uint64_t)off, 0));
        HD_TRY(static_branch(m, sg));
    }
    for (int l = 0; l < m->nlanes; ++l) HIP_TRY(hipStreamSynchronize(m->lane[l].stream));
    m->cl = 0;
    m->s_row0 = row0;
    m->in_session = true;
    return HD_OK;
}

extern "C" HdStatus hd_sample_begin(HdModel* m, const int32_t* tokens, const int32_t* region, const int32_t* chain,
                                    const int32_t* order, const int32_t* T, int32_t B, int32_t Tmax, uint32_t flags,
                                    uint64_t seed, uint64_t row0, const float* q_noise,
                                    const uint8_t* enc_masks, const uint8_t* conv_masks) {
    const HdStatus s = sample_begin_impl(m, tokens, region, chain, order, T, B, Tmax, flags, seed, row0, q_noise, enc_masks, conv_masks);
    if (s != HD_OK && m && !m->in_session) {     // a failure half-way through the lane loop must not leave lane state behind
        m->cl = 0; m->nlanes = 1; m->sB = 0; m->timed = false;
    }
    return s;
}

extern "C" HdStatus hd_sample_restart(HdModel* m, uint64_t seed) {
    if (!m || !m->in_session) return fail(HD_ERR_STATE, "hd_sample_restart: no open session");
    if (m->sB == 0) { m->s_seed = seed; m->s_steps = 0; return HD_OK; }
    HIP_TRY(hipSetDevice(m->device));
    // a guard that fired in the sample being discarded still switches its kernels off (and is counted) before the flags are cleared:
    // the next sample must not run on kernels that just failed (the begin / run / restart / run / ... / end pattern of bench.py)
    if (m->s_steps > 0) HD_TRY(check_guards(m, m->nlanes, nullptr));
    m->s_seed = seed; m->s_steps = 0; m->s_dirty = false;
    for (int l = 0; l < m->nlanes; ++l) {
        m->cl = l;
        HdModel::Lane& ln = m->lane[l];
        HIP_TRY(hipMemcpyAsync(ln.ws.tokens, ln.ws.tokens0, (size_t)ln.B * m->L * sizeof(int32_t), hipMemcpyDeviceToDevice, ln.stream));
        HD_TRY(set_run_state(m, seed, m->s_row0 + (uint64_t)ln.row_off, 0));
    }
    m->cl = 0;
    return HD_OK;
}

extern "C" HdStatus hd_sample_run(HdModel* m, int32_t t0, int32_t t1) {
    if (!m || !m->in_session) return fail(HD_ERR_STATE, "hd_sample_run: no open session");
    if (t0 < 0 || t1 < t0 || t1 > m->sTmax) return fail(HD_ERR_INVALID, "hd_sample_run: steps [%d,%d) outside [0,%d]", t0, t1, m->sTmax);
    if (m->sB == 0 || t1 == t0) return HD_OK;
    HIP_TRY(hipSetDevice(m->device));
    if (t1 > m->s_steps) m->s_steps = t1;
    const int dm = drop_mode_of(m, m->sflags);
    const bool use_graph = !(m->sflags & HD_NO_GRAPH) && dm != DROP_INJECT;
    for (int l = 0; l < m->nlanes; ++l) {
        m->cl = l;
        HdModel::Lane& ln = m->lane[l];
        const Segs sg = make_segs(m, ln.B);
        hipLaunchKernelGGL(set_step_k, dim3(1), dim3(1), 0, ln.stream, ln.rs, (uint32_t)t0);
        if (!use_graph) continue;
        const uint32_t gflags = m->sflags & HD_NO_PRUNE;
        if (!ln.graph_exec || ln.graph_B != ln.B || ln.graph_flags != gflags || ln.graph_drop != dm || ln.graph_q != m->s_has_q ||
            ln.graph_Tmax != m->sTmax || ln.graph_qB != m->sB || ln.graph_qoff != ln.row_off ||
            ln.graph_qptr != (m->s_has_q ? m->qnoise : nullptr) || ln.graph_x3 != kernel_set(m)) {
            ln.drop_graphs();
            HIP_TRY(hipStreamSynchronize(ln.stream));
            HIP_TRY(hipStreamBeginCapture(ln.stream, hipStreamCaptureModeThreadLocal));
            HdStatus s = one_step(m, sg, dm, nullptr, nullptr);
            hipError_t e = hipStreamEndCapture(ln.stream, &ln.graph);
            if (s != HD_OK) { m->cl = 0; return s; }
            if (e != hipSuccess) { m->cl = 0; return fail(HD_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e)); }
            HIP_TRY(hipGraphInstantiate(&ln.graph_exec, ln.graph, nullptr, nullptr, 0));
            ln.graph_B = ln.B; ln.graph_flags = gflags; ln.graph_drop = dm; ln.graph_q = m->s_has_q; ln.graph_Tmax = m->sTmax;
            ln.graph_qB = m->sB; ln.graph_qoff = ln.row_off; ln.graph_qptr = m->s_has_q ? m->qnoise : nullptr;
            ln.graph_x3 = kernel_set(m);
        }
    }
    for (int l = 0; l < m->nlanes; ++l) HIP_TRY(hipEventRecord(m->lane[l].ev0, m->lane[l].stream));
    // HD_LOOP_GRAPH (or HUDIFF_LOOP_GRAPH=1): the whole [t0, t1) loop of a lane is one graph -- a chain of t1 - t0 child-graph
    // nodes of the captured step (the step counter lives on the device, so every step is the same node) -- kept for the next
    // sample with the same number of steps.  Default: the step graph is launched t1 - t0 times, which measured 2.4 % faster
    // (the lanes interleave more freely between step graphs than inside two 20 000-node graphs).
    const bool loop_graph = m->opt[HD_OPT_LOOP_GRAPH] != 0 || (m->sflags & HD_LOOP_GRAPH);
    if (use_graph && loop_graph && t1 - t0 > 1) {
        for (int l = 0; l < m->nlanes; ++l) {
            HdModel::Lane& ln = m->lane[l];
            if (!ln.loop_exec || ln.loop_steps != t1 - t0) {
                if (ln.loop_exec) { hipGraphExecDestroy(ln.loop_exec); ln.loop_exec = nullptr; }
                if (ln.loop_graph) { hipGraphDestroy(ln.loop_graph); ln.loop_graph = nullptr; }
                HIP_TRY(hipGraphCreate(&ln.loop_graph, 0));
                hipGraphNode_t prev = nullptr;
                for (int t = t0; t < t1; ++t) {
                    hipGraphNode_t node = nullptr;
                    HIP_TRY(hipGraphAddChildGraphNode(&node, ln.loop_graph, prev ? &prev : nullptr, prev ? 1 : 0, ln.graph));
                    prev = node;
                }
                HIP_TRY(hipGraphInstantiate(&ln.loop_exec, ln.loop_graph, nullptr, nullptr, 0));
                ln.loop_steps = t1 - t0;
            }
        }
        for (int l = 0; l < m->nlanes; ++l) HIP_TRY(hipGraphLaunch(m->lane[l].loop_exec, m->lane[l].stream));
    } else if (use_graph) {
        // the lanes are fed alternately; on the device they run concurrently and drift freely (no cross-lane edges)
        for (int t = t0; t < t1; ++t)
            for (int l = 0; l < m->nlanes; ++l) HIP_TRY(hipGraphLaunch(m->lane[l].graph_exec, m->lane[l].stream));
    } else {
        for (int t = t0; t < t1; ++t)
            for (int l = 0; l < m->nlanes; ++l) {
                m->cl = l;
                HdModel::Lane& ln = m->lane[l];
                const Segs sg = make_segs(m, ln.B);
                const size_t es = (size_t)m->cfg.n_encoder_layers * ln.B * m->L * m->d, cs = (size_t)m->cfg.dual_layers * ln.B * m->L * m->D;
                HD_TRY(one_step(m, sg, dm, dm == DROP_INJECT ? ln.ws.enc_masks + (size_t)t * es : nullptr,
                                dm == DROP_INJECT ? ln.ws.conv_masks + (size_t)t * cs : nullptr));
            }
    }
    for (int l = 0; l < m->nlanes; ++l) HIP_TRY(hipEventRecord(m->lane[l].ev1, m->lane[l].stream));
    m->cl = 0;
    m->timed = true;
    m->last_steps = t1 - t0;
    return HD_OK;
}

extern "C" HdStatus hd_sync(HdModel* m) {
    if (!m) return fail(HD_ERR_INVALID, "hd_sync: null model");
    HIP_TRY(hipSetDevice(m->device));
    for (auto& ln : m->lane) HIP_TRY(hipStreamSynchronize(ln.stream));
    // inside a session the guards are looked at here too: the steps run so far are then marked invalid (hd_sample_end repeats them)
    // and the failing kernels are off for whatever is enqueued next
    if (m->in_session && m->sB > 0 && m->s_steps > 0) HD_TRY(check_guards(m, m->nlanes, nullptr));
    return HD_OK;
}

static HdStatus sample_collect(HdModel* m, int32_t* tokens, bool* numeric) {
    for (int l = 0; l < m->nlanes; ++l) {
        HdModel::Lane& ln = m->lane[l];
        HIP_TRY(hipMemcpyAsync(tokens + (size_t)ln.row_off * m->L, ln.ws.tokens, (size_t)ln.B * m->L * sizeof(int32_t), hipMemcpyDeviceToHost, ln.stream));
    }
    return check_guards(m, m->nlanes, numeric);
}

extern "C" HdStatus hd_sample_tokens(HdModel* m, int32_t* tokens) {
    if (!m || !m->in_session) return fail(HD_ERR_STATE, "hd_sample_tokens: no open session");
    if (m->sB == 0) return HD_OK;
    if (!tokens) return fail(HD_ERR_INVALID, "hd_sample_tokens: null tokens");
    HIP_TRY(hipSetDevice(m->device));
    HD_TRY(sample_collect(m, tokens, nullptr));
    if (m->s_dirty) return fail(HD_ERR_STATE, "hd_sample_tokens: a guard of the split-precision kernels fired during these steps; their tokens "
                                               "are invalid (hd_sample_end repeats the sample)");
    return HD_OK;
}

static HdStatus sample_end_impl(HdModel* m, int32_t* tokens, bool* numeric) {
    HIP_TRY(hipSetDevice(m->device));
    HD_TRY(sample_collect(m, tokens, numeric));
    m->last_call_repeated = false;
    // Guards (range, ln_sync; see check_guards): some split-precision kernel could not be trusted during this sample.  The whole
    // sample is repeated on the kernels that do not need the guard -- same resident inputs, same noise key, same steps -- and the
    // handle stays on them.  At most one repeat per guard.
    for (int attempt = 0; m->s_dirty && attempt < 2; ++attempt) {
        const int steps = m->s_steps;
        m->s_dirty = false;                          // (hd_sample_restart would otherwise look at the same flags again)
        m->s_steps = 0;
        HD_TRY(hd_sample_restart(m, m->s_seed));
        if (steps > 0) HD_TRY(hd_sample_run(m, 0, steps));
        HD_TRY(sample_collect(m, tokens, numeric));
        m->last_call_repeated = true;
    }
    if (m->s_dirty) return fail(HD_ERR_STATE, "hd_sample_end: a guard of the split-precision kernels fired on the fp32 kernels (internal error)");
    return HD_OK;
}

extern "C" HdStatus hd_sample_end(HdModel* m, int32_t* tokens) {
    if (!m || !m->in_session) return fail(HD_ERR_STATE, "hd_sample_end: no open session");
    if (m->sB == 0) { m->in_session = false; return HD_OK; }
    if (!tokens) { m->in_session = false; return fail(HD_ERR_INVALID, "hd_sample_end: null tokens"); }
    bool numeric = false;
    const HdStatus s = sample_end_impl(m, tokens, &numeric);
    m->in_session = false;                           // on every path: a failed end never leaves a half-open session behind
    m->s_dirty = false;
    m->cl = 0;
    if (s != HD_OK) return s;
    if (numeric)
        return fail(HD_ERR_NUMERIC, "hd_sample: non-finite logits (NaN / inf) at some denoiser step -- weights or inputs out of range; "
                                    "the reference's torch.multinomial raises at this point");
    return HD_OK;
}

extern "C" HdStatus hd_precision_report(HdModel* m, HdPrecisionInfo* out, size_t size) {
    if (!m || !out) return fail(HD_ERR_INVALID, "hd_precision_report: null argument");
    HdPrecisionInfo r{};
    r.precision = m->finalized ? m->precision : m->precision_req;
    r.split_built = (m->x3 ? 1 : 0) | (m->attn_x3 ? 2 : 0);
    r.split_in_use = (m->finalized && split_active(m)) ? 1 : 0;
    r.lnsync_in_use = (m->finalized && m->x3 && !m->x3_suspended && m->lnsync_level > 0) ? 1 : 0;
    r.range_fallbacks = m->range_fallbacks;
    r.lnsync_fallbacks = m->lnsync_fallbacks;
    r.last_call_repeated = m->last_call_repeated ? 1 : 0;
    r.lnsync_cross_xcd = m->lnsync_cross_xcd ? 1 : 0;
    memcpy(out, &r, size < sizeof(r) ? size : sizeof(r));
    return HD_OK;
}

extern "C" HdStatus hd_precision_reset(HdModel* m) {
    if (!m) return fail(HD_ERR_INVALID, "hd_precision_reset: null model");
    if (m->in_session) return fail(HD_ERR_STATE, "hd_precision_reset: a sampling session is open");
    if (m->x3_suspended || m->lnsync_level != m->lnsync_level_cfg) {
        m->x3_suspended = false;
        m->lnsync_level = m->lnsync_level_cfg;
        for (auto& ln : m->lane) ln.drop_graphs();
    }
    return HD_OK;
}

extern "C" HdStatus hd_precision_info(HdModel* m, int32_t* split_built, int32_t* split_in_use, int64_t* range_fallbacks) {
    if (!m) return fail(HD_ERR_INVALID, "hd_precision_info: null model");
    if (split_built) *split_built = (m->x3 ? 1 : 0) | (m->attn_x3 ? 2 : 0);
    if (split_in_use) *split_in_use = split_active(m) ? 1 : 0;
    if (range_fallbacks) *range_fallbacks = m->range_fallbacks;
    return HD_OK;
}

extern "C" HdStatus hd_set_option(HdModel* m, int32_t option, int64_t value) {
    if (!m) return fail(HD_ERR_INVALID, "hd_set_option: null model");
    if (option < 0 || option >= HD_OPT_COUNT) return fail(HD_ERR_INVALID, "hd_set_option: unknown option %d", option);
    const OptDef& d = OPTS[option];
    if (!opt_legal(d, value)) return fail(HD_ERR_INVALID, "hd_set_option: value %lld is not legal for option %d (%s)", (long long)value, option, d.env);
    if (m->in_session) return fail(HD_ERR_STATE, "hd_set_option: a sampling session is open");
    if (d.create_only && m->finalized) return fail(HD_ERR_STATE, "hd_set_option: option %d (%s) is fixed at hd_finalize", option, d.env);
    if (m->opt[option] == value) return HD_OK;
    m->opt[option] = value;
    if (option == HD_OPT_LNSYNC_LEVEL && m->lnsync_level == m->lnsync_level_cfg) m->lnsync_level = (int)value;      // (a handle whose guard fired keeps level 0 until hd_precision_reset)
    if (option == HD_OPT_LNSYNC_LEVEL) m->lnsync_level_cfg = (int)value;
    if (m->finalized) {
        hipSetDevice(m->device);
        for (auto& ln : m->lane) { if (ln.stream) hipStreamSynchronize(ln.stream); ln.drop_graphs(); }      // captured with the old choice
    }
    return HD_OK;
}

extern "C" HdStatus hd_get_option(HdModel* m, int32_t option, int64_t* value) {
    if (!m || !value) return fail(HD_ERR_INVALID, "hd_get_option: null argument");
    if (option < 0 || option >= HD_OPT_COUNT) return fail(HD_ERR_INVALID, "hd_get_option: unknown option %d", option);
    *value = m->opt[option];
    return HD_OK;
}

extern "C" HdStatus hd_sample(HdModel* m, int32_t* tokens, const int32_t* region, const int32_t* chain,
                              const int32_t* order, const int32_t* T, int32_t B, int32_t Tmax, uint32_t flags,
                              uint64_t seed, uint64_t row0, const float* q_noise,
                              const uint8_t* enc_masks, const uint8_t* conv_masks) {
    HD_TRY(hd_sample_begin(m, tokens, region, chain, order, T, B, Tmax, flags, seed, row0, q_noise, enc_masks, conv_masks));
    int tmax_eff = 0;
    for (int b = 0; b < B; ++b) tmax_eff = T[b] > tmax_eff ? T[b] : tmax_eff;
    HdStatus s = hd_sample_run(m, 0, tmax_eff);
    if (s != HD_OK) { m->in_session = false; m->cl = 0; return s; }
    return hd_sample_end(m, tokens);
}

extern "C" HdStatus hd_last_run_ms(HdModel* m, float* ms, int32_t* steps) {
    if (!m || !ms) return fail(HD_ERR_INVALID, "hd_last_run_ms: null argument");
    if (!m->timed) return fail(HD_ERR_STATE, "hd_last_run_ms: no timed run");
    HIP_TRY(hipSetDevice(m->device));
    float best = 0.f;
    for (int l = 0; l < m->nlanes; ++l) {       // from the first lane's start event to the last lane's end event
        float t = 0.f;
        HIP_TRY(hipEventSynchronize(m->lane[l].ev1));
        HIP_TRY(hipEventElapsedTime(&t, m->lane[0].ev0, m->lane[l].ev1));
        best = t > best ? t : best;
    }
    *ms = best;
    if (steps) *steps = m->last_steps;
    return HD_OK;
}

// ---- debugging aids (used by tests to localise a parity failure; not part of the product path) ---------
extern "C" HdStatus hd_debug_stop_after(HdModel* m, int32_t stage) {
    if (!m) return fail(HD_ERR_INVALID, "hd_debug_stop_after: null model");
    m->debug_stop_after = stage;   // 1: after the token encoder (+static add), 2+n: before attention block n
    return HD_OK;
}

extern "C" HdStatus hd_debug_fail_next_lnsync(HdModel* m) {
    if (!m) return fail(HD_ERR_INVALID, "hd_debug_fail_next_lnsync: null model");
    m->debug_lnsync_fail = true;   // the ln_sync meetings of the kernels launched / captured next give up after one poll
    if (m->finalized) {
        HIP_TRY(hipSetDevice(m->device));
        for (auto& ln : m->lane) { if (ln.stream) hipStreamSynchronize(ln.stream); ln.drop_graphs(); }
    }
    return HD_OK;
}

extern "C" HdStatus hd_debug_scatter_lnsync(HdModel* m, int32_t on) {
    if (!m) return fail(HD_ERR_INVALID, "hd_debug_scatter_lnsync: null model");
    if (!m->finalized) return fail(HD_ERR_STATE, "hd_debug_scatter_lnsync: call hd_finalize first");
    if (m->in_session) return fail(HD_ERR_STATE, "hd_debug_scatter_lnsync: a sampling session is open");
    m->debug_lnsync_scatter = on != 0;
    HIP_TRY(hipSetDevice(m->device));                // hd_set_option's sequence: nothing captured with the old placement may still be in flight
    for (auto& ln : m->lane) { if (ln.stream) hipStreamSynchronize(ln.stream); ln.drop_graphs(); }
    return HD_OK;
}

extern "C" HdStatus hd_debug_read(HdModel* m, const char* name, int32_t B, float* out, int64_t n_floats) {
    if (!m || !name || !out) return fail(HD_ERR_INVALID, "hd_debug_read: null argument");
    HIP_TRY(hipSetDevice(m->device));
    const Workspace& ws = cur(m).ws;
    const std::string k(name);
    const float* src = nullptr;
    int width = 0;
    bool x16 = false;                                             // rows are X16 split rows (x16_hi(): per 16 columns 16 fp16 high parts, then 16 low parts)
    if (k == "FEAT") { src = ws.FEAT; width = m->D; }
    else if (k == "Y") { src = ws.Y; width = m->D; }
    else if (k == "X") { src = ws.X; width = m->d; }
    else if (k == "POS") { src = ws.POS; width = m->d; }
    else if (k == "EXTRA") { src = ws.EXTRA; width = m->d; }
    // "AT": at = x + A1(x) behind the first attention of a block, at + A2(LN1(at)) behind the second.  On the split route the second sum
    // exists in split form only (ws.ATX: FF1 is its one reader), so it is decoded from there (hi + lo: 22 significand bits) -- the name
    // means the same tensor on every route.  "ATX" / "YX" / "O": decoded split rows as well when the split route wrote them.
    else if (k == "AT") { src = ws.at_in_atx ? ws.ATX : ws.AT; x16 = ws.at_in_atx; width = m->D; }
    else if (k == "ATX") { src = ws.ATX; x16 = true; width = m->D; }
    else if (k == "YX") { src = ws.YX; x16 = true; width = m->D; }
    else if (k == "O") { src = ws.O; x16 = ws.o_is_split; width = m->A; }
    else if (k == "QKV") { src = ws.QKV; width = 3 * m->A; }
    else return fail(HD_ERR_INVALID, "hd_debug_read: unknown buffer %s", name);
    if (!src) return fail(HD_ERR_STATE, "hd_debug_read: buffer %s does not exist on this handle's route", name);
    if (B > ws.capB || n_floats != (int64_t)B * m->L * width) return fail(HD_ERR_INVALID, "hd_debug_read: size mismatch");
    // rows are segment-major on the device; return them as [B, L, width]
    std::vector<float> tmp((size_t)n_floats);
    HIP_TRY(hipStreamSynchronize(cur(m).stream));
    HIP_TRY(hipMemcpy(tmp.data(), src, (size_t)n_floats * sizeof(float), hipMemcpyDeviceToHost));
    if (x16) {
        std::vector<float> row((size_t)width);
        for (size_t r = 0; r < (size_t)B * m->L; ++r) {
            const _Float16* h = reinterpret_cast<const _Float16*>(tmp.data() + r * width);
            for (int c = 0; c < width; ++c) row[c] = (float)h[x16_hi(c)] + (float)h[x16_hi(c) + X16_LO];
            memcpy(tmp.data() + r * width, row.data(), sizeof(float) * width);
        }
    }
    const Segs sg = make_segs(m, B);
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < m->L; ++l)
            memcpy(out + ((size_t)b * m->L + l) * width, tmp.data() + (size_t)sg.row(b, l) * width, sizeof(float) * width);
    return HD_OK;
}
