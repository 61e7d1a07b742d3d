// C-ABI entry points of libmpe_hip.so (see include/mpe.h).  Host-side orchestration only:
// argument checks, workspace, weight upload, and the stream-ordered launch sequences.
#include <cmath>
#include <cstdarg>
#include <new>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mpe_internal.h"

using namespace mpe;

namespace {

int fail(mpe_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define HIPCHK(ctx, expr)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, MPE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Every entry point runs on the device the context was created on (one process per GPU is the
// deployment, but a second context on another device in the same process must also work).
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const mpe_ctx *ctx) {
        if (!ctx) return;
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != ctx->device) {
            prev = cur;
            (void)hipSetDevice(ctx->device);
        }
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

void dev_free(mpe_ctx *ctx, void *p) {
    if (!p) return;
    for (size_t i = 0; i < ctx->owned.size(); ++i)
        if (ctx->owned[i] == p) {
            (void)hipFree(p);
            ctx->owned.erase(ctx->owned.begin() + i);
            return;
        }
}

template <typename T>
int dev_alloc(mpe_ctx *ctx, T **p, size_t count, bool zero = true) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(ctx, MPE_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    if (zero) {
        e = hipMemset(q, 0, bytes);
        if (e != hipSuccess) return fail(ctx, MPE_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    }
    ctx->owned.push_back(q);
    *p = static_cast<T *>(q);
    return MPE_OK;
}

void free_linear(mpe_ctx *ctx, Linear *L) {
    dev_free(ctx, L->w);
    dev_free(ctx, L->b);
    dev_free(ctx, L->w16);
    dev_free(ctx, L->w3);
    *L = Linear();
}

int upload_linear(mpe_ctx *ctx, const float *w, const float *b, int out_dim, int in_dim, Linear *L) {
    if (!w || out_dim <= 0 || in_dim <= 0) return fail(ctx, MPE_ERR_INVALID, "bad linear layer %dx%d", out_dim, in_dim);
    free_linear(ctx, L);                   // a layer set twice does not keep its first copies
    L->in_dim = in_dim;
    L->out_dim = out_dim;
    L->ldw = round_up(in_dim, LD_ALIGN);
    const int rows = weight_rows(out_dim);
    int rc = dev_alloc(ctx, &L->w, (size_t)rows * L->ldw);
    if (rc) return rc;
    rc = dev_alloc(ctx, &L->b, (size_t)rows);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy2D(L->w, (size_t)L->ldw * sizeof(float), w, (size_t)in_dim * sizeof(float),
                            (size_t)in_dim * sizeof(float), out_dim, hipMemcpyHostToDevice));
    if (b) HIPCHK(ctx, hipMemcpy(L->b, b, (size_t)out_dim * sizeof(float), hipMemcpyHostToDevice));
    return MPE_OK;
}

int check_batch(mpe_ctx *ctx, const mpe_batch *b) {
    if (!ctx) return MPE_ERR_INVALID;
    if (!b) return fail(ctx, MPE_ERR_INVALID, "batch is NULL");
    if (b->n_frames < 0 || b->n_heads < 0 || b->n_edge_nodes < 0) return fail(ctx, MPE_ERR_INVALID, "negative batch size");
    if (b->n_frames > ctx->cfg.max_frames || b->n_heads > ctx->cfg.max_heads ||
        b->n_edge_nodes > ctx->cfg.max_edge_nodes)
        return fail(ctx, MPE_ERR_CAPACITY, "batch (%d frames, %d heads, %d edge-nodes) exceeds capacity (%d, %d, %d)",
                    b->n_frames, b->n_heads, b->n_edge_nodes, ctx->cfg.max_frames, ctx->cfg.max_heads,
                    ctx->cfg.max_edge_nodes);
    if (b->n_frames > 0 && (!b->d_frame_head_off || !b->d_frame_en_off || (!b->d_en_pair && (!b->d_slot_cam || !b->d_slot_n))))
        return fail(ctx, MPE_ERR_INVALID, "batch offset tables missing");
    if (b->d_en_pair && ctx->x_m_cap <= 0)
        return fail(ctx, MPE_ERR_UNSUPPORTED, "explicit edge-node lists need max_heads_per_frame <= 1024 and node ids that fit 16 bits "
                    "(max_heads_per_frame = %d)", ctx->cfg.max_heads_per_frame);
    if (b->n_heads > 0 && (!b->d_head_cam || !b->d_joint_mask || !b->d_tri_mask || !b->d_xy || !b->d_vp))
        return fail(ctx, MPE_ERR_INVALID, "batch skeleton arrays missing");
    return MPE_OK;
}

struct GemmProf {
    mpe_ctx *ctx;
    hipStream_t s;
    bool on;
    size_t idx;
    GemmProf(mpe_ctx *c, hipStream_t st, double flop, int dev_m_n, int dev_m_k, int kind = 0) : ctx(c), s(st), on(false), idx(0) {
        if (!c->profiling) return;
        if (c->prof_used == c->prof.size()) {
            ProfileRec r{};
            if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
            c->prof.push_back(r);
        }
        idx = c->prof_used++;
        c->prof[idx].flop = flop;
        c->prof[idx].dev_n = dev_m_n;
        c->prof[idx].dev_k = dev_m_k;
        c->prof[idx].kind = kind;
        on = true;
        (void)hipEventRecord(c->prof[idx].start, s);
    }
    ~GemmProf() {
        if (on) (void)hipEventRecord(ctx->prof[idx].stop, s);
    }
};

int linear(mpe_ctx *ctx, hipStream_t s, const float *A, int lda, const Linear &L, float *C, int ldc, int m,
           const int32_t *d_m, bool leaky, float slope, bool acc64 = false, const int32_t *a_rows = nullptr,
           const int32_t *c_rows = nullptr, double flop_override = -1.0, const AttnCoef *coef = nullptr,
           bool *coef_done = nullptr, bool out_half = false) {
    if (coef_done) *coef_done = false;
    if (m <= 0) return MPE_OK;
    if (lda < L.ldw) return fail(ctx, MPE_ERR_INVALID, "activation stride %d < padded K %d", lda, L.ldw);
    const bool host_m = !d_m || flop_override >= 0.0;
    GemmProf gp(ctx, s, flop_override >= 0.0 ? flop_override : (d_m ? 0.0 : 2.0 * m * (double)L.out_dim * L.in_dim),
                host_m ? 0 : L.out_dim, host_m ? 0 : L.in_dim);
    HIPCHK(ctx, launch_linear(s, A, lda, L.w, L.ldw, L.b, C, ldc, m, d_m, L.out_dim, L.ldw, leaky, slope, acc64, a_rows,
                              c_rows, coef, coef_done, out_half));
    return MPE_OK;
}

// MPE_LATENCY_PATH=0: small batches through the batch path's own small-batch kernels (read per call: tests toggle it)
bool latency_path_on() {
    const char *e = getenv("MPE_LATENCY_PATH");
    return !(e && e[0] == '0');
}

unsigned short f32_to_bf16(float f);
int ensure_bf16_weights(mpe_ctx *ctx, Linear *L);
int ensure_split_weights(mpe_ctx *ctx, hipStream_t s, Linear *L);

// GEMM of a GAT layer: fp32 MFMA (parity) or, in the reduced-precision mode, bf16 MFMA with an
// optional fp16 result (`out_half`: C is the same buffer seen as fp16 rows, ldc in halves)
int gat_linear(mpe_ctx *ctx, hipStream_t s, const float *A, int lda, Linear &L, float *C, int ldc, int m,
               const int32_t *d_m, bool leaky, float slope, bool out_half, const int32_t *a_rows = nullptr,
               const int32_t *c_rows = nullptr, double flop_override = -1.0, const AttnCoef *coef = nullptr,
               bool *coef_done = nullptr) {
    if (coef_done) *coef_done = false;
    static const int mink = getenv("MPE_GAT_ACC64_MINK") ? atoi(getenv("MPE_GAT_ACC64_MINK")) : 512;
    const bool sb_f64 = ctx->gat_acc64 || (mink > 0 && L.in_dim > mink);
    // (fp16 result rows exist in the split TILE kernel only: the fc2 launches at batch sizes the tile kernel takes, without f64
    // sums; every other launch of the fp16-attention mode stays on the fp32 MFMA, whose tile and wave-per-tile kernels all store
    // fp16 rows)
    const bool sb_half_ok = !out_half || (!leaky && !sb_f64 && linear_sb16_uses_tile_kernel(m, L.out_dim, false));
    // (launches with gathered rows -- layer-0 fc1 per camera -- stay on the fp32 MFMA; the grouped layer-0 launch does not come here)
    // In the explicit f64-sum mode (mpe_set_precision GAT 1 on top of the split form) layer 0's fc2 keeps the fp32 MFMA with a
    // flush per 32-deep stage: the split form flushes every second stage, and on the K = 902 sum of the steep layer-0 features
    // that cadence left one 5x4 fixture frame 1.46x the reference's own distance from the f64 network where the mode promises
    // <= 1 (tests/test_gpu_stages.py::test_score_noise_against_the_f64_network; 0.79 with the flush per stage).
    const bool l0_fc2_f64_mode = ctx->gat_acc64 && &L == &ctx->gat[0].fc2;
    if (ctx->gat_split && !ctx->gat_reduced && sb_half_ok && !a_rows && !c_rows && L.w != ctx->l0_w && !l0_fc2_f64_mode) {
        // split-bf16 form (gemm_sb16.hip): fp32-accurate products on the bf16 matrix pipe; f64 sums where the fp32 path has them
        const bool f64 = sb_f64;
        if (m <= 0) return MPE_OK;
        int rc = ensure_split_weights(ctx, s, &L);
        if (rc) return rc;
        if (lda < L.ldw) return fail(ctx, MPE_ERR_INVALID, "activation stride %d < padded K %d", lda, L.ldw);
        const bool host_m = !d_m || flop_override >= 0.0;
        GemmProf gp(ctx, s, flop_override >= 0.0 ? flop_override : (d_m ? 0.0 : 2.0 * m * (double)L.out_dim * L.in_dim),
                    host_m ? 0 : L.out_dim, host_m ? 0 : L.in_dim, 1);
        HIPCHK(ctx, launch_linear_sb16(s, A, lda, L.w3, (size_t)weight_rows(L.out_dim) * L.ldw, L.ldw, L.b, C, ldc, m, d_m, L.out_dim, L.ldw,
                                       leaky, slope, f64, f64 ? nullptr : coef, coef_done, out_half));
        return MPE_OK;
    }
    if (!ctx->gat_reduced) {
        // Long sums (K > 512: fc2 of layer 0, K = 902 / 1082, on head rows only -- no measurable cost) always
        // run with f64 running sums: a single fp32 chain of that length was the largest contribution to the
        // score noise (ARPLAB frames of random shape: 3.2e-5 from the reference with it, 2.3e-5 without, where
        // the reference's own fp32 scores sit 1.8e-5 from the float64 network).  MPE_GAT_ACC64_MINK overrides
        // the threshold (0 = never), mpe_set_precision(ctx, 1, .) extends it to every GAT GEMM.
        const bool acc64 = sb_f64;
        // out_half here = the fp16-attention mode (fp32 MFMA GEMM, result rows stored as fp16)
        return linear(ctx, s, A, lda, L, C, ldc, m, d_m, leaky, slope, acc64, a_rows, c_rows, flop_override, coef, coef_done, out_half);
    }
    if (m <= 0) return MPE_OK;
    int rc = ensure_bf16_weights(ctx, &L);
    if (rc) return rc;
    if (lda < L.ldw) return fail(ctx, MPE_ERR_INVALID, "activation stride %d < padded K %d", lda, L.ldw);
    const bool host_m = !d_m || flop_override >= 0.0;
    GemmProf gp(ctx, s, flop_override >= 0.0 ? flop_override : (d_m ? 0.0 : 2.0 * m * (double)L.out_dim * L.in_dim),
                host_m ? 0 : L.out_dim, host_m ? 0 : L.in_dim, 2);
    HIPCHK(ctx, launch_linear_bf16(s, A, lda, L.w16, L.ldw16, L.b, C, ldc, m, d_m, L.out_dim, L.ldw16, leaky, slope,
                                   L.ldw, out_half, a_rows, c_rows));
    return MPE_OK;
}

void drop_gat_workspace(mpe_ctx *ctx) {
    float **bufs[] = {&ctx->x0, &ctx->h0, &ctx->xdense, &ctx->hdense, &ctx->act[0], &ctx->act[1], &ctx->act[2],
                      &ctx->a12, &ctx->en0_ft2, &ctx->en0_a, &ctx->xc};
    for (float **p : bufs) {
        dev_free(ctx, *p);
        *p = nullptr;
    }
    dev_free(ctx, ctx->cam_count);
    dev_free(ctx, ctx->cam_list);
    ctx->cam_count = ctx->cam_list = nullptr;
    for (int i = 0; i < 2; ++i) {
        dev_free(ctx, ctx->gat_pl[i]);
        ctx->gat_pl[i] = nullptr;
    }
    for (int c = 0; c < MPE_MAX_CAMERAS; ++c) {
        dev_free(ctx, ctx->l0_fc1[c].w16);          // w / b are views into l0_w / l0_b
        ctx->l0_fc1[c] = Linear();
    }
    dev_free(ctx, ctx->l0_w);
    dev_free(ctx, ctx->l0_b);
    ctx->l0_w = ctx->l0_b = nullptr;
    ctx->en0_ready = false;
    ctx->gat_ws_ready = false;
}

int ensure_gat_workspace(mpe_ctx *ctx) {
    if (ctx->gat_ws_ready) return MPE_OK;
    drop_gat_workspace(ctx);               // leftovers of an attempt that failed half way
    for (int l = 0; l < ctx->gat_layers; ++l)
        if (!ctx->gat_ready[l]) return fail(ctx, MPE_ERR_STATE, "GAT layer %d has no weights", l);
    if (ctx->gat_layers <= 0) return fail(ctx, MPE_ERR_STATE, "GAT parameters not set");
    const int V = ctx->cfg.n_cameras, J = ctx->cfg.n_joints;
    const int F = 2 + V * J * 10;
    if (ctx->gat[0].in_dim != F)
        return fail(ctx, MPE_ERR_INVALID, "GAT input width %d != 2 + V*J*10 = %d", ctx->gat[0].in_dim, F);
    ctx->feat_ld = round_up(F, LD_ALIGN);
    int widest = 0;
    for (int l = 0; l < ctx->gat_layers; ++l) {
        const GatLayer &g = ctx->gat[l];
        if (l > 0 && g.in_dim != ctx->gat[l - 1].heads * ctx->gat[l - 1].out_dim)
            return fail(ctx, MPE_ERR_INVALID, "GAT layer %d input width mismatch", l);
        if (g.heads > 16) return fail(ctx, MPE_ERR_INVALID, "at most 16 attention heads supported");
        if (l > 0) widest = widest > g.in_dim ? widest : g.in_dim;
        widest = widest > g.heads * g.out_dim ? widest : g.heads * g.out_dim;
    }
    ctx->act_ld = round_up(widest, LD_ALIGN);
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->x0, (size_t)ctx->cfg.max_heads * ctx->feat_ld))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->h0, (size_t)ctx->cfg.max_heads * ctx->feat_ld))) return rc;
    for (int i = 0; i < 3; ++i)
        if ((rc = dev_alloc(ctx, &ctx->act[i], (size_t)ctx->max_nodes * ctx->act_ld))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->a12, (size_t)ctx->max_nodes * 32))) return rc;
    // layer-0 constants of the edge-nodes: every edge-node row is the one-hot e_1, so
    // fc1(e_1) = W1[:,1] + b1 and everything downstream of it is a model constant.
    const GatLayer &g0 = ctx->gat[0];
    std::vector<float> w1((size_t)g0.in_dim * g0.fc1.ldw), b1(g0.in_dim), c1(ctx->feat_ld, 0.f);
    HIPCHK(ctx, hipMemcpy(w1.data(), g0.fc1.w, w1.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(b1.data(), g0.fc1.b, b1.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int n = 0; n < g0.in_dim; ++n) {
        const float v = w1[(size_t)n * g0.fc1.ldw + 1] + b1[n];
        c1[n] = v > 0.f ? v : v * ctx->gat_alpha;
    }
    float *d_c1 = nullptr;
    if ((rc = dev_alloc(ctx, &d_c1, (size_t)ctx->feat_ld))) return rc;
    struct Tmp {                           // the staging row is only needed inside this function
        mpe_ctx *c;
        float *p;
        ~Tmp() { dev_free(c, p); }
    } tmp{ctx, d_c1};
    if ((rc = dev_alloc(ctx, &ctx->en0_ft2, (size_t)ctx->act_ld))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->en0_a, 32))) return rc;
    HIPCHK(ctx, hipMemcpy(d_c1, c1.data(), c1.size() * sizeof(float), hipMemcpyHostToDevice));
    const bool was = ctx->profiling;
    ctx->profiling = false;
    rc = linear(ctx, nullptr, d_c1, ctx->feat_ld, g0.fc2, ctx->en0_ft2, ctx->act_ld, 1, nullptr, false, 0.f);
    ctx->profiling = was;
    if (rc) return rc;
    HIPCHK(ctx, launch_attn_coef(nullptr, ctx->en0_ft2, ctx->act_ld, 1, g0.heads, g0.out_dim, g0.attn_l, g0.attn_r,
                                 ctx->en0_a));
    HIPCHK(ctx, hipDeviceSynchronize());
    ctx->en0_ready = true;
    // layer-0 fc1 grouped by camera: a head row is e_0 plus the J*10 block of its own camera,
    // so fc1(row) = W1[:, block_c] . feat + (b1 + W1[:,0]); K shrinks from F to J*10.
    {
        const int blk = J * 10;
        ctx->l0_ld = round_up(blk, LD_ALIGN);
        if ((rc = dev_alloc(ctx, &ctx->xc, (size_t)ctx->cfg.max_heads * ctx->l0_ld))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->cam_count, (size_t)V))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->cam_list, (size_t)V * ctx->cfg.max_heads))) return rc;
        // the V per-camera matrices live in ONE allocation at a constant stride so that a single
        // grouped launch can address them (launch_linear_grouped); l0_fc1[c] are views into it
        const int rows = weight_rows(g0.in_dim);
        const size_t per = (size_t)rows * ctx->l0_ld;
        if ((rc = dev_alloc(ctx, &ctx->l0_w, per * V))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->l0_b, (size_t)rows))) return rc;
        std::vector<float> wc((size_t)g0.in_dim * blk), bc(g0.in_dim);
        for (int n = 0; n < g0.in_dim; ++n) bc[n] = b1[n] + w1[(size_t)n * g0.fc1.ldw + 0];
        HIPCHK(ctx, hipMemcpy(ctx->l0_b, bc.data(), bc.size() * sizeof(float), hipMemcpyHostToDevice));
        for (int c = 0; c < V; ++c) {
            for (int n = 0; n < g0.in_dim; ++n)
                memcpy(&wc[(size_t)n * blk], &w1[(size_t)n * g0.fc1.ldw + 2 + c * blk], blk * sizeof(float));
            Linear &L = ctx->l0_fc1[c];
            L = Linear();
            L.w = ctx->l0_w + per * c;
            L.b = ctx->l0_b;
            L.in_dim = blk;
            L.out_dim = g0.in_dim;
            L.ldw = ctx->l0_ld;
            HIPCHK(ctx, hipMemcpy2D(L.w, (size_t)L.ldw * sizeof(float), wc.data(), (size_t)blk * sizeof(float),
                                    (size_t)blk * sizeof(float), g0.in_dim, hipMemcpyHostToDevice));
        }
        // K shrinks from F to J*10 (exact: the skipped products are zeros), one launch for all cameras
        ctx->l0_grouped = true;
        if (const char *e = getenv("MPE_L0_GROUPED")) ctx->l0_grouped = atoi(e) != 0;
    }
    ctx->gat_ws_ready = true;              // last step: a failure above leaves the flag false and is retried
    return MPE_OK;
}

unsigned short f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);          // round to nearest even
    return (unsigned short)(u >> 16);
}

int ensure_bf16_weights(mpe_ctx *ctx, Linear *L) {
    if (L->w16) return MPE_OK;
    const int rows = weight_rows(L->out_dim);
    L->ldw16 = round_up(L->in_dim, 128);
    std::vector<float> w((size_t)rows * L->ldw);
    HIPCHK(ctx, hipMemcpy(w.data(), L->w, w.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<unsigned short> wb((size_t)rows * L->ldw16, 0);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < L->in_dim; ++k) wb[(size_t)r * L->ldw16 + k] = f32_to_bf16(w[(size_t)r * L->ldw + k]);
    int rc = dev_alloc(ctx, &L->w16, wb.size(), false);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy(L->w16, wb.data(), wb.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    return MPE_OK;
}

// the three bf16 planes of a weight matrix (split on the device from the padded fp32 copy, stream-ordered)
int ensure_split_weights(mpe_ctx *ctx, hipStream_t s, Linear *L) {
    if (L->w3) return MPE_OK;
    const size_t count = (size_t)weight_rows(L->out_dim) * L->ldw;
    int rc = dev_alloc(ctx, &L->w3, 3 * count, false);
    if (rc) return rc;
    HIPCHK(ctx, launch_split_planes(s, L->w, count, L->w3));
    return MPE_OK;
}

int ensure_mlp_workspace(mpe_ctx *ctx) {
    if (ctx->mlp_ws_ready) return MPE_OK;
    float **bufs[] = {&ctx->mlp_rows, &ctx->mlp_act[0], &ctx->mlp_act[1]};
    for (float **p : bufs) {
        dev_free(ctx, *p);
        *p = nullptr;
    }
    if (ctx->mlp_layers <= 0) return fail(ctx, MPE_ERR_STATE, "MLP parameters not set");
    for (int l = 0; l < ctx->mlp_layers; ++l)
        if (!ctx->mlp_ready[l]) return fail(ctx, MPE_ERR_STATE, "MLP layer %d has no weights", l);
    const int V = ctx->cfg.n_cameras, J = ctx->cfg.n_joints, npj = ctx->cfg.numbers_per_joint;
    if (ctx->mlp[0].in_dim != V * J * npj)
        return fail(ctx, MPE_ERR_INVALID, "MLP input width %d != V*J*numbers_per_joint = %d", ctx->mlp[0].in_dim,
                    V * J * npj);
    int widest = 0;
    for (int l = 0; l < ctx->mlp_layers; ++l) {
        if (l > 0 && ctx->mlp[l].in_dim != ctx->mlp[l - 1].out_dim)
            return fail(ctx, MPE_ERR_INVALID, "MLP layer %d input width mismatch", l);
        widest = widest > ctx->mlp[l].out_dim ? widest : ctx->mlp[l].out_dim;
    }
    ctx->mlp_ld_in = round_up(ctx->mlp[0].in_dim, 128);       // 128: K stage of the bf16 variant
    ctx->mlp_ld_hidden = round_up(widest, 128);
    const size_t rows = (size_t)ctx->cfg.max_frames * ctx->cfg.max_persons_per_frame;
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->mlp_rows, rows * ctx->mlp_ld_in))) return rc;
    for (int i = 0; i < 2; ++i)
        if ((rc = dev_alloc(ctx, &ctx->mlp_act[i], rows * ctx->mlp_ld_hidden))) return rc;
    ctx->mlp_ws_ready = true;
    return MPE_OK;
}

// Where the input rows of a GAT layer come from
enum GatInput {
    GAT_IN_IMPLICIT,   // layer 0, production: head rows featurised on the device, edge-node rows constant
    GAT_IN_DENSE0,     // layer 0, caller-provided dense rows already copied into ctx->xdense
    GAT_IN_ACT         // layer > 0: rows in ctx->act[0]
};

// fc1 -> LeakyReLU(alpha) -> fc2 of layer l (gat2.py:53-55): leaves ft2 in a->ft2 / n_rows_ft2
int gat_layer_linear(mpe_ctx *ctx, hipStream_t s, const mpe_batch *b, int l, GatInput in, AggArgs *a, int *n_rows_ft2) {
    GatLayer &g = ctx->gat[l];
    const int V = ctx->cfg.n_cameras, J = ctx->cfg.n_joints;
    const int n_nodes = b->n_heads + b->n_edge_nodes;
    const bool red = ctx->gat_reduced;
    const bool half_rows = red || ctx->gat_attn_fp16;        // ft2 as fp16 rows for the attention stage
    const int ld_ft = half_rows ? 2 * ctx->act_ld : ctx->act_ld;   // fp16 rows keep the byte stride of the fp32 rows
    int rc;
    // fc2 also emits the attention coefficients a1|a2 from its epilogue when the layer's shape allows
    const AttnCoef coef{g.attn_l, g.attn_r, ctx->a12, g.heads, g.out_dim};
    const bool no_epi = getenv("MPE_NO_COEF_EPILOGUE") != nullptr;           // read per call: tests toggle it
    const AttnCoef *cp = no_epi ? nullptr : &coef;
    bool done = false;
    if (in == GAT_IN_DENSE0) {
        if ((rc = gat_linear(ctx, s, ctx->xdense, ctx->feat_ld, g.fc1, ctx->hdense, ctx->feat_ld, n_nodes, nullptr,
                             true, ctx->gat_alpha, false)))
            return rc;
        if ((rc = gat_linear(ctx, s, ctx->hdense, ctx->feat_ld, g.fc2, ctx->act[2], ld_ft, n_nodes, nullptr, false,
                             0.f, half_rows, nullptr, nullptr, -1.0, cp, &done)))
            return rc;
        a->ft2 = ctx->act[2];
        *n_rows_ft2 = n_nodes;
    } else if (in == GAT_IN_IMPLICIT) {
        // heads only: edge-node rows are the layer-0 constants
        if (ctx->l0_grouped && !red && linear_uses_tile_kernel(b->n_heads, g.in_dim)) {
            // ONE grouped launch: camera c's heads (device-side count, rows gathered from the compact
            // features and scattered back to head order) x camera c's J*10-column block of fc1
            GemmProf gp(ctx, s, 2.0 * b->n_heads * (double)g.in_dim * (J * 10), 0, 0);
            HIPCHK(ctx, launch_linear_grouped(s, ctx->xc, ctx->l0_ld, ctx->l0_w, ctx->l0_ld,
                                              (long)weight_rows(g.in_dim) * ctx->l0_ld, ctx->l0_b, ctx->h0, ctx->feat_ld,
                                              b->n_heads, ctx->cam_count, V, ctx->cam_list, ctx->cfg.max_heads, g.in_dim,
                                              ctx->l0_ld, true, ctx->gat_alpha, ctx->gat_acc64));
        } else if (ctx->l0_grouped) {
            // small batches / reduced precision: one launch per camera over that camera's heads
            const double flop_each = 2.0 * b->n_heads * (double)g.in_dim * (J * 10) / V;
            for (int c = 0; c < V; ++c)
                if ((rc = gat_linear(ctx, s, ctx->xc, ctx->l0_ld, ctx->l0_fc1[c], ctx->h0, ctx->feat_ld, b->n_heads,
                                     ctx->cam_count + c, true, ctx->gat_alpha, false,
                                     ctx->cam_list + (size_t)c * ctx->cfg.max_heads,
                                     ctx->cam_list + (size_t)c * ctx->cfg.max_heads, flop_each)))
                    return rc;
        } else if ((rc = gat_linear(ctx, s, ctx->x0, ctx->feat_ld, g.fc1, ctx->h0, ctx->feat_ld, b->n_heads, nullptr,
                                    true, ctx->gat_alpha, false)))
            return rc;
        if ((rc = gat_linear(ctx, s, ctx->h0, ctx->feat_ld, g.fc2, ctx->act[1], ld_ft, b->n_heads, nullptr, false,
                             0.f, half_rows, nullptr, nullptr, -1.0, cp, &done)))
            return rc;
        a->ft2 = ctx->act[1];
        *n_rows_ft2 = b->n_heads;
        a->en_const_ft2 = ctx->en0_ft2;
        a->en_const_a = ctx->en0_a;
    } else {
        if ((rc = gat_linear(ctx, s, ctx->act[0], ctx->act_ld, g.fc1, ctx->act[1], ctx->act_ld, n_nodes, nullptr,
                             true, ctx->gat_alpha, false)))
            return rc;
        if ((rc = gat_linear(ctx, s, ctx->act[1], ctx->act_ld, g.fc2, ctx->act[2], ld_ft, n_nodes, nullptr, false,
                             0.f, half_rows, nullptr, nullptr, -1.0, cp, &done)))
            return rc;
        a->ft2 = ctx->act[2];
        *n_rows_ft2 = n_nodes;
    }
    if (!done && cp && g.out_dim == 40 && !half_rows && ctx->gat_split && !red) {
        // the split-bf16 form with f64 sums (layer-0 fc2: K = 902) has no coefficient epilogue: a1 | a2 by the coefficient kernel
        // (canonical order = the epilogue's, gat.hip:coef40), so that the attention stage still finds them ready
        HIPCHK(ctx, launch_attn_coef(s, a->ft2, ld_ft, *n_rows_ft2, g.heads, g.out_dim, g.attn_l, g.attn_r, ctx->a12, 0));
        done = true;
    }
    a->a12_ready = done ? 1 : 0;
    return MPE_OK;
}

AggArgs gat_agg_args(const mpe_ctx *ctx, int l) {
    const GatLayer &g = ctx->gat[l];
    AggArgs a{};
    a.heads = g.heads;
    a.out_dim = g.out_dim;
    a.alpha = ctx->gat_alpha;
    a.out_slope = ctx->gat_hidden_slope;
    const bool half_rows = ctx->gat_reduced || ctx->gat_attn_fp16;
    a.ft_half = half_rows ? 1 : 0;
    a.ld = half_rows ? 2 * ctx->act_ld : ctx->act_ld;
    a.a12 = ctx->a12;
    return a;
}

int gat_attention(mpe_ctx *ctx, hipStream_t s, const mpe_batch *b, int l, const AggArgs &a, int n_rows_ft2) {
    const GatLayer &g = ctx->gat[l];
    HIPCHK(ctx, launch_gat_attention(s, *b, ctx->cfg.n_cameras, ctx->cfg.max_heads_per_frame, ctx->node_off,
                                     ctx->head_frame, ctx->en_frame, ctx->en_pair, g.attn_l, g.attn_r, ctx->a12, a,
                                     n_rows_ft2, ctx->head_src, explicit_deg_cap(ctx->cfg.max_heads_per_frame), ctx->x_m_cap));
    return MPE_OK;
}

int gat_topology(mpe_ctx *ctx, hipStream_t s, const mpe_batch *b) {
    if (b->d_en_pair && (size_t)b->n_frames * ctx->cfg.max_heads_per_frame * explicit_deg_cap(ctx->cfg.max_heads_per_frame) > ctx->head_src_cap)
        return fail(ctx, MPE_ERR_CAPACITY, "in-edge table too small for %d frames with explicit edge-node lists", b->n_frames);
    HIPCHK(ctx, launch_topology(s, *b, ctx->cfg.n_cameras, ctx->node_off, ctx->head_frame, ctx->en_frame, ctx->en_pair,
                                ctx->cfg.max_heads_per_frame, ctx->d_status, ctx->head_src,
                                explicit_deg_cap(ctx->cfg.max_heads_per_frame), ctx->x_m_cap));
    return MPE_OK;
}

// ---- small batches (a few frames; the reference's call pattern is one frame per call, test/metrics_from_model.py:120-300) ----
// (sixteen: measured against the batch path with one context on one stream -- 9 frames 336 against 429 us per call, 12: 389 / 456,
// 16: 427 / 449; 24: 577 / 531, 32: 657 / 558: the plane-fed GEMMs run one workgroup per CU and take a round per 256 of them)
constexpr int LAT_MAX_FRAMES = 16;
constexpr int LAT_MAX_NODES = 16384;

// Can this batch take the latency launches?  Default precision modes, implicit topology, rows featurised on the device, a network
// whose layers have instantiations (the deployed one: train_skeleton_matching.py:40-57) -- anything else keeps the batch path.
bool lat_gat_ok(mpe_ctx *ctx, const mpe_batch *b, bool dense_in) {
    if (!latency_path_on() || dense_in || b->d_en_pair || b->n_frames > LAT_MAX_FRAMES) return false;
    if (b->n_heads + b->n_edge_nodes > LAT_MAX_NODES || b->n_heads <= 0) return false;
    if (!ctx->gat_split || ctx->gat_acc64 || ctx->gat_reduced || ctx->gat_attn_fp16 || !ctx->l0_grouped) return false;
    // (cross-check switches of the batch path: a run that asks for the tile kernels at every batch size, for the coefficient kernel or
    // for another f64 threshold means the batch path's kernels)
    if (getenv("MPE_NO_COEF_EPILOGUE") || getenv("MPE_GAT_ACC64_MINK") || getenv("MPE_SKINNY_WAVES") || getenv("MPE_GEMM_NARROW")) return false;
    if (!lat_l0a_available(ctx->cfg.n_joints, ctx->l0_ld)) return false;
    const GatLayer &g0 = ctx->gat[0];
    if (g0.out_dim != 40 || g0.in_dim <= 512) return false;      // layer 0: fc2 with f64 sums + the coefficient kernel, as the batch path
    for (int l = 1; l < ctx->gat_layers; ++l) {
        const GatLayer &g = ctx->gat[l];
        if (!lat_gemm_available(g.fc1.ldw, g.in_dim, false, 0) || !lat_gemm_available(g.fc2.ldw, g.heads * g.out_dim, true, g.out_dim)) return false;
        if (g.in_dim > 512) return false;                        // (such a layer would run with f64 sums)
    }
    const size_t shm = (size_t)2 * ((size_t)16 * (ctx->cfg.max_heads_per_frame + 1) + (ctx->cfg.max_heads_per_frame + 1) + 1 +
                                    (size_t)(ctx->cfg.n_cameras + 1) * (ctx->cfg.n_cameras + 1)) * sizeof(float);
    return shm <= 40 * 1024;
}

int ensure_lat_workspace(mpe_ctx *ctx) {
    if (ctx->gat_pl[0]) return MPE_OK;
    ctx->lat_rows = ctx->max_nodes < LAT_MAX_NODES ? ctx->max_nodes : LAT_MAX_NODES;
    ctx->gat_pl_plane = (size_t)ctx->lat_rows * ctx->act_ld;
    int rc;
    for (int i = 0; i < 2; ++i)
        if ((rc = dev_alloc(ctx, &ctx->gat_pl[i], 3 * ctx->gat_pl_plane))) return rc;       // zeroed: pad columns meet zero weights
    return MPE_OK;
}

// The matching network for a small batch in 3 + 3 (L - 1) launches instead of ~26: front + layer-0 fc1 (k_lat_l0a), layer-0 fc2 as the
// batch path has it, then per layer attention (both halves, one launch) -> fc1 -> fc2 with the activations
// travelling as bf16 planes between them (k_lat_gemm).  Same arithmetic, same summation orders, same bits as the batch kernels.
// *tail (optional): the attention stage of the LAST layer is left to the caller's tail launch (k_lat_tail: scores + clustering in one
// launch) when the layer's coefficients come from the fc2 epilogue; *tail says whether it was, and describes the stage.
struct LatTail {
    bool deferred = false;
    const float *ft2 = nullptr;
    int ld = 0;
    float alpha = 0.f, out_slope = 0.f;
    int out_mode = 0;
};

int run_gat_lat(mpe_ctx *ctx, hipStream_t s, const mpe_batch *b, float *d_scores_en, float *d_scores_heads, LatTail *tail = nullptr) {
    if (tail) tail->deferred = false;
    int rc = ensure_lat_workspace(ctx);
    if (rc) return rc;
    const int V = ctx->cfg.n_cameras, J = ctx->cfg.n_joints, hmax = ctx->cfg.max_heads_per_frame;
    const int n_nodes = b->n_heads + b->n_edge_nodes;
    if (n_nodes > ctx->lat_rows) return fail(ctx, MPE_ERR_CAPACITY, "latency path: %d nodes exceed %d", n_nodes, ctx->lat_rows);
    GatLayer &g0 = ctx->gat[0];
    uint16_t *head_src = getenv("MPE_NO_HEAD_SRC_TABLE") ? nullptr : ctx->head_src;          // read per call: tests toggle it
    {
        GemmProf gp(ctx, s, 2.0 * b->n_heads * (double)g0.in_dim * (J * 10), 0, 0);
        HIPCHK(ctx, launch_lat_l0a(s, ctx->d_cfg, *b, V, J, ctx->node_off, ctx->head_frame, ctx->en_frame, ctx->en_pair, ctx->head_src, hmax,
                                   ctx->d_status, ctx->l0_w, (long)weight_rows(g0.in_dim) * ctx->l0_ld, ctx->l0_ld, ctx->l0_b, g0.in_dim, ctx->h0,
                                   ctx->feat_ld, ctx->gat_alpha));
    }
    bool done = false;
    if ((rc = gat_linear(ctx, s, ctx->h0, ctx->feat_ld, g0.fc2, ctx->act[1], ctx->act_ld, b->n_heads, nullptr, false, 0.f, false, nullptr, nullptr,
                         -1.0, nullptr, &done)))
        return rc;
    const int L = ctx->gat_layers;
    for (int l = 0; l < L; ++l) {
        GatLayer &g = ctx->gat[l];
        AggArgs a = gat_agg_args(ctx, l);
        a.attn_l = g.attn_l;
        a.attn_r = g.attn_r;
        bool coef = false;                                   // a1 | a2 already in a12?  (otherwise the attention kernel computes them from the rows)
        if (l == 0) {
            a.ft2 = ctx->act[1];
            a.en_const_ft2 = ctx->en0_ft2;
            a.en_const_a = ctx->en0_a;
        } else {
            if ((rc = ensure_split_weights(ctx, s, &g.fc1)) || (rc = ensure_split_weights(ctx, s, &g.fc2))) return rc;
            const int hd = g.heads * g.out_dim;
            if (lat_gemm_fusable(g.fc1.ldw, g.in_dim, g.fc2.ldw, hd, g.out_dim)) {
                // the last layer: fc1 and the one-row fc2 in one launch (k_lat_gemm<.., FUSE2>)
                GemmProf gp(ctx, s, 2.0 * n_nodes * (double)g.in_dim * (g.in_dim + hd), 0, 0, 1);
                HIPCHK(ctx, launch_lat_gemm_fused(s, ctx->gat_pl[0], ctx->act_ld, ctx->gat_pl_plane, g.fc1.w3, (size_t)weight_rows(g.fc1.out_dim) * g.fc1.ldw,
                                                  g.fc1.ldw, g.fc1.b, n_nodes, g.in_dim, ctx->gat_alpha, g.fc2.w3,
                                                  (size_t)weight_rows(g.fc2.out_dim) * g.fc2.ldw, g.fc2.ldw, g.fc2.b, ctx->act[2], ctx->act_ld, g.attn_l, g.attn_r,
                                                  ctx->a12));
                coef = true;
            } else {
            {
                GemmProf gp(ctx, s, 2.0 * n_nodes * (double)g.in_dim * g.in_dim, 0, 0, 1);
                HIPCHK(ctx, launch_lat_gemm(s, ctx->gat_pl[0], ctx->act_ld, ctx->gat_pl_plane, g.fc1.w3, (size_t)weight_rows(g.fc1.out_dim) * g.fc1.ldw,
                                            g.fc1.ldw, g.fc1.b, nullptr, 0, ctx->gat_pl[1], ctx->act_ld, ctx->gat_pl_plane, n_nodes, g.in_dim, g.fc1.ldw,
                                            false, ctx->gat_alpha, nullptr, nullptr, nullptr, 0));
            }
            {
                GemmProf gp(ctx, s, 2.0 * n_nodes * (double)hd * g.in_dim, 0, 0, 1);
                HIPCHK(ctx, launch_lat_gemm(s, ctx->gat_pl[1], ctx->act_ld, ctx->gat_pl_plane, g.fc2.w3, (size_t)weight_rows(g.fc2.out_dim) * g.fc2.ldw,
                                            g.fc2.ldw, g.fc2.b, ctx->act[2], ctx->act_ld, nullptr, 0, 0, n_nodes, hd, g.fc2.ldw, true, 0.f, g.attn_l,
                                            g.attn_r, ctx->a12, g.out_dim, &coef));
            }
            }
            a.ft2 = ctx->act[2];
        }
        a.a12_ready = coef ? 1 : 0;
        if (l == L - 1) {
            a.out_mode = ctx->gat_out_mode;
            a.score_mode = 1;
            a.out = d_scores_en;
            a.out_heads = d_scores_heads;
            a.ld_out = 1;
            if (tail && coef && !d_scores_heads && g.heads == 1 && g.out_dim == 1) {
                tail->deferred = true;
                tail->ft2 = a.ft2;
                tail->ld = a.ld;
                tail->alpha = a.alpha;
                tail->out_slope = a.out_slope;
                tail->out_mode = a.out_mode;
                return MPE_OK;
            }
        } else {
            a.out_mode = 0;
            a.out = nullptr;
            a.out_pl = ctx->gat_pl[0];
            a.out_pl_plane = ctx->gat_pl_plane;
            a.ld_out = ctx->act_ld;
        }
        HIPCHK(ctx, launch_lat_attention(s, *b, V, hmax, ctx->node_off, ctx->head_frame, ctx->en_frame, ctx->en_pair, a, head_src));
    }
    return MPE_OK;
}

int run_gat(mpe_ctx *ctx, hipStream_t s, const mpe_batch *b, float *d_scores_en, float *d_scores_heads,
            const float *d_feats = nullptr, int ld_feats = 0) {
    int rc = ensure_gat_workspace(ctx);
    if (rc) return rc;
    if (lat_gat_ok(ctx, b, d_feats != nullptr)) return run_gat_lat(ctx, s, b, d_scores_en, d_scores_heads);
    const int V = ctx->cfg.n_cameras, J = ctx->cfg.n_joints;
    const int n_nodes = b->n_heads + b->n_edge_nodes;
    if ((rc = gat_topology(ctx, s, b))) return rc;
    const bool dense_in = d_feats != nullptr;   // caller-provided N x F rows (GAT2.forward(inputs, g))
    if (dense_in) {
        if (ld_feats < ctx->gat[0].in_dim) return fail(ctx, MPE_ERR_INVALID, "feature stride too small");
        if (!ctx->xdense) {
            int rc2 = dev_alloc(ctx, &ctx->xdense, (size_t)ctx->max_nodes * ctx->feat_ld);
            if (rc2) return rc2;
            if ((rc2 = dev_alloc(ctx, &ctx->hdense, (size_t)ctx->max_nodes * ctx->feat_ld))) return rc2;
        }
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->xdense, (size_t)ctx->feat_ld * sizeof(float), d_feats,
                                     (size_t)ld_feats * sizeof(float), (size_t)ctx->gat[0].in_dim * sizeof(float),
                                     n_nodes, hipMemcpyDeviceToDevice, s));
    } else if (ctx->l0_grouped) {
        HIPCHK(ctx, launch_head_features(s, ctx->d_cfg, *b, J, ctx->xc, ctx->l0_ld, 0, 0, false, ctx->cam_count, V));
        HIPCHK(ctx, launch_group_heads(s, b->n_heads, V, b->d_head_cam, ctx->cam_count, ctx->cam_list,
                                       ctx->cfg.max_heads, true));
    } else {
        HIPCHK(ctx, launch_head_features(s, ctx->d_cfg, *b, J, ctx->x0, ctx->feat_ld, 0, 0, true));
    }
    const int L = ctx->gat_layers;
    for (int l = 0; l < L; ++l) {
        const bool last = l == L - 1;
        AggArgs a = gat_agg_args(ctx, l);
        int n_rows_ft2 = 0;
        if ((rc = gat_layer_linear(ctx, s, b, l, l > 0 ? GAT_IN_ACT : dense_in ? GAT_IN_DENSE0 : GAT_IN_IMPLICIT, &a,
                                   &n_rows_ft2)))
            return rc;
        if (last) {
            a.out_mode = ctx->gat_out_mode;
            a.score_mode = 1;
            a.out = d_scores_en;
            a.out_heads = d_scores_heads;
            a.ld_out = 1;
        } else {
            a.out_mode = 0;
            a.out = ctx->act[0];
            a.ld_out = ctx->act_ld;
        }
        if ((rc = gat_attention(ctx, s, b, l, a, n_rows_ft2))) return rc;
    }
    return MPE_OK;
}

// stage-level entry points: caller rows [n_nodes][ld] <-> workspace rows
int copy_rows_in(mpe_ctx *ctx, hipStream_t s, float *dst, int ld_dst, const float *src, int ld_src, int cols, int rows) {
    if (rows <= 0) return MPE_OK;
    // the GEMMs read whole padded rows: columns beyond `cols` must be finite (they meet zero weights)
    HIPCHK(ctx, hipMemsetAsync(dst, 0, (size_t)rows * ld_dst * sizeof(float), s));
    HIPCHK(ctx, hipMemcpy2DAsync(dst, (size_t)ld_dst * sizeof(float), src, (size_t)ld_src * sizeof(float),
                                 (size_t)cols * sizeof(float), rows, hipMemcpyDeviceToDevice, s));
    return MPE_OK;
}

}  // namespace

extern "C" {

const char *mpe_version(void) { return "mpe-hip 0.1 (gfx950)"; }

const char *mpe_last_error(const mpe_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mpe_create(const mpe_config *cfg, mpe_ctx **out) {
    if (!cfg || !out) return MPE_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_cameras < 1 || cfg->n_cameras > MPE_MAX_CAMERAS || cfg->n_joints < 1 ||
        cfg->n_joints > MPE_MAX_JOINTS || cfg->max_frames < 1 || cfg->max_heads < 1 || cfg->max_edge_nodes < 1 ||
        cfg->max_heads_per_frame < 2 || cfg->max_heads_per_frame > 32767 || cfg->max_persons_per_frame < 1 ||
        !cfg->Kinv || !cfg->K || !cfg->T_i || !cfg->P || !cfg->dist)
        return MPE_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MPE_ERR_HIP;
    mpe_ctx *ctx = new (std::nothrow) mpe_ctx();
    if (!ctx) return MPE_ERR_NOMEM;
    ctx->cfg = *cfg;
    if (hipGetDevice(&ctx->device) != hipSuccess) {
        delete ctx;
        return MPE_ERR_HIP;
    }
    DevCfg &h = ctx->hcfg;
    memset(&h, 0, sizeof h);
    h.V = cfg->n_cameras;
    h.J = cfg->n_joints;
    h.W = cfg->image_width;
    h.H = cfg->image_height;
    h.npj = cfg->numbers_per_joint;
    h.min_views = cfg->min_views;
    h.median_axis = cfg->median_axis;
    h.used_joint_mask = cfg->used_joint_mask;
    h.threshold = cfg->threshold;
    h.median_window = cfg->median_window;
    for (int c = 0; c < h.V; ++c) {
        memcpy(h.Kinv[c], cfg->Kinv + 9 * c, 9 * sizeof(float));
        memcpy(h.K[c], cfg->K + 9 * c, 9 * sizeof(float));
        memcpy(h.T_i[c], cfg->T_i + 16 * c, 16 * sizeof(float));
        memcpy(h.P[c], cfg->P + 12 * c, 12 * sizeof(double));
        memcpy(h.dist[c], cfg->dist + 5 * c, 5 * sizeof(double));
    }
    ctx->cfg.Kinv = ctx->cfg.K = ctx->cfg.T_i = nullptr;
    ctx->cfg.P = ctx->cfg.dist = nullptr;
    ctx->max_nodes = cfg->max_heads + cfg->max_edge_nodes;
    int rc = MPE_OK;
    do {
        if ((rc = dev_alloc(ctx, &ctx->d_cfg, 1))) break;
        if ((rc = dev_alloc(ctx, &ctx->d_status, 1))) break;
        if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_status), sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { rc = MPE_ERR_NOMEM; break; }
        if (hipMemcpy(ctx->d_cfg, &h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) { rc = MPE_ERR_HIP; break; }
        if ((rc = dev_alloc(ctx, &ctx->head_frame, (size_t)cfg->max_heads))) break;
        if ((rc = dev_alloc(ctx, &ctx->en_frame, (size_t)cfg->max_edge_nodes))) break;
        if ((rc = dev_alloc(ctx, &ctx->en_pair, (size_t)cfg->max_edge_nodes * 2))) break;
        if ((rc = dev_alloc(ctx, &ctx->node_off, (size_t)cfg->max_frames + 1))) break;
        ctx->cl_keys_per_frame = cluster_keys_per_frame(cfg->max_heads_per_frame);
        {
            // in-edge source table of the heads: implicit topology [hmax][hmax + 1] per frame; explicit edge-node lists
            // (mpe_batch::d_en_pair) [hmax][2 hmax], available while a frame's node ids fit the table's 16 bits.  A frame
            // of an explicit list may hold as many edge-nodes as the clustering scratch has keys.
            const size_t hm = (size_t)cfg->max_heads_per_frame;
            const bool x_ok = hm <= 1024 && hm + ctx->cl_keys_per_frame <= 65535;
            ctx->x_m_cap = x_ok ? (int)ctx->cl_keys_per_frame : 0;
            size_t per = head_src_entries(cfg->max_heads_per_frame, cfg->n_cameras);
            if (x_ok && hm * explicit_deg_cap(cfg->max_heads_per_frame) > per) per = hm * explicit_deg_cap(cfg->max_heads_per_frame);
            ctx->head_src_cap = per * cfg->max_frames;
            if (per && (rc = dev_alloc(ctx, &ctx->head_src, ctx->head_src_cap, false))) break;
        }
        ctx->cl_scratch_per_frame = cluster_scratch_per_frame(cfg->max_heads_per_frame);
        if ((rc = dev_alloc(ctx, &ctx->cl_keys, ctx->cl_keys_per_frame * cfg->max_frames, false))) break;
        if ((rc = dev_alloc(ctx, &ctx->cl_scratch, ctx->cl_scratch_per_frame * cfg->max_frames, false))) break;
        if ((rc = dev_alloc(ctx, &ctx->mlp_count, 1))) break;
        if ((rc = dev_alloc(ctx, &ctx->scores_tmp, (size_t)cfg->max_edge_nodes))) break;
        if ((rc = dev_alloc(ctx, &ctx->person_off, (size_t)cfg->max_frames + 1))) break;
        if ((rc = dev_alloc(ctx, &ctx->valid_tmp, (size_t)cfg->max_frames * cfg->max_persons_per_frame))) break;
    } while (0);
    if (rc) {
        mpe_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return MPE_OK;
}

void mpe_destroy(mpe_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard dg(ctx);
    (void)hipDeviceSynchronize();
    for (void *p : ctx->owned) (void)hipFree(p);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    for (auto &r : ctx->prof) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    if (ctx->tot_start) (void)hipEventDestroy(ctx->tot_start);
    if (ctx->tot_stop) (void)hipEventDestroy(ctx->tot_stop);
    delete ctx;
}

int mpe_set_gat_params(mpe_ctx *ctx, int32_t n_layers, float alpha, float hidden_slope) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    if (n_layers < 2 || n_layers > MPE_MAX_GAT_LAYERS) return fail(ctx, MPE_ERR_INVALID, "bad GAT layer count %d", n_layers);
    if (ctx->gat_ws_ready) return fail(ctx, MPE_ERR_STATE, "GAT weights are frozen after the first batch");
    ctx->gat_layers = n_layers;
    ctx->gat_alpha = alpha;
    ctx->gat_hidden_slope = hidden_slope;
    return MPE_OK;
}

int mpe_set_gat_layer(mpe_ctx *ctx, int32_t layer, int32_t in_dim, int32_t heads, int32_t out_dim, const float *fc1_w,
                      const float *fc1_b, const float *fc2_w, const float *fc2_b, const float *attn_l,
                      const float *attn_r) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    if (layer < 0 || layer >= ctx->gat_layers) return fail(ctx, MPE_ERR_INVALID, "GAT layer index %d out of range", layer);
    if (ctx->gat_ws_ready) return fail(ctx, MPE_ERR_STATE, "GAT weights are frozen after the first batch");
    if (!fc1_w || !fc1_b || !fc2_w || !fc2_b || !attn_l || !attn_r || heads < 1 || out_dim < 1)
        return fail(ctx, MPE_ERR_INVALID, "GAT layer %d: missing tensor", layer);
    GatLayer &g = ctx->gat[layer];
    dev_free(ctx, g.attn_l);               // a layer set twice does not keep its first copies
    dev_free(ctx, g.attn_r);
    g.attn_l = g.attn_r = nullptr;
    ctx->gat_ready[layer] = false;
    g.in_dim = in_dim;
    g.heads = heads;
    g.out_dim = out_dim;
    int rc;
    if ((rc = upload_linear(ctx, fc1_w, fc1_b, in_dim, in_dim, &g.fc1))) return rc;
    if ((rc = upload_linear(ctx, fc2_w, fc2_b, heads * out_dim, in_dim, &g.fc2))) return rc;
    if ((rc = dev_alloc(ctx, &g.attn_l, (size_t)heads * out_dim))) return rc;
    if ((rc = dev_alloc(ctx, &g.attn_r, (size_t)heads * out_dim))) return rc;
    HIPCHK(ctx, hipMemcpy(g.attn_l, attn_l, (size_t)heads * out_dim * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(g.attn_r, attn_r, (size_t)heads * out_dim * sizeof(float), hipMemcpyHostToDevice));
    ctx->gat_ready[layer] = true;
    return MPE_OK;
}

int mpe_set_mlp_params(mpe_ctx *ctx, int32_t n_layers, float slope) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    if (n_layers < 1 || n_layers > MPE_MAX_MLP_LAYERS) return fail(ctx, MPE_ERR_INVALID, "bad MLP layer count %d", n_layers);
    if (ctx->mlp_ws_ready) return fail(ctx, MPE_ERR_STATE, "MLP weights are frozen after the first batch");
    ctx->mlp_layers = n_layers;
    ctx->mlp_slope = slope;
    return MPE_OK;
}

int mpe_set_mlp_layer(mpe_ctx *ctx, int32_t layer, int32_t in_dim, int32_t out_dim, const float *w, const float *b) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    if (layer < 0 || layer >= ctx->mlp_layers) return fail(ctx, MPE_ERR_INVALID, "MLP layer index %d out of range", layer);
    if (ctx->mlp_ws_ready) return fail(ctx, MPE_ERR_STATE, "MLP weights are frozen after the first batch");
    if (!w || !b) return fail(ctx, MPE_ERR_INVALID, "MLP layer %d: missing tensor", layer);
    int rc = upload_linear(ctx, w, b, out_dim, in_dim, &ctx->mlp[layer]);
    if (rc) return rc;
    ctx->mlp_ready[layer] = true;
    return MPE_OK;
}

int mpe_upload_linear(mpe_ctx *ctx, const float *w, const float *b, int32_t out_dim, int32_t in_dim, float **d_w,
                      float **d_b, int32_t *ldw) {
    if (!ctx || !d_w || !d_b || !ldw) return MPE_ERR_INVALID;
    Linear L;
    int rc = upload_linear(ctx, w, b, out_dim, in_dim, &L);
    if (rc) return rc;
    *d_w = L.w;
    *d_b = L.b;
    *ldw = L.ldw;
    return MPE_OK;
}

int mpe_free_device(mpe_ctx *ctx, void *d_ptr) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    for (size_t i = 0; i < ctx->owned.size(); ++i)
        if (ctx->owned[i] == d_ptr) {
            (void)hipFree(d_ptr);
            ctx->owned.erase(ctx->owned.begin() + i);
            return MPE_OK;
        }
    return fail(ctx, MPE_ERR_INVALID, "pointer not owned by this context");
}

int mpe_linear(mpe_ctx *ctx, void *stream, const float *d_a, int32_t lda, const float *d_w, int32_t ldw,
               const float *d_bias, float *d_c, int32_t ldc, int32_t m, const int32_t *d_m, int32_t n, int32_t k,
               int32_t slope_on, float slope) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    if (!d_a || !d_w || !d_bias || !d_c || m < 0 || n < 1 || k < 1)
        return fail(ctx, MPE_ERR_INVALID, "mpe_linear: bad argument");
    if (ldw % LD_ALIGN || ldw < k || lda < ldw || ldc % 4 || ldc < n)
        return fail(ctx, MPE_ERR_INVALID, "mpe_linear: strides must satisfy ldw%%32==0, lda>=ldw>=k, ldc%%4==0");
    Linear L;
    L.w = const_cast<float *>(d_w);
    L.b = const_cast<float *>(d_bias);
    L.in_dim = k;
    L.out_dim = n;
    L.ldw = ldw;
    if (slope_on & 32) {
        if (lda % 4 || (reinterpret_cast<uintptr_t>(d_a) & 15) || (reinterpret_cast<uintptr_t>(d_w) & 15))
            return fail(ctx, MPE_ERR_INVALID, "mpe_linear (f64 form): rows must be 16-byte aligned (lda %% 4 == 0)");
        HIPCHK(ctx, launch_linear_f64(static_cast<hipStream_t>(stream), d_a, lda, d_w, ldw, d_bias, d_c, ldc, m, d_m, n, ldw, (slope_on & 1) != 0, slope));
        return MPE_OK;
    }
    if (slope_on & 4) {
        // split-bf16 arithmetic (gemm_sb16.hip) on caller-provided weights: the planes are made for this call (a stage-level
        // entry point for tests; the batch entry points keep theirs with the context)
        hipStream_t s = static_cast<hipStream_t>(stream);
        // (d_w must hold weight_rows(n) = round_up(n, 16) + 208 rows of ldw floats, zero padded, as mpe_upload_linear prepares
        // them: the planes are made of all of them, and the loader's 32-bit per-lane offset covers 3 planes of that size)
        const size_t count = (size_t)weight_rows(n) * ldw;
        if (3 * count * sizeof(unsigned short) >= ((size_t)1 << 32))
            return fail(ctx, MPE_ERR_INVALID, "mpe_linear (split form): n x ldw too large for the 32-bit plane offsets");
        unsigned short *planes = nullptr;
        HIPCHK(ctx, hipMalloc(reinterpret_cast<void **>(&planes), 3 * count * sizeof(unsigned short)));
        hipError_t e = launch_split_planes(s, d_w, count, planes);
        if (e == hipSuccess)
            e = launch_linear_sb16(s, d_a, lda, planes, count, ldw, d_bias, d_c, ldc, m, d_m, n, ldw, (slope_on & 1) != 0, slope, (slope_on & 8) == 0,
                                   nullptr, nullptr, false, (slope_on & 16) ? 1 : 2);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(planes);
        HIPCHK(ctx, e);
        return MPE_OK;
    }
    return linear(ctx, static_cast<hipStream_t>(stream), d_a, lda, L, d_c, ldc, m, d_m, (slope_on & 1) != 0, slope,
                  (slope_on & 2) != 0);
}

int mpe_head_features(mpe_ctx *ctx, void *stream, const mpe_batch *b, float *d_feat) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;        // an empty batch has nothing to write (outputs may be NULL)
    DeviceGuard dg(ctx);
    if (!d_feat) return fail(ctx, MPE_ERR_INVALID, "d_feat is NULL");
    HIPCHK(ctx, launch_head_features(static_cast<hipStream_t>(stream), ctx->d_cfg, *b, ctx->cfg.n_joints, d_feat,
                                     ctx->cfg.n_joints * 10, 0, 0, false));
    return MPE_OK;
}

int mpe_dense_rows(mpe_ctx *ctx, void *stream, const mpe_batch *b, float *d_rows, int32_t ld) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;
    DeviceGuard dg(ctx);
    const int F = 2 + ctx->cfg.n_cameras * ctx->cfg.n_joints * 10;
    if (!d_rows || ld < F) return fail(ctx, MPE_ERR_INVALID, "mpe_dense_rows: d_rows is NULL or ld < %d", F);
    if (b->n_frames != 1) return fail(ctx, MPE_ERR_UNSUPPORTED, "mpe_dense_rows: one graph per call (node order = heads, then edge-nodes)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIPCHK(ctx, launch_head_features(s, ctx->d_cfg, *b, ctx->cfg.n_joints, d_rows, ld, 0, 0, true));
    HIPCHK(ctx, launch_onehot_rows(s, d_rows + (size_t)b->n_heads * ld, b->n_edge_nodes, ld, 1));
    return MPE_OK;
}

int mpe_gat_forward(mpe_ctx *ctx, void *stream, const mpe_batch *b, const float *d_feats, int32_t ld_feats,
                    float *d_scores_en, float *d_scores_heads) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;        // an empty batch has nothing to write (outputs may be NULL)
    DeviceGuard dg(ctx);
    if (!d_scores_en) return fail(ctx, MPE_ERR_INVALID, "d_scores_en is NULL");
    return run_gat(ctx, static_cast<hipStream_t>(stream), b, d_scores_en, d_scores_heads, d_feats, ld_feats);
}

int mpe_set_gat_output(mpe_ctx *ctx, int32_t mode) {
    if (!ctx) return MPE_ERR_INVALID;
    if (mode != 1 && mode != 2) return fail(ctx, MPE_ERR_INVALID, "GAT output mode: 1 = sigmoid, 2 = identity");
    ctx->gat_out_mode = mode;
    return MPE_OK;
}

static int report_status(mpe_ctx *ctx, int32_t st) {
    if (st & 2)
        return fail(ctx, MPE_ERR_INVALID, "an explicit edge-node list held a pair outside its frame (or h1 == h2), or repeated pairs "
                    "beyond the in-degree capacity 2 * max_heads_per_frame; such pairs were replaced / dropped");
    if (st & 1)
        return fail(ctx, MPE_ERR_CAPACITY, "a frame holds more than max_heads_per_frame = %d skeletons (or, with an explicit edge-node list, more "
                    "than %d edge-nodes); its scores are zero and it produced no persons", ctx->cfg.max_heads_per_frame, ctx->x_m_cap);
    return MPE_OK;
}

// mpe_sync_status in two halves, for a caller that has more to do before it waits (the per-frame mirrors: the read-back of the status
// word joins the chain it belongs to instead of starting a chain of its own after the wait): mpe_status_queue orders the read-back and
// the reset behind everything queued so far, mpe_status_wait synchronises and reports it.
int mpe_status_queue(mpe_ctx *ctx, void *stream) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ctx->status_queued) {                       // an earlier half nobody waited for: its word must not be lost
        HIPCHK(ctx, hipStreamSynchronize(s));
        ctx->status_carry |= *ctx->h_status;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t), s));
    ctx->status_queued = true;
    return MPE_OK;
}

int mpe_status_wait(mpe_ctx *ctx, void *stream) {
    if (!ctx) return MPE_ERR_INVALID;
    if (!ctx->status_queued) {
        const int rc = mpe_status_queue(ctx, stream);
        if (rc) return rc;
    }
    DeviceGuard dg(ctx);
    HIPCHK(ctx, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    ctx->status_queued = false;
    const int32_t st = *ctx->h_status | ctx->status_carry;
    ctx->status_carry = 0;
    return report_status(ctx, st);
}

int mpe_sync_status(mpe_ctx *ctx, void *stream) {
    const int rc = mpe_status_queue(ctx, stream);
    return rc ? rc : mpe_status_wait(ctx, stream);
}

static int stage_layer_checks(mpe_ctx *ctx, const mpe_batch *b, int32_t layer, const void *in, int32_t ld_in, int cols_in,
                              const void *out, int32_t ld_out, int cols_out) {
    if (layer < 0 || layer >= ctx->gat_layers) return fail(ctx, MPE_ERR_INVALID, "GAT layer index %d out of range", layer);
    if (!in || !out || ld_in < cols_in || ld_out < cols_out)
        return fail(ctx, MPE_ERR_INVALID, "stage entry point: need ld_in >= %d and ld_out >= %d", cols_in, cols_out);
    (void)b;
    return MPE_OK;
}

int mpe_gat_layer(mpe_ctx *ctx, void *stream, const mpe_batch *b, int32_t layer, const float *d_in, int32_t ld_in,
                  float *d_out, int32_t ld_out, int32_t activation) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;        // an empty batch has nothing to write (outputs may be NULL)
    DeviceGuard dg(ctx);
    if ((rc = ensure_gat_workspace(ctx))) return rc;
    if (activation < 0 || activation > 2) return fail(ctx, MPE_ERR_INVALID, "activation: 0 LeakyReLU, 1 sigmoid, 2 none");
    if (layer < 0 || layer >= ctx->gat_layers) return fail(ctx, MPE_ERR_INVALID, "GAT layer index %d out of range", layer);
    const GatLayer &g = ctx->gat[layer];
    const int hd = g.heads * g.out_dim;
    if ((rc = stage_layer_checks(ctx, b, layer, d_in, ld_in, g.in_dim, d_out, ld_out, hd))) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_nodes = b->n_heads + b->n_edge_nodes;
    if ((rc = gat_topology(ctx, s, b))) return rc;
    if (layer == 0) {
        if (!ctx->xdense) {
            if ((rc = dev_alloc(ctx, &ctx->xdense, (size_t)ctx->max_nodes * ctx->feat_ld))) return rc;
            if ((rc = dev_alloc(ctx, &ctx->hdense, (size_t)ctx->max_nodes * ctx->feat_ld))) return rc;
        }
        if ((rc = copy_rows_in(ctx, s, ctx->xdense, ctx->feat_ld, d_in, ld_in, g.in_dim, n_nodes))) return rc;
    } else if ((rc = copy_rows_in(ctx, s, ctx->act[0], ctx->act_ld, d_in, ld_in, g.in_dim, n_nodes)))
        return rc;
    AggArgs a = gat_agg_args(ctx, layer);
    int n_rows_ft2 = 0;
    if ((rc = gat_layer_linear(ctx, s, b, layer, layer == 0 ? GAT_IN_DENSE0 : GAT_IN_ACT, &a, &n_rows_ft2))) return rc;
    a.out_mode = activation;
    a.out = ctx->act[0];
    a.ld_out = ctx->act_ld;
    if ((rc = gat_attention(ctx, s, b, layer, a, n_rows_ft2))) return rc;
    if (n_nodes > 0)
        HIPCHK(ctx, hipMemcpy2DAsync(d_out, (size_t)ld_out * sizeof(float), ctx->act[0], (size_t)ctx->act_ld * sizeof(float),
                                     (size_t)hd * sizeof(float), n_nodes, hipMemcpyDeviceToDevice, s));
    return MPE_OK;
}

int mpe_edge_softmax_aggregate(mpe_ctx *ctx, void *stream, const mpe_batch *b, int32_t layer, const float *d_ft2,
                               int32_t ld_ft2, float *d_out, int32_t ld_out) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;        // an empty batch has nothing to write (outputs may be NULL)
    DeviceGuard dg(ctx);
    if ((rc = ensure_gat_workspace(ctx))) return rc;
    if (layer < 0 || layer >= ctx->gat_layers) return fail(ctx, MPE_ERR_INVALID, "GAT layer index %d out of range", layer);
    if (ctx->gat_reduced || ctx->gat_attn_fp16)
        return fail(ctx, MPE_ERR_STATE, "mpe_edge_softmax_aggregate takes fp32 rows (a reduced-precision mode is on)");
    const GatLayer &g = ctx->gat[layer];
    const int hd = g.heads * g.out_dim;
    if ((rc = stage_layer_checks(ctx, b, layer, d_ft2, ld_ft2, hd, d_out, ld_out, hd))) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_nodes = b->n_heads + b->n_edge_nodes;
    if ((rc = gat_topology(ctx, s, b))) return rc;
    if ((rc = copy_rows_in(ctx, s, ctx->act[2], ctx->act_ld, d_ft2, ld_ft2, hd, n_nodes))) return rc;
    AggArgs a = gat_agg_args(ctx, layer);
    a.ft2 = ctx->act[2];
    a.out_mode = 2;
    a.out = ctx->act[0];
    a.ld_out = ctx->act_ld;
    if ((rc = gat_attention(ctx, s, b, layer, a, n_nodes))) return rc;
    if (n_nodes > 0)
        HIPCHK(ctx, hipMemcpy2DAsync(d_out, (size_t)ld_out * sizeof(float), ctx->act[0], (size_t)ctx->act_ld * sizeof(float),
                                     (size_t)hd * sizeof(float), n_nodes, hipMemcpyDeviceToDevice, s));
    return MPE_OK;
}

int mpe_set_threshold(mpe_ctx *ctx, float threshold) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    if (!(threshold >= 0.f)) return fail(ctx, MPE_ERR_INVALID, "threshold must be >= 0");
    ctx->hcfg.threshold = threshold;
    ctx->cfg.threshold = threshold;
    HIPCHK(ctx, hipMemcpy(ctx->d_cfg, &ctx->hcfg, sizeof ctx->hcfg, hipMemcpyHostToDevice));
    return MPE_OK;
}

int mpe_cluster_batch(mpe_ctx *ctx, void *stream, const mpe_batch *b, const float *d_scores, int32_t *d_persons,
                      int32_t *d_n_persons) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;        // an empty batch has nothing to write (outputs may be NULL)
    DeviceGuard dg(ctx);
    if (!d_scores || !d_persons || !d_n_persons) return fail(ctx, MPE_ERR_INVALID, "mpe_cluster_batch: NULL output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = gat_topology(ctx, s, b))) return rc;
    HIPCHK(ctx, launch_cluster(s, ctx->d_cfg, *b, ctx->en_pair, d_scores, ctx->cfg.max_persons_per_frame,
                               ctx->cfg.max_heads_per_frame, ctx->cl_keys, ctx->cl_keys_per_frame, ctx->cl_scratch,
                               ctx->cl_scratch_per_frame, d_persons, d_n_persons));
    return MPE_OK;
}

int mpe_match_batch(mpe_ctx *ctx, void *stream, const mpe_batch *b, float *d_scores, int32_t *d_persons,
                    int32_t *d_n_persons) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;        // an empty batch has nothing to write (outputs may be NULL)
    DeviceGuard dg(ctx);
    if (!d_persons || !d_n_persons) return fail(ctx, MPE_ERR_INVALID, "mpe_match_batch: NULL output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *scores = d_scores;
    if (!scores) {
        scores = ctx->scores_tmp;
    }
    const int hmax = ctx->cfg.max_heads_per_frame;
    if ((rc = ensure_gat_workspace(ctx))) return rc;
    for (int k = 0; k < 2; ++k)                               // whatever route this call takes: pairs solved for an older batch in these arrays are stale
        if (ctx->pair_key[k] == b->d_xy) ctx->pair_key[k] = nullptr;
    if (lat_gat_ok(ctx, b, false) && lat_tail_available(hmax, ctx->cl_keys_per_frame) && b->n_edge_nodes > 0) {
        // small batch: the last layer's scores and the clustering in ONE launch, and beside them every cross-camera pair of the batch
        // solved for the row kernel of mpe_mlp3d_batch (which recognises the batch by the tag left here)
        LatTail tail;
        if ((rc = run_gat_lat(ctx, s, b, scores, nullptr, &tail))) return rc;
        if (tail.deferred) {
            const int J = ctx->cfg.n_joints;
            const int slot = ctx->pair_next;
            const size_t need = (size_t)LAT_MAX_FRAMES * hmax * hmax * J * 3;
            // (the buffer is indexed by the batch's head number: a batch whose frames are within capacity has at most 16 hmax of them;
            // anything else -- flagged on the device, rejected by Engine.check_capacity -- gets no pair solves and no tag)
            const bool pairs_ok = (size_t)b->n_heads <= (size_t)LAT_MAX_FRAMES * hmax;
            if (pairs_ok && !ctx->pair_pts[slot] && (rc = dev_alloc(ctx, &ctx->pair_pts[slot], need, false))) return rc;
            for (int k = 0; k < 2; ++k)                       // one tag per input array at most: an older batch in the same arrays is gone
                if (k == slot || ctx->pair_key[k] == b->d_xy) ctx->pair_key[k] = nullptr;
            HIPCHK(ctx, launch_lat_tail(s, ctx->d_cfg, *b, J, ctx->en_pair, ctx->en_frame, tail.ft2, tail.ld, ctx->a12, tail.alpha, tail.out_slope,
                                        tail.out_mode, ctx->node_off, scores, ctx->cfg.max_persons_per_frame, hmax, ctx->cl_keys_per_frame, d_persons,
                                        d_n_persons, pairs_ok ? ctx->pair_pts[slot] : nullptr));
            if (pairs_ok) {
                ctx->pair_key[slot] = b->d_xy;
                ctx->pair_heads[slot] = b->n_heads;
                ctx->pair_en[slot] = b->n_edge_nodes;
                ctx->pair_next = slot ^ 1;
            }
            return MPE_OK;
        }
    } else if ((rc = run_gat(ctx, s, b, scores, nullptr))) {
        return rc;
    }
    HIPCHK(ctx, launch_cluster(s, ctx->d_cfg, *b, ctx->en_pair, scores, ctx->cfg.max_persons_per_frame,
                               ctx->cfg.max_heads_per_frame, ctx->cl_keys, ctx->cl_keys_per_frame, ctx->cl_scratch,
                               ctx->cl_scratch_per_frame, d_persons, d_n_persons));
    return MPE_OK;
}

int mpe_mlp_input_rows(mpe_ctx *ctx, void *stream, const mpe_batch *b, const int32_t *d_persons,
                       const int32_t *d_n_persons, float *d_rows, int32_t ld_rows, uint8_t *d_valid) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;        // an empty batch has nothing to write (outputs may be NULL)
    DeviceGuard dg(ctx);
    const int width = ctx->cfg.n_cameras * ctx->cfg.n_joints * ctx->cfg.numbers_per_joint;
    if (!d_persons || !d_n_persons || !d_rows || ld_rows < width)
        return fail(ctx, MPE_ERR_INVALID, "mpe_mlp_input_rows: bad argument");
    HIPCHK(ctx, launch_mlp_rows(static_cast<hipStream_t>(stream), ctx->d_cfg, ctx->cfg.n_cameras, ctx->cfg.n_joints, *b,
                                d_persons, d_n_persons, nullptr, ctx->cfg.max_persons_per_frame, d_rows, ld_rows,
                                d_valid));
    return MPE_OK;
}

static int mlp_chain(mpe_ctx *ctx, hipStream_t s, const float *x, int ld_x, int m, const int32_t *d_m, float **y_out,
                     int *ld_y, const DecodeEpi *dec = nullptr, bool *dec_done = nullptr) {
    if (dec_done) *dec_done = false;
    const float *in = x;
    int ld_in = ld_x;
    int rc;
    for (int l = 0; l < ctx->mlp_layers; ++l) {
        float *out = ctx->mlp_act[l & 1];
        const bool last = l == ctx->mlp_layers - 1;
        if (ctx->mlp_f64mm) {
            Linear &L = ctx->mlp[l];
            if (ld_in < L.ldw) return fail(ctx, MPE_ERR_INVALID, "activation stride %d < padded K %d", ld_in, L.ldw);
            GemmProf gp(ctx, s, d_m ? 0.0 : 2.0 * m * (double)L.out_dim * L.in_dim, d_m ? L.out_dim : 0, d_m ? L.in_dim : 0, 3);
            HIPCHK(ctx, launch_linear_f64(s, in, ld_in, L.w, L.ldw, L.b, out, ctx->mlp_ld_hidden, m, d_m, L.out_dim, L.ldw, !last, ctx->mlp_slope));
        } else if (ctx->mlp_split) {
            Linear &L = ctx->mlp[l];
            if ((rc = ensure_split_weights(ctx, s, &L))) return rc;
            if (ld_in < L.ldw) return fail(ctx, MPE_ERR_INVALID, "activation stride %d < padded K %d", ld_in, L.ldw);
            GemmProf gp(ctx, s, d_m ? 0.0 : 2.0 * m * (double)L.out_dim * L.in_dim, d_m ? L.out_dim : 0, d_m ? L.in_dim : 0, 1);
            HIPCHK(ctx, launch_linear_sb16(s, in, ld_in, L.w3, (size_t)weight_rows(L.out_dim) * L.ldw, L.ldw, L.b, out, ctx->mlp_ld_hidden, m,
                                           d_m, L.out_dim, L.ldw, !last, ctx->mlp_slope, true, nullptr, nullptr, false, ctx->mlp_flush,
                                           last ? dec : nullptr, last ? dec_done : nullptr));
        } else if (ctx->mlp_bf16) {
            Linear &L = ctx->mlp[l];
            if ((rc = ensure_bf16_weights(ctx, &L))) return rc;
            if (ld_in < L.ldw16) return fail(ctx, MPE_ERR_INVALID, "bf16 GEMM needs an input stride >= %d", L.ldw16);
            GemmProf gp(ctx, s, d_m ? 0.0 : 2.0 * m * (double)L.out_dim * L.in_dim, d_m ? L.out_dim : 0, d_m ? L.in_dim : 0, 2);
            HIPCHK(ctx, launch_linear_bf16(s, in, ld_in, L.w16, L.ldw16, L.b, out, ctx->mlp_ld_hidden, m, d_m, L.out_dim,
                                           L.ldw16, !last, ctx->mlp_slope));
        } else if ((rc = linear(ctx, s, in, ld_in, ctx->mlp[l], out, ctx->mlp_ld_hidden, m, d_m, !last, ctx->mlp_slope,
                                ctx->mlp_acc64)))
            return rc;
        in = out;
        ld_in = ctx->mlp_ld_hidden;
    }
    *y_out = const_cast<float *>(in);
    *ld_y = ld_in;
    return MPE_OK;
}

int mpe_mlp_forward(mpe_ctx *ctx, void *stream, const float *d_x, int32_t ld_x, int32_t m, float *d_y, int32_t ld_y) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    int rc = ensure_mlp_workspace(ctx);
    if (rc) return rc;
    if (!d_x || !d_y || m < 0) return fail(ctx, MPE_ERR_INVALID, "mpe_mlp_forward: bad argument");
    if ((size_t)m > (size_t)ctx->cfg.max_frames * ctx->cfg.max_persons_per_frame)
        return fail(ctx, MPE_ERR_CAPACITY, "mpe_mlp_forward: %d rows exceed capacity", m);
    if (ld_x < ctx->mlp_ld_in) return fail(ctx, MPE_ERR_INVALID, "mpe_mlp_forward: ld_x must be >= %d (zero padded)", ctx->mlp_ld_in);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *y;
    int ldy;
    if ((rc = mlp_chain(ctx, s, d_x, ld_x, m, nullptr, &y, &ldy))) return rc;
    const int n_out = ctx->mlp[ctx->mlp_layers - 1].out_dim;
    HIPCHK(ctx, hipMemcpy2DAsync(d_y, (size_t)ld_y * sizeof(float), y, (size_t)ldy * sizeof(float),
                                 (size_t)n_out * sizeof(float), m, hipMemcpyDeviceToDevice, s));
    return MPE_OK;
}

int mpe_mlp3d_batch(mpe_ctx *ctx, void *stream, const mpe_batch *b, const int32_t *d_persons,
                    const int32_t *d_n_persons, float *d_poses, uint8_t *d_valid) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;        // an empty batch has nothing to write (outputs may be NULL)
    DeviceGuard dg(ctx);
    if ((rc = ensure_mlp_workspace(ctx))) return rc;
    if (!d_persons || !d_n_persons || !d_poses) return fail(ctx, MPE_ERR_INVALID, "mpe_mlp3d_batch: NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int pcap = ctx->cfg.max_persons_per_frame;
    const int n_out = ctx->mlp[ctx->mlp_layers - 1].out_dim;
    if (n_out != ctx->cfg.n_joints * 3)
        return fail(ctx, MPE_ERR_INVALID, "MLP output width %d != 3*J", n_out);
    // small batches: the persons' prefix inside the row kernel, the decode inside the last layer's launch (two launches fewer)
    const bool small = latency_path_on() && b->n_frames <= LAT_MAX_FRAMES;
    if (small) {
        // the batch whose pairs mpe_match_batch left solved (same input arrays, same sizes)?  then the row kernel fetches them
        const double *pairs = nullptr;
        for (int k = 0; k < 2; ++k)
            if (ctx->pair_key[k] && ctx->pair_key[k] == b->d_xy && ctx->pair_heads[k] == b->n_heads && ctx->pair_en[k] == b->n_edge_nodes)
                pairs = ctx->pair_pts[k];
        HIPCHK(ctx, launch_mlp_rows(s, ctx->d_cfg, ctx->cfg.n_cameras, ctx->cfg.n_joints, *b, d_persons, d_n_persons, nullptr, pcap, ctx->mlp_rows,
                                    ctx->mlp_ld_in, d_valid ? d_valid : ctx->valid_tmp, ctx->person_off, ctx->mlp_count, d_poses, n_out, pairs,
                                    ctx->cfg.max_heads_per_frame));
    } else {
        HIPCHK(ctx, launch_person_scan(s, b->n_frames, pcap, d_n_persons, ctx->person_off, ctx->mlp_count));
        HIPCHK(ctx, launch_mlp_rows(s, ctx->d_cfg, ctx->cfg.n_cameras, ctx->cfg.n_joints, *b, d_persons, d_n_persons,
                                    ctx->person_off, pcap, ctx->mlp_rows, ctx->mlp_ld_in, d_valid ? d_valid : ctx->valid_tmp));
    }
    float *y;
    int ldy;
    const DecodeEpi dec{ctx->person_off, b->n_frames, pcap, n_out, 10.f, d_poses};
    bool decoded = false;
    if ((rc = mlp_chain(ctx, s, ctx->mlp_rows, ctx->mlp_ld_in, b->n_frames * pcap, ctx->mlp_count, &y, &ldy, small ? &dec : nullptr, &decoded))) return rc;
    if (!decoded) HIPCHK(ctx, launch_decode(s, b->n_frames, pcap, n_out, 10.f, d_n_persons, ctx->person_off, y, ldy, d_poses));
    return MPE_OK;
}

int mpe_triangulate_batch(mpe_ctx *ctx, void *stream, const mpe_batch *b, const int32_t *d_persons,
                          const int32_t *d_n_persons, double *d_poses, uint8_t *d_joint_valid, uint32_t flags) {
    int rc = check_batch(ctx, b);
    if (rc) return rc;
    if (b->n_frames == 0) return MPE_OK;        // an empty batch has nothing to write (outputs may be NULL)
    DeviceGuard dg(ctx);
    if (!d_persons || !d_n_persons || !d_poses || !d_joint_valid)
        return fail(ctx, MPE_ERR_INVALID, "mpe_triangulate_batch: NULL argument");
    HIPCHK(ctx, launch_triangulate(static_cast<hipStream_t>(stream), ctx->d_cfg, ctx->cfg.n_cameras, ctx->cfg.n_joints,
                                   *b, d_persons, d_n_persons, ctx->cfg.max_persons_per_frame, d_poses, d_joint_valid,
                                   (flags & 1u) ? 0xFFFFFFFFu : ctx->cfg.used_joint_mask, (flags & 2u) != 0));
    return MPE_OK;
}

int mpe_dlt_pairs(mpe_ctx *ctx, void *stream, const double *d_pts, const int32_t *d_cams, int32_t n, double *d_out) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    if (!d_pts || !d_cams || !d_out || n < 0) return fail(ctx, MPE_ERR_INVALID, "mpe_dlt_pairs: bad argument");
    HIPCHK(ctx, launch_dlt_pairs(static_cast<hipStream_t>(stream), ctx->d_cfg, d_pts, d_cams, n, d_out));
    return MPE_OK;
}

int mpe_set_precision(mpe_ctx *ctx, int32_t gat_acc64, int32_t mlp_acc64) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    if (gat_acc64 < 0 || gat_acc64 > 6 || mlp_acc64 < 0 || mlp_acc64 > 5)
        return fail(ctx, MPE_ERR_INVALID, "precision modes: GAT 0..6, MLP 0..5");
    // GAT modes 4 / 5: modes 0 / 1 with the GEMMs of layers >= 1 in the split-bf16 form (4 = the default); 6 = mode 3 likewise
    // (fp16 rows: from the split tile kernel's coefficient epilogue where gat_linear finds it applicable, the fp32 MFMA otherwise)
    ctx->gat_split = gat_acc64 >= 4;
    if (gat_acc64 >= 4) gat_acc64 = gat_acc64 == 4 ? 0 : gat_acc64 == 5 ? 1 : 3;
    ctx->gat_acc64 = gat_acc64 == 1;
    ctx->gat_reduced = gat_acc64 == 2;
    ctx->gat_attn_fp16 = gat_acc64 == 3;
    ctx->mlp_acc64 = mlp_acc64 == 1;
    ctx->mlp_bf16 = mlp_acc64 == 2;
    ctx->mlp_split = mlp_acc64 == 3 || mlp_acc64 == 4;
    ctx->mlp_f64mm = mlp_acc64 == 5;               // MLP mode 5: exact products + f64 accumulation on the f64 matrix pipe
    ctx->mlp_flush = mlp_acc64 == 4 ? 1 : 2;       // MLP mode 4: the split form with an f64 flush per K stage (maximum accuracy)
    return MPE_OK;
}

int mpe_profile_enable(mpe_ctx *ctx, int32_t on) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    ctx->profiling = on != 0;
    if (on == 1) ctx->prof_used = 0;        // 1 = start afresh, 2 = resume (keep the records), 0 = pause / off
    return MPE_OK;
}

int mpe_profile_read(mpe_ctx *ctx, double *gemm_ms, double *gemm_flop, int64_t *gemm_launches, double *total_ms) {
    if (!ctx) return MPE_ERR_INVALID;
    DeviceGuard dg(ctx);
    HIPCHK(ctx, hipDeviceSynchronize());
    int32_t dev_m = 0;
    HIPCHK(ctx, hipMemcpy(&dev_m, ctx->mlp_count, sizeof dev_m, hipMemcpyDeviceToHost));
    double ms[3] = {0, 0, 0}, flop[3] = {0, 0, 0};
    int64_t cnt[3] = {0, 0, 0};
    for (size_t i = 0; i < ctx->prof_used; ++i) {
        if (ctx->prof[i].kind == 3) continue;        // f64 matrix-pipe launches (MLP mode 5): not part of the roofline figures
        const int k = ctx->prof[i].kind == 1 ? 1 : ctx->prof[i].kind == 2 ? 2 : 0;       // fp32 MFMA | split-bf16 | plain bf16 (reduced modes)
        float t = 0;
        if (hipEventElapsedTime(&t, ctx->prof[i].start, ctx->prof[i].stop) == hipSuccess) ms[k] += t;
        flop[k] += ctx->prof[i].flop;
        if (ctx->prof[i].dev_n) flop[k] += 2.0 * dev_m * (double)ctx->prof[i].dev_n * ctx->prof[i].dev_k;
        ++cnt[k];
    }
    if (gemm_ms) *gemm_ms = ms[0];
    if (gemm_flop) *gemm_flop = flop[0];
    if (gemm_launches) *gemm_launches = cnt[0];
    ctx->sb_ms = ms[1];
    ctx->sb_flop = flop[1];
    ctx->sb_launches = cnt[1];
    ctx->bf_ms = ms[2];
    ctx->bf_flop = flop[2];
    ctx->bf_launches = cnt[2];
    if (total_ms) *total_ms = 0;
    ctx->prof_used = 0;
    return MPE_OK;
}

int mpe_profile_read_split(mpe_ctx *ctx, double *ms, double *flop, int64_t *launches) {
    if (!ctx) return MPE_ERR_INVALID;
    if (ms) *ms = ctx->sb_ms;
    if (flop) *flop = ctx->sb_flop;
    if (launches) *launches = ctx->sb_launches;
    return MPE_OK;
}

int mpe_profile_read_bf16(mpe_ctx *ctx, double *ms, double *flop, int64_t *launches) {
    if (!ctx) return MPE_ERR_INVALID;
    if (ms) *ms = ctx->bf_ms;
    if (flop) *flop = ctx->bf_flop;
    if (launches) *launches = ctx->bf_launches;
    return MPE_OK;
}

}  // extern "C"
