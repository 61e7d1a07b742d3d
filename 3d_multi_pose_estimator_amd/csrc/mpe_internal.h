// Internal declarations shared by the HIP translation units of libmpe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/mpe.h"

namespace mpe {

constexpr int GEMM_BM = 128;   // activation rows per workgroup
constexpr int GEMM_BN = 80;    // output features per workgroup (5 MFMA tiles of 16)
constexpr int GEMM_BK = 32;    // K depth per LDS stage
constexpr int GEMM_BN_MAX = 208;   // widest feature tile of the GEMM variants (13 MFMA tiles)
// rows a weight matrix / bias vector is padded to: any feature tile that starts below out_dim
// stays inside the allocation
inline int weight_rows(int out_dim) { return (out_dim + 15) / 16 * 16 + GEMM_BN_MAX; }
constexpr int LD_ALIGN = 32;   // every activation / weight row stride is a multiple of this

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct Linear {                // one nn.Linear, zero padded on the device
    float *w = nullptr;        // [weight_rows(out)][ldw]
    float *b = nullptr;        // [weight_rows(out)]
    unsigned short *w16 = nullptr;   // optional bf16 copy [rows][ldw16] (reduced-precision mode)
    int ldw16 = 0;
    unsigned short *w3 = nullptr;    // optional: the three bf16 planes of w, [3][weight_rows(out)][ldw] (split-bf16 mode, gemm_sb16.hip)
    int in_dim = 0, out_dim = 0, ldw = 0;
};

struct GatLayer {
    Linear fc1, fc2;
    float *attn_l = nullptr;   // [heads*out_dim]
    float *attn_r = nullptr;
    int in_dim = 0, heads = 0, out_dim = 0;
};

struct DevCfg {                // calibration + scalars, lives in device memory
    int32_t V, J, W, H, npj, min_views, median_axis;
    uint32_t used_joint_mask;
    float threshold, median_window;
    float Kinv[MPE_MAX_CAMERAS][9];
    float K[MPE_MAX_CAMERAS][9];
    float T_i[MPE_MAX_CAMERAS][16];
    double P[MPE_MAX_CAMERAS][12];
    double dist[MPE_MAX_CAMERAS][5];
};

// hipFuncSetAttribute (dynamic-LDS opt-in) is per DEVICE: remember which devices have it
struct PerDeviceFlag {
    bool done[64] = {};
    static int dev() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d & 63;
    }
    bool test() const { return done[dev()]; }
    void set() { done[dev()] = true; }
};

struct ProfileRec {
    hipEvent_t start, stop;
    double flop;      // known on the host, or
    int dev_n, dev_k;  // 2 * (device-side row count) * dev_n * dev_k
    int kind;          // 0 = fp32 MFMA launch (gemm.hip), 1 = split-bf16 launch (gemm_sb16.hip)
};

}  // namespace mpe

struct mpe_ctx {
    mpe_config cfg;
    int device = 0;                // HIP device the context was created on (every entry point switches to it)
    int32_t *h_status = nullptr;   // page-locked twin of the status word (mpe_sync_status reads it back without a staging copy)
    bool status_queued = false;    // mpe_status_queue has ordered a read-back nobody has waited for yet
    int32_t status_carry = 0;      // bits of read-backs that were overtaken by a later mpe_status_queue
    int32_t *d_status = nullptr;   // device-side status word: bit 0 = a frame exceeded max_heads_per_frame
    bool gat_ws_ready = false;     // set only after the LAST step of ensure_gat_workspace succeeded
    bool mlp_ws_ready = false;
    int gat_out_mode = 1;          // activation of the last GAT layer: 1 = sigmoid, 2 = identity (logits)
    mpe::DevCfg hcfg;
    mpe::DevCfg *d_cfg = nullptr;
    std::string err;
    // weights
    int gat_layers = 0;
    float gat_alpha = 0.15f, gat_hidden_slope = 0.01f;
    mpe::GatLayer gat[MPE_MAX_GAT_LAYERS];
    bool gat_ready[MPE_MAX_GAT_LAYERS] = {};
    float *en0_ft2 = nullptr;      // layer-0 transformed feature shared by all edge-nodes [ld]
    float *en0_a = nullptr;        // its a1 | a2 coefficients [2*heads]
    bool en0_ready = false;
    int mlp_layers = 0;
    float mlp_slope = 0.1f;
    bool mlp_acc64 = true;         // fp32 MFMA with f64 running sums per K stage in the MLP GEMMs (MLP mode 1; the parity mode of rounds 1-3)
    bool mlp_bf16 = false;         // reduced precision: bf16 MFMA for the MLP GEMMs
    bool mlp_f64mm = false;        // MLP mode 5: every launch on the f64 matrix pipe (gemm_f64.hip)
    int mlp_flush = 2;             // MLP modes 3 / 4: K stages per f64 flush of the split form (2 = default, 1 = the maximum-accuracy mode)
    bool mlp_split = true;         // DEFAULT (MLP mode 3): fp32-accurate MLP GEMMs on the bf16 MFMA (three bf16 planes per operand, six products, f64 sums every second stage)
    bool gat_acc64 = false;
    bool gat_reduced = false;      // reduced precision: bf16 MFMA GEMMs + fp16 feature rows in the attention stage
    bool gat_attn_fp16 = false;    // configs[4] as BASELINE words it: fp16 feature rows (ft2) in the attention stage, GEMMs stay fp32
    bool gat_split = true;         // DEFAULT (GAT mode 4): fc1 / fc2 of layers >= 1 in the split-bf16 form (gemm_sb16.hip); layer 0 (head rows only,
                                   // gathered / grouped launches) stays on the fp32 MFMA
    mpe::Linear mlp[MPE_MAX_MLP_LAYERS];
    bool mlp_ready[MPE_MAX_MLP_LAYERS] = {};
    // workspace (sized at create / grown when weights define the widths)
    int max_nodes = 0;
    int feat_ld = 0;               // padded F
    int act_ld = 0;                // padded widest hidden activation of the GAT
    float *x0 = nullptr;           // [max_heads][feat_ld] dense head rows
    float *h0 = nullptr;           // [max_heads][feat_ld] fc1 output of layer 0
    float *xdense = nullptr;       // [max_nodes][feat_ld] caller-provided dense rows (API mirror only)
    float *hdense = nullptr;
    // layer 0 grouped by camera: only the own-camera block of a head row is non-zero
    float *xc = nullptr;           // [max_heads][l0_ld] compact J*10 features
    int l0_ld = 0;
    mpe::Linear l0_fc1[MPE_MAX_CAMERAS];   // fc1 restricted to the camera's column block, bias + W[:,0] (views into l0_w)
    float *l0_w = nullptr;         // [V][weight_rows(in_dim)][l0_ld] the per-camera matrices, contiguous (grouped launch)
    float *l0_b = nullptr;         // [weight_rows(in_dim)] their common bias
    int32_t *cam_count = nullptr;  // [V]
    int32_t *cam_list = nullptr;   // [V][max_heads] head indices of each camera
    bool l0_grouped = true;
    float *act[3] = {nullptr, nullptr, nullptr};   // [max_nodes][act_ld]
    float *a12 = nullptr;          // [max_nodes][2*16]
    // small batches: every cross-camera pair of a batch solved beside the clustering (cluster.hip: k_lat_tail) for the row kernel of the
    // SAME batch to fetch: two buffers [heads][hmax][J][3] f64, each tagged with the batch it holds (the batch's d_xy pointer and sizes)
    double *pair_pts[2] = {nullptr, nullptr};
    const void *pair_key[2] = {nullptr, nullptr};
    int pair_heads[2] = {0, 0}, pair_en[2] = {0, 0}, pair_next = 0;
    unsigned short *gat_pl[2] = {nullptr, nullptr};   // small batches (lat.hip): activations between attention -> fc1 -> fc2 as three bf16 planes [3][lat_rows][act_ld]
    size_t gat_pl_plane = 0;       // plane stride in elements
    int lat_rows = 0;
    int32_t *head_frame = nullptr; // [max_heads]
    int32_t *en_frame = nullptr;   // [max_edge_nodes]
    int32_t *en_pair = nullptr;    // [max_edge_nodes][2] frame-local head ids
    uint16_t *head_src = nullptr;  // [max_frames][hmax][hmax + 1] in-edge sources of the heads (small frames only); explicit
                                   // pair lists: [n_frames][hmax][2 hmax] in the same allocation
    size_t head_src_cap = 0;       // entries allocated
    int x_m_cap = 0;               // explicit pair lists: edge-nodes a frame may hold (0 = mode unavailable on this context)
    int32_t *node_off = nullptr;   // [max_frames+1]
    uint64_t *cl_keys = nullptr;   // clustering scratch
    int32_t *cl_scratch = nullptr;
    size_t cl_keys_per_frame = 0, cl_scratch_per_frame = 0;
    int mlp_ld_in = 0, mlp_ld_hidden = 0;
    float *mlp_rows = nullptr;     // [max_frames*Pcap][mlp_ld_in]
    float *mlp_act[2] = {nullptr, nullptr};
    int32_t *mlp_count = nullptr;
    float *scores_tmp = nullptr;    // [max_edge_nodes]
    int32_t *person_off = nullptr;  // [max_frames+1]
    uint8_t *valid_tmp = nullptr;   // [max_frames*Pcap]
    std::vector<void *> owned;     // everything to hipFree at destroy
    // profiling
    bool profiling = false;
    std::vector<mpe::ProfileRec> prof;
    size_t prof_used = 0;
    hipEvent_t tot_start = nullptr, tot_stop = nullptr;
    bool tot_valid = false;
    double sb_ms = 0, sb_flop = 0;      // the split-bf16 launches of the records last read by mpe_profile_read
    int64_t sb_launches = 0;
    double bf_ms = 0, bf_flop = 0;      // likewise the plain bf16 launches (reduced-precision modes)
    int64_t bf_launches = 0;
};

namespace mpe {

// gemm.hip
struct AttnCoef {              // fc2 of a graph-attention layer: also emit a1|a2 (gat2.py:57-58) from the epilogue
    const float *attn_l, *attn_r;   // [heads*out_dim]
    float *a12;                     // [rows][32]: a1[0..15] | a2[0..15]
    int heads, out_dim;
};
// the decode of the MLP output (metrics_from_model.py:281-294) from the epilogue of the last layer's launch: row m of the compacted
// rows is person m - person_off[f] of the frame f with person_off[f] <= m < person_off[f + 1]
struct DecodeEpi {
    const int32_t *person_off;     // [n_frames + 1]
    int n_frames, pcap, n_out;
    float scale;
    float *poses;                  // [n_frames][pcap][n_out]
};
hipError_t launch_linear_sb16(hipStream_t s, const float *A, int lda, const unsigned short *W3, size_t w_plane, int ldw,
                              const float *bias, float *C, int ldc, int m_cap, const int32_t *d_m, int n, int k_pad, bool leaky,
                              float slope, bool f64 = true, const AttnCoef *coef = nullptr, bool *coef_done = nullptr,
                              bool out_half = false,      // out_half (fp16 ft2 rows, configs[4]): tile kernel, fp32-chain launches without LeakyReLU only; hipErrorInvalidValue otherwise
                              int flush_stages = 2,       // f64 launches: K stages per f64 flush (2 = default, 1 = the maximum-accuracy mode)
                              const DecodeEpi *dec = nullptr, bool *dec_done = nullptr);   // the K-split kernel can also store the decoded poses (*dec_done says whether it did)
bool linear_sb16_uses_tile_kernel(int m_cap, int n, bool f64);
int device_cu_count();        // gemm_sb16.hip: CUs of the current device (asked once per device)

// lat.hip: the small-batch ("latency") forms -- the arithmetic of the split-bf16 kernels (same bits), shortest serial depth
bool lat_gemm_available(int k_pad, int n, bool fc2, int out_dim);
hipError_t launch_lat_gemm(hipStream_t s, const unsigned short *Apl, int lda, size_t a_plane, const unsigned short *W3, size_t w_plane, int ldw,
                           const float *bias, float *C, int ldc, unsigned short *Cpl, int ldcp, size_t c_plane, int m, int n, int k_pad, bool fc2,
                           float slope, const float *attn_l, const float *attn_r, float *a12, int out_dim, bool *coef_done = nullptr);
bool lat_gemm_fusable(int k1_pad, int n1, int k2_pad, int n2, int out_dim2);
hipError_t launch_lat_gemm_fused(hipStream_t s, const unsigned short *Apl, int lda, size_t a_plane, const unsigned short *W3, size_t w_plane, int ldw,
                                 const float *bias, int m, int n, float slope, const unsigned short *W3b, size_t w_plane_b, int ldw_b, const float *bias_b,
                                 float *C2, int ldc2, const float *attn_l, const float *attn_r, float *a12);
// gemm_f64.hip: exact products, f64 accumulation on the f64 matrix pipe (MLP mode 5: the f64-evaluated network, not the fast path; the slope rule is in include/mpe.h)
hipError_t launch_linear_f64(hipStream_t s, const float *A, int lda, const float *W, int ldw, const float *bias, float *C, int ldc,
                             int m_cap, const int32_t *d_m, int n, int k_pad, bool leaky, float slope);
hipError_t launch_linear(hipStream_t s, const float *A, int lda, const float *W, int ldw, const float *bias,
                         float *C, int ldc, int m_cap, const int32_t *d_m, int n, int k_pad, bool leaky,
                         float slope, bool acc64, const int32_t *a_rows = nullptr, const int32_t *c_rows = nullptr,
                         const AttnCoef *coef = nullptr, bool *coef_done = nullptr, bool out_half = false);
// one launch for `n_grp` GEMMs that share the activation / result buffers: group g takes the rows
// row_lists[g * grp_stride + 0 .. grp_count[g]) (device-side counts) and the weights W + g * w_grp_stride
hipError_t launch_linear_grouped(hipStream_t s, const float *A, int lda, const float *W, int ldw, long w_grp_stride,
                                 const float *bias, float *C, int ldc, int m_cap, const int32_t *grp_count, int n_grp,
                                 const int32_t *row_lists, int grp_stride, int n, int k_pad, bool leaky, float slope, bool acc64 = false);
bool linear_uses_tile_kernel(int m_cap, int n);
hipError_t launch_onehot_rows(hipStream_t s, float *rows, int n_rows, int ld, int col);
hipError_t launch_group_heads(hipStream_t s, int n_heads, int V, const int32_t *head_cam, int32_t *cam_count,
                              int32_t *cam_list, int list_stride, bool counts_zeroed = false);

hipError_t launch_linear_bf16(hipStream_t s, const float *A, int lda, const unsigned short *Wb, int ldw,
                              const float *bias, float *C, int ldc, int m_cap, const int32_t *d_m, int n, int k_pad,
                              bool leaky, float slope, int k_lim = 0, bool out_half = false,
                              const int32_t *a_rows = nullptr, const int32_t *c_rows = nullptr);

// gemm_sb16.hip: fp32-accurate nn.Linear on the bf16 MFMA (three bf16 planes per operand, f64 flush every second stage)
hipError_t launch_split_planes(hipStream_t s, const float *w, size_t count, unsigned short *planes);
// gat.hip
hipError_t launch_topology(hipStream_t s, const mpe_batch &b, int V, int32_t *node_off, int32_t *head_frame,
                           int32_t *en_frame, int32_t *en_pair, int max_heads_per_frame, int32_t *status,
                           uint16_t *head_src, int x_deg_cap = 0, int x_m_cap = 0);
// explicit pair lists (mpe_batch::d_en_pair): row stride of the in-edge source table = largest in-degree of a head
inline int explicit_deg_cap(int max_heads_per_frame) { return 2 * max_heads_per_frame; }
// entries of the per-frame in-edge source table of the heads, or 0 when frames of that capacity do not get one
size_t head_src_entries(int max_heads_per_frame, int V);
hipError_t launch_head_features(hipStream_t s, const DevCfg *cfg, const mpe_batch &b, int J, float *feat,
                                int ld_feat, int col0, int stride_cam, bool dense, int32_t *zero_counts = nullptr,
                                int n_zero = 0);
hipError_t launch_attn_coef(hipStream_t s, const float *ft2, int ld, int n_rows, int heads, int out_dim,
                            const float *attn_l, const float *attn_r, float *a12, int ft_half = 0);
struct AggArgs {
    const float *ft2;          // rows of transformed features (fp16 rows when ft_half, reduced precision)
    int ld;                    // row stride in elements of that type
    int ft_half;
    const float *a12;          // [rows][32]: a1[0..15] | a2[0..15]
    int a12_ready;             // a12 already holds this layer's coefficients (fc2 epilogue)
    int heads, out_dim;
    float alpha, out_slope;    // attention LeakyReLU slope; activation applied to the output
    int out_mode;              // 0 = LeakyReLU(out_slope), 1 = sigmoid, 2 = identity
    const float *en_const_ft2; // layer 0: one row shared by all edge-nodes (ft2 holds head rows only)
    const float *en_const_a;   // layer 0: its a1|a2
    float *out;                // [n_nodes][ld_out] (or scores when score_mode)
    int ld_out;
    int score_mode;            // last layer: write out[edge-node index] (and heads to out_heads)
    float *out_heads;
    const float *attn_l, *attn_r;   // [heads*out_dim]: for the kernels that compute a1 | a2 themselves when a12_ready == 0 (general path, small batches)
    unsigned short *out_pl;    // small batches (lat.hip): the output as three bf16 planes [3][rows][ld_out] INSTEAD of fp32 rows
    size_t out_pl_plane;       // plane stride in elements
};
hipError_t launch_aggregate(hipStream_t s, const mpe_batch &b, int V, int max_heads_per_frame,
                            const int32_t *node_off, const int32_t *head_frame, const int32_t *en_frame,
                            const int32_t *en_pair, const AggArgs &a, const uint16_t *head_src, int x_deg_cap = 0);

bool lat_l0a_available(int J, int l0_ld);
hipError_t launch_lat_l0a(hipStream_t s, const DevCfg *cfg, const mpe_batch &b, int V, int J, int32_t *node_off, int32_t *head_frame,
                          int32_t *en_frame, int32_t *en_pair, uint16_t *head_src, int hmax, int32_t *status, const float *l0_w,
                          long w_cam_stride, int l0_ld, const float *l0_b, int n_out, float *h0, int ld_h0, float alpha);
hipError_t launch_lat_attention(hipStream_t s, const mpe_batch &b, int V, int max_heads_per_frame, const int32_t *node_off,
                                const int32_t *head_frame, const int32_t *en_frame, const int32_t *en_pair, const AggArgs &a,
                                const uint16_t *head_src);
hipError_t launch_gat_attention(hipStream_t s, const mpe_batch &b, int V, int max_heads_per_frame,
                                const int32_t *node_off, const int32_t *head_frame, const int32_t *en_frame,
                                const int32_t *en_pair, const float *attn_l, const float *attn_r, float *a12,
                                const AggArgs &a, int n_rows_ft2, const uint16_t *head_src, int x_deg_cap = 0,
                                int x_m_cap = 0);

// cluster.hip
size_t cluster_keys_per_frame(int max_heads_per_frame);
size_t cluster_scratch_per_frame(int max_heads_per_frame);
hipError_t launch_cluster(hipStream_t s, const DevCfg *cfg, const mpe_batch &b, const int32_t *en_pair,
                          const float *scores, int pcap, int max_heads_per_frame, uint64_t *keys,
                          size_t keys_per_frame, int32_t *scratch, size_t scratch_per_frame,
                          int32_t *persons, int32_t *n_persons);

// the small-batch tail launch (cluster.hip: k_lat_tail): last layer's scores + clustering per frame, every cross-camera pair solved beside it
bool lat_tail_available(int hmax, size_t keys_per_frame);
hipError_t launch_lat_tail(hipStream_t s, const DevCfg *cfg, const mpe_batch &b, int J, const int32_t *en_pair, const int32_t *en_frame,
                           const float *ft2, int ld, const float *a12, float alpha, float out_slope, int out_mode, const int32_t *node_off,
                           float *scores, int pcap, int hmax, size_t keys_per_frame, int32_t *persons, int32_t *n_persons, double *pair_pts);

// pose3d.hip
hipError_t launch_person_scan(hipStream_t s, int n_frames, int pcap, const int32_t *n_persons, int32_t *person_off,
                              int32_t *total);
hipError_t launch_mlp_rows(hipStream_t s, const DevCfg *cfg, int V, int J, const mpe_batch &b,
                           const int32_t *persons, const int32_t *n_persons, const int32_t *person_off, int pcap,
                           float *rows, int ld_rows, uint8_t *valid, int32_t *scan_out = nullptr, int32_t *total_out = nullptr,
                           float *zero_poses = nullptr, int n_out = 0, const double *pair_pts = nullptr, int pair_hmax = 0);

hipError_t launch_triangulate(hipStream_t s, const DevCfg *cfg, int V, int J, const mpe_batch &b,
                              const int32_t *persons, const int32_t *n_persons, int pcap, double *poses,
                              uint8_t *joint_valid, uint32_t out_mask, bool positive_ids_only = false);
hipError_t launch_dlt_pairs(hipStream_t s, const DevCfg *cfg, const double *pts, const int32_t *cams, int n,
                            double *out);
hipError_t launch_decode(hipStream_t s, int n_frames, int pcap, int n_out, float scale, const int32_t *n_persons,
                         const int32_t *person_off, const float *y, int ld_y, float *poses);
int cluster_table_cap(int hmax);

}  // namespace mpe
