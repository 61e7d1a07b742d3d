/*
 * mpe.h — C ABI of libmpe_hip.so, the MI355X (gfx950) implementation of the per-frame
 * inference path of gnns4hri/3D_multi_pose_estimator:
 *
 *   2D skeletons per camera -> graph featurisation -> 5-layer graph attention network
 *   -> greedy person clustering -> { MLP 3D regression | pairwise-DLT triangulation }.
 *
 * The reference has no FFI: its operator API is the set of Python symbols that
 * test/metrics_from_model.py:12-24 and test/metrics_from_triangulation.py:13-23 import.
 * Each entry point below names the reference code it replaces; the Python mirror of
 * those symbols (the .py files of 3d_multi_pose_estimator_amd) is a thin ctypes layer over this
 * file (see INTEGRATION.md for the binding a maintainer would add to the reference).
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions across the boundary;
 *   - every call returns 0 on success or a negative mpe_status; mpe_last_error(ctx)
 *     gives the message of the last failure on that context;
 *   - pointers named d_* are DEVICE pointers owned by the caller (e.g. torch
 *     tensor.data_ptr()); everything else is host memory, copied during the call;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all batch
 *     entry points are asynchronous with respect to the host and allocate nothing;
 *   - one mpe_ctx may be used from one thread at a time.
 */
#ifndef MPE_H
#define MPE_H

#include <stddef.h>     /* size_t */
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPE_MAX_CAMERAS 32      /* camera sets are kept as 32-bit masks in the clustering kernel */
#define MPE_MAX_JOINTS 32       /* joint presence is a 32-bit mask per skeleton */
#define MPE_MAX_GAT_LAYERS 8
#define MPE_MAX_MLP_LAYERS 16

typedef enum {
    MPE_OK = 0,
    MPE_ERR_INVALID = -1,       /* bad argument / shape mismatch                      */
    MPE_ERR_CAPACITY = -2,      /* batch exceeds the capacity given to mpe_create     */
    MPE_ERR_STATE = -3,         /* weights not uploaded yet                           */
    MPE_ERR_HIP = -4,           /* a HIP runtime call failed                          */
    MPE_ERR_NOMEM = -5,
    MPE_ERR_UNSUPPORTED = -6    /* input the device-side parser leaves to the host parser */
} mpe_status;

typedef struct mpe_ctx mpe_ctx;

/* Static configuration = reference `parameters` (parameters.py:12-45) plus the calibration
 * globals the hot-path modules derive at import time (graph_generator.py:32-52,
 * pose_estimator_dataset_from_json.py:28-47).  Cameras are indexed by their position in
 * parameters.used_cameras_skeleton_matching (== used_cameras == camera_names for the
 * shipped presets). */
typedef struct {
    int32_t n_cameras;          /* V_cfg                                               */
    int32_t n_joints;           /* J = len(parameters.joint_list) (18)                 */
    int32_t image_width;        /* parameters.image_width                              */
    int32_t image_height;       /* parameters.image_height                             */
    int32_t numbers_per_joint;  /* parameters.numbers_per_joint (14)                   */
    int32_t min_views;          /* parameters.min_number_of_views                      */
    int32_t median_axis;        /* parameters.axes_3D['Y'][0]                          */
    uint32_t used_joint_mask;   /* bit j set for j in parameters.used_joints           */
    float threshold;            /* CLASSIFICATION_THRESHOLD (0.5)                      */
    float median_window;        /* 0.05 m (pose_estimator_utils.py:73)                 */
    /* capacities of one batch (workspace is sized from these at mpe_create) */
    int32_t max_frames;
    int32_t max_heads;          /* total 2D skeletons in a batch                       */
    int32_t max_edge_nodes;     /* total cross-camera skeleton pairs in a batch        */
    int32_t max_heads_per_frame;
    int32_t max_persons_per_frame;  /* Pcap, >= floor(max_heads_per_frame / min_views) */
    /* calibration, row-major, host pointers */
    const float *Kinv;          /* [V][9]  torch.inverse(camera_matrix) f32            */
    const float *K;             /* [V][9]  camera_matrix f32                           */
    const float *T_i;           /* [V][16] get_transform(cam,"root") cast to f32       */
    const double *P;            /* [V][12] get_transform("root",cam)[0:3,:] f64        */
    const double *dist;         /* [V][5]  k1,k2,p1,p2,k3 f64 (OpenCV order)           */
} mpe_config;

/* One batch of frames in packed, structure-of-arrays form.  A "head" is one 2D skeleton
 * with at least one joint; heads of a frame are numbered in the reference's order
 * (cameras in the frame dict's key order, then list order; graph_generator.py:583-601).
 * A "slot" is one camera of that dict order.  All pointers are device pointers. */
typedef struct {
    int32_t n_frames;
    int32_t n_heads;                 /* total heads in the batch                        */
    int32_t n_edge_nodes;            /* total edge-nodes (cross-slot head pairs)        */
    const int32_t *d_frame_head_off; /* [n_frames+1] exclusive prefix of heads          */
    const int32_t *d_frame_en_off;   /* [n_frames+1] exclusive prefix of edge-nodes     */
    const int32_t *d_slot_cam;       /* [n_frames][V] camera index of slot s, -1 unused */
    const int32_t *d_slot_n;         /* [n_frames][V] number of heads in slot s         */
    const int32_t *d_head_cam;       /* [n_heads] camera index                          */
    const uint32_t *d_joint_mask;    /* [n_heads] bit j: joint j present in the dict    */
    const uint32_t *d_tri_mask;      /* [n_heads] bit j: present and values[0] > 0      */
    const double *d_xy;              /* [n_heads][J][2] pixel x,y (values[1], values[2])*/
    const float *d_vp;               /* [n_heads][J][2] values[3] (valid), values[4]    */
    /* Optional EXPLICIT edge-node list, NULL = the implicit one of process_test (every cross-slot head pair once,
     * graph_generator.py:854-864).  [n_edge_nodes][2] frame-local head ids (h1, h2) of edge-node m, frames back to back
     * as d_frame_en_off says; edge-node X = (h1,h2) carries the edges (h1,X),(X,h1),(h2,X),(X,h2),(X,X) in that order
     * (add_edge_node_to_graph, :627-656).  This is the topology of MergedMultipleHumansDataset.process_training
     * (:672-810; mode 'test_generated' of test/sm_metrics_without_gt.py:108): heads grouped by person instead of by
     * camera slot and one edge-node per ORDERED head pair -- or any other pair list.  The heads may then come in any
     * order; d_slot_cam / d_slot_n are not read and may be NULL.  Per-frame limits, checked on the device and reported by
     * mpe_sync_status: heads <= max_heads_per_frame, edge-nodes <= the power of two >= max(512, max_heads_per_frame^2 / 2 + 1),
     * every pair inside its frame with h1 != h2 (else MPE_ERR_INVALID).  A process_training-style graph holds one edge-node per ORDERED
     * cross-camera head pair, up to H^2 (V - 1) / V of them: with V = 5 a frame of H heads fits while 0.8 H^2 <= that power of two
     * (H <= 35 at max_heads_per_frame = 40, whose capacity is 1024) -- a larger frame is reported (MPE_ERR_CAPACITY), never
     * truncated; raise max_heads_per_frame to make room.  Available on contexts with
     * max_heads_per_frame <= 1024 whose per-frame node ids fit 16 bits (MPE_ERR_UNSUPPORTED otherwise). */
    const int32_t *d_en_pair;
} mpe_batch;

/* ---- environment switches of the library (diagnostics; none is needed in production) -------------------------------------------
 * The list is FROZEN (round 5; round 6 added the two MPE_LATENCY_* switches with the small-batch launches): these are all the variables
 * csrc/ reads.  Read once per process unless marked "per call".
 *   kernel selection, each a cross-check path the GPU suite is run under (tools/run_switch_matrix.sh):
 *     MPE_SKINNY_WAVES=<n>        16 x 16 tiles up to which the wave-per-tile GEMM kernels run (default 1024; 0 = tile kernels always)
 *     MPE_GEMM_NARROW=0           narrow outputs (<= 16 / 64 features) on the tile kernels as well
 *     MPE_GAT_ACC64_MINK=<k>      GAT launches with K > k get f64 running sums (default 512: fc2 of layer 0; 0 = never -- that CHANGES the
 *                                 numerics: the 2e-5 score bound rests on these sums, profiles/r05_switch_matrix.txt; not a cross-check path)
 *     MPE_L0_GROUPED=0            layer-0 fc1 dense over the whole 902-wide row instead of per camera block
 *     MPE_NO_COEF_EPILOGUE        (per call) attention coefficients from k_attn_coef instead of the fc2 epilogue
 *     MPE_NO_FUSED_ATTENTION      (per call) the general attention kernels for every frame size
 *     MPE_FUSED_NO_OVERLAP        (per call) plain staging in k_gat_fused
 *     MPE_NO_HEAD_SRC_TABLE       (per call) in-edge sources derived in the kernels instead of read from the per-batch table
 *     MPE_CLUSTER_KERNEL=wave|block|lds|big   (per call) clustering kernel
 *     MPE_HALF_VEC=4              fp16 rows of the general attention kernels read 4 columns per thread instead of 8
 *     MPE_JSON_WGS=<n>            workgroups of the device-side JSON walk
 *     MPE_LATENCY_PATH=0          (per call) batches of at most 16 frames through the batch path's own small-batch kernels instead of the
 *                                 latency launches of csrc/lat.hip / gat.hip (k_lat_l0a, k_lat_gemm, k_lat_attention; persons' prefix and
 *                                 decode folded into their neighbours): same bits either way (tests/test_gpu_latency.py)
 *   host packer (threads, timing prints): MPE_PACK_THREADS, MPE_SCAN_THREADS, MPE_SCAN_CHUNK_KB, MPE_PACK_NO_SIMD, MPE_PACK_TIMING,
 *     MPE_STAGE_TIMING
 * Gone since round 5 (their code left the library): MPE_GEMM_TUNE, MPE_GEMM_BN, MPE_GEMM_LOADER, MPE_SB_GAT_MW, MPE_SB_PERS,
 * MPE_SB_LDS_PAD, and the compile-time ablation switches MPE_EXP / MPE_SBEXP.  `make exp EXPFLAGS=-DMPE_SB_CLOCK` builds the one
 * diagnostic variant left: in-kernel clock stamps of the split-bf16 tile kernel (tools/sb_clock_probe.py).
 * Round 6 built and removed two experiments with their switches (MPE_LATENCY_MLP, MPE_MLP_CHAIN, mpe_linear flag bit 6): the records are
 * profiles/r06_mlp_fp32_weights_experiment.txt and profiles/r06_mlp_chain_experiment.txt, the code is in commit 102284c. */

/* ---- lifetime ------------------------------------------------------------------------ */
int mpe_create(const mpe_config *cfg, mpe_ctx **out);
void mpe_destroy(mpe_ctx *ctx);
const char *mpe_last_error(const mpe_ctx *ctx);
const char *mpe_version(void);

/* ---- weights (host pointers, copied once; ctx owns padded device copies) ---------------
 * GAT2 state-dict tensors of layer l (gat2.py:18-48): fc1.weight [in][in], fc1.bias [in],
 * fc2.weight [heads*out][in], fc2.bias [heads*out], attn_l / attn_r [heads][out]. */
int mpe_set_gat_params(mpe_ctx *ctx, int32_t n_layers, float alpha, float hidden_slope);
int mpe_set_gat_layer(mpe_ctx *ctx, int32_t layer, int32_t in_dim, int32_t heads, int32_t out_dim,
                      const float *fc1_w, const float *fc1_b, const float *fc2_w, const float *fc2_b,
                      const float *attn_l, const float *attn_r);
/* PoseEstimatorMLP (utils/mlp.py:8-28): layer l = Linear(in,out) [+ LeakyReLU(slope)] */
int mpe_set_mlp_params(mpe_ctx *ctx, int32_t n_layers, float slope);
int mpe_set_mlp_layer(mpe_ctx *ctx, int32_t layer, int32_t in_dim, int32_t out_dim,
                      const float *w, const float *b);

/* Arithmetic of the GEMMs.  0 = one fp32 MFMA chain over the whole K; 1 = fp32 MFMA, every 32-deep K stage flushed into f64
 * running sums, so a dot product carries about one rounding, like a blocked CPU sgemm.  MLP mode 3 (the MLP DEFAULT since round
 * 4) = the same accuracy class on the bf16 matrix pipe: every fp32 operand is taken as the exact sum of three bf16 numbers, the
 * six significant partial products run on v_mfma_f32_16x16x32_bf16 with fp32 accumulators flushed into f64 sums every second
 * stage (csrc/gemm_sb16.hip; measured error against exactly evaluated dot products: that of mode 1 or below, 1.35x faster
 * launches).  MLP mode 4 = mode 3 with an f64 flush after EVERY stage: the maximum-accuracy form (rms error of a launch 0.13-0.18
 * instead of 0.24-0.26 ulp of its output scale; the MLP launches take ~7 % longer).  MLP mode 5 = the f64-evaluated network: exact
 * fp32 x fp32 products accumulated in f64 over the whole K on the f64 matrix pipe (v_mfma_f64_16x16x4_f64, csrc/gemm_f64.hip), bias and
 * LeakyReLU in f64, fp32 rounding between layers; several times slower, for parity work.  Its LeakyReLU multiplies by the network's
 * parameter as a double: the seven-digit decimal that rounds to the fp32 slope given to mpe_set_mlp_params (0.1 for
 * nn.LeakyReLU(0.1), utils/mlp.py:11), or that fp32 value itself when no such decimal exists -- the network, not a replay of the
 * reference's fp32 kernel (which multiplies by (float)0.1).  Defaults: GAT 4 (below), MLP 3 (the MLP's K is up to 3072 and its 3D output is compared with the reference at the
 * micrometre level: DESIGN.md section 5; mode 1 stays selectable).  MLP mode 2 is the
 * reduced-precision variant of BASELINE.json configs[4]: weights and staged activations in
 * bf16, v_mfma_f32_16x16x32_bf16 with fp32 accumulation (~3 significant digits; not parity).
 * GAT mode 2 is the other half of that config: fc1/fc2 on the bf16 MFMA and the transformed
 * features (ft2) stored as fp16 rows for the attention stage (coefficients, softmax and sums
 * stay fp32; the layer-0 edge-node constants stay fp32).  GAT mode 3 is configs[4] as BASELINE.json words it
 * ("fp16 GATv2 attention + bf16 MLP" with MLP mode 2): only the ft2 rows are fp16 -- fc2 stores them from its fp32
 * results, the attention coefficients a1/a2 still come from the fp32 values in the GEMM epilogue -- and fc1/fc2 stay
 * on the fp32 MFMA.  GAT modes 4 (the GAT DEFAULT since round 4), 5 and 6 are modes 0, 1 and 3 with fc1 / fc2 of the layers
 * >= 1 in the split-bf16 form of MLP mode 3 (fp32-accurate: rms error of a launch at or below the fp32 MFMA chain's, 1.5x
 * faster launches; without f64 sums where mode 0 has none); layer 0's gathered launches stay on the fp32 MFMA.  In mode 6 the
 * fc2 launches store their fp16 rows from the split tile kernel when the batch is large enough for it (more than
 * MPE_SKINNY_WAVES = 1024 16 x 16 tiles, more than one 16-wide output tile); every other fp16-row launch (small batches, the
 * 1-wide last layer, f64-sum launches) stays on the fp32 MFMA, whose kernels all store fp16 rows. */
int mpe_set_precision(mpe_ctx *ctx, int32_t gat_acc64, int32_t mlp_acc64);

/* ---- batch entry points ---------------------------------------------------------------
 * mpe_match_batch replaces, per frame: MergedMultipleHumansDataset(mode='test', alt='3')
 * (graph_generator.py:813-876), GAT2.forward (gat2.py:137-149) and
 * get_person_proposal_from_network_output (skeleton_matching_utils.py:12-132).
 *   d_scores   [n_edge_nodes]  sigmoid output of every edge-node (may be NULL)
 *   d_persons  [n_frames][Pcap][V] frame-local head id per camera, -1 = None
 *   d_n_persons[n_frames]
 * A batch of zero frames is accepted by every batch entry point and does nothing. */
int mpe_match_batch(mpe_ctx *ctx, void *stream, const mpe_batch *b,
                    float *d_scores, int32_t *d_persons, int32_t *d_n_persons);

/* 3D stage A: PoseEstimatorDataset dict branch (pose_estimator_dataset_from_json.py:237-298,
 * incl. get_3D_from_triangulation :63-101) + PoseEstimatorMLP + x10 decode
 * (metrics_from_model.py:243-294).
 *   d_poses [n_frames][Pcap][J][3] f32 metres, d_valid [n_frames][Pcap] 1 = person row kept
 * Small batches (<= 16 frames): when mpe_match_batch has just run on the SAME batch arrays (same d_xy pointer, same head and edge-node
 * counts), it has already solved every cross-camera skeleton pair of the batch beside its clustering launch, and the row kernel here
 * fetches them instead of solving (same function, same arguments: same bits).  The arrays must not change in between -- which the
 * call order of the pipeline implies anyway. */
int mpe_mlp3d_batch(mpe_ctx *ctx, void *stream, const mpe_batch *b,
                    const int32_t *d_persons, const int32_t *d_n_persons,
                    float *d_poses, uint8_t *d_valid);

/* 3D stage B: caller gather + triangulate (metrics_from_triangulation.py:234-272,
 * pose_estimator_utils.py:52-75).
 *   d_poses [n_frames][Pcap][J][3] f64, d_joint_valid [n_frames][Pcap][J] 1 = joint emitted
 *   flags bit 0: emit every triangulated joint (what `triangulate` itself returns); otherwise
 *   joints outside parameters.used_joints come back as zeros, as the caller's copy does.
 *   flags bit 1: gather only joints whose values[0] (the joint id) is > 0, as the caller in
 *   test/reprojection_error.py:296-300 does (joint 0 is then never triangulated). */
int mpe_triangulate_batch(mpe_ctx *ctx, void *stream, const mpe_batch *b,
                          const int32_t *d_persons, const int32_t *d_n_persons,
                          double *d_poses, uint8_t *d_joint_valid, uint32_t flags);

/* ---- stage-level entry points (parity tests, Python mirrors of single reference symbols) */
/* C[M][N] = act(A[M][K] * W[N][K]^T + bias): nn.Linear (+ LeakyReLU when slope_on != 0).
 * Row strides in elements; A and C device pointers, W/bias device pointers prepared by
 * mpe_upload_linear (zero padded).  d_m, if not NULL, overrides M with a device-side count.
 * flags: bit 0 = apply LeakyReLU(slope), bit 1 = f64 running sums (see mpe_set_precision), bit 2 = the split-bf16 form
 * (csrc/gemm_sb16.hip; the planes are made for the call), with bit 3 = without its f64 sums and bit 4 = an f64 flush per K stage;
 * bit 5 = the f64 matrix-pipe form (csrc/gemm_f64.hip). */
int mpe_upload_linear(mpe_ctx *ctx, const float *w, const float *b, int32_t out_dim, int32_t in_dim,
                      float **d_w, float **d_b, int32_t *ldw);
int mpe_free_device(mpe_ctx *ctx, void *d_ptr);
int mpe_linear(mpe_ctx *ctx, void *stream, const float *d_a, int32_t lda, const float *d_w, int32_t ldw,
               const float *d_bias, float *d_c, int32_t ldc, int32_t m, const int32_t *d_m,
               int32_t n, int32_t k, int32_t flags, float slope);

/* HumanGraphFromView.initializeWithAlternative3 (graph_generator.py:444-508): the J*10
 * non-zero block of every head row: d_feat [n_heads][J][10]. */
int mpe_head_features(mpe_ctx *ctx, void *stream, const mpe_batch *b, float *d_feat);
/* graph.ndata['h'] of ONE frame's graph as the reference builds it (graph_generator.py:444-508, 629-631): the dense
 * [n_heads + n_edge_nodes][ld >= 2 + V*J*10] rows in node order -- head rows (column 0 = 1, the camera's J*10 block),
 * then the edge-node rows (one-hot at column 1); pad columns are zeroed.  For callers that ask for the matrix itself. */
int mpe_dense_rows(mpe_ctx *ctx, void *stream, const mpe_batch *b, float *d_rows, int32_t ld);

/* GAT2.forward over the batch (gat2.py:137-149).  d_feats == NULL: node rows are featurised
 * on the device from the packed skeletons (the production path; edge-node rows are constant
 * and de-duplicated).  d_feats != NULL: caller-provided dense [n_nodes][ld_feats] rows in
 * node order (frame by frame: heads, then edge-nodes), as GAT2.forward(inputs, g) receives.
 * d_scores_en [n_edge_nodes]; optional d_scores_heads [n_heads] (the reference also evaluates
 * the last layer at head nodes; unused downstream). */
int mpe_gat_forward(mpe_ctx *ctx, void *stream, const mpe_batch *b, const float *d_feats, int32_t ld_feats,
                    float *d_scores_en, float *d_scores_heads);
/* Activation of the last GAT layer: 1 = sigmoid (final_activation = nn.Sigmoid(), the deployed
 * model, train_skeleton_matching.py:34), 2 = identity (final_activation = None, gat2.py:146-148). */
int mpe_set_gat_output(mpe_ctx *ctx, int32_t mode);

/* One GraphAttention2 layer (gat2.py:50-76) plus the activation GAT2.forward applies to its
 * flattened output (:141-147).  d_in [n_nodes][ld_in] holds the layer's input rows in node order
 * (frame by frame: heads, then edge-nodes; in_dim columns used), d_out [n_nodes][ld_out] receives
 * heads*out_dim columns.  activation: 0 = LeakyReLU(hidden slope), 1 = sigmoid, 2 = none.
 * Layer 0 takes dense F-wide rows (no de-duplication of the constant edge-node rows). */
int mpe_gat_layer(mpe_ctx *ctx, void *stream, const mpe_batch *b, int32_t layer, const float *d_in, int32_t ld_in,
                  float *d_out, int32_t ld_out, int32_t activation);

/* The graph half of a layer only -- what the reference delegates to DGL (gat2.py:57-66, 78-88):
 * a1/a2 = <ft2, attn_l/r> (the two torch.bmm), apply_edges(LeakyReLU(a1[src] + a2[dst])),
 * edge_softmax over the in-edges of every destination, update_all(u_mul_e, sum).
 * d_ft2 [n_nodes][ld_ft2] = fc2 output (heads*out_dim columns) -> d_out [n_nodes][ld_out], no
 * activation.  Uses attn_l / attn_r of `layer`. */
int mpe_edge_softmax_aggregate(mpe_ctx *ctx, void *stream, const mpe_batch *b, int32_t layer, const float *d_ft2,
                               int32_t ld_ft2, float *d_out, int32_t ld_out);

/* Per-frame capacity.  The host side of this ABI sees batch totals only; a frame that holds
 * more than mpe_config.max_heads_per_frame skeletons (and therefore possibly more edge-nodes
 * than the per-frame LDS / scratch budget) is detected on the device: its scores come back as
 * zeros, it yields n_persons = 0, and a sticky status bit is raised.  mpe_sync_status
 * synchronises `stream`, returns MPE_ERR_CAPACITY if any batch since the last call contained
 * such a frame (MPE_ERR_INVALID if an explicit edge-node list held a bad pair; MPE_OK otherwise) and clears the bits.  Limits that mpe_create enforces:
 * max_heads_per_frame < 32768, n_cameras <= 32, n_joints <= 32, attention heads <= 16. */
int mpe_sync_status(mpe_ctx *ctx, void *stream);
/* The same in two halves: mpe_status_queue orders the read-back (into page-locked memory) and the reset of the word behind everything
 * queued on `stream` so far and returns at once; mpe_status_wait synchronises `stream` and reports what the read-back saw (it queues
 * one itself if none is pending).  Bits raised by work queued after mpe_status_queue are reported by the next call. */
int mpe_status_queue(mpe_ctx *ctx, void *stream);
int mpe_status_wait(mpe_ctx *ctx, void *stream);

/* CLASSIFICATION_THRESHOLD of get_person_proposal_from_network_output (default from mpe_config) */
int mpe_set_threshold(mpe_ctx *ctx, float threshold);

/* get_person_proposal_from_network_output on caller-provided scores. */
int mpe_cluster_batch(mpe_ctx *ctx, void *stream, const mpe_batch *b, const float *d_scores,
                      int32_t *d_persons, int32_t *d_n_persons);

/* MLP input rows only: d_rows [n_frames*Pcap][ld_rows] f32 (row r = frame*Pcap + p). */
int mpe_mlp_input_rows(mpe_ctx *ctx, void *stream, const mpe_batch *b, const int32_t *d_persons,
                       const int32_t *d_n_persons, float *d_rows, int32_t ld_rows, uint8_t *d_valid);
/* PoseEstimatorMLP.forward on d_x [m][ld_x] -> d_y [m][out_dim] (no x10). */
int mpe_mlp_forward(mpe_ctx *ctx, void *stream, const float *d_x, int32_t ld_x, int32_t m,
                    float *d_y, int32_t ld_y);

/* cv2.undistortPoints + cv2.triangulatePoints for explicit pairs (pose_estimator_utils.py:63-67):
 * d_pts [n][2][2] pixel points, d_cams [n][2] camera indices -> d_out [n][3] f64. */
int mpe_dlt_pairs(mpe_ctx *ctx, void *stream, const double *d_pts, const int32_t *d_cams,
                  int32_t n, double *d_out);

/* ---- host-side packer (no GPU involved) ---------------------------------------------------
 * Frame JSON in the reference's wire format (list of frames; frame = {camera: ["<JSON text of
 * the skeleton list>", timestamp, 'no_image', bodies_3D]}; skeleton = {joint id: [id, x, y,
 * valid, prob], optional "ID"}; panoptic_conversor/get_joints_from_panoptic_model_multi.py:
 * 231-236,281,287) -> the host arrays of an mpe_batch, in the reference's head order
 * (graph_generator.py:573-605).  Replaces json.load + json.loads per camera + the Python
 * loops of load_people_view_graph.  Frames frame_start, frame_start+frame_step, ... (at most
 * max_frames, 0 = all) are parsed by n_threads threads (0 = all cores). */
typedef struct mpe_packed mpe_packed;
typedef struct {
    int32_t n_frames, n_heads, n_edge_nodes, n_cameras, n_joints;
    const int32_t *frame_head_off, *frame_en_off, *slot_cam, *slot_n, *head_cam, *skeleton_index;
    const uint32_t *joint_mask, *tri_mask;
    const double *xy;
    const float *vp;
} mpe_packed_arrays;
int mpe_pack_json(const char *json, size_t len, const char *const *camera_names, int32_t n_cameras,
                  int32_t n_joints, int32_t frame_start, int32_t frame_step, int32_t max_frames,
                  int32_t n_threads, mpe_packed **out);
int mpe_packed_view(const mpe_packed *pk, mpe_packed_arrays *view);
/* The same parse straight into caller-provided arrays -- e.g. views of ONE page-locked buffer
 * that then travels to the device in a single copy (no intermediate host copies).  Capacities
 * max_frames / max_heads are the caller's array sizes: frame_head_off / frame_en_off hold
 * max_frames + 1 entries, slot_cam / slot_n max_frames * n_cameras, the per-head arrays max_heads
 * (xy / vp max_heads * n_joints * 2).  At most min(max_frames argument, dst->max_frames) frames are
 * parsed; MPE_ERR_CAPACITY if their skeletons exceed dst->max_heads. */
typedef struct {
    int32_t max_frames, max_heads;
    int32_t *frame_head_off, *frame_en_off, *slot_cam, *slot_n, *head_cam, *skeleton_index;
    uint32_t *joint_mask, *tri_mask;
    double *xy;
    float *vp;
} mpe_pack_dst;
int mpe_pack_json_into(const char *json, size_t len, const char *const *camera_names, int32_t n_cameras,
                       int32_t n_joints, int32_t frame_start, int32_t frame_step, int32_t max_frames,
                       int32_t n_threads, const mpe_pack_dst *dst, int32_t *n_frames, int32_t *n_heads,
                       int32_t *n_edge_nodes);
/* ONE frame as the reference's per-frame callers hold it after json.load (test/metrics_from_model.py:182-199; what
 * MergedMultipleHumansDataset(mode='test') receives, graph_generator.py:813-876): per configured camera the TEXT of its skeleton list
 * (frame[cam][0]), in the frame dict's order; cams[i] = that camera's index in the configured list.  Same grammar, head order and numbers
 * as mpe_pack_json_into gives for the document [frame]; dst as there with max_frames >= 1.  extents (NULL or dst->max_heads * 2 entries)
 * receives per head the byte offsets [begin, end) of its skeleton object inside its camera text (the per-frame callers hand single
 * skeletons on as text, metrics_from_model.py:250-252).  MPE_ERR_INVALID with mpe_pack_last_error for text the packer declines,
 * MPE_ERR_CAPACITY beyond dst->max_heads. */
int mpe_pack_views_into(const char *const *texts, const size_t *lens, const int32_t *cams, int32_t n_views, int32_t n_cameras,
                        int32_t n_joints, const mpe_pack_dst *dst, int32_t *n_heads, int32_t *n_edge_nodes, int32_t *extents);
/* A document that is consumed in windows (frame_start, max_frames): the index scans the document for
 * its frame extents ONCE, in a background thread it starts at creation, ahead of the windows; a window
 * is parsed as its frames are published (n_threads = 0: as many workers as the process may use: the
 * smaller of the hardware threads, the affinity mask and the cgroup CPU quota; MPE_PACK_THREADS
 * overrides).  A malformed document fails the first window that reaches the damage.  `json` must stay
 * valid and unchanged while the index lives; mpe_json_index_free joins the scan thread.  One index may
 * be used from one caller thread at a time. */
typedef struct mpe_json_index mpe_json_index;
int mpe_json_index_create(const char *json, size_t len, mpe_json_index **out);
void mpe_json_index_free(mpe_json_index *ix);
int mpe_pack_indexed_into(mpe_json_index *ix, const char *const *camera_names, int32_t n_cameras, int32_t n_joints,
                          int32_t frame_start, int32_t frame_step, int32_t max_frames, int32_t n_threads,
                          const mpe_pack_dst *dst, int32_t *n_frames, int32_t *n_heads, int32_t *n_edge_nodes);
void mpe_packed_free(mpe_packed *pk);
const char *mpe_pack_last_error(void);

/* ---- device-side second-level parse (csrc/jsonparse.hip) -----------------------------
 * The host keeps the first level of the wire format only: mpe_json_stage_window walks the frames of a window
 * (extents from the index) down to the camera entries, records for every configured camera the extent of the
 * STRING that holds its skeleton list (frame dict key order: graph_generator.py:583-601) and copies those
 * strings, 16-byte aligned, into `text_dst` -- one page-locked buffer that travels to the device in one copy.
 * mpe_json_parse_device parses the strings there into the arrays of `out` (device pointers; capacities as for
 * mpe_pack_dst) and `d_skeleton_index`, replacing json.loads per camera + load_people_view_graph
 * (graph_generator.py:573-605) like the host packer, whose arrays it reproduces bit for bit.
 *   skeletons_per_string_cap  rows of the staging arena per string (a string with more skeleton objects goes to the host)
 *   d_scratch  mpe_json_scratch_bytes(n_entries, skeletons_per_string_cap, n_joints) bytes of device memory
 *   d_totals   [4]: n_heads, n_edge_nodes, status (0 ok, bit 0: a string needs the host parser, bit 1: more
 *              heads than head_cap), largest number of heads in a frame
 * MPE_ERR_UNSUPPORTED from the staging call and a non-zero status both mean: pack this window with
 * mpe_pack_indexed_into (the host parser defines the accepted language and the error messages). */
typedef struct {
    int32_t frame, cam;            /* frame of the window, camera index (position in camera_names) */
    uint32_t begin, end;           /* the string's body in the staged text                         */
} mpe_json_entry;
int mpe_json_stage_window(mpe_json_index *ix, const char *const *camera_names, int32_t n_cameras, int32_t frame_start,
                          int32_t frame_step, int32_t max_frames, int32_t n_threads, char *text_dst, size_t text_cap,
                          mpe_json_entry *entries, int32_t entry_cap, int32_t *frame_entry_off, int32_t *n_frames,
                          int32_t *n_entries, size_t *text_bytes);
size_t mpe_json_scratch_bytes(int32_t n_entries_cap, int32_t skeletons_per_string_cap, int32_t n_joints);
int mpe_json_parse_device(mpe_ctx *ctx, void *stream, const char *d_text, const mpe_json_entry *d_entries,
                          const int32_t *d_frame_entry_off, int32_t n_entries, int32_t n_frames, int32_t head_cap,
                          int32_t skeletons_per_string_cap, void *d_scratch, size_t scratch_bytes, const mpe_batch *out,
                          int32_t *d_skeleton_index, int32_t *d_totals);


/* Timing probe for bench.py: average duration (ms) of the dominant GEMM launches measured
 * with HIP events on the launch stream during the last mpe_match_batch / mpe_mlp3d_batch
 * when profiling is enabled; see bench.py. */
int mpe_profile_enable(mpe_ctx *ctx, int32_t on);      /* 1 = on, records cleared; 2 = on, records kept (resume); 0 = off (records kept until read) */
int mpe_profile_read(mpe_ctx *ctx, double *gemm_ms, double *gemm_flop, int64_t *gemm_launches,
                     double *total_ms);
/* mpe_profile_read reports the launches on the fp32 MFMA (the dominant kernel, k_linear_dma); the split-bf16 launches of the
 * same records (MLP mode 3: k_linear_sb*, fp32-equivalent FLOPs = 2 M N K, executed on the bf16 MFMA as six products) are kept
 * apart and read here, after mpe_profile_read. */
int mpe_profile_read_split(mpe_ctx *ctx, double *ms, double *flop, int64_t *launches);
/* ... and the plain bf16 launches of the reduced-precision modes (configs[4]): one bf16 product per product, priced against the bf16 peak */
int mpe_profile_read_bf16(mpe_ctx *ctx, double *ms, double *flop, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* MPE_H */
