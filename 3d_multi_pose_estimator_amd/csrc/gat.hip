// Graph side of the skeleton-matching network, frame-batched and with implicit topology.
//
// The reference materialises, per frame, a DGL graph with a dense N x 902 feature matrix
// (graph_generator.py:813-876) and runs apply_edges / edge_softmax / update_all on it
// (gat2.py:57-66).  For graph alternative '3' the topology is a pure function of the
// per-camera skeleton counts, so no graph object exists here: the in-edges of a node follow
// from slot_n[frame][V] (k_topology writes the pair list of the edge-nodes, k_head_sources the
// in-edge sources of the heads, once per batch; every layer's attention kernels read those):
//
//   head h (slot s, index i):   (h,h), then every edge-node X that pairs h with a head of
//                               another slot, in ascending X  (edge ids are created per
//                               edge-node in ascending order, graph_generator.py:632-651)
//   edge-node X = (h1,h2):      (h1,X), (h2,X), (X,X)
//
// Sums over in-edges run in that order (= DGL's edge order for a destination).
#include <cstdlib>

#include "mpe_internal.h"
#include "sb_common.h"

namespace mpe {

// ---------------------------------------------------------------------------------------
// topology tables: node_off[f], head_frame[h], en_frame[m], en_pair[m] = (h1,h2) frame-local
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void topology_body(int f, int n_frames, int V, const int32_t *__restrict__ head_off,
                                              const int32_t *__restrict__ en_off, const int32_t *__restrict__ slot_n,
                                              int32_t *__restrict__ node_off, int32_t *__restrict__ head_frame,
                                              int32_t *__restrict__ en_frame, int32_t *__restrict__ en_pair, int hmax,
                                              int32_t *__restrict__ status) {
    __shared__ int s_n[MPE_MAX_CAMERAS], s_start[MPE_MAX_CAMERAS];
    const int h0 = head_off[f], H = head_off[f + 1] - h0;
    const int e0 = en_off[f], M = en_off[f + 1] - e0;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int s = 0; s < V; ++s) {
            s_n[s] = slot_n[f * V + s];
            s_start[s] = acc;
            acc += s_n[s];
        }
        node_off[f] = h0 + e0;
        // per-frame capacity (the host side of the C ABI sees batch totals only): such a frame is
        // skipped by the attention and clustering kernels and reported by mpe_sync_status
        if (H > hmax && status) atomicOr(status, 1);
        if (f == n_frames - 1) node_off[n_frames] = head_off[n_frames] + en_off[n_frames];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += blockDim.x) head_frame[h0 + i] = f;
    for (int i = threadIdx.x; i < M; i += blockDim.x) en_frame[e0 + i] = f;
    int base = 0;
    for (int a = 0; a < V; ++a)
        for (int b = a + 1; b < V; ++b) {
            const int na = s_n[a], nb = s_n[b], cnt = na * nb;
            for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
                const int i = t / nb, j = t - i * nb;
                en_pair[2 * (size_t)(e0 + base + t) + 0] = s_start[a] + i;
                en_pair[2 * (size_t)(e0 + base + t) + 1] = s_start[b] + j;
            }
            base += cnt;
        }
}

__global__ void k_topology(int n_frames, int V, const int32_t *__restrict__ head_off,
                           const int32_t *__restrict__ en_off, const int32_t *__restrict__ slot_n,
                           int32_t *__restrict__ node_off, int32_t *__restrict__ head_frame,
                           int32_t *__restrict__ en_frame, int32_t *__restrict__ en_pair, int hmax,
                           int32_t *__restrict__ status) {
    topology_body(blockIdx.x, n_frames, V, head_off, en_off, slot_n, node_off, head_frame, en_frame, en_pair, hmax, status);
}

__global__ void k_head_sources(int V, int hmax, const int32_t *__restrict__ head_off,
                               const int32_t *__restrict__ slot_n, uint16_t *__restrict__ head_src);

// Per-frame table of the in-edge sources of the heads (k_head_sources, [hmax][hmax + 1] uint16 per
// frame): the attention kernels read it instead of deriving the sources again in every workgroup of
// every layer.  Frames whose node ids do not fit 16 bits get none (the kernels then fall back to the
// arithmetic).
size_t head_src_entries(int max_heads_per_frame, int V) {
    const size_t h = (size_t)max_heads_per_frame;
    const size_t nodes = h + h * h * (size_t)(V - 1) / (2 * (size_t)V) + 1;
    return nodes <= 65535 ? h * (h + 1) : 0;
}

// Explicit edge-node lists (mpe_batch::d_en_pair): the graphs of process_training (graph_generator.py:672-810, one
// edge-node per ORDERED head pair, heads grouped by person) or any other pair list.  One workgroup per frame: the
// caller's pairs are checked and copied into the context's table (a pair outside [0, H) or with h1 == h2 is replaced
// by a memory-safe one and raises status bit 1 << 1), then the in-edge sources of every head are listed in ascending
// edge-node order -- head h: (h,h), then every X whose pair holds h; that is ascending edge id, the reference's
// creation order (:632-651) -- into the same per-frame table the implicit mode uses, here [hmax][deg_cap].  A head
// with more than deg_cap - 1 edge-nodes (only possible with repeated pairs) loses the surplus and raises bit 1 << 1;
// a frame with more than hmax heads or m_cap edge-nodes gets an all-0xFFFF table and raises bit 0 (skipped downstream).
constexpr int XT_CHUNK = 2048;
__global__ __launch_bounds__(256) void k_topology_explicit(int n_frames, const int32_t *__restrict__ head_off,
                                                           const int32_t *__restrict__ en_off,
                                                           const int32_t *__restrict__ pairs_in,
                                                           int32_t *__restrict__ node_off, int32_t *__restrict__ head_frame,
                                                           int32_t *__restrict__ en_frame, int32_t *__restrict__ en_pair,
                                                           uint16_t *__restrict__ head_src, int hmax, int deg_cap, int m_cap,
                                                           int32_t *__restrict__ status) {
    __shared__ uint32_t s_pr[XT_CHUNK];
    const int f = blockIdx.x, t = threadIdx.x;
    const int h0 = head_off[f], H = head_off[f + 1] - h0;
    const int e0 = en_off[f], M = en_off[f + 1] - e0;
    const bool ok = H <= hmax && M <= m_cap;
    if (t == 0) {
        node_off[f] = h0 + e0;
        if (f == n_frames - 1) node_off[n_frames] = head_off[n_frames] + en_off[n_frames];
        if (!ok && status) atomicOr(status, 1);
    }
    for (int i = t; i < H; i += blockDim.x) head_frame[h0 + i] = f;
    int flag = 0;
    uint16_t *tab = head_src + (size_t)f * hmax * deg_cap;
    // one thread per head row of the table; deg = entries written so far
    int deg_a[4];                                   // rows t, t + 256, ... (hmax <= 1024 = 4 x 256 rows in explicit mode: checked by the caller)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = t + r * 256;
        deg_a[r] = 0;
        if (ok && h < H && h < hmax) {
            tab[(size_t)h * deg_cap] = (uint16_t)h;
            deg_a[r] = 1;
        }
    }
    for (int c0 = 0; c0 < M; c0 += XT_CHUNK) {
        const int cn = min(XT_CHUNK, M - c0);
        __syncthreads();
        for (int i = t; i < cn; i += blockDim.x) {
            const size_t m = (size_t)e0 + c0 + i;
            int h1 = pairs_in[2 * m], h2 = pairs_in[2 * m + 1];
            if (h1 < 0 || h1 >= H || h2 < 0 || h2 >= H || h1 == h2) {
                flag = 2;
                h1 = 0;
                h2 = H > 1 ? 1 : 0;
            }
            en_frame[m] = f;
            en_pair[2 * m] = h1;
            en_pair[2 * m + 1] = h2;
            s_pr[i] = ((uint32_t)h1 << 16) | (uint32_t)h2;
        }
        __syncthreads();
        if (!ok) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = t + r * 256;
            if (h >= H || h >= hmax) continue;
            uint16_t *row = tab + (size_t)h * deg_cap;
            int deg = deg_a[r];
            for (int i = 0; i < cn; ++i) {
                const uint32_t pr = s_pr[i];
                if ((int)(pr >> 16) == h || (int)(pr & 0xFFFFu) == h) {
                    if (deg < deg_cap) row[deg++] = (uint16_t)(H + c0 + i);
                    else flag = 2;
                }
            }
            deg_a[r] = deg;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = t + r * 256;
        if (h >= hmax) continue;
        uint16_t *row = tab + (size_t)h * deg_cap;
        for (int e = deg_a[r]; e < deg_cap; ++e) row[e] = 0xFFFF;
    }
    if (flag && status) atomicOr(status, flag);
}

hipError_t launch_topology(hipStream_t s, const mpe_batch &b, int V, int32_t *node_off, int32_t *head_frame,
                           int32_t *en_frame, int32_t *en_pair, int max_heads_per_frame, int32_t *status,
                           uint16_t *head_src, int x_deg_cap, int x_m_cap) {
    if (b.n_frames <= 0) return hipSuccess;
    if (b.d_en_pair) {
        if (!head_src || x_deg_cap <= 0 || max_heads_per_frame > 1024) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_topology_explicit, dim3(b.n_frames), dim3(256), 0, s, b.n_frames, b.d_frame_head_off,
                           b.d_frame_en_off, b.d_en_pair, node_off, head_frame, en_frame, en_pair, head_src,
                           max_heads_per_frame, x_deg_cap, x_m_cap, status);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_topology, dim3(b.n_frames), dim3(128), 0, s, b.n_frames, V, b.d_frame_head_off,
                       b.d_frame_en_off, b.d_slot_n, node_off, head_frame, en_frame, en_pair, max_heads_per_frame,
                       status);
    if (head_src && head_src_entries(max_heads_per_frame, V))
        hipLaunchKernelGGL(k_head_sources, dim3(b.n_frames), dim3(256), (size_t)max_heads_per_frame * sizeof(int), s, V,
                           max_heads_per_frame, b.d_frame_head_off, b.d_slot_n, head_src);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// head featurisation (graph_generator.py:444-508)
// ---------------------------------------------------------------------------------------
// Per present joint: [i, j, valid, prob, cx, cy, cz, vx, vy, vz] with
//   i = (x - W/2)/(W/2), j = (H/2 - y)/(H/2) evaluated in f64 and rounded to f32,
//   c = camera centre (T_i @ [0,0,0,1]), v = (T_i @ [Kinv @ [x,y,1]; 0])[0:3] in f32.
// dense = true writes the whole zero-padded F-wide row (col 0 = 1, own-camera block at
// 2 + cam*J*10); dense = false writes only the J*10 block.
// the ten numbers of joint j of a head (zeros when the skeleton does not hold it)
__device__ __forceinline__ void head_joint_features(const DevCfg *__restrict__ cfg, int cam, uint32_t mask, int j, const double *__restrict__ pxy,
                                                    const float *__restrict__ pvp, float *o) {
#pragma clang fp contract(off)
    if (mask >> j & 1u) {
        const double x = pxy[0], y = pxy[1];
        const double hw = cfg->W / 2.0, hh = cfg->H / 2.0;
        o[0] = (float)((x - hw) / hw);
        o[1] = (float)((hh - y) / hh);
        o[2] = pvp[0];
        o[3] = pvp[1];
        const float *Ki = cfg->Kinv[cam];
        const float *T = cfg->T_i[cam];
        const float px = (float)x, py = (float)y;
        float pix[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float a = Ki[r * 3 + 0] * px;
            a = __builtin_fmaf(Ki[r * 3 + 1], py, a);
            a = __builtin_fmaf(Ki[r * 3 + 2], 1.0f, a);
            pix[r] = a;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o[4 + r] = T[r * 4 + 3];      // T_i @ [0,0,0,1]
            float a = T[r * 4 + 0] * pix[0];
            a = __builtin_fmaf(T[r * 4 + 1], pix[1], a);
            a = __builtin_fmaf(T[r * 4 + 2], pix[2], a);
            o[7 + r] = a;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 10; ++k) o[k] = 0.f;
    }
}

__global__ void k_head_features(const DevCfg *__restrict__ cfg, int n_heads, int J,
                                const int32_t *__restrict__ head_cam, const uint32_t *__restrict__ joint_mask,
                                const double *__restrict__ xy, const float *__restrict__ vp,
                                float *__restrict__ feat, int ld_feat, int dense, int32_t *__restrict__ zero_counts,
                                int n_zero) {
#pragma clang fp contract(off)
    // the per-camera head counts of k_group_heads (the next launch) start from zero: cleared here instead of by a
    // memset node of their own
    if (zero_counts && blockIdx.x == 0 && (int)threadIdx.x < n_zero) zero_counts[threadIdx.x] = 0;
    const int h = blockIdx.x;
    if (h >= n_heads) return;
    __shared__ float s_f[MPE_MAX_JOINTS * 10];
    const int cam = head_cam[h];
    const int t = threadIdx.x;
    if (t < J) {
        float o[10];
        head_joint_features(cfg, cam, joint_mask[h], t, xy + ((size_t)h * J + t) * 2, vp + ((size_t)h * J + t) * 2, o);
#pragma unroll
        for (int k = 0; k < 10; ++k) s_f[t * 10 + k] = o[k];
    }
    __syncthreads();
    const int blk = J * 10;
    if (dense) {
        const int lo = 2 + cam * blk;
        float *row = feat + (size_t)h * ld_feat;
        for (int c = t; c < ld_feat; c += blockDim.x) {
            float v = 0.f;
            if (c == 0) v = 1.f;
            else if (c >= lo && c < lo + blk) v = s_f[c - lo];
            row[c] = v;
        }
    } else {
        float *row = feat + (size_t)h * ld_feat;
        for (int c = t; c < blk; c += blockDim.x) row[c] = s_f[c];
    }
}

hipError_t launch_head_features(hipStream_t s, const DevCfg *cfg, const mpe_batch &b, int J, float *feat,
                                int ld_feat, int, int, bool dense, int32_t *zero_counts, int n_zero) {
    if (b.n_heads <= 0) return zero_counts ? hipMemsetAsync(zero_counts, 0, (size_t)n_zero * sizeof(int32_t), s) : hipSuccess;
    hipLaunchKernelGGL(k_head_features, dim3(b.n_heads), dim3(128), 0, s, cfg, b.n_heads, J, b.d_head_cam,
                       b.d_joint_mask, b.d_xy, b.d_vp, feat, ld_feat, dense ? 1 : 0, zero_counts, n_zero);
    return hipGetLastError();
}

// rows that are one-hot at `col` (the input rows of the edge-nodes, graph_generator.py:629-631: one-hot at column 1)
__global__ void k_onehot_rows(float *__restrict__ rows, int n_rows, int ld, int col) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_rows * ld) return;
    rows[i] = (int)(i % ld) == col ? 1.f : 0.f;
}

hipError_t launch_onehot_rows(hipStream_t s, float *rows, int n_rows, int ld, int col) {
    if (n_rows <= 0) return hipSuccess;
    const size_t n = (size_t)n_rows * ld;
    hipLaunchKernelGGL(k_onehot_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rows, n_rows, ld, col);
    return hipGetLastError();
}

// heads of every camera (order inside a camera is irrelevant: rows are independent).  A
// workgroup counts its 1024 heads per camera in LDS and reserves one range per camera with a
// single global atomic (20 000 heads on 5 cameras would otherwise serialise on 5 addresses).
__global__ __launch_bounds__(1024) void k_group_heads(int n_heads, const int32_t *__restrict__ head_cam,
                                                      int32_t *__restrict__ cam_count, int32_t *__restrict__ cam_list,
                                                      int list_stride) {
    __shared__ int s_cnt[MPE_MAX_CAMERAS], s_base[MPE_MAX_CAMERAS];
    const int t = threadIdx.x;
    if (t < MPE_MAX_CAMERAS) s_cnt[t] = 0;
    __syncthreads();
    const int h = blockIdx.x * blockDim.x + t;
    int c = -1, pos = 0;
    if (h < n_heads) {
        c = head_cam[h];
        pos = atomicAdd(&s_cnt[c], 1);
    }
    __syncthreads();
    if (t < MPE_MAX_CAMERAS && s_cnt[t] > 0) s_base[t] = atomicAdd(&cam_count[t], s_cnt[t]);
    __syncthreads();
    if (c >= 0) cam_list[(size_t)c * list_stride + s_base[c] + pos] = h;
}

hipError_t launch_group_heads(hipStream_t s, int n_heads, int V, const int32_t *head_cam, int32_t *cam_count,
                              int32_t *cam_list, int list_stride, bool counts_zeroed) {
    hipError_t e = counts_zeroed ? hipSuccess : hipMemsetAsync(cam_count, 0, (size_t)V * sizeof(int32_t), s);
    if (e != hipSuccess || n_heads <= 0) return e;
    hipLaunchKernelGGL(k_group_heads, dim3((n_heads + 1023) / 1024), dim3(1024), 0, s, n_heads, head_cam, cam_count,
                       cam_list, list_stride);
    return hipGetLastError();
}

// Feature rows of the attention stage are fp32, or fp16 in the reduced-precision mode
// (BASELINE.json configs[4]); `half` is uniform per launch.  Arithmetic stays fp32.
__device__ __forceinline__ float ld_ft(const float *base, size_t idx, int half) {
    return half ? (float)reinterpret_cast<const _Float16 *>(base)[idx] : base[idx];
}

template <int VEC>
__device__ __forceinline__ void ld_ftv(const float *base, size_t idx, int half, float *out) {
    if (half) {
        typedef _Float16 hv __attribute__((ext_vector_type(VEC)));
        const hv v = *reinterpret_cast<const hv *>(reinterpret_cast<const _Float16 *>(base) + idx);
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] = (float)v[k];
    } else {
        typedef float fv __attribute__((ext_vector_type(VEC)));
        const fv v = *reinterpret_cast<const fv *>(base + idx);
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] = v[k];
    }
}

template <>
__device__ __forceinline__ void ld_ftv<1>(const float *base, size_t idx, int half, float *out) {
    out[0] = ld_ft(base, idx, half);
}

// ---------------------------------------------------------------------------------------
// attention coefficients a1 = <ft2[n,h,:], attn_l[h,:]>, a2 with attn_r  (gat2.py:57-58)
//
// Canonical summation order for 40-wide heads = the order of the fc2 GEMM epilogue
// (k_linear_dma<.., A12>, gemm.hip), so that a coefficient has the same bits whichever kernel
// produced it: four fma chains s_0..s_3 over the features that lane group q of the MFMA layout
// owns inside an 80-wide tile (two heads per tile, hence the head's parity), then
// (s_0 + s_1) + (s_2 + s_3).  Other head widths: one fma chain over d.
// ---------------------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ void coef40(F ft, const float *__restrict__ al, const float *__restrict__ ar, int parity,
                                       float &o1, float &o2) {
    float s1[4], s2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x1 = 0.f, x2 = 0.f;
        auto step = [&](int d) {
            const float v = ft(d);
            x1 = __builtin_fmaf(v, al[d], x1);
            x2 = __builtin_fmaf(v, ar[d], x2);
        };
        if (parity == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) step(16 * t + 4 * q + i);
            if (q < 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) step(32 + 4 * q + i);
            }
        } else {
            if (q >= 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) step(4 * q - 8 + i);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) step(8 + 4 * q + i);
#pragma unroll
            for (int i = 0; i < 4; ++i) step(24 + 4 * q + i);
        }
        s1[q] = x1;
        s2[q] = x2;
    }
    o1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    o2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
}

constexpr int COEF_ROWS = 32;      // rows per workgroup (fewer when the rows are wide: 48 KiB of LDS at most)

__global__ __launch_bounds__(256) void k_attn_coef(const float *__restrict__ ft2, int ld, int n_rows, int heads,
                                                   int out_dim, const float *__restrict__ attn_l,
                                                   const float *__restrict__ attn_r, float *__restrict__ a12,
                                                   int ft_half, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float s_ft[];   // [rows_per_wg][hdp], hdp = hd rounded up to 4 (+4: bank spread)
    const int hd = heads * out_dim;
    const int hd4 = (hd + 3) / 4, hdp = hd4 * 4 + 4;
    const int r0 = blockIdx.x * rows_per_wg;
    const int nr = min(rows_per_wg, n_rows - r0);
    if (!ft_half && ld % 4 == 0 && hd4 * 4 <= ld) {
        // fp32 rows: 16-byte loads (the row stride covers the rounded-up width)
        for (int i = threadIdx.x; i < nr * hd4; i += blockDim.x) {
            const int r = i / hd4, c4 = i - r * hd4;
            *reinterpret_cast<float4 *>(s_ft + r * hdp + c4 * 4) =
                *reinterpret_cast<const float4 *>(ft2 + (size_t)(r0 + r) * ld + c4 * 4);
        }
    } else if (ft_half && ld % 8 == 0) {
        // fp16 rows (configs[4]): 16-byte loads of eight halves, scalar loads for the last hd % 8 columns (the all-scalar
        // loop below took 626 us per launch at 23 x 10 against 399 us for the fp32 rows it was meant to halve)
        const int hd8 = hd / 8, tail = hd - hd8 * 8;
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const _Float16 *fh = reinterpret_cast<const _Float16 *>(ft2);
        for (int i = threadIdx.x; i < nr * hd8; i += blockDim.x) {
            const int r = i / hd8, c8 = i - r * hd8;
            const h8 v = *reinterpret_cast<const h8 *>(fh + (size_t)(r0 + r) * ld + c8 * 8);
            float *d = s_ft + r * hdp + c8 * 8;
            *reinterpret_cast<float4 *>(d) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
            *reinterpret_cast<float4 *>(d + 4) = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
        }
        for (int i = threadIdx.x; i < nr * tail; i += blockDim.x) {
            const int r = i / tail, c = hd8 * 8 + (i - r * tail);
            s_ft[r * hdp + c] = (float)fh[(size_t)(r0 + r) * ld + c];
        }
    } else {
        for (int i = threadIdx.x; i < nr * hd; i += blockDim.x) {
            const int r = i / hd, c = i - r * hd;
            s_ft[r * hdp + c] = ld_ft(ft2, (size_t)(r0 + r) * ld + c, ft_half);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nr * heads; i += blockDim.x) {
        const int r = i / heads, h = i - r * heads;
        const float *fv = s_ft + r * hdp + h * out_dim;
        float x1 = 0.f, x2 = 0.f;
        if (out_dim == 40) {
            coef40([&](int d) { return fv[d]; }, attn_l + h * 40, attn_r + h * 40, h & 1, x1, x2);
        } else {
            const float *al = attn_l + h * out_dim, *ar = attn_r + h * out_dim;
            for (int d = 0; d < out_dim; ++d) {
                x1 = __builtin_fmaf(fv[d], al[d], x1);
                x2 = __builtin_fmaf(fv[d], ar[d], x2);
            }
        }
        a12[(size_t)(r0 + r) * 32 + h] = x1;
        a12[(size_t)(r0 + r) * 32 + 16 + h] = x2;
    }
}

hipError_t launch_attn_coef(hipStream_t s, const float *ft2, int ld, int n_rows, int heads, int out_dim,
                            const float *attn_l, const float *attn_r, float *a12, int ft_half) {
    if (n_rows <= 0) return hipSuccess;
    const size_t row_bytes = (size_t)((heads * out_dim + 3) / 4 * 4 + 4) * sizeof(float);
    int rows = (int)((size_t)48 * 1024 / row_bytes);
    rows = rows < 1 ? 1 : rows > COEF_ROWS ? COEF_ROWS : rows;
    const int grid = (n_rows + rows - 1) / rows;
    hipLaunchKernelGGL(k_attn_coef, dim3(grid), dim3(256), rows * row_bytes, s, ft2, ld, n_rows, heads, out_dim, attn_l,
                       attn_r, a12, ft_half, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// edge attention + softmax over in-edges + weighted sum + activation
// (gat2.py:61-66,78-88 and the activation of GAT2.forward :141-147)
//
// Edge-node destinations (89 % of the rows at 5x4) have exactly three in-edges (h1, h2, self):
// every lane recomputes its head's 3-way softmax in registers.  Head destinations (in-degree
// 1 + heads of the other cameras, 221 at 23x10) use a WAVE per (destination, attention head):
//   * the source of in-edge e is arithmetic, no list is built: e = 0 is the self loop, e >= 1 is
//     the (e-1)-th head of the other cameras in head order, and the edge-node that joins them
//     follows from the per-frame table of pair bases -- ascending e is ascending edge id, the
//     reference's edge creation order (graph_generator.py:632-651);
//   * max, exp and sum are wave reductions of fixed shape: lane j accumulates the in-edges
//     e = j, j+64, ... in ascending order, then an xor butterfly (32,16,8,4,2,1).  The reference
//     delegates this sum to DGL's edge_softmax, whose order is not pinned by anything in the
//     reference; the tree differs from a sequential sum by a few ulp (parity bound 2e-5).
//   * the weighted sum of the source rows stays sequential in edge order per output column.
// The fused kernel (frames whose slice fits in LDS) and the general kernels share these orders,
// so a frame gives the same bits on either path (tested).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float agg_activate(float x, int mode, float slope) {
    if (mode == 0) return x > 0.f ? x : x * slope;
    if (mode == 1) return 1.f / (1.f + expf(-x));
    return x;
}

// Butterfly over a lane group of G = 64, 32 or 16 lanes.  The canonical shape is the 64-lane one;
// a smaller group is used only when every in-degree of the launch fits in it (deg <= G), and then
// gives the same bits: the lanes it leaves out would hold 0 (sum) / -inf (max).
template <int G>
__device__ __forceinline__ float group_sum(float x) {
    if (G > 32) x = x + __shfl_xor(x, 32);
    if (G > 16) x = x + __shfl_xor(x, 16);
    x = x + __shfl_xor(x, 8);
    x = x + __shfl_xor(x, 4);
    x = x + __shfl_xor(x, 2);
    x = x + __shfl_xor(x, 1);
    return x;
}

template <int G>
__device__ __forceinline__ float group_max(float x) {
    if (G > 32) x = fmaxf(x, __shfl_xor(x, 32));
    if (G > 16) x = fmaxf(x, __shfl_xor(x, 16));
    x = fmaxf(x, __shfl_xor(x, 8));
    x = fmaxf(x, __shfl_xor(x, 4));
    x = fmaxf(x, __shfl_xor(x, 2));
    x = fmaxf(x, __shfl_xor(x, 1));
    return x;
}

__device__ __forceinline__ float wave_sum(float x) { return group_sum<64>(x); }
__device__ __forceinline__ float wave_max(float x) { return group_max<64>(x); }

// Per-frame topology scalars in LDS: slot prefix sums and the first edge-node id of every
// camera-slot pair (p < q), lexicographic = creation order (graph_generator.py:854-864).
struct FrameTopo {
    int *start;   // [V + 1] first head of slot s
    int *base;    // [V * V] frame-local node id of the first edge-node of pair (p, q), p < q
};

template <typename SN>
__device__ __forceinline__ void build_topo(FrameTopo tp, SN sn, int V, int H, int t, int nthreads) {
    if (t == 0) {
        int acc = 0;
        for (int s = 0; s < V; ++s) {
            tp.start[s] = acc;
            acc += sn[s];
        }
        tp.start[V] = acc;
    }
    for (int pq = t; pq < V * V; pq += nthreads) {
        const int p = pq / V, q = pq - p * V;
        if (p >= q) continue;
        int b = H;
        for (int a = 0; a <= p; ++a) {
            const int na = sn[a];
            const int qe = a == p ? q : V;
            for (int c = a + 1; c < qe; ++c) b += na * sn[c];
        }
        tp.base[pq] = b;
    }
}

// source node (frame-local id) of in-edge e of head v
template <typename SN>
__device__ __forceinline__ int head_in_edge(const FrameTopo &tp, SN sn, int V, int v, int s, int e) {
    if (e == 0) return v;
    const int st = tp.start[s], ns = sn[s], i = v - st;
    int u = e - 1;
    if (u >= st) u += ns;                       // the (e-1)-th head that is not in slot s
    int p = 0;
    while (p + 1 < V && u >= tp.start[p + 1]) ++p;
    const int k = u - tp.start[p];
    return p < s ? tp.base[p * V + s] + k * ns + i : tp.base[s * V + p] + i * sn[p] + k;
}

__device__ __forceinline__ int slot_of_head(const FrameTopo &tp, int V, int v) {
    int s = 0;
    while (s + 1 < V && v >= tp.start[s + 1]) ++s;
    return s;
}

// In-edge sources of every head of a frame, [hmax][hmax + 1] per frame: entry (h, e) = frame-local id
// of the source node of in-edge e of head h, 0xFFFF for e >= in-degree and for rows h >= H.  Frames
// beyond hmax heads (flagged by k_topology) get an all-0xFFFF table.
__device__ __forceinline__ void head_sources_body(int f, int *s_slot, int V, int hmax, const int32_t *__restrict__ head_off,
                                                  const int32_t *__restrict__ slot_n, uint16_t *__restrict__ head_src) {
    __shared__ int s_topo[MPE_MAX_CAMERAS + 1 + MPE_MAX_CAMERAS * MPE_MAX_CAMERAS];
    __shared__ int s_n[MPE_MAX_CAMERAS];
    const int t = threadIdx.x;
    const int H = head_off[f + 1] - head_off[f];
    const int max_deg = hmax + 1;
    FrameTopo tp;
    tp.start = s_topo;
    tp.base = s_topo + V + 1;
    if (t < V) s_n[t] = slot_n[(size_t)f * V + t];
    __syncthreads();
    build_topo(tp, s_n, V, H, t, blockDim.x);
    __syncthreads();
    const bool ok = H <= hmax;
    if (ok)
        for (int h = t; h < H; h += blockDim.x) s_slot[h] = slot_of_head(tp, V, h);
    __syncthreads();
    uint16_t *dst = head_src + (size_t)f * hmax * max_deg;
    for (int i = t; i < hmax * max_deg; i += blockDim.x) {
        const int h = i / max_deg, e = i - h * max_deg;
        int u = 0xFFFF;
        if (ok && h < H) {
            const int sl = s_slot[h], st = tp.start[sl], ns = s_n[sl];
            if (e == 0) {
                u = h;
            } else if (e < 1 + H - ns) {
                int o = e - 1;
                if (o >= st) o += ns;               // the (e-1)-th head that is not in slot sl
                const int p = s_slot[o], k = o - tp.start[p], ii = h - st;
                u = p < sl ? tp.base[p * V + sl] + k * ns + ii : tp.base[sl * V + p] + ii * s_n[p] + k;
            }
        }
        dst[i] = (uint16_t)u;
    }
}

__global__ __launch_bounds__(256) void k_head_sources(int V, int hmax, const int32_t *__restrict__ head_off,
                                                      const int32_t *__restrict__ slot_n,
                                                      uint16_t *__restrict__ head_src) {
    extern __shared__ int s_slot[];                 // [hmax] camera slot of every head
    head_sources_body(blockIdx.x, s_slot, V, hmax, head_off, slot_n, head_src);
}

// ---------------------------------------------------------------------------------------
// Small batches: the front of the matching stage in ONE launch (k_topology + k_head_sources + k_head_features + k_group_heads +
// the per-camera fc1 launches of layer 0 are nine launches of ~4.5 us each for one 5 x 4 frame).  Workgroups [0, n_frames) write
// the topology tables of their frame (first read by the attention stage, two launches later); the others are fc1 of layer 0
// (gat2.py:53-54 on head rows, K = the J * 10 columns of the head's own camera: see the layer-0 de-duplication in DESIGN.md 2) for one
// (camera, 16 heads of that camera, 64 output features): the heads of a camera are found from the frames' slot tables (ascending
// head order), their feature rows are computed in place (k_head_features' numbers) into LDS, and four waves multiply them with the
// camera's weight block -- k_linear_skinny's arithmetic: v_mfma_f32_16x16x4_f32, k ascending, one fp32 chain -- all of a wave's
// weight fragments requested before the features are computed.
// ---------------------------------------------------------------------------------------
constexpr int L0A_NK = 6;           // K stages of 32: J * 10 <= 192
constexpr int L0A_XS = 196;         // LDS row stride of the feature tile (floats): 16 rows 16 B apart in the banks

__global__ __launch_bounds__(256) void k_lat_l0a(const DevCfg *__restrict__ cfg, int n_frames, int n_heads, int V, int J,
                                                 const int32_t *__restrict__ head_off, const int32_t *__restrict__ en_off,
                                                 const int32_t *__restrict__ slot_cam, const int32_t *__restrict__ slot_n,
                                                 const uint32_t *__restrict__ joint_mask, const double *__restrict__ xy,
                                                 const float *__restrict__ vp, int32_t *__restrict__ node_off,
                                                 int32_t *__restrict__ head_frame, int32_t *__restrict__ en_frame,
                                                 int32_t *__restrict__ en_pair, uint16_t *__restrict__ head_src, int hmax,
                                                 int32_t *__restrict__ status, const float *__restrict__ l0_w, long w_cam_stride,
                                                 int l0_ld, const float *__restrict__ l0_b, int n_out, float *__restrict__ h0,
                                                 int ld_h0, float alpha) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    extern __shared__ int s_dyn_i[];
    if ((int)blockIdx.x < n_frames) {
        topology_body(blockIdx.x, n_frames, V, head_off, en_off, slot_n, node_off, head_frame, en_frame, en_pair, hmax, status);
        if (head_src) {
            __syncthreads();
            head_sources_body(blockIdx.x, s_dyn_i, V, hmax, head_off, slot_n, head_src);
        }
        return;
    }
    __shared__ __attribute__((aligned(16))) float s_x[16 * L0A_XS];
    __shared__ int s_cnt[MPE_MAX_CAMERAS], s_row[16], s_job[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ntiles = (n_out + 15) / 16, ncg = (ntiles + 3) / 4;
    const int job = blockIdx.x - n_frames;
    const int rtg = job / ncg, cg = job - rtg * ncg;
    // heads per camera, from the slot tables (slot s of frame f holds slot_n heads of camera slot_cam, heads of a frame in slot order)
    if (t < V) {
        int c = 0;
        for (int i = 0; i < n_frames * V; ++i) c += slot_cam[i] == t ? slot_n[i] : 0;
        s_cnt[t] = c;
    }
    __syncthreads();
    if (t == 0) {
        int c = 0, acc = 0;
        for (; c < V; ++c) {
            const int tl = (s_cnt[c] + 15) / 16;
            if (rtg < acc + tl) break;
            acc += tl;
        }
        s_job[0] = c;               // camera (V = no such row tile)
        s_job[1] = rtg - acc;       // row tile within the camera
    }
    __syncthreads();
    const int cam = s_job[0];
    if (cam >= V) return;
    // this wave's weight fragments: one burst, before anything that depends on the rows
    const int ct = cg * 4 + wave;
    const int fq = lane >> 4, fr = lane & 15;
    f32x4 wv[L0A_NK][2];
    {
        const int ctc = ct < ntiles ? ct : ntiles - 1;
        const float *pw = l0_w + (size_t)cam * w_cam_stride + (size_t)(ctc * 16 + fr) * l0_ld + 8 * fq;
#pragma unroll
        for (int kt = 0; kt < L0A_NK; ++kt) {
            const int ko = (kt * 32 < l0_ld ? kt : 0) * 32;
            wv[kt][0] = *reinterpret_cast<const f32x4 *>(pw + ko);
            wv[kt][1] = *reinterpret_cast<const f32x4 *>(pw + ko + 4);
        }
    }
    if (t < 16) {
        // the (16 r + t)-th head of the camera, in ascending head order
        int want = s_job[1] * 16 + t, found = -1;
        if (want < s_cnt[cam]) {
            for (int f = 0; f < n_frames && found < 0; ++f) {
                int start = 0;
                for (int sl = 0; sl < V; ++sl) {
                    const int n = slot_n[f * V + sl];
                    if (slot_cam[f * V + sl] == cam) {
                        if (want < n) {
                            found = head_off[f] + start + want;
                            break;
                        }
                        want -= n;
                    }
                    start += n;
                }
            }
        }
        s_row[t] = found;
    }
    for (int i = t; i < 16 * (L0A_XS - J * 10); i += 256) {             // columns behind the J * 10 features: zeros (they meet zero weights)
        const int r = i / (L0A_XS - J * 10), c = J * 10 + (i - r * (L0A_XS - J * 10));
        s_x[r * L0A_XS + c] = 0.f;
    }
    __syncthreads();
    for (int i = t; i < 16 * J; i += 256) {
        const int r = i / J, j = i - r * J;
        const int h = s_row[r];
        float o[10];
        if (h >= 0) {
            head_joint_features(cfg, cam, joint_mask[h], j, xy + ((size_t)h * J + j) * 2, vp + ((size_t)h * J + j) * 2, o);
        } else {
#pragma unroll
            for (int k = 0; k < 10; ++k) o[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) s_x[r * L0A_XS + j * 10 + k] = o[k];
    }
    __syncthreads();
    if (ct >= ntiles) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nk = l0_ld / 32;
#pragma unroll
    for (int kt = 0; kt < L0A_NK; ++kt) {
        if (kt < nk) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(&s_x[fr * L0A_XS + kt * 32 + 8 * fq]);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(&s_x[fr * L0A_XS + kt * 32 + 8 * fq + 4]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kt][0][q], a0[q], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kt][1][q], a1[q], acc, 0, 0, 0);
        }
    }
    const int h = s_row[fr];
    if (h < 0) return;
    const int nb = ct * 16 + fq * 4;
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(l0_b + nb);
    f32x4 v = acc + bv;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * alpha;
    float *dst = h0 + (size_t)h * ld_h0 + nb;
    if (nb + 3 < n_out) {
        *reinterpret_cast<f32x4 *>(dst) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (nb + i < n_out) dst[i] = v[i];
    }
}

bool lat_l0a_available(int J, int l0_ld) { return J * 10 <= L0A_NK * 32 && l0_ld <= L0A_NK * 32 && L0A_XS >= l0_ld; }

hipError_t launch_lat_l0a(hipStream_t s, const DevCfg *cfg, const mpe_batch &b, int V, int J, int32_t *node_off, int32_t *head_frame,
                          int32_t *en_frame, int32_t *en_pair, uint16_t *head_src, int hmax, int32_t *status, const float *l0_w,
                          long w_cam_stride, int l0_ld, const float *l0_b, int n_out, float *h0, int ld_h0, float alpha) {
    if (b.n_frames <= 0) return hipSuccess;
    if (b.d_en_pair || !lat_l0a_available(J, l0_ld)) return hipErrorInvalidValue;
    if (!head_src_entries(hmax, V)) head_src = nullptr;
    const int ntiles = (n_out + 15) / 16, ncg = (ntiles + 3) / 4;
    const int row_tiles = V + (b.n_heads + 15) / 16;                   // >= sum over the cameras of ceil(heads of the camera / 16)
    const int grid = b.n_frames + (b.n_heads > 0 ? row_tiles * ncg : 0);
    hipLaunchKernelGGL(k_lat_l0a, dim3(grid), dim3(256), (size_t)hmax * sizeof(int), s, cfg, b.n_frames, b.n_heads, V, J, b.d_frame_head_off,
                       b.d_frame_en_off, b.d_slot_cam, b.d_slot_n, b.d_joint_mask, b.d_xy, b.d_vp, node_off, head_frame, en_frame, en_pair,
                       head_src, hmax, status, l0_w, w_cam_stride, l0_ld, l0_b, n_out, h0, ld_h0, alpha);
    return hipGetLastError();
}

// Weighted sum over the in-edges of a head destination, canonical order: eight accumulators, the
// j-th over the in-edges e = j, j+8, ... in ascending order, combined as
// ((a0+a1)+(a2+a3)) + ((a4+a5)+(a6+a7)).  Eight independent chains keep eight row loads in flight
// (a head of the 23 x 10 rig has 221 in-edges); for the three in-edges of an edge-node the same
// rule reads (v1*w1 + v2*w2) + v3*w3.  `row(e, out)` fetches VEC columns of the e-th source row,
// `wt(e)` its softmax weight.
template <int VEC, typename RowFn, typename WFn>
__device__ __forceinline__ void weighted_sum8(int deg, RowFn row, WFn wt, float *out) {
#pragma clang fp contract(off)
    float acc[8][VEC];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[j][k] = 0.f;
    int e = 0;
    for (; e + 8 <= deg; e += 8) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {             // two batches of four rows: half the live registers
            float fv[4][VEC], w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                row(e + hf * 4 + j, fv[j]);
                w[j] = wt(e + hf * 4 + j);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float m = fv[j][k] * w[j];
                    acc[hf * 4 + j][k] = acc[hf * 4 + j][k] + m;
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (e + j < deg) {
            float fv[VEC];
            row(e + j, fv);
            const float w = wt(e + j);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float m = fv[k] * w;
                acc[j][k] = acc[j][k] + m;
            }
        }
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        out[k] = ((acc[0][k] + acc[1][k]) + (acc[2][k] + acc[3][k])) + ((acc[4][k] + acc[5][k]) + (acc[6][k] + acc[7][k]));
}

constexpr int EN_ROWS = 16;     // edge-node rows per workgroup

// Phase A: one thread per (row, attention head) computes the 3-way softmax weights and one per row
// the three source rows; phase B: one thread per (row, VEC columns) does the weighted sum from them.
// rows of the activated output: fp32 rows, or (small batches, lat.hip: the next layer's fc1 takes its activations as the three
// bf16 planes split8 would make of them) planes only
template <int VEC>
__device__ __forceinline__ void agg_store_row(const AggArgs &a, size_t row, int c, const float *o) {
    if (a.out_pl) {
        unsigned short *d = a.out_pl + row * a.ld_out + c;
        if (VEC == 4) {
            sb::f32x4 v = {o[0], o[1], o[2], o[3]};
            sb::store_planes4(d, a.out_pl_plane, v);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) sb::store_planes1(d + k, a.out_pl_plane, o[k]);
        }
        return;
    }
    typedef float vecf __attribute__((ext_vector_type(VEC)));
    vecf v;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = o[k];
    *reinterpret_cast<vecf *>(a.out + row * a.ld_out + c) = v;
}

// a1 | a2 of one attention head of a feature row, when no GEMM epilogue left them in a12 (small batches: the layer-0 rows and the
// 30-wide heads): the canonical orders -- coef40 for 40-wide heads (parity = the head's place in its 80-wide tile), one chain otherwise
__device__ __forceinline__ void row_coef(const float *__restrict__ fv, const AggArgs &a, int hh, float &x1, float &x2) {
    const float *al = a.attn_l + hh * a.out_dim, *ar = a.attn_r + hh * a.out_dim;
    if (a.out_dim == 40) {
        // the forty values first (ten 16-byte requests in flight), then the chains: left to the chains, every value is a request of
        // its own behind the previous one's wait
        float v[40];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const float4 x = *reinterpret_cast<const float4 *>(fv + 4 * q);
            v[4 * q] = x.x;
            v[4 * q + 1] = x.y;
            v[4 * q + 2] = x.z;
            v[4 * q + 3] = x.w;
        }
        coef40([&](int d) { return v[d]; }, al, ar, hh & 1, x1, x2);
    } else if (a.out_dim == 30) {
        float v[30];                                       // (a 30-wide head starts at a multiple of 120 bytes: 8-byte requests)
#pragma unroll
        for (int q = 0; q < 15; ++q) {
            const float2 x = *reinterpret_cast<const float2 *>(fv + 2 * q);
            v[2 * q] = x.x;
            v[2 * q + 1] = x.y;
        }
        x1 = x2 = 0.f;
#pragma unroll
        for (int d = 0; d < 30; ++d) {
            x1 = __builtin_fmaf(v[d], al[d], x1);
            x2 = __builtin_fmaf(v[d], ar[d], x2);
        }
    } else {
        x1 = x2 = 0.f;
        for (int d = 0; d < a.out_dim; ++d) {
            x1 = __builtin_fmaf(fv[d], al[d], x1);
            x2 = __builtin_fmaf(fv[d], ar[d], x2);
        }
    }
}

template <int VEC, int ER = 16>
__device__ __forceinline__ void aggregate_en_body(int wg, int n_en, const int32_t *__restrict__ head_off,
                                                  const int32_t *__restrict__ en_off,
                                                  const int32_t *__restrict__ node_off,
                                                  const int32_t *__restrict__ en_frame,
                                                  const int32_t *__restrict__ en_pair, const AggArgs &a, int hmax) {
#pragma clang fp contract(off)
    __shared__ float s_w[ER][16][3];                // softmax weights of (h1, h2, self) per attention head
    __shared__ int s_over[ER];                      // row of a frame beyond max_heads_per_frame (flagged by k_topology): score 0
    __shared__ long s_src[ER][3];                   // ft2 row of h1, h2, self (self: -1 at layer 0 = the shared row)
    __shared__ long s_dst[ER];                      // output row
    const int hd = a.heads * a.out_dim, heads = a.heads;
    const int per_row = (hd + VEC - 1) / VEC;       // a VEC-column group may straddle two attention heads
    const int m0 = wg * ER;
    const int rows = min(ER, n_en - m0);
    const bool l0 = a.en_const_ft2 != nullptr;
    for (int i = threadIdx.x; i < rows * heads; i += blockDim.x) {
        const int r = i / heads, hh = i - r * heads;
        const int m = m0 + r;
        const int f = en_frame[m];
        const int hb = head_off[f], H = head_off[f + 1] - hb;
        const int nb = node_off[f];
        const int v = H + (m - en_off[f]);
        const int h1 = en_pair[2 * (size_t)m], h2 = en_pair[2 * (size_t)m + 1];
        const float *ra1, *ra2, *ra3;     // a1|a2 rows of h1, h2, self
        if (l0) {
            ra1 = a.a12 + (size_t)(hb + h1) * 32;
            ra2 = a.a12 + (size_t)(hb + h2) * 32;
            ra3 = a.en_const_a;
        } else {
            ra1 = a.a12 + (size_t)(nb + h1) * 32;
            ra2 = a.a12 + (size_t)(nb + h2) * 32;
            ra3 = a.a12 + (size_t)(nb + v) * 32;
        }
        float a2v, c1, c2, c3;
        if (a.a12_ready) {
            a2v = ra3[16 + hh];
            c1 = ra1[hh];
            c2 = ra2[hh];
            c3 = ra3[hh];
        } else {
            // no coefficients in a12: from the rows themselves (fp32 rows; the shared layer-0 row has its own constants)
            const size_t rb = (size_t)(l0 ? hb : nb);
            float unused;
            row_coef(a.ft2 + (rb + h1) * a.ld + hh * a.out_dim, a, hh, c1, unused);
            row_coef(a.ft2 + (rb + h2) * a.ld + hh * a.out_dim, a, hh, c2, unused);
            if (l0) {
                c3 = a.en_const_a[hh];
                a2v = a.en_const_a[16 + hh];
            } else {
                row_coef(a.ft2 + (rb + v) * a.ld + hh * a.out_dim, a, hh, c3, a2v);
            }
        }
        float e1 = c1 + a2v, e2 = c2 + a2v, e3 = c3 + a2v;
        e1 = e1 > 0.f ? e1 : e1 * a.alpha;
        e2 = e2 > 0.f ? e2 : e2 * a.alpha;
        e3 = e3 > 0.f ? e3 : e3 * a.alpha;
        const float mx = fmaxf(fmaxf(e1, e2), e3);
        const float x1 = expf(e1 - mx), x2 = expf(e2 - mx), x3 = expf(e3 - mx);
        const float sum = (x1 + x2) + x3;
        s_w[r][hh][0] = x1 / sum;
        s_w[r][hh][1] = x2 / sum;
        s_w[r][hh][2] = x3 / sum;
        if (hh == 0) {
            s_src[r][0] = (long)(l0 ? hb : nb) + h1;
            s_src[r][1] = (long)(l0 ? hb : nb) + h2;
            s_src[r][2] = l0 ? -1 : (long)nb + v;
            s_dst[r] = (long)nb + v;
            s_over[r] = H > hmax ? 1 : 0;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * per_row; i += blockDim.x) {
        const int r = i / per_row;
        const int c = (i - r * per_row) * VEC;
        const int hh = c / a.out_dim;
        float v1[VEC], v2[VEC], v3[VEC];  // feature rows
        ld_ftv<VEC>(a.ft2, (size_t)s_src[r][0] * a.ld + c, a.ft_half, v1);
        ld_ftv<VEC>(a.ft2, (size_t)s_src[r][1] * a.ld + c, a.ft_half, v2);
        if (l0) ld_ftv<VEC>(a.en_const_ft2, (size_t)c, 0, v3);
        else ld_ftv<VEC>(a.ft2, (size_t)s_src[r][2] * a.ld + c, a.ft_half, v3);
        float w1[VEC], w2[VEC], w3[VEC];
        const bool one_head = VEC == 1 || (c + VEC <= hd && (c + VEC - 1) / a.out_dim == hh);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            // columns behind the row's width (padding of the last group) reuse the last head: finite, never read
            int hk = hh;
            if (!one_head) {
                hk = (c + k) / a.out_dim;
                hk = hk < heads ? hk : heads - 1;
            }
            w1[k] = s_w[r][hk][0];
            w2[k] = s_w[r][hk][1];
            w3[k] = s_w[r][hk][2];
        }
        float o[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float acc = v1[k] * w1[k];
            acc = acc + v2[k] * w2[k];
            acc = acc + v3[k] * w3[k];
            o[k] = agg_activate(acc, a.out_mode, a.out_slope);
        }
        if (a.score_mode) a.out[m0 + r] = s_over[r] ? 0.f : o[0];
        else agg_store_row<VEC>(a, (size_t)s_dst[r], c, o);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_aggregate_en(int n_en, const int32_t *__restrict__ head_off,
                                                      const int32_t *__restrict__ en_off,
                                                      const int32_t *__restrict__ node_off,
                                                      const int32_t *__restrict__ en_frame,
                                                      const int32_t *__restrict__ en_pair, AggArgs a, int hmax) {
    aggregate_en_body<VEC, EN_ROWS>(blockIdx.x, n_en, head_off, en_off, node_off, en_frame, en_pair, a, hmax);
}

// Head destinations, general path: AGG_ROWS head rows per workgroup, two waves per row for the
// topology scalars, the in-edge sources and the softmax weights (wave reductions; the two waves
// split the attention heads), then all threads accumulate the weighted source rows column by
// column (weighted_sum8).
constexpr int AGG_ROWS = 2;

template <int VEC>
__device__ __forceinline__ void aggregate_heads_body(
    int wg, float *s_dyn, int n_heads, int V, int max_deg, const int32_t *__restrict__ head_off, const int32_t *__restrict__ en_off,
    const int32_t *__restrict__ slot_n, const int32_t *__restrict__ node_off,
    const int32_t *__restrict__ head_frame, const AggArgs &a, const uint16_t *__restrict__ head_src, int src_stride, int hmax) {
#pragma clang fp contract(off)
    const int heads = a.heads, hd = a.heads * a.out_dim;
    float *s_w = s_dyn;                                                          // [AGG_ROWS][heads][max_deg]
    int *s_src = reinterpret_cast<int *>(s_w + (size_t)AGG_ROWS * heads * max_deg);   // [AGG_ROWS][max_deg]
    int *s_topo = s_src + AGG_ROWS * max_deg;                                    // [AGG_ROWS][V + 1 + V * V]
    __shared__ int s_deg[AGG_ROWS], s_f[AGG_ROWS], s_v[AGG_ROWS], s_H[AGG_ROWS];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool l0 = a.en_const_ft2 != nullptr;
    const int r_own = wave & (AGG_ROWS - 1), half = wave / AGG_ROWS;      // row of this wave, which half of the heads
    const int gh = wg * AGG_ROWS + r_own;
    int deg = 0, f = 0, hb = 0, H = 0, v = 0;
    int32_t nb = 0;
    const int32_t *sn = slot_n;
    FrameTopo tp;
    tp.start = s_topo + r_own * (V + 1 + V * V);
    tp.base = tp.start + V + 1;
    bool live = false;
    if (gh < n_heads) {
        f = head_frame[gh];
        hb = head_off[f];
        H = head_off[f + 1] - hb;
        live = H <= hmax;                       // frames beyond max_heads_per_frame are skipped (flagged by k_topology)
        v = gh - hb;
        nb = node_off[f];
        sn = slot_n + (size_t)f * V;
        if (live && half == 0 && !head_src) build_topo(tp, sn, V, H, lane, 64);
    }
    __syncthreads();
    if (live) {
        if (head_src) {
            // in-edge sources from the per-frame table (0xFFFF behind the in-degree); both waves of a
            // row derive the in-degree, the first one fills the LDS list
            const uint16_t *tab = head_src + ((size_t)f * hmax + v) * src_stride;
            int *src = s_src + r_own * max_deg;
            int dg = 0;
            for (int e = lane; e < src_stride; e += 64) {
                const int u = tab[e];
                if (u != 0xFFFF) {
                    if (half == 0) src[e] = u;
                    dg = e + 1;
                }
            }
            deg = (int)wave_max((float)dg);
        } else {
            const int sl = slot_of_head(tp, V, v);
            deg = 1 + H - sn[sl];
            if (half == 0) {
                int *src = s_src + r_own * max_deg;
                for (int e = lane; e < deg; e += 64) src[e] = head_in_edge(tp, sn, V, v, sl, e);
            }
        }
        if (half == 0 && lane == 0) {
            s_f[r_own] = f;
            s_v[r_own] = v;
            s_H[r_own] = H;
        }
    }
    if (half == 0 && lane == 0) s_deg[r_own] = deg;
    __syncthreads();
    float *s_a2 = reinterpret_cast<float *>(s_topo + AGG_ROWS * (V + 1 + V * V));           // [AGG_ROWS][16] (only without coefficients in a12)
    if (!a.a12_ready) {
        // No coefficients in a12 (small batches, lat.hip): ALL of them first, one (in-edge, attention head) pair per thread of the
        // row's two waves -- a1 of every source into its slot of the weight table, a2 of the row itself beside it -- so that the
        // requests of a whole row are in flight together (inside the per-head loop below they would be two round trips per head).
        if (live) {
            const int *src = s_src + r_own * max_deg;
            const size_t rb = (size_t)(l0 ? hb : nb);
            for (int i = lane + 64 * half; i < (deg + 1) * heads; i += 64 * (4 / AGG_ROWS)) {
                const int e = i / heads, hh = i - e * heads;
                float c1, c2;
                if (e == deg) {
                    row_coef(a.ft2 + (rb + v) * a.ld + hh * a.out_dim, a, hh, c1, c2);
                    s_a2[r_own * 16 + hh] = c2;
                } else {
                    const int u = src[e];
                    if (l0 && u >= H) c1 = a.en_const_a[hh];
                    else row_coef(a.ft2 + (rb + u) * a.ld + hh * a.out_dim, a, hh, c1, c2);
                    s_w[((size_t)r_own * heads + hh) * max_deg + e] = c1;
                }
            }
        }
        __syncthreads();
    }
    if (live) {
        const int *src = s_src + r_own * max_deg;
        const float *a_dst = l0 ? a.a12 + (size_t)(hb + v) * 32 : a.a12 + (size_t)(nb + v) * 32;
        for (int hh = half; hh < heads; hh += 4 / AGG_ROWS) {
            const float a2v = a.a12_ready ? a_dst[16 + hh] : s_a2[r_own * 16 + hh];
            float *w = s_w + ((size_t)r_own * heads + hh) * max_deg;
            float mx = -INFINITY;
            for (int e = lane; e < deg; e += 64) {
                const int u = src[e];
                float a1u;
                if (a.a12_ready) {
                    const float *a_src;
                    if (l0) a_src = u >= H ? a.en_const_a : a.a12 + (size_t)(hb + u) * 32;
                    else a_src = a.a12 + (size_t)(nb + u) * 32;
                    a1u = a_src[hh];
                } else {
                    a1u = w[e];
                }
                float x = a1u + a2v;
                x = x > 0.f ? x : x * a.alpha;
                w[e] = x;
                mx = fmaxf(mx, x);
            }
            mx = wave_max(mx);
            float sum = 0.f;
            for (int e = lane; e < deg; e += 64) {
                const float ex = expf(w[e] - mx);
                w[e] = ex;
                sum = sum + ex;
            }
            sum = wave_sum(sum);
            for (int e = lane; e < deg; e += 64) w[e] = w[e] / sum;
        }
    }
    __syncthreads();

    // out[v][c] = sum_e w[e][h(c)] * ft2[src_e][c], then activation; VEC consecutive columns (of one
    // attention head) per thread
    const int per_row = hd / VEC;
    for (int i = t; i < AGG_ROWS * per_row; i += blockDim.x) {
        const int r = i / per_row, c = (i - r * per_row) * VEC;
        const int dg = s_deg[r];
        if (dg == 0) continue;
        const int ff = s_f[r], vv = s_v[r], HH = s_H[r];
        const int hh = c / a.out_dim;
        const int32_t nbb = node_off[ff], hbb = head_off[ff];
        const float *w = s_w + ((size_t)r * heads + hh) * max_deg;
        const int *src = s_src + r * max_deg;
        float acc[VEC];
        weighted_sum8<VEC>(
            dg,
            [&](int e, float *fv) {
                const int u = src[e];
                if (l0 && u >= HH) ld_ftv<VEC>(a.en_const_ft2, (size_t)c, 0, fv);
                else ld_ftv<VEC>(a.ft2, (size_t)((l0 ? hbb : nbb) + u) * a.ld + c, a.ft_half, fv);
            },
            [&](int e) { return w[e]; }, acc);
        if (a.score_mode) {
            a.out_heads[(size_t)hbb + vv] = agg_activate(acc[0], a.out_mode, a.out_slope);
        } else {
            float o[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = agg_activate(acc[k], a.out_mode, a.out_slope);
            agg_store_row<VEC>(a, (size_t)(nbb + vv), c, o);
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_aggregate_heads(
    int n_heads, int V, int max_deg, const int32_t *__restrict__ head_off, const int32_t *__restrict__ en_off,
    const int32_t *__restrict__ slot_n, const int32_t *__restrict__ node_off,
    const int32_t *__restrict__ head_frame, AggArgs a, const uint16_t *__restrict__ head_src, int src_stride, int hmax) {
    extern __shared__ float s_dyn[];
    aggregate_heads_body<VEC>(blockIdx.x, s_dyn, n_heads, V, max_deg, head_off, en_off, slot_n, node_off, head_frame, a, head_src, src_stride, hmax);
}

// Small batches (a few frames: the reference's own call pattern is one frame per call): BOTH halves of the attention stage in ONE
// launch -- workgroups [0, n_en_wgs) take edge-node rows, the rest head rows -- with the kernels' own bodies, hence their bits.
// Two edge-node rows per workgroup: every thread then has ONE group of columns (with sixteen rows a thread walks seven groups one
// after the other, each a round trip to L2: 11 us per launch for one 5 x 4 frame).
constexpr int LAT_EN_ROWS = 2;
template <int VEC>
__global__ __launch_bounds__(256) void k_lat_attention(
    int n_en_wgs, int n_en, int n_heads, int V, int max_deg, const int32_t *__restrict__ head_off, const int32_t *__restrict__ en_off,
    const int32_t *__restrict__ slot_n, const int32_t *__restrict__ node_off, const int32_t *__restrict__ head_frame,
    const int32_t *__restrict__ en_frame, const int32_t *__restrict__ en_pair, AggArgs a, const uint16_t *__restrict__ head_src,
    int src_stride, int hmax) {
    extern __shared__ float s_dyn[];
    if ((int)blockIdx.x < n_en_wgs) {
        aggregate_en_body<VEC, LAT_EN_ROWS>(blockIdx.x, n_en, head_off, en_off, node_off, en_frame, en_pair, a, hmax);
    } else if (!a.score_mode || a.out_heads) {
        aggregate_heads_body<VEC>(blockIdx.x - n_en_wgs, s_dyn, n_heads, V, max_deg, head_off, en_off, slot_n, node_off, head_frame, a, head_src,
                                  src_stride, hmax);
    }
}

// ---------------------------------------------------------------------------------------
// Fused attention stage for frames whose head slice fits in LDS: one workgroup per
// (frame, attention head).  It loads ft2[:, head, :] of the frame once (N x D' floats, dense
// 16-byte chunks: every LDS access of the hot loops is a ds_read/write_b128), takes a1/a2 from the
// fc2 epilogue (or computes them in the canonical order when the GEMM did not), and writes the
// activated output slice -- ft2 is read once and nothing else touches HBM.
// ---------------------------------------------------------------------------------------
// floats of the small LDS tables of k_gat_fused, rounded up to 1 KiB (the feature image follows)
__host__ __device__ inline size_t fused_tables_floats(int hmax, int max_deg, int V, int n_cap, int m_cap) {
    const size_t deg = (size_t)max_deg;
    const size_t n = 2 * (size_t)n_cap + 2 * hmax * deg + (size_t)m_cap * 4 + (V + 1) + (size_t)V * V + MPE_MAX_CAMERAS + hmax;
    return (n + 255) & ~(size_t)255;
}

// Workgroup barrier that orders LDS traffic only: global loads and LDS-DMA pieces issued before it
// stay in flight (a plain __syncthreads() carries a full workgroup fence, i.e. s_waitcnt vmcnt(0)).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The same without any fence: this wave's LDS operations are complete (lgkmcnt), outstanding
// global loads / LDS-DMA pieces are NOT waited for.  Only for code whose LDS reads after the barrier
// never touch bytes an in-flight DMA piece writes.
__device__ __forceinline__ void lds_barrier_raw() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int FUSED_PIECES = 8;     // LDS-DMA pieces per wave of the overlapped staging (8 x 4 x 1 KiB = 32 KiB image)

// Where a workgroup of k_gat_fused spends its life (DIAGNOSTIC builds only: make exp EXPFLAGS=-DMPE_FUSED_CLOCK, tools/fused_clock_probe.py):
// thread 0 of every workgroup stores the 100 MHz clock between its phase boundaries into the workgroup's own slot (plain stores: a
// shared counter would serialise 34 000 workgroups per forward on one address and measure itself).
#ifdef MPE_FUSED_CLOCK
__device__ unsigned int g_fused_stamp[16384][4];
#define FUSED_STAMP(SLOT)                                                                \
    do {                                                                                 \
        if (threadIdx.x == 0) {                                                          \
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();            \
            g_fused_stamp[blockIdx.x & 16383][SLOT] = (unsigned int)(now_ - fused_t_);   \
            fused_t_ = now_;                                                             \
        }                                                                                \
    } while (0)
#else
#define FUSED_STAMP(SLOT) do { } while (0)
#endif

template <int VEC, int G>
__global__ __launch_bounds__(256) void k_gat_fused(int V, int max_deg, int hmax, int n_cap, int m_cap,
                                                   const int32_t *__restrict__ head_off,
                                                   const int32_t *__restrict__ en_off,
                                                   const int32_t *__restrict__ slot_n,
                                                   const int32_t *__restrict__ node_off,
                                                   const int32_t *__restrict__ en_pair,
                                                   const float *__restrict__ attn_l,
                                                   const float *__restrict__ attn_r, AggArgs a, int overlap,
                                                   const uint16_t *__restrict__ head_src) {
#pragma clang fp contract(off)
    typedef float vecf __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(1024))) float s_dyn[];
    // XCD-aware order: workgroups with equal (id % 8) share an L2; the attention heads of one
    // frame read neighbouring 160-byte pieces of the same rows, so they go to the same XCD
    const int bid = blockIdx.x, nwg = gridDim.x;
#ifdef MPE_FUSED_CLOCK
    unsigned long long fused_t_ = __builtin_amdgcn_s_memrealtime();
#endif
    const int xcd = bid & 7, xq = nwg >> 3, xr = nwg & 7;
    const int vid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int f = vid / a.heads, hh = vid - f * a.heads;
    const int D = a.out_dim;
    const int hb = head_off[f], H = head_off[f + 1] - hb;
    const int eb = en_off[f], M = en_off[f + 1] - eb;
    const int N = H + M, nb = node_off[f];
    if (M <= 0) return;                             // frames without a graph produce nothing
    if (H > hmax || N > n_cap || M > m_cap) {
        // frame beyond max_heads_per_frame (k_topology raised the status flag): defined output
        if (a.score_mode)
            for (int m = threadIdx.x; m < M; m += blockDim.x) a.out[(size_t)eb + m] = 0.f;
        return;
    }
    const int Dp = D;                               // dense rows: 16-byte chunks when VEC = 4 (then D % 4 == 0)
    // LDS: the small tables first, the feature image last (1 KiB aligned, 1 KiB of slack behind it:
    // the last DMA piece of the image may run past its end)
    float *s_a1 = s_dyn;                            // [n_cap]
    float *s_a2 = s_a1 + n_cap;                     // [n_cap]
    float *s_wh = s_a2 + n_cap;                     // [hmax][max_deg] softmax weights of heads
    int *s_src = reinterpret_cast<int *>(s_wh + (size_t)hmax * max_deg);            // [hmax][max_deg]
    int *s_pair = s_src + (size_t)hmax * max_deg;                                   // [m_cap] h1 << 16 | h2
    float *s_wen = reinterpret_cast<float *>(s_pair + m_cap);                       // [m_cap][3] softmax weights of edge-nodes
    FrameTopo tp;
    tp.start = reinterpret_cast<int *>(s_wen + (size_t)m_cap * 3);                  // [V + 1]
    tp.base = tp.start + V + 1;                     // [V * V]
    int *s_sn = tp.base + V * V;                    // [V] heads per camera slot of this frame
    int *s_deg = s_sn + MPE_MAX_CAMERAS;            // [hmax] in-degree of the heads
    float *s_ft = s_dyn + fused_tables_floats(hmax, max_deg, V, n_cap, m_cap);       // [n_cap][Dp]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool l0 = a.en_const_ft2 != nullptr;
    const int c0 = hh * D;
    const int DV = D / VEC;
    const int32_t *sn = slot_n + (size_t)f * V;

    // phase 1: feature slice -> LDS (layer 0: ft2 holds head rows only, edge-nodes share one row),
    // coefficients of this attention head, pair table, topology scalars.  fp32 rows of whole 16-byte
    // chunks go straight into LDS (global_load_lds_dwordx4: a wave-instruction fills 1 KiB = 64
    // consecutive chunks of the dense image, every lane with its own source address), so all of a
    // thread's requests are in flight at once; other shapes are staged through registers, four
    // requests at a time.
    if (VEC == 4 && (overlap & 1)) {
        // Small frames with coefficients from the GEMM (the production case).  Every global load the
        // softmax phase depends on is issued FIRST, then exactly FUSED_PIECES image pieces per wave;
        // vmcnt counts in issue order, so `s_waitcnt vmcnt(FUSED_PIECES)` means "my table values are
        // here" while the image is still landing, and the softmax phase runs underneath it (raw
        // barriers).  The five loads, the eight LDS-DMA pieces and the counted wait are ONE asm
        // statement with early-clobber outputs: the compiler sees the five values defined only at its
        // end, so it can neither read, copy nor spill a destination register that has not landed, and it
        // cannot schedule anything into the window (the round-2 form used separate asm statements and
        // was correct by convention only; a run-time branch inside that window once gave NaNs in 8 % of
        // the frames).  tests/test_host_logic.py::test_fused_attention_window_isa pins the emitted ISA.
        const int total = N * DV;
        const size_t row0 = (size_t)(l0 ? hb : nb);
        const int tn = t < N ? t : N - 1, tm = t < M ? t : M - 1;
        const int n_src2 = hmax * max_deg / 2;                   // dwords (entry pairs) of the frame's source table
        const int ts = t < n_src2 ? t : n_src2 - 1;
        const float *pr_ = (l0 && tn >= H) ? a.en_const_a : a.a12 + (row0 + tn) * 32;
        const float *p1 = pr_ + hh, *p2 = pr_ + 16 + hh;
        const int32_t *pe = en_pair + 2 * (size_t)(eb + tm);
        const uint32_t *ps = reinterpret_cast<const uint32_t *>(head_src + (size_t)f * hmax * max_deg) + ts;
        const float *src[FUSED_PIECES];
#pragma unroll
        for (int u = 0; u < FUSED_PIECES; ++u) {
            int c = wave * 64 + u * 256 + lane;
            c = c < total ? c : total - 1;                   // pieces past the image repeat its last chunk (slack space)
            const int node = c / DV, d = (c - node * DV) * 4;
            src[u] = (l0 && node >= H) ? a.en_const_ft2 + (c0 + d) : a.ft2 + (row0 + node) * a.ld + c0 + d;
        }
        // LDS byte address of this wave's first piece; piece u lands 4 KiB (256 chunks) further
        const unsigned m0v = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(__attribute__((address_space(3))) float *)s_ft + (unsigned)(wave * 64) * 16u);
        float r1, r2;
        int e1, e2;
        uint32_t sv;
        unsigned m0_keep;
        static_assert(FUSED_PIECES == 8, "the asm block below issues eight pieces and waits for vmcnt(8)");
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            "global_load_dword %[r1], %[p1], off\n\t"
            "global_load_dword %[r2], %[p2], off\n\t"
            "global_load_dword %[e1], %[pe], off\n\t"
            "global_load_dword %[e2], %[pe], off offset:4\n\t"
            "global_load_dword %[sv], %[ps], off\n\t"
            "s_mov_b32 m0, %[m0v]\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %[s0], off\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %[s1], off\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %[s2], off\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %[s3], off\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %[s4], off\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %[s5], off\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %[s6], off\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %[s7], off\n\t"
            "s_mov_b32 m0, %[keep]\n\t"
            "s_waitcnt vmcnt(8)"
            : [r1] "=&v"(r1), [r2] "=&v"(r2), [e1] "=&v"(e1), [e2] "=&v"(e2), [sv] "=&v"(sv), [keep] "=&s"(m0_keep)
            : [p1] "v"(p1), [p2] "v"(p2), [pe] "v"(pe), [ps] "v"(ps), [m0v] "s"(m0v), [s0] "v"(src[0]), [s1] "v"(src[1]),
              [s2] "v"(src[2]), [s3] "v"(src[3]), [s4] "v"(src[4]), [s5] "v"(src[5]), [s6] "v"(src[6]), [s7] "v"(src[7])
            : "memory", "scc");
        if (t < N) {
            s_a1[t] = r1;
            s_a2[t] = r2;
        }
        if (t < M) s_pair[t] = (e1 << 16) | e2;
        if (t < n_src2) {
            const int lo = sv & 0xFFFF, hi = sv >> 16;
            s_src[2 * t] = lo == 0xFFFF ? -1 : lo;
            s_src[2 * t + 1] = hi == 0xFFFF ? -1 : hi;
        }
        FUSED_STAMP(0);                    // issue of the loads + landing of the table values
        lds_barrier_raw();
    } else {
    if (VEC == 4 && !a.ft_half) {
        const int total = N * DV;
        for (int c0_ = wave * 64; c0_ < total; c0_ += 256) {
            int c = c0_ + lane;
            c = c < total ? c : total - 1;                   // tail lanes repeat the last chunk (same bytes, same slot)
            const int node = c / DV, d = (c - node * DV) * 4;
            const float *src = (l0 && node >= H) ? a.en_const_ft2 + (c0 + d)
                                                  : a.ft2 + (size_t)((l0 ? hb : nb) + node) * a.ld + c0 + d;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(s_ft + (size_t)c0_ * 4), 16, 0, 0);
        }
    } else if (!a.ft_half) {
        // fp32 rows that are not whole 16-byte chunks (30-wide attention heads): the same dense image,
        // written by 4-byte LDS-DMA pieces (a wave-instruction fills 64 consecutive floats)
        const int total = N * D;
        for (int c0_ = wave * 64; c0_ < total; c0_ += 256) {
            int c = c0_ + lane;
            c = c < total ? c : total - 1;
            const int node = c / D, d = c - node * D;
            const float *src = (l0 && node >= H) ? a.en_const_ft2 + (c0 + d)
                                                  : a.ft2 + (size_t)((l0 ? hb : nb) + node) * a.ld + c0 + d;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(s_ft + (size_t)c0_), 4, 0, 0);
        }
    } else {
        const int total = N * DV;
        for (int i0 = t; i0 < total; i0 += 4 * blockDim.x) {
            float v[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int i = i0 + u * blockDim.x;
                i = i < total ? i : total - 1;
                const int node = i / DV, d = (i - node * DV) * VEC;
                if (l0 && node >= H) ld_ftv<VEC>(a.en_const_ft2, (size_t)(c0 + d), 0, v[u]);
                else ld_ftv<VEC>(a.ft2, (size_t)((l0 ? hb : nb) + node) * a.ld + c0 + d, a.ft_half, v[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * blockDim.x;
                if (i < total) {
                    const int node = i / DV, d = (i - node * DV) * VEC;
                    vecf o;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) o[k] = v[u][k];
                    *reinterpret_cast<vecf *>(s_ft + node * Dp + d) = o;
                }
            }
        }
    }
    if (a.a12_ready) {
        for (int node = t; node < N; node += blockDim.x) {
            const float *r = (l0 && node >= H) ? a.en_const_a : a.a12 + (size_t)((l0 ? hb : nb) + node) * 32;
            s_a1[node] = r[hh];
            s_a2[node] = r[16 + hh];
        }
    }
    for (int m = t; m < M; m += blockDim.x)
        s_pair[m] = (en_pair[2 * (size_t)(eb + m)] << 16) | en_pair[2 * (size_t)(eb + m) + 1];
    if (head_src) {
        const uint16_t *tab = head_src + (size_t)f * hmax * max_deg;
        for (int i = t; i < H * max_deg; i += blockDim.x) {
            const int u = tab[i];
            s_src[i] = u == 0xFFFF ? -1 : u;
        }
        __syncthreads();
    } else {
        if (t < V) s_sn[t] = sn[t];
        __syncthreads();
        build_topo(tp, s_sn, V, H, t, blockDim.x);
        __syncthreads();
        for (int i = t; i < H * max_deg; i += blockDim.x) {
            const int h = i / max_deg, e = i - h * max_deg;
            const int sl = slot_of_head(tp, V, h);
            s_src[i] = e < 1 + H - s_sn[sl] ? head_in_edge(tp, s_sn, V, h, sl, e) : -1;
        }
        __syncthreads();
    }
    }
    if (!a.a12_ready) {
        // a1 = <ft, attn_l[head]>, a2 = <ft, attn_r[head]> (the GEMM epilogue did not provide them)
        for (int node = t; node < N; node += blockDim.x) {
            const float *fv = s_ft + node * Dp;
            float x1 = 0.f, x2 = 0.f;
            if (D == 40) {
                coef40([&](int d) { return fv[d]; }, attn_l + c0, attn_r + c0, hh & 1, x1, x2);
            } else {
                for (int d = 0; d < D; ++d) {
                    x1 = __builtin_fmaf(fv[d], attn_l[c0 + d], x1);
                    x2 = __builtin_fmaf(fv[d], attn_r[c0 + d], x2);
                }
            }
            s_a1[node] = x1;
            s_a2[node] = x2;
        }
        __syncthreads();
    }
    // phase 2: softmax weights.  Edge-node X: in-edges (h1, h2, X), one thread each.  Head h: one
    // lane group per head, fixed-shape reductions.
    for (int m = t; m < M; m += blockDim.x) {
        const int pr = s_pair[m];
        const int h1 = pr >> 16, h2 = pr & 0xFFFF, x = H + m;
        const float a2v = s_a2[x];
        float e1 = s_a1[h1] + a2v, e2 = s_a1[h2] + a2v, e3 = s_a1[x] + a2v;
        e1 = e1 > 0.f ? e1 : e1 * a.alpha;
        e2 = e2 > 0.f ? e2 : e2 * a.alpha;
        e3 = e3 > 0.f ? e3 : e3 * a.alpha;
        const float mx = fmaxf(fmaxf(e1, e2), e3);
        const float x1 = expf(e1 - mx), x2 = expf(e2 - mx), x3 = expf(e3 - mx);
        const float sum = (x1 + x2) + x3;
        s_wen[m * 3 + 0] = x1 / sum;
        s_wen[m * 3 + 1] = x2 / sum;
        s_wen[m * 3 + 2] = x3 / sum;
    }
    if (!a.score_mode || a.out_heads) {
        const int gl = t & (G - 1);
        for (int h = t / G; h < H; h += 256 / G) {
            float *w = s_wh + h * max_deg;
            const int *src = s_src + h * max_deg;           // in-edge sources, -1 behind the in-degree
            const float a2v = s_a2[h];
            float mx = -INFINITY;
            int deg = 0;
            for (int e = gl; e < max_deg; e += G) {
                const int u = src[e];
                if (u >= 0) {
                    float x = s_a1[u] + a2v;
                    x = x > 0.f ? x : x * a.alpha;
                    w[e] = x;
                    mx = fmaxf(mx, x);
                    deg = e + 1;
                }
            }
            mx = group_max<G>(mx);
            deg = (int)group_max<G>((float)deg);
            float sum = 0.f;
            for (int e = gl; e < deg; e += G) {
                const float ex = expf(w[e] - mx);
                w[e] = ex;
                sum = sum + ex;
            }
            sum = group_sum<G>(sum);
            for (int e = gl; e < deg; e += G) w[e] = w[e] / sum;
            if (gl == 0) s_deg[h] = deg;
        }
    }
    FUSED_STAMP(1);                        // softmax phase (under the landing image)
    __syncthreads();
    FUSED_STAMP(2);                        // what was left of the image's landing + the barrier
    // phase 3: weighted sums in edge order, activation, store
    const int first = (a.score_mode && !a.out_heads) ? H : 0;
    if (VEC == 4 && !a.score_mode && (DV & 1) == 0 && !(overlap & 2)) {
        // The production shape (40- / 80-wide heads, feature rows out).  An item of the edge-node rows is TWO 16-byte column groups
        // of one row: its pair, its three weights and its index arithmetic are fetched once for eight columns instead of once for
        // four -- measured with in-kernel stamps (tools/fused_clock_probe.py, profiles/r06_fused_phases.txt): phase 3 is 6.5 of a
        // workgroup's 17.3 us and is instruction-bound (seven passes of ~70 instructions per thread), the image has long landed.
        // Head rows (up to 17 sources per item) keep one group per item.  Per column the arithmetic is what it was: same bits.
        const int DV2 = DV / 2;
        const int n_head_items = H * DV, n_items = n_head_items + M * DV2;
        for (int i = t; i < n_items; i += blockDim.x) {
            if (i >= n_head_items) {
                const int j = i - n_head_items;
                const int m = j / DV2, d = (j - m * DV2) * 8;
                const int node = H + m;
                const int pr = s_pair[m];
                const int h1 = pr >> 16, h2 = pr & 0xFFFF;
                const float w1 = s_wen[m * 3 + 0], w2 = s_wen[m * 3 + 1], w3 = s_wen[m * 3 + 2];
                const float *p1 = s_ft + h1 * Dp + d, *p2 = s_ft + h2 * Dp + d, *p3 = s_ft + node * Dp + d;
                float *dst = a.out + (size_t)(nb + node) * a.ld_out + c0 + d;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const vecf f1 = *reinterpret_cast<const vecf *>(p1 + 4 * g);
                    const vecf f2 = *reinterpret_cast<const vecf *>(p2 + 4 * g);
                    const vecf f3 = *reinterpret_cast<const vecf *>(p3 + 4 * g);
                    vecf o;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float acc = f1[k] * w1;
                        acc = acc + f2[k] * w2;
                        acc = acc + f3[k] * w3;
                        o[k] = agg_activate(acc, a.out_mode, a.out_slope);
                    }
                    *reinterpret_cast<vecf *>(dst + 4 * g) = o;
                }
            } else {
                const int node = i / DV, d = (i - node * DV) * VEC;
                const int deg = s_deg[node];
                const int *src = s_src + node * max_deg;
                const float *w = s_wh + node * max_deg;
                float acc[VEC];
                weighted_sum8<VEC>(
                    deg,
                    [&](int e, float *fv) {
                        const vecf x = *reinterpret_cast<const vecf *>(s_ft + src[e] * Dp + d);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) fv[k] = x[k];
                    },
                    [&](int e) { return w[e]; }, acc);
                vecf o;
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = agg_activate(acc[k], a.out_mode, a.out_slope);
                *reinterpret_cast<vecf *>(a.out + (size_t)(nb + node) * a.ld_out + c0 + d) = o;
            }
        }
        FUSED_STAMP(3);
        return;
    }
    for (int i = first * DV + t; i < N * DV; i += blockDim.x) {
        const int node = i / DV, d = (i - node * DV) * VEC;
        vecf o;
        if (node >= H) {
            const int m = node - H;
            const int pr = s_pair[m];
            const int h1 = pr >> 16, h2 = pr & 0xFFFF;
            const float w1 = s_wen[m * 3 + 0], w2 = s_wen[m * 3 + 1], w3 = s_wen[m * 3 + 2];
            const vecf f1 = *reinterpret_cast<const vecf *>(s_ft + h1 * Dp + d);
            const vecf f2 = *reinterpret_cast<const vecf *>(s_ft + h2 * Dp + d);
            const vecf f3 = *reinterpret_cast<const vecf *>(s_ft + node * Dp + d);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float acc = f1[k] * w1;
                acc = acc + f2[k] * w2;
                acc = acc + f3[k] * w3;
                o[k] = agg_activate(acc, a.out_mode, a.out_slope);
            }
        } else {
            const int deg = s_deg[node];
            const int *src = s_src + node * max_deg;
            const float *w = s_wh + node * max_deg;
            float acc[VEC];
            weighted_sum8<VEC>(
                deg,
                [&](int e, float *fv) {
                    const vecf x = *reinterpret_cast<const vecf *>(s_ft + src[e] * Dp + d);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) fv[k] = x[k];
                },
                [&](int e) { return w[e]; }, acc);
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = agg_activate(acc[k], a.out_mode, a.out_slope);
        }
        if (a.score_mode) {
            if (node >= H) a.out[(size_t)eb + (node - H)] = o[0];
            else a.out_heads[(size_t)hb + node] = o[0];
        } else {
            *reinterpret_cast<vecf *>(a.out + (size_t)(nb + node) * a.ld_out + c0 + d) = o;
        }
    }
    FUSED_STAMP(3);                        // phase 3: weighted sums, activation, stores issued
}

#ifdef MPE_FUSED_CLOCK
}  // namespace mpe
extern "C" int mpe_debug_fused_stamps(unsigned int *out, int clear) {            // [16384][4]      // diagnostic builds only; not part of include/mpe.h
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mpe::g_fused_stamp), sizeof(mpe::g_fused_stamp)) != hipSuccess) return -1;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(mpe::g_fused_stamp)) != hipSuccess || hipMemset(p, 0, sizeof(mpe::g_fused_stamp)) != hipSuccess) return -1;
    }
    return 0;
}
namespace mpe {
#endif

// LDS bytes of k_gat_fused for frames of up to `hmax` heads.  Implicit topology: at most hmax^2 (V-1) / 2V edge-nodes
// and hmax + 1 in-edges per head; explicit pair lists (x_m_cap > 0): the caller's caps.
static size_t fused_lds_bytes(int hmax, int V, int out_dim, int *n_cap, int *m_cap, int *max_deg, int x_deg_cap, int x_m_cap) {
    const int mc = x_m_cap > 0 ? x_m_cap : hmax * hmax * (V - 1) / (2 * V) + 1;
    const int nc = hmax + mc;
    const int deg = x_m_cap > 0 ? x_deg_cap : hmax + 1;
    const int Dp = out_dim;
    size_t image = (size_t)nc * Dp + 256;
    if (out_dim % 4 == 0 && image < (size_t)FUSED_PIECES * 1024)
        image = (size_t)FUSED_PIECES * 1024;                                           // overlapped staging issues whole pieces
    const size_t bytes = (fused_tables_floats(hmax, deg, V, nc, mc) + image) * sizeof(float);
    *n_cap = nc;
    *m_cap = mc;
    *max_deg = deg;
    return bytes;
}

static int agg_vec(const AggArgs &a) {
    if (!a.score_mode && a.ld % 4 == 0 && a.ld_out % 4 == 0) {
        if (a.out_dim % 4 == 0) return 4;
        if (a.out_dim % 2 == 0) return 2;
    }
    return 1;
}

// fp16 rows in the general kernels: 8 columns per thread (one 16-byte load) or 4 (8-byte loads, twice the threads per row)
static bool half_vec8() {
    static const bool v = !(getenv("MPE_HALF_VEC") && atoi(getenv("MPE_HALF_VEC")) == 4);
    return v;
}

constexpr size_t FUSED_LDS_LIMIT = 160 * 1024;      // all of a CU's LDS (one workgroup per CU at the limit)

hipError_t launch_gat_attention(hipStream_t s, const mpe_batch &b, int V, int max_heads_per_frame,
                                const int32_t *node_off, const int32_t *head_frame, const int32_t *en_frame,
                                const int32_t *en_pair, const float *attn_l, const float *attn_r, float *a12,
                                const AggArgs &a, int n_rows_ft2, const uint16_t *head_src, int x_deg_cap, int x_m_cap) {
    int n_cap, m_cap, max_deg;
    const bool xpl = b.d_en_pair != nullptr;              // explicit pair list: the in-edge sources exist as a table only
    if (xpl && (!head_src || x_m_cap <= 0)) return hipErrorInvalidValue;
    if (!xpl) x_deg_cap = x_m_cap = 0;
    if (!xpl && getenv("MPE_NO_HEAD_SRC_TABLE")) head_src = nullptr;       // read per call: tests toggle it
    const size_t shm = fused_lds_bytes(max_heads_per_frame, V, a.out_dim, &n_cap, &m_cap, &max_deg, x_deg_cap, x_m_cap);
    const bool no_fuse = getenv("MPE_NO_FUSED_ATTENTION") != nullptr;       // read per call: tests toggle it
    if (shm <= FUSED_LDS_LIMIT && !no_fuse && b.n_frames > 0) {
        const int vec = agg_vec(a);
        // lane-group width of the head softmax: the largest in-degree a head can have
        // A lane takes the in-edges j, j + G, ... in ascending order before the butterfly.  With at most 2 G in-edges that is the
        // canonical 64-lane tree itself (its first level adds lane j to lane j + 32, its second the result to lane j + 16's: for
        // in-degrees up to 32 the first addition of lane j is x_j + x_(j+16), whichever form runs), so sixteen lanes serve heads of
        // up to 32 in-edges and thirty-two lanes heads of up to 64 -- sixteen heads per pass instead of eight at 5 x 4
        // (tests/test_gpu_stages.py::test_attention_paths_give_identical_bits compares with the 64-lane general kernel).
        const int grp = max_deg <= 32 ? 16 : max_deg <= 64 ? 32 : 64;
        // overlapped staging: coefficients from the GEMM, fp32 rows of 16-byte chunks, tables of at most one
        // entry per thread, image of at most FUSED_PIECES x 256 chunks
        if (!xpl && !head_src_entries(max_heads_per_frame, V)) head_src = nullptr;
        // (the source table then travels as at most one dword = two entries per thread)
        // (bit 1: MPE_FUSED_NO_OVERLAP also selects the one-group-per-item form of phase 3 -- the switch is the kernel's plain form)
        const int overlap = ((!xpl && vec == 4 && a.a12_ready && !a.ft_half && n_cap <= 256 && m_cap <= 256 && head_src &&
                              max_heads_per_frame * (max_heads_per_frame + 1) <= 512 &&
                              n_cap * (a.out_dim / 4) <= FUSED_PIECES * 256 && !getenv("MPE_FUSED_NO_OVERLAP")) ? 1 : 0) |
                            (getenv("MPE_FUSED_NO_OVERLAP") ? 2 : 0);
        const void *fn = nullptr;
#define MPE_FUSED(V_, G_)                                                                                     \
    do {                                                                                                      \
        fn = reinterpret_cast<const void *>(k_gat_fused<V_, G_>);                                             \
        if (shm > 64 * 1024) {                                                                                \
            hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS_LIMIT); \
            if (e_ != hipSuccess) return e_;                                                                  \
        }                                                                                                     \
        hipLaunchKernelGGL((k_gat_fused<V_, G_>), dim3(b.n_frames * a.heads), dim3(256), shm, s, V,            \
                           max_deg, max_heads_per_frame, n_cap, m_cap, b.d_frame_head_off, b.d_frame_en_off, b.d_slot_n, \
                           node_off, en_pair, attn_l, attn_r, a, overlap, head_src);                          \
    } while (0)
#define MPE_FUSED_G(V_)                     \
    do {                                    \
        if (grp == 16) MPE_FUSED(V_, 16);   \
        else if (grp == 32) MPE_FUSED(V_, 32); \
        else MPE_FUSED(V_, 64);             \
    } while (0)
        if (vec == 4) MPE_FUSED_G(4);
        else if (vec == 2) MPE_FUSED_G(2);
        else MPE_FUSED_G(1);
#undef MPE_FUSED_G
#undef MPE_FUSED
        return hipGetLastError();
    }
    AggArgs a2 = a;
    a2.a12 = a12;
    if (!a.a12_ready) {
        hipError_t e = launch_attn_coef(s, a.ft2, a.ld, n_rows_ft2, a.heads, a.out_dim, attn_l, attn_r, a12, a.ft_half);
        if (e != hipSuccess) return e;
        a2.a12_ready = 1;                  // (the general kernels compute the coefficients themselves when this reads 0: small batches only)
    }
    return launch_aggregate(s, b, V, max_heads_per_frame, node_off, head_frame, en_frame, en_pair, a2, head_src, x_deg_cap);
}

// the general kernels' two halves as one launch (small batches; implicit topology)
hipError_t launch_lat_attention(hipStream_t s, const mpe_batch &b, int V, int max_heads_per_frame, const int32_t *node_off,
                                const int32_t *head_frame, const int32_t *en_frame, const int32_t *en_pair, const AggArgs &a,
                                const uint16_t *head_src) {
    if (b.d_en_pair || a.ft_half || (!a.a12_ready && (!a.attn_l || !a.attn_r))) return hipErrorInvalidValue;
    const int vec = agg_vec(a);
    if (!head_src_entries(max_heads_per_frame, V)) head_src = nullptr;
    const int n_en_wgs = (b.n_edge_nodes + LAT_EN_ROWS - 1) / LAT_EN_ROWS;
    const bool heads_on = b.n_heads > 0 && (!a.score_mode || a.out_heads);
    const int n_h_wgs = heads_on ? (b.n_heads + AGG_ROWS - 1) / AGG_ROWS : 0;
    if (n_en_wgs + n_h_wgs == 0) return hipSuccess;
    const int src_stride = max_heads_per_frame + 1;
    int max_deg = src_stride < 3 ? 3 : src_stride;
    const size_t shm = (size_t)AGG_ROWS * ((size_t)a.heads * max_deg + max_deg + (V + 1) + (size_t)V * V + 16) * sizeof(float);
    if (shm > 48 * 1024) return hipErrorInvalidValue;                  // (the caller keeps such rigs on the batch path)
    const int hd4 = (a.heads * a.out_dim + 3) / 4 * 4;
    // (one vector width for both halves: the head half needs whole heads per group; per-column arithmetic does not depend on it)
    (void)hd4;
#define MPE_LA(V_)                                                                                                              \
    hipLaunchKernelGGL(k_lat_attention<V_>, dim3(n_en_wgs + n_h_wgs), dim3(256), shm, s, n_en_wgs, b.n_edge_nodes, b.n_heads, V, max_deg, \
                       b.d_frame_head_off, b.d_frame_en_off, b.d_slot_n, node_off, head_frame, en_frame, en_pair, a, head_src, src_stride, \
                       max_heads_per_frame)
    if (vec == 4) MPE_LA(4);
    else if (vec == 2) MPE_LA(2);
    else MPE_LA(1);
#undef MPE_LA
    return hipGetLastError();
}

hipError_t launch_aggregate(hipStream_t s, const mpe_batch &b, int V, int max_heads_per_frame,
                            const int32_t *node_off, const int32_t *head_frame, const int32_t *en_frame,
                            const int32_t *en_pair, const AggArgs &a, const uint16_t *head_src, int x_deg_cap) {
    const int vec = agg_vec(a);
    const bool xpl = b.d_en_pair != nullptr;
    if (xpl && (!head_src || x_deg_cap <= 0)) return hipErrorInvalidValue;
    if (!xpl && !head_src_entries(max_heads_per_frame, V)) head_src = nullptr;
    if (b.n_edge_nodes > 0) {
        const unsigned blocks = (unsigned)((b.n_edge_nodes + EN_ROWS - 1) / EN_ROWS);
#define MPE_EN(V_)                                                                                      \
    hipLaunchKernelGGL(k_aggregate_en<V_>, dim3(blocks), dim3(256), 0, s, b.n_edge_nodes, b.d_frame_head_off, \
                       b.d_frame_en_off, node_off, en_frame, en_pair, a, max_heads_per_frame)
        // the edge-node kernel takes 16-byte groups whatever the head width (a group may straddle two
        // heads), as long as the rounded-up row fits the row strides
        const int hd4 = (a.heads * a.out_dim + 3) / 4 * 4;
        const bool wide = !a.score_mode && !a.ft_half && a.ld % 4 == 0 && a.ld_out % 4 == 0 && hd4 <= a.ld && hd4 <= a.ld_out &&
                          !a.en_const_ft2;
        if (vec == 4 && a.ft_half && a.out_dim % 8 == 0 && half_vec8()) MPE_EN(8);    // fp16 rows: 8 columns = one 16-byte load
        else if (vec == 4 || wide) MPE_EN(4);
        else if (vec == 2) MPE_EN(2);
        else MPE_EN(1);
#undef MPE_EN
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (b.n_heads > 0 && (!a.score_mode || a.out_heads)) {
        const int src_stride = xpl ? x_deg_cap : max_heads_per_frame + 1;     // row stride of the in-edge source table
        int max_deg = src_stride;
        if (max_deg < 3) max_deg = 3;
        const size_t shm = (size_t)AGG_ROWS * ((size_t)a.heads * max_deg + max_deg + (V + 1) + (size_t)V * V + 16) * sizeof(float);
        if (shm > 64 * 1024) {
            static PerDeviceFlag attr;
            if (!attr.test()) {
                hipError_t e = hipSuccess;
                const void *fns[4] = {reinterpret_cast<const void *>(k_aggregate_heads<8>), reinterpret_cast<const void *>(k_aggregate_heads<4>),
                                      reinterpret_cast<const void *>(k_aggregate_heads<2>), reinterpret_cast<const void *>(k_aggregate_heads<1>)};
                for (const void *fn : fns)
                    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS_LIMIT);
                if (e != hipSuccess) return e;
                attr.set();
            }
        }
        const int grid = (b.n_heads + AGG_ROWS - 1) / AGG_ROWS;
#define MPE_HEADS(V_)                                                                                   \
    hipLaunchKernelGGL(k_aggregate_heads<V_>, dim3(grid), dim3(256), shm, s, b.n_heads, V, max_deg,         \
                       b.d_frame_head_off, b.d_frame_en_off, b.d_slot_n, node_off, head_frame, a, head_src,       \
                       src_stride, max_heads_per_frame)
        if (vec == 4 && a.ft_half && a.out_dim % 8 == 0 && half_vec8()) MPE_HEADS(8);
        else if (vec == 4) MPE_HEADS(4);
        else if (vec == 2) MPE_HEADS(2);
        else MPE_HEADS(1);
#undef MPE_HEADS
    }
    return hipGetLastError();
}

}  // namespace mpe
