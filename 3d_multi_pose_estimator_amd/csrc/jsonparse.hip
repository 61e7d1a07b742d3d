// Device-side second-level parse of the frame JSON ("on-device parser/packer").
//
// Wire format (reference panoptic_conversor/get_joints_from_panoptic_model_multi.py:231-236,287;
// consumers test/metrics_from_model.py:128-140,182-191): a list of frames, frame = {camera: [ "<JSON
// text of the skeleton list>", timestamp, 'no_image', bodies_3D ]}, skeleton = {"<joint id>": [id, x,
// y, valid, prob], ..., optional "ID": n}.  The reference parses every frame twice in Python.  Here the
// host keeps the first level only (frame extents by the AVX2 scanner, then per frame the camera keys
// and the extent of each used camera's skeleton STRING -- packer.cpp:mpe_json_stage_window, which also
// copies those strings, ~55 % of the bytes, into one page-locked staging buffer); the strings travel to
// the device in one copy and are parsed there into the nine arrays of mpe_batch:
//
//   k_json_braces    one WAVE per string: every lane takes 16 bytes of a 1 KiB piece, the braces are
//                    found by byte compares, numbered by a wave prefix sum -> the extent of every
//                    skeleton object of the string (skeleton k = k-th '{' .. k-th '}')
//   k_json_skeleton  one THREAD per skeleton: walks its members -- the host parser's grammar, statement
//                    by statement -- converts the numbers (exact decimal -> binary64, el_double.h) and
//                    writes a row of a staging arena indexed by (string, k); also checks the list
//                    syntax between its skeleton and the next
//   k_json_layout    per frame: slot tables, heads (= skeletons with >= 1 joint), edge-nodes; prefix sums
//   k_json_compact   rows of the non-empty skeletons -> the batch arrays, in the reference's head order
//                    (camera key order, then list order: graph_generator.py:583-601)
//
// The device accepts the CANONICAL shapes only (what json.dumps writes: digit keys, "ID" with a
// scalar, lists of >= 5 plain numbers of <= 19 significant digits, no braces inside values); anything
// else -- literals, nested values, numbers the exact fast path declines, more skeletons in one string
// than the staging arena has rows for -- raises a status bit and the caller packs that window with the
// host packer, which remains the definition of the accepted language and of the error messages.
// tests/test_gpu_json.py: the arrays equal the host packer's bit for bit.
#include <cstdlib>

#include "el_double.h"
#include "mpe_internal.h"

namespace mpe {

enum { JS_OK = 0, JS_FALLBACK = 1, JS_CAPACITY = 2 };

// ---- byte cursor with an 8-byte register window (the strings are 16-byte aligned in the staging buffer) ----
struct JCur {
    const char *t;          // 8-byte aligned base
    int i, n;
    int wb;                 // window base (multiple of 8), -8 = empty
    uint64_t w;
    __device__ JCur(const char *t_, int i_, int n_) : t(t_), i(i_), n(n_), wb(-8), w(0) {}
    __device__ int at(int k) {
        if (k >= n || k < 0) return 0;
        const int b = k & ~7;
        if (b != wb) {
            w = *reinterpret_cast<const uint64_t *>(t + b);     // may read up to 7 bytes past n: inside the padded staging buffer
            wb = b;
        }
        return (int)((w >> ((k & 7) * 8)) & 0xFF);
    }
    __device__ int ch() { return at(i); }
    __device__ int ch1() { return at(i + 1); }
    __device__ void ws() {
        for (;;) {
            const int c = at(i);
            if (c == ' ' || c == '\n' || c == '\t' || c == '\r') ++i;
            else break;
        }
    }
    __device__ bool quote() { return at(i) == '\\' && at(i + 1) == '"'; }      // inner text: a quote is \"
    // byte source of el_parse_number_src
    __device__ int peek(int k) { return at(i + k); }
    __device__ void skip(int k) { i += k; }
};

// One skeleton object, text[b .. e) = "{ ... }".  Returns false if it is not canonical.  Writes the row `r` of the
// staging arena (xy / vp rows are cleared first only when a joint is seen: empty skeletons leave garbage, flagged by jm = 0).
__device__ bool walk_skeleton(const char *t, int b, int e, int J, uint32_t *jm_out, uint32_t *tm_out, double *xy, float *vp) {
    JCur c(t, b, e);
    if (c.ch() != '{') return false;
    ++c.i;
    uint32_t jm = 0, tm = 0;
    c.ws();
    if (c.ch() == '}') {
        ++c.i;
    } else {
        for (;;) {
            c.ws();
            if (!c.quote()) return false;
            c.i += 2;
            int j = 0, kd = 0;
            bool is_id = false;
            if (c.ch() == 'I' && c.ch1() == 'D') {
                is_id = true;
                c.i += 2;
            } else {
                for (int d = c.ch(); d >= '0' && d <= '9' && kd < 9; d = c.ch()) {
                    j = j * 10 + (d - '0');
                    ++kd;
                    ++c.i;
                }
                if (kd == 0) return false;
            }
            if (!c.quote()) return false;
            c.i += 2;
            c.ws();
            if (c.ch() != ':') return false;
            ++c.i;
            c.ws();
            if (is_id) {
                if (c.quote()) {                              // a plain string
                    c.i += 2;
                    for (int d = c.ch(); d != 0 && d != '\\' && d != '"'; d = c.ch()) ++c.i;
                    if (!c.quote()) return false;
                    c.i += 2;
                } else {
                    double d;
                    bool ok;
                    if (el_parse_number_src(c, &d, &ok) == 0) return false;
                }
            } else {
                if (j >= J) return false;
                if (c.ch() != '[') return false;
                ++c.i;
                double v[5] = {0, 0, 0, 0, 0};
                int cnt = 0;
                for (;;) {
                    c.ws();
                    if (c.ch() == ']') {
                        ++c.i;
                        break;
                    }
                    double d = 0;
                    bool ok = true;
                    if (el_parse_number_src(c, &d, &ok) == 0 || !ok) return false;
                    if (cnt < 5) v[cnt] = d;
                    ++cnt;
                    c.ws();
                    const int nx = c.ch();
                    if (nx == ',') ++c.i;
                    else if (nx != ']') return false;
                }
                if (cnt < 5) return false;
                if (jm == 0) {
                    for (int q = 0; q < J * 2; ++q) {
                        xy[q] = 0.0;
                        vp[q] = 0.f;
                    }
                }
                xy[j * 2 + 0] = v[1];
                xy[j * 2 + 1] = v[2];
                vp[j * 2 + 0] = (float)v[3];
                vp[j * 2 + 1] = (float)v[4];
                jm |= 1u << j;
                if (v[0] > 0.) tm |= 1u << j;
            }
            c.ws();
            if (c.ch() == ',') {
                ++c.i;
                continue;
            }
            if (c.ch() == '}') {
                ++c.i;
                break;
            }
            return false;
        }
    }
    if (c.i != e) return false;                              // the object closes where the brace scan said (no nested braces)
    *jm_out = jm;
    *tm_out = tm;
    return true;
}

// text[b .. e) must be blanks around exactly one `sep` (0 = blanks only)
__device__ bool gap_is(const char *t, int n, int b, int e, int sep) {
    JCur c(t, b, n);
    c.ws();
    if (sep) {
        if (c.i >= e || c.ch() != sep) return false;
        ++c.i;
        c.ws();
    }
    return c.i == e;
}

struct SkExt {
    int32_t b, e;
};

__global__ __launch_bounds__(64) void k_json_braces(const char *__restrict__ text, const mpe_json_entry *__restrict__ entries,
                                                     int n_entries, int kcap, SkExt *__restrict__ ext, int32_t *__restrict__ n_sk,
                                                     int32_t *__restrict__ totals) {
    const int e = blockIdx.x, lane = threadIdx.x;
    if (e >= n_entries) return;
    const mpe_json_entry en = entries[e];
    const char *t = text + en.begin;
    const int n = (int)(en.end - en.begin);
    SkExt *my = ext + (size_t)e * kcap;
    int n_open = 0, n_close = 0;
    for (int base = 0; base < n; base += 1024) {
        const int off = base + lane * 16;
        uint32_t wv[4] = {0, 0, 0, 0};
        if (off < n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(t + off);     // strings start 16-byte aligned; the tail lies inside the padded buffer
            wv[0] = v.x; wv[1] = v.y; wv[2] = v.z; wv[3] = v.w;
        }
        uint32_t mo = 0, mc = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t by = (wv[k >> 2] >> ((k & 3) * 8)) & 0xFF;
            if (off + k < n) {
                mo |= (by == '{' ? 1u : 0u) << k;
                mc |= (by == '}' ? 1u : 0u) << k;
            }
        }
        int po = __popc(mo), pc = __popc(mc);
        int so = po, sc = pc;                                             // inclusive wave scans
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int a = __shfl_up(so, d), b2 = __shfl_up(sc, d);
            if (lane >= d) {
                so += a;
                sc += b2;
            }
        }
        int io = n_open + so - po, ic = n_close + sc - pc;
        while (mo) {
            const int k = __ffs(mo) - 1;
            mo &= mo - 1;
            if (io < kcap) my[io].b = off + k;
            ++io;
        }
        while (mc) {
            const int k = __ffs(mc) - 1;
            mc &= mc - 1;
            if (ic < kcap) my[ic].e = off + k + 1;
            ++ic;
        }
        n_open += __shfl(so, 63);
        n_close += __shfl(sc, 63);
    }
    if (lane == 0) {
        n_sk[e] = n_open < kcap ? n_open : kcap;
        if (n_open != n_close || n_open > kcap) atomicOr(&totals[2], JS_FALLBACK);
    }
}

// thread (entry, k): skeleton k of the string; also the list syntax around it.  The (entry, k) pairs are walked k-major
// (gid = k * n_entries + entry): strings hold a few skeletons each, far fewer than the capacity kcap, so the live pairs are
// the first n_entries * (a few) indices and whole waves are either live or skip at once; grid-stride over a capped grid
// (side_grid below) because the kernel shares the GPU with the previous window's GEMMs.
__global__ __launch_bounds__(256) void k_json_skeleton(const char *__restrict__ text, const mpe_json_entry *__restrict__ entries,
                                                        int n_entries, int kcap, int J, const SkExt *__restrict__ ext,
                                                        const int32_t *__restrict__ n_sk, uint32_t *__restrict__ r_jm,
                                                        uint32_t *__restrict__ r_tm, double *__restrict__ r_xy, float *__restrict__ r_vp,
                                                        int32_t *__restrict__ totals) {
    const long total = (long)n_entries * kcap;
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
        const int k = (int)(gid / n_entries), e = (int)(gid - (long)k * n_entries);
        const int cnt = n_sk[e];
        if (k >= (cnt > 0 ? cnt : 1)) continue;
        const mpe_json_entry en = entries[e];
        const char *t = text + en.begin;
        const int n = (int)(en.end - en.begin);
        bool ok = true;
        if (cnt == 0) {
            // "[]" with blanks
            JCur c(t, 0, n);
            c.ws();
            ok = c.ch() == '[';
            ++c.i;
            c.ws();
            ok = ok && c.ch() == ']';
            ++c.i;
            c.ws();
            ok = ok && c.i == n;
        } else {
            const SkExt *my = ext + (size_t)e * kcap;
            const int b = my[k].b, en_ = my[k].e;
            ok = b < en_ && en_ <= n;
            if (ok && k == 0) ok = gap_is(t, n, 0, b, '[');
            const size_t r = (size_t)e * kcap + k;
            uint32_t jm = 0, tm = 0;
            if (ok) ok = walk_skeleton(t, b, en_, J, &jm, &tm, r_xy + r * J * 2, r_vp + r * J * 2);
            // every path writes its row: a skeleton rejected before the walk must not leave the masks of the PREVIOUS window
            // in the scratch (k_json_layout would count them as heads and could report a capacity overflow instead of the
            // hand-over to the host parser)
            r_jm[r] = ok ? jm : 0;
            r_tm[r] = ok ? tm : 0;
            if (ok) ok = k + 1 < cnt ? (my[k + 1].b >= en_ && gap_is(t, n, en_, my[k + 1].b, ',')) : gap_is(t, n, en_, n, ']');
        }
        if (!ok) atomicOr(&totals[2], JS_FALLBACK);
    }
}

// per frame: slot tables, heads and edge-nodes; exclusive prefix sums over the frames; head base per entry
__global__ __launch_bounds__(1024) void k_json_layout(const mpe_json_entry *__restrict__ entries,
                                                       const int32_t *__restrict__ frame_entry_off, int n_frames, int V, int kcap,
                                                       const int32_t *__restrict__ n_sk, const uint32_t *__restrict__ r_jm, int head_cap,
                                                       int32_t *__restrict__ frame_head_off, int32_t *__restrict__ frame_en_off,
                                                       int32_t *__restrict__ slot_cam, int32_t *__restrict__ slot_n,
                                                       int32_t *__restrict__ head_base, int32_t *__restrict__ totals) {
    __shared__ long long s_h[1024], s_e[1024];
    __shared__ int s_max[1024];
    const int t = threadIdx.x, per = (n_frames + 1023) / 1024;
    const int f0 = t * per < n_frames ? t * per : n_frames, f1 = f0 + per < n_frames ? f0 + per : n_frames;
    auto heads_of = [&](int e) {
        int c = 0;
        const int cnt = n_sk[e];
        for (int k = 0; k < cnt; ++k) c += r_jm[(size_t)e * kcap + k] != 0;
        return c;
    };
    long long hs = 0, es = 0;
    int mx = 0;
    for (int f = f0; f < f1; ++f) {
        const int e0 = frame_entry_off[f], ne = frame_entry_off[f + 1] - e0;
        long long h = 0, sq = 0;
        for (int s = 0; s < V; ++s) {
            const int nn = s < ne ? heads_of(e0 + s) : 0;
            slot_cam[(size_t)f * V + s] = s < ne ? entries[e0 + s].cam : -1;
            slot_n[(size_t)f * V + s] = nn;
            h += nn;
            sq += (long long)nn * nn;
        }
        hs += h;
        es += (h * h - sq) / 2;                                // pairs of skeletons from different cameras
        mx = mx > (int)h ? mx : (int)h;
    }
    s_h[t] = hs;
    s_e[t] = es;
    s_max[t] = mx;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                       // inclusive scan of the per-thread sums
        const long long a = t >= d ? s_h[t - d] : 0, b = t >= d ? s_e[t - d] : 0;
        const int m = t >= d ? s_max[t - d] : 0;
        __syncthreads();
        s_h[t] += a;
        s_e[t] += b;
        s_max[t] = s_max[t] > m ? s_max[t] : m;
        __syncthreads();
    }
    long long hb = s_h[t] - hs, eb = s_e[t] - es;              // exclusive
    for (int f = f0; f < f1; ++f) {
        const int e0 = frame_entry_off[f], ne = frame_entry_off[f + 1] - e0;
        frame_head_off[f] = (int32_t)hb;
        frame_en_off[f] = (int32_t)eb;
        long long h = 0, sq = 0;
        for (int s = 0; s < ne; ++s) {
            head_base[e0 + s] = (int32_t)(hb + h);
            const int nn = slot_n[(size_t)f * V + s];
            h += nn;
            sq += (long long)nn * nn;
        }
        hb += h;
        eb += (h * h - sq) / 2;
    }
    if (t == 1023) {
        frame_head_off[n_frames] = (int32_t)s_h[1023];
        frame_en_off[n_frames] = (int32_t)s_e[1023];
        totals[0] = (int32_t)s_h[1023];
        totals[1] = (int32_t)s_e[1023];
        totals[3] = s_max[1023];
        if (s_h[1023] > head_cap || s_e[1023] > 0x7FFFFFFFLL) atomicOr(&totals[2], JS_CAPACITY);
    }
}

// staging rows of the non-empty skeletons -> batch arrays in head order
__global__ __launch_bounds__(256) void k_json_compact(const mpe_json_entry *__restrict__ entries, int n_entries, int kcap, int J,
                                                      const int32_t *__restrict__ n_sk, const uint32_t *__restrict__ r_jm,
                                                      const uint32_t *__restrict__ r_tm, const double *__restrict__ r_xy,
                                                      const float *__restrict__ r_vp, const int32_t *__restrict__ head_base,
                                                      const int32_t *__restrict__ totals, int32_t *__restrict__ head_cam,
                                                      int32_t *__restrict__ skeleton_index, uint32_t *__restrict__ joint_mask,
                                                      uint32_t *__restrict__ tri_mask, double *__restrict__ xy, float *__restrict__ vp) {
    if (totals[2] != JS_OK) return;
    const long total = (long)n_entries * kcap;
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
        const int k = (int)(gid / n_entries), e = (int)(gid - (long)k * n_entries);
        if (k >= n_sk[e]) continue;
        const size_t r = (size_t)e * kcap + k;
        if (!r_jm[r]) continue;
        int rank = 0;
        for (int q = 0; q < k; ++q) rank += r_jm[(size_t)e * kcap + q] != 0;
        const size_t h = (size_t)head_base[e] + rank;
        head_cam[h] = entries[e].cam;
        skeleton_index[h] = k;
        joint_mask[h] = r_jm[r];
        tri_mask[h] = r_tm[r];
        for (int q = 0; q < J * 2; ++q) {
            xy[h * J * 2 + q] = r_xy[r * J * 2 + q];
            vp[h * J * 2 + q] = r_vp[r * J * 2 + q];
        }
    }
}

}  // namespace mpe

using namespace mpe;

extern "C" size_t mpe_json_scratch_bytes(int32_t n_entries_cap, int32_t skeletons_per_string_cap, int32_t n_joints) {
    const size_t e = (size_t)(n_entries_cap > 0 ? n_entries_cap : 1), k = (size_t)(skeletons_per_string_cap > 0 ? skeletons_per_string_cap : 1);
    const size_t rows = e * k;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    return up(e * 4) + up(e * 4) + up(rows * sizeof(SkExt)) + up(rows * 4) + up(rows * 4) + up(rows * (size_t)n_joints * 2 * 8) +
           up(rows * (size_t)n_joints * 2 * 4);
}

// Grid of the two thread-per-skeleton kernels.  They run beside the previous window's GEMMs (pipeline.py), so the grid is
// capped: one skeleton is ~1.4 KB of serial byte work per thread and the kernel's duration is that latency, not the
// number of workgroups -- measured with Engine.stream_json, 24 windows of 1000 frames (20 000 skeletons each): cap 16 ->
// 44.0k frames/s, 48 -> 88.6k, 128 -> 132.4k, uncapped (79 workgroups... of 256 threads would do; "uncapped" = one thread
// per skeleton slot) -> 122.6k.  MPE_JSON_WGS overrides the cap.
static unsigned side_grid(long threads) {
    static const long cap = getenv("MPE_JSON_WGS") ? atol(getenv("MPE_JSON_WGS")) : 128;
    const long want = (threads + 255) / 256;
    return (unsigned)(want < cap ? want : (cap > 0 ? cap : 1));
}

extern "C" int mpe_json_parse_device(mpe_ctx *ctx, void *stream, const char *d_text, const mpe_json_entry *d_entries,
                                     const int32_t *d_frame_entry_off, int32_t n_entries, int32_t n_frames, int32_t head_cap,
                                     int32_t skeletons_per_string_cap, void *d_scratch, size_t scratch_bytes, const mpe_batch *out,
                                     int32_t *d_skeleton_index, int32_t *d_totals) {
    if (!ctx || !out || !d_totals || !d_scratch || !d_skeleton_index || n_entries < 0 || n_frames < 0 || head_cap < 0 ||
        skeletons_per_string_cap < 1)
        return MPE_ERR_INVALID;
    if (n_frames > 0 && (!d_frame_entry_off || !out->d_frame_head_off || !out->d_frame_en_off || !out->d_slot_cam || !out->d_slot_n))
        return MPE_ERR_INVALID;
    if (n_entries > 0 && (!d_text || !d_entries || !out->d_head_cam || !out->d_joint_mask || !out->d_tri_mask || !out->d_xy || !out->d_vp))
        return MPE_ERR_INVALID;
    const int V = ctx->cfg.n_cameras, J = ctx->cfg.n_joints, kcap = skeletons_per_string_cap;
    if (mpe_json_scratch_bytes(n_entries, kcap, J) > scratch_bytes) return MPE_ERR_CAPACITY;
    if (hipSetDevice(ctx->device) != hipSuccess) return MPE_ERR_HIP;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // carve the scratch
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t e = (size_t)(n_entries > 0 ? n_entries : 1), rows = e * (size_t)kcap;
    char *p = static_cast<char *>(d_scratch);
    int32_t *n_sk = reinterpret_cast<int32_t *>(p); p += up(e * 4);
    int32_t *head_base = reinterpret_cast<int32_t *>(p); p += up(e * 4);
    SkExt *ext = reinterpret_cast<SkExt *>(p); p += up(rows * sizeof(SkExt));
    uint32_t *r_jm = reinterpret_cast<uint32_t *>(p); p += up(rows * 4);
    uint32_t *r_tm = reinterpret_cast<uint32_t *>(p); p += up(rows * 4);
    double *r_xy = reinterpret_cast<double *>(p); p += up(rows * (size_t)J * 2 * 8);
    float *r_vp = reinterpret_cast<float *>(p);
    if (hipMemsetAsync(d_totals, 0, 4 * sizeof(int32_t), s) != hipSuccess) return MPE_ERR_HIP;
    if (n_frames == 0) return MPE_OK;
    if (n_entries > 0) {
        hipLaunchKernelGGL(k_json_braces, dim3((unsigned)n_entries), dim3(64), 0, s, d_text, d_entries, n_entries, kcap, ext, n_sk, d_totals);
        hipLaunchKernelGGL(k_json_skeleton, dim3(side_grid((long)n_entries * kcap)), dim3(256), 0, s, d_text, d_entries, n_entries, kcap, J, ext,
                           n_sk, r_jm, r_tm, r_xy, r_vp, d_totals);
    }
    hipLaunchKernelGGL(k_json_layout, dim3(1), dim3(1024), 0, s, d_entries, d_frame_entry_off, n_frames, V, kcap, n_sk, r_jm, head_cap,
                       const_cast<int32_t *>(out->d_frame_head_off), const_cast<int32_t *>(out->d_frame_en_off),
                       const_cast<int32_t *>(out->d_slot_cam), const_cast<int32_t *>(out->d_slot_n), head_base, d_totals);
    if (n_entries > 0) {
        hipLaunchKernelGGL(k_json_compact, dim3(side_grid((long)n_entries * kcap)), dim3(256), 0, s, d_entries, n_entries, kcap, J, n_sk, r_jm,
                           r_tm, r_xy, r_vp, head_base, d_totals, const_cast<int32_t *>(out->d_head_cam), d_skeleton_index,
                           const_cast<uint32_t *>(out->d_joint_mask), const_cast<uint32_t *>(out->d_tri_mask),
                           const_cast<double *>(out->d_xy), const_cast<float *>(out->d_vp));
    }
    return hipGetLastError() == hipSuccess ? MPE_OK : MPE_ERR_HIP;
}
