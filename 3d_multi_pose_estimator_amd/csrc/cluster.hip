// Greedy person clustering, one wavefront per frame.
//
// Restates get_person_proposal_from_network_output (reference
// utils/skeleton_matching_utils.py:12-132) on the implicit topology, bit for bit:
//   * a matching exists for edge-node X=(h1,h2) when score[X] > threshold (:49-55); which of
//     the two heads is `a` follows CPython's iteration order of the 2-element set (:53);
//   * stable sort by score, descending (:60) -> 64-bit keys (inverted score bits, edge-node
//     index): matchings are created in edge-node order, so ties keep creation order;
//   * camera-uniqueness tests and human-index bookkeeping (:61-108) with camera sets as
//     32-bit masks, including the quirk that merging two humans drops the absorbed group's
//     camera list (:97-102);
//   * connected components in node-insertion order (:117-130) with networkx 3.4.2's
//     _plain_bfs, whose result is a Python set: iteration order of that set decides which
//     head wins when a component holds two heads of one camera, so CPython's set
//     (open addressing, LINEAR_PROBES 9, PERTURB_SHIFT 5, growth x4) is emulated exactly.
//
// Work split inside the wave: all 64 lanes build the keys and run a bitonic sort in LDS;
// lane 0 then replays the sequential greedy rules and the component walk on LDS-resident
// state (a few hundred dependent steps of ~64-cycle LDS latency instead of HBM latency).
// Frames whose edge-node count does not fit the LDS budget use the global-scratch variant.
#include <cstdlib>
#include <cstring>

#include "mpe_internal.h"
#include "dlt_common.h"

namespace mpe {

namespace {

struct PySet {             // CPython 3.10 set of small non-negative ints (hash(i) == i)
    int32_t *tab;          // current table (-1 = empty)
    int32_t *alt;          // spare table for growth
    int mask, fill;
};

__device__ inline int pyset_probe(const int32_t *tab, int mask, int key, bool stop_on_equal, bool *found) {
    unsigned perturb = (unsigned)key;
    int i = key & mask;
    *found = false;
    while (true) {
        int probes = (i + 9 <= mask) ? 9 : 0;
        int j = i;
        while (true) {
            const int cur = tab[j];
            if (cur == -1) return j;
            if (stop_on_equal && cur == key) {
                *found = true;
                return j;
            }
            ++j;
            if (probes == 0) break;
            --probes;
        }
        perturb >>= 5;
        i = (int)(((unsigned)i * 5u + 1u + perturb) & (unsigned)mask);
    }
}

__device__ inline void pyset_init(PySet &s) {
    s.mask = 7;
    s.fill = 0;
    for (int i = 0; i < 8; ++i) s.tab[i] = -1;
}

__device__ inline bool pyset_contains(const PySet &s, int key) {
    bool found;
    pyset_probe(s.tab, s.mask, key, true, &found);
    return found;
}

__device__ inline void pyset_add(PySet &s, int key) {
    bool found;
    const int j = pyset_probe(s.tab, s.mask, key, true, &found);
    if (found) return;
    s.tab[j] = key;
    s.fill++;
    if (s.fill * 5 < s.mask * 3) return;
    const int minused = s.fill * 4;
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    for (int i = 0; i < newsize; ++i) s.alt[i] = -1;
    for (int i = 0; i <= s.mask; ++i) {
        const int k = s.tab[i];
        if (k != -1) {
            bool f2;
            s.alt[pyset_probe(s.alt, newsize - 1, k, false, &f2)] = k;
        }
    }
    int32_t *t = s.tab;
    s.tab = s.alt;
    s.alt = t;
    s.mask = newsize - 1;
}

// order of list({h1, h2}): the element in the lower slot of an 8-entry table comes first
__device__ inline bool pair_first_is_h1(int h1, int h2) {
    const int s1 = h1 & 7;
    unsigned perturb = (unsigned)h2;
    int i = h2 & 7;
    while (i == s1) {
        perturb >>= 5;
        i = (int)(((unsigned)i * 5u + 1u + perturb) & 7u);
    }
    return s1 < i;
}

constexpr uint64_t KEY_NONE = ~0ull;

__device__ inline uint64_t make_key(float score, float thr, int m) {
    // Python compares float(score) > threshold; NaN fails the test
    if (!(score > thr)) return KEY_NONE;
    return ((uint64_t)(0xFFFFFFFFu - __float_as_uint(score)) << 32) | (uint32_t)m;
}

// the sequential part, shared by both variants; every array lives in `W` (LDS or global)
struct Work {
    int32_t *seen, *order, *linked, *human, *cfh, *ea, *eb, *done, *lvl, *nxt, *off, *adj, *tabA, *tabB;
};

__device__ inline Work carve(int32_t *base, int hmax, int table_cap) {
    Work w;
    w.seen = base;
    w.order = w.seen + hmax;
    w.linked = w.order + hmax;
    w.human = w.linked + hmax;
    w.cfh = w.human + hmax;
    w.ea = w.cfh + hmax;
    w.eb = w.ea + hmax;
    w.done = w.eb + hmax;
    w.lvl = w.done + hmax;
    w.nxt = w.lvl + hmax;
    w.off = w.nxt + hmax;            // [hmax + 1] CSR offsets of the accepted-edge graph
    w.adj = w.off + hmax + 1;        // [2 * hmax] neighbours (accepted edges form a forest: < hmax edges)
    w.tabA = w.adj + 2 * hmax;
    w.tabB = w.tabA + table_cap;
    return w;
}

// networkx 3.4.2 _plain_bfs from `v` over the CSR adjacency (w.off / w.adj), result in `set`
__device__ inline void bfs_component(const Work &w, int v, int n_nodes, PySet &set) {
    pyset_init(set);
    pyset_add(set, v);
    int nl = 1;
    w.lvl[0] = v;
    bool full = set.fill == n_nodes;
    int32_t *L = w.lvl, *N = w.nxt;
    while (nl > 0 && !full) {
        int nn = 0;
        for (int li = 0; li < nl; ++li) {
            const int x = L[li];
            for (int i = w.off[x], i1 = w.off[x + 1]; i < i1; ++i) {
                const int wv = w.adj[i];
                if (!pyset_contains(set, wv)) {
                    pyset_add(set, wv);
                    N[nn++] = wv;
                }
            }
            if (set.fill == n_nodes) { full = true; break; }
        }
        int32_t *t = L; L = N; N = t;
        nl = nn;
    }
}

// adjacency lists in edge-creation order (the iteration order of networkx's adj dicts).
// Every accepted edge joins two different groups, so ne < H.
__device__ inline void build_csr(const Work &w, int H, int ne) {
    for (int h = 0; h < H; ++h) w.seen[h] = 0;
    for (int e = 0; e < ne; ++e) {
        w.seen[w.ea[e]]++;
        w.seen[w.eb[e]]++;
    }
    int acc = 0;
    for (int h = 0; h < H; ++h) {
        const int d = w.seen[h];
        w.off[h] = acc;
        w.seen[h] = acc;
        acc += d;
    }
    w.off[H] = acc;
    for (int e = 0; e < ne; ++e) {
        const int a = w.ea[e], b = w.eb[e];
        w.adj[w.seen[a]++] = b;
        w.adj[w.seen[b]++] = a;
    }
}

// greedy merge over the sorted keys + connected components; returns the number of persons
template <typename PairFn>
__device__ inline int greedy_and_components(const Work &w, const uint64_t *keys, int n_keys, PairFn pair_of, int H,
                                            int M, const int32_t *cam, int V, int min_views, int pcap,
                                            int32_t *out) {
    for (int h = 0; h < H; ++h) {
        w.seen[h] = 0;
        w.human[h] = -1;
        w.done[h] = 0;
        w.linked[h] = (int32_t)(1u << cam[h]);
    }
    // G.add_node order = first appearance in the edge scan (h1 then h2 of every edge-node)
    int n_nodes = 0;
    for (int m = 0; m < M && n_nodes < H; ++m) {
        int h1, h2;
        pair_of(m, h1, h2);
        if (!w.seen[h1]) { w.seen[h1] = 1; w.order[n_nodes++] = h1; }
        if (!w.seen[h2]) { w.seen[h2] = 1; w.order[n_nodes++] = h2; }
    }
    int cur = 0, ne = 0;
    for (int k = 0; k < n_keys; ++k) {
        const uint64_t key = keys[k];
        if (key == KEY_NONE) break;
        int h1, h2;
        pair_of((int)(key & 0xFFFFFFFFu), h1, h2);
        const bool first1 = pair_first_is_h1(h1, h2);
        const int a = first1 ? h1 : h2, b = first1 ? h2 : h1;
        const uint32_t ba = 1u << cam[a], bb = 1u << cam[b];
        if (((uint32_t)w.linked[b] & ba) || ((uint32_t)w.linked[a] & bb)) continue;
        const int ha = w.human[a], hb = w.human[b];
        if (ha >= 0 && ((uint32_t)w.cfh[ha] & bb)) continue;
        if (hb >= 0 && ((uint32_t)w.cfh[hb] & ba)) continue;
        if (ha < 0 && hb < 0) {
            w.human[a] = cur;
            w.human[b] = cur;
            w.cfh[cur] = (int32_t)(ba | bb);
            ++cur;
        } else if (ha >= 0 && hb < 0) {
            w.human[b] = ha;
            w.cfh[ha] = (int32_t)((uint32_t)w.cfh[ha] | bb);
        } else if (hb >= 0 && ha < 0) {
            w.human[a] = hb;
            w.cfh[hb] = (int32_t)((uint32_t)w.cfh[hb] | ba);
        } else {
            if ((uint32_t)w.cfh[hb] & (uint32_t)w.cfh[ha]) continue;
            for (int n = 0; n < H; ++n)
                if (w.human[n] == hb) w.human[n] = ha;
            // the absorbed group's camera list is dropped, not merged (reference :97-102)
        }
        w.ea[ne] = a;
        w.eb[ne] = b;
        ++ne;
        w.linked[a] = (int32_t)((uint32_t)w.linked[a] | bb);
        w.linked[b] = (int32_t)((uint32_t)w.linked[b] | ba);
    }
    build_csr(w, H, ne);
    int np = 0;
    for (int oi = 0; oi < n_nodes; ++oi) {
        const int v = w.order[oi];
        if (w.done[v]) continue;
        PySet set{w.tabA, w.tabB, 7, 0};
        bfs_component(w, v, n_nodes, set);
        const int cnt = set.fill;
        int32_t *pout = (cnt >= min_views && np < pcap) ? out + (size_t)np * V : nullptr;
        for (int i = 0; i <= set.mask; ++i) {
            const int h = set.tab[i];
            if (h < 0) continue;
            w.done[h] = 1;
            if (pout) pout[cam[h]] = h;
        }
        if (pout) ++np;
    }
    return np;
}

__device__ inline void sift_down(uint64_t *k, int start, int end) {
    int root = start;
    while (true) {
        int child = 2 * root + 1;
        if (child > end) break;
        if (child + 1 <= end && k[child] < k[child + 1]) ++child;
        if (k[root] < k[child]) {
            const uint64_t t = k[root];
            k[root] = k[child];
            k[child] = t;
            root = child;
        } else {
            break;
        }
    }
}

}  // namespace

int cluster_table_cap(int hmax) {
    // largest table CPython reaches for a set of `hmax` elements
    int mask = 7, fill = 0;
    for (int i = 0; i < hmax; ++i) {
        ++fill;
        if (fill * 5 >= mask * 3) {
            int minused = fill * 4, ns = 8;
            while (ns <= minused) ns <<= 1;
            mask = ns - 1;
        }
    }
    return mask + 1;
}

size_t cluster_keys_per_frame(int hmax) {
    // >= the largest edge-node count of a frame; a power of two (>= 512) because the workgroup
    // kernel pads its bitonic sort to one when it sorts in this scratch
    size_t need = (size_t)hmax * hmax / 2 + 1, n = 512;
    while (n < need) n <<= 1;
    return n;
}

size_t cluster_scratch_per_frame(int hmax) {
    return (size_t)13 * hmax + 1 + 2 * (size_t)cluster_table_cap(hmax);
}

// ---- LDS variant: keys [n_pow2] u64 | pairs [n_pow2] u32 | work arrays ------------------
__global__ __launch_bounds__(64) void k_cluster_lds(const DevCfg *__restrict__ cfg, int n_frames,
                                                    const int32_t *__restrict__ head_off,
                                                    const int32_t *__restrict__ en_off,
                                                    const int32_t *__restrict__ head_cam,
                                                    const int32_t *__restrict__ en_pair,
                                                    const float *__restrict__ scores, int pcap, int hmax,
                                                    int table_cap, int n_pow2, int32_t *__restrict__ persons,
                                                    int32_t *__restrict__ n_persons) {
    extern __shared__ uint64_t s_keys[];
    const int f = blockIdx.x;
    if (f >= n_frames) return;
    const int V = cfg->V;
    const int lane = threadIdx.x;
    const int h0 = head_off[f], H = head_off[f + 1] - h0;
    const int e0 = en_off[f], M = en_off[f + 1] - e0;
    int32_t *out = persons + (size_t)f * pcap * V;
    for (int i = lane; i < pcap * V; i += 64) out[i] = -1;
    if (M <= 0 || H > hmax || M > n_pow2) {
        if (lane == 0) n_persons[f] = 0;
        return;
    }
    uint32_t *s_pair = reinterpret_cast<uint32_t *>(s_keys + n_pow2);
    int32_t *s_work = reinterpret_cast<int32_t *>(s_pair + n_pow2);
    int32_t *s_cam = s_work + 13 * hmax + 1 + 2 * table_cap;
    const float thr = cfg->threshold;
    // smallest power of two covering this frame's edge-nodes
    int n = 64;
    while (n < M) n <<= 1;
    for (int m = lane; m < n; m += 64) {
        uint64_t key = KEY_NONE;
        if (m < M) {
            const int h1 = en_pair[2 * (size_t)(e0 + m)], h2 = en_pair[2 * (size_t)(e0 + m) + 1];
            s_pair[m] = ((uint32_t)h1 << 16) | (uint32_t)h2;
            key = make_key(scores[e0 + m], thr, m);
        }
        s_keys[m] = key;
    }
    for (int h = lane; h < H; h += 64) s_cam[h] = head_cam[h0 + h];
    __syncthreads();
    // bitonic sort, ascending (KEY_NONE sinks to the end)
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < n / 2; t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
                const int p = i | j;
                const bool up = (i & k) == 0;
                const uint64_t x = s_keys[i], y = s_keys[p];
                if ((x > y) == up) {
                    s_keys[i] = y;
                    s_keys[p] = x;
                }
            }
            __syncthreads();
        }
    if (lane != 0) return;
    Work w = carve(s_work, hmax, table_cap);
    auto pair_of = [&](int m, int &h1, int &h2) {
        const uint32_t pr = s_pair[m];
        h1 = (int)(pr >> 16);
        h2 = (int)(pr & 0xFFFFu);
    };
    n_persons[f] = greedy_and_components(w, s_keys, n, pair_of, H, M, s_cam, V, cfg->min_views, pcap, out);
}

// ---- wave-register variant (<= 64 heads per frame) ----------------------------------------
// The sequential rules run on the scalar unit: per-head state lives one head per lane in
// VGPRs (camera, linked-camera mask, human index; cameras-of-human one human per lane; the
// accepted edges one edge per lane) and every step reads/writes it with v_readlane /
// a one-lane select at a uniform index -- a few cycles instead of a ~100-cycle LDS round trip per
// dependent access.  The pending matchings of a 64-key chunk are tested against the current
// state in parallel (a rejected matching changes nothing), so the serial part is one round per
// ACCEPTED matching; the relabelling of a merged group is one vector select; the neighbours of
// a node are a ballot over the edge lanes (ascending lane = edge-creation order); CPython's set
// table (8 -> 32 -> 128 slots for <= 64 keys) is two VGPRs and is only walked for components
// that hold two heads of one camera.  Same rules, same orders, same results as
// greedy_and_components.
#define MPE_RL(v, i) __builtin_amdgcn_readlane((int)(v), (i))
// write `val` into lane `idx` (uniform): a compare + select on the vector unit
__device__ inline int wl(int old, int val, int idx) { return (int)threadIdx.x == idx ? val : old; }

struct RegSet {
    int t0, t1;            // table slots 0..63 / 64..127, one per lane (-1 = empty)
    int mask, fill;
};

__device__ inline int rs_get(const RegSet &s, int j) { return j < 64 ? MPE_RL(s.t0, j) : MPE_RL(s.t1, j - 64); }

__device__ inline void rs_put(RegSet &s, int j, int key) {
    if (j < 64) s.t0 = wl(s.t0, key, j);
    else s.t1 = wl(s.t1, key, j - 64);
}

__device__ inline int rs_probe(const RegSet &s, int key, bool stop_on_equal, bool *found) {
    unsigned perturb = (unsigned)key;
    int i = key & s.mask;
    *found = false;
    while (true) {
        int probes = (i + 9 <= s.mask) ? 9 : 0;
        int j = i;
        while (true) {
            const int cur = rs_get(s, j);
            if (cur == -1) return j;
            if (stop_on_equal && cur == key) {
                *found = true;
                return j;
            }
            ++j;
            if (probes == 0) break;
            --probes;
        }
        perturb >>= 5;
        i = (int)(((unsigned)i * 5u + 1u + perturb) & (unsigned)s.mask);
    }
}

// returns true when `key` was not in the set yet
__device__ inline bool rs_add(RegSet &s, int key) {
    bool found;
    const int j = rs_probe(s, key, true, &found);
    if (found) return false;
    rs_put(s, j, key);
    s.fill++;
    if (s.fill * 5 < s.mask * 3) return true;
    const int minused = s.fill * 4;
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    RegSet n{-1, -1, newsize - 1, s.fill};
    // re-insert in table order (CPython set_table_resize); slots beyond the old mask hold -1
    for (int half = 0; half < 2; ++half) {
        const int t = half ? s.t1 : s.t0;
        unsigned long long bits = __ballot(t != -1);
        while (bits) {
            const int i = __builtin_ctzll(bits);
            bits &= bits - 1;
            bool f2;
            const int k = MPE_RL(t, i);
            rs_put(n, rs_probe(n, k, false, &f2), k);
        }
    }
    s = n;
    return true;
}

// (the body of k_cluster_wave: one wave, lane = threadIdx.x < 64; also run by the small-batch tail launch k_lat_tail)
__device__ __forceinline__ void cluster_wave_body(uint64_t *s_keys, int f, const DevCfg *__restrict__ cfg, int n_frames,
                                                  const int32_t *__restrict__ head_off, const int32_t *__restrict__ en_off,
                                                  const int32_t *__restrict__ head_cam, const int32_t *__restrict__ en_pair,
                                                  const float *__restrict__ scores, int pcap, int n_pow2, int hmax,
                                                  int32_t *__restrict__ persons, int32_t *__restrict__ n_persons) {
    if (f >= n_frames) return;
    const int V = cfg->V;
    const int lane = threadIdx.x;
    const int h0 = head_off[f], H = head_off[f + 1] - h0;
    const int e0 = en_off[f], M = en_off[f + 1] - e0;
    int32_t *out = persons + (size_t)f * pcap * V;
    if (M <= 0 || H > 64 || H > hmax || M > n_pow2) {
        for (int i = lane; i < pcap * V; i += 64) out[i] = -1;
        if (lane == 0) n_persons[f] = 0;
        return;
    }
    uint32_t *s_pair = reinterpret_cast<uint32_t *>(s_keys + n_pow2);
    const float thr = cfg->threshold;
    const int min_views = cfg->min_views;
    // Only the matchings above the threshold are sorted (compacted to the front in index order by wave ballots; the keys are
    // unique, so what the sort returns does not depend on where it found them): a 5 x 4 frame has 160 candidates and a few dozen
    // matchings, i.e. 21 compare-exchange passes over 64 keys instead of 36 over 256 -- a third of this kernel's time for one frame.
    int n_valid = 0;
    for (int m0 = 0; m0 < M; m0 += 64) {
        const int m = m0 + lane;
        uint64_t key = KEY_NONE;
        if (m < M) {
            const int h1 = en_pair[2 * (size_t)(e0 + m)], h2 = en_pair[2 * (size_t)(e0 + m) + 1];
            s_pair[m] = ((uint32_t)h1 << 16) | (uint32_t)h2;
            key = make_key(scores[e0 + m], thr, m);
        }
        const unsigned long long live = __ballot(key != KEY_NONE);
        if (key != KEY_NONE) s_keys[n_valid + __popcll(live & ((1ull << lane) - 1ull))] = key;
        n_valid += __popcll(live);
    }
    int n = 64;
    while (n < n_valid) n <<= 1;
    for (int m = n_valid + lane; m < n; m += 64) s_keys[m] = KEY_NONE;
    __syncthreads();
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < n / 2; t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const bool up = (i & k) == 0;
                const uint64_t x = s_keys[i], y = s_keys[p];
                if ((x > y) == up) {
                    s_keys[i] = y;
                    s_keys[p] = x;
                }
            }
            __syncthreads();
        }

    // ---- per-lane state ----
    int32_t *s_aux = reinterpret_cast<int32_t *>(s_pair + n_pow2);      // [4][64] scratch ints
    int32_t *s_first = s_aux, *s_ord = s_aux + 64, *s_gor = s_aux + 128, *s_gcnt = s_aux + 192;
    const int camv = lane < H ? head_cam[h0 + lane] : 0;
    int linkedv = (int)(1u << camv);
    int humanv = -1, cfhv = 0, eav = -1, ebv = -1;

    // G.add_node order = first appearance in the edge scan (h1 then h2 of every edge-node):
    // first[h] = min(2m + position), rank of first[h] among the heads that appear
    s_first[lane] = 0x7FFFFFFF;
    s_gor[lane] = 0;
    s_gcnt[lane] = 0;
    __syncthreads();
    for (int m = lane; m < M; m += 64) {
        const uint32_t pr = s_pair[m];
        atomicMin(&s_first[pr >> 16], 2 * m);
        atomicMin(&s_first[pr & 0xFFFFu], 2 * m + 1);
    }
    __syncthreads();
    const int firstv = s_first[lane];
    const int n_nodes = __popcll(__ballot(firstv != 0x7FFFFFFF));
    {
        int rank = 0;
        for (int l = 0; l < H; ++l) rank += MPE_RL(firstv, l) < firstv ? 1 : 0;
        if (firstv != 0x7FFFFFFF) s_ord[rank] = lane;
    }
    __syncthreads();
    const int orderv = lane < n_nodes ? s_ord[lane] : 0;

    // greedy merge over the sorted keys, 64 at a time.  A rejected matching changes nothing, so
    // all pending keys of the chunk are tested against the current state at once (one key per
    // lane); the first lane that passes is exactly the next matching the sequential rules accept.
    int cur = 0, ne = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const uint64_t key = s_keys[k0 + lane];
        const bool valid = key != KEY_NONE;
        const uint32_t pr = valid ? s_pair[(uint32_t)key] : 0u;
        const int h1 = (int)(pr >> 16), h2 = (int)(pr & 0xFFFFu);
        const bool first1 = pair_first_is_h1(h1, h2);
        const int a_l = first1 ? h1 : h2, b_l = first1 ? h2 : h1;
        const uint32_t ba_l = 1u << __shfl(camv, a_l), bb_l = 1u << __shfl(camv, b_l);
        const int cnt = __popcll(__ballot(valid));             // sorted: the valid keys are a prefix
        int start = 0;
        while (true) {
            const uint32_t la_l = (uint32_t)__shfl(linkedv, a_l), lb_l = (uint32_t)__shfl(linkedv, b_l);
            const int ha_l = __shfl(humanv, a_l), hb_l = __shfl(humanv, b_l);
            // the cross-lane reads stay outside any per-lane condition: ds_bpermute returns 0 for
            // source lanes that are masked off
            const uint32_t ca_x = (uint32_t)__shfl(cfhv, ha_l >= 0 ? ha_l : 0);
            const uint32_t cb_x = (uint32_t)__shfl(cfhv, hb_l >= 0 ? hb_l : 0);
            const uint32_t ca_l = ha_l >= 0 ? ca_x : 0u, cb_l = hb_l >= 0 ? cb_x : 0u;
            const bool rej = (lb_l & ba_l) || (la_l & bb_l) || (ca_l & bb_l) || (cb_l & ba_l) ||
                             (ha_l >= 0 && hb_l >= 0 && (cb_l & ca_l));
            const unsigned long long pass = __ballot(valid && lane >= start && !rej);
            if (!pass) break;
            const int q = __builtin_ctzll(pass);
            const int a = MPE_RL(a_l, q), b = MPE_RL(b_l, q);
            const uint32_t ba = (uint32_t)MPE_RL(ba_l, q), bb = (uint32_t)MPE_RL(bb_l, q);
            const uint32_t la = (uint32_t)MPE_RL(la_l, q), lb = (uint32_t)MPE_RL(lb_l, q);
            const int ha = MPE_RL(ha_l, q), hb = MPE_RL(hb_l, q);
            const uint32_t ca = (uint32_t)MPE_RL(ca_l, q), cb = (uint32_t)MPE_RL(cb_l, q);
            if (ha < 0 && hb < 0) {
                humanv = wl(humanv, cur, a);
                humanv = wl(humanv, cur, b);
                cfhv = wl(cfhv, (int)(ba | bb), cur);
                ++cur;
            } else if (ha >= 0 && hb < 0) {
                humanv = wl(humanv, ha, b);
                cfhv = wl(cfhv, (int)(ca | bb), ha);
            } else if (hb >= 0 && ha < 0) {
                humanv = wl(humanv, hb, a);
                cfhv = wl(cfhv, (int)(cb | ba), hb);
            } else {
                humanv = humanv == hb ? ha : humanv;       // the absorbed group's camera list is dropped (:97-102)
            }
            eav = wl(eav, a, ne);
            ebv = wl(ebv, b, ne);
            ++ne;
            linkedv = wl(linkedv, (int)(la | bb), a);
            linkedv = wl(linkedv, (int)(lb | ba), b);
            start = q + 1;
        }
        if (cnt < 64) break;
    }

    // connected components in node-insertion order.  The accepted edges join heads of one human
    // index, so a component is the set of lanes with equal humanv (or a single unmatched head).
    // Only the ITERATION ORDER of networkx's result set can matter, and only when two heads of
    // one camera share a component (possible through the dropped camera list above): then the
    // BFS + CPython set emulation decides; otherwise each camera slot has one candidate.
    if (humanv >= 0) {
        atomicOr(&s_gor[humanv], 1 << camv);
        atomicAdd(&s_gcnt[humanv], 1);
    }
    for (int i = lane; i < pcap * V; i += 64) out[i] = -1;
    __threadfence_block();                                   // the -1 fill lands before any person row
    __syncthreads();
    const bool plainv = __popc((unsigned)s_gor[lane]) == s_gcnt[lane];   // lane = human index
    unsigned long long done = 0;
    int np = 0;
    for (int oi = 0; oi < n_nodes; ++oi) {
        const int v = MPE_RL(orderv, oi);
        if ((done >> v) & 1ull) continue;
        const int g = MPE_RL(humanv, v);
        if (g < 0 || MPE_RL((int)plainv, g)) {
            const unsigned long long memb = g >= 0 ? __ballot(humanv == g) : (1ull << v);
            done |= memb;
            if (__popcll(memb) >= min_views && np < pcap) {
                if ((memb >> lane) & 1ull) out[(size_t)np * V + camv] = lane;
                ++np;
            }
            continue;
        }
        // networkx _plain_bfs + CPython set order
        RegSet set{-1, -1, 7, 0};
        rs_add(set, v);
        int lvlv = 0, nxtv = 0;
        lvlv = wl(lvlv, v, 0);
        int nl = 1;
        bool full = set.fill == n_nodes;
        while (nl > 0 && !full) {
            int nn = 0;
            for (int li = 0; li < nl; ++li) {
                const int x = MPE_RL(lvlv, li);
                unsigned long long nb = __ballot(eav == x || ebv == x);   // unused edge lanes hold -1
                while (nb) {
                    const int e = __builtin_ctzll(nb);
                    nb &= nb - 1;
                    const int ae = MPE_RL(eav, e), be = MPE_RL(ebv, e);
                    const int wv = ae == x ? be : ae;
                    if (rs_add(set, wv)) nxtv = wl(nxtv, wv, nn++);
                }
                if (set.fill == n_nodes) {
                    full = true;
                    break;
                }
            }
            const int t = lvlv;
            lvlv = nxtv;
            nxtv = t;
            nl = nn;
        }
        const bool emit = set.fill >= min_views && np < pcap;
        int res = -1;                                            // lane c: the head of camera c
        for (int half = 0; half < 2; ++half) {
            const int t = half ? set.t1 : set.t0;
            unsigned long long bits = __ballot(t != -1);
            while (bits) {
                const int i = __builtin_ctzll(bits);
                bits &= bits - 1;
                const int h = MPE_RL(t, i);
                done |= 1ull << h;
                res = wl(res, h, MPE_RL(camv, h));               // later in set order wins, as in the reference
            }
        }
        if (emit) {
            if (lane < V && res >= 0) out[(size_t)np * V + lane] = res;
            ++np;
        }
    }
    if (lane == 0) n_persons[f] = np;
}

__global__ __launch_bounds__(64) void k_cluster_wave(const DevCfg *__restrict__ cfg, int n_frames,
                                                     const int32_t *__restrict__ head_off,
                                                     const int32_t *__restrict__ en_off,
                                                     const int32_t *__restrict__ head_cam,
                                                     const int32_t *__restrict__ en_pair,
                                                     const float *__restrict__ scores, int pcap, int n_pow2,
                                                     int hmax, int32_t *__restrict__ persons,
                                                     int32_t *__restrict__ n_persons) {
    extern __shared__ uint64_t s_keys[];
    cluster_wave_body(s_keys, blockIdx.x, cfg, n_frames, head_off, en_off, head_cam, en_pair, scores, pcap, n_pow2, hmax, persons, n_persons);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Small batches: the tail of the matching stage in ONE launch.  Workgroup f < n_frames (its first wave): the scores of the frame's
// edge-nodes -- the attention stage of the LAST layer (one attention head, one feature: gat2.py:57-66 + the final activation,
// :145-148), k_lat_attention's edge-node half written out for one column: same expressions, same order, same bits -- and then the
// frame's clustering (cluster_wave_body).  The other workgroups solve, BESIDE that (the clustering is 20 us of sequential rules on
// one wave; the rest of the chip is idle), every cross-camera skeleton pair of the batch for every joint: one DLT solve per
// (edge-node, joint) with the arguments in camera order, as the row kernel's person_front would call it
// (pose_estimator_dataset_from_json.py:63-101) -- which of them a person needs is only known after the clustering, that they
// are all of this form is known before.  k_mlp_rows then fetches its pairs (pair_pts) instead of solving them: 21 -> 8 us.
// ---------------------------------------------------------------------------------------------------------------------------
struct TailScores {                // the last layer's attention stage (AggArgs of that layer, the fields an N x 1 layer needs)
    const float *ft2;              // [n_nodes][ld]: column 0
    int ld;
    const float *a12;              // [n_nodes][32]: a1 at [0], a2 at [16]
    float alpha, out_slope;
    int out_mode;
    const int32_t *node_off;
    float *scores;                 // [n_edge_nodes] (written here, read by the clustering of the same workgroup)
};

__global__ __launch_bounds__(256) void k_lat_tail(const DevCfg *__restrict__ cfg, int n_frames, int n_en, int J,
                                                  const int32_t *__restrict__ head_off, const int32_t *__restrict__ en_off,
                                                  const int32_t *__restrict__ head_cam, const int32_t *__restrict__ en_pair,
                                                  const int32_t *__restrict__ en_frame, TailScores ts, int pcap, int n_pow2, int hmax,
                                                  int32_t *__restrict__ persons, int32_t *__restrict__ n_persons,
                                                  const uint32_t *__restrict__ tri_mask, const double *__restrict__ xy,
                                                  double *__restrict__ pair_pts) {
#pragma clang fp contract(off)
    extern __shared__ uint64_t s_keys[];
    if ((int)blockIdx.x < n_frames) {
        if (threadIdx.x >= 64) return;                          // (one wave: the other three leave before any barrier)
        const int f = blockIdx.x, lane = threadIdx.x;
        const int hb = head_off[f], H = head_off[f + 1] - hb;
        const int e0 = en_off[f], M = en_off[f + 1] - e0;
        const int nb = ts.node_off[f];
        for (int m = lane; m < M; m += 64) {
            // gat.hip: aggregate_en_body<1> for one attention head of width one, score mode
            const int h1 = en_pair[2 * (size_t)(e0 + m)], h2 = en_pair[2 * (size_t)(e0 + m) + 1], v = H + m;
            const float *ra1 = ts.a12 + (size_t)(nb + h1) * 32, *ra2 = ts.a12 + (size_t)(nb + h2) * 32, *ra3 = ts.a12 + (size_t)(nb + v) * 32;
            const float a2v = ra3[16];
            float e1 = ra1[0] + a2v, e2 = ra2[0] + a2v, e3 = ra3[0] + a2v;
            e1 = e1 > 0.f ? e1 : e1 * ts.alpha;
            e2 = e2 > 0.f ? e2 : e2 * ts.alpha;
            e3 = e3 > 0.f ? e3 : e3 * ts.alpha;
            const float mx = fmaxf(fmaxf(e1, e2), e3);
            const float x1 = expf(e1 - mx), x2 = expf(e2 - mx), x3 = expf(e3 - mx);
            const float sum = (x1 + x2) + x3;
            const float w1 = x1 / sum, w2 = x2 / sum, w3 = x3 / sum;
            const float v1 = ts.ft2[(size_t)(nb + h1) * ts.ld], v2 = ts.ft2[(size_t)(nb + h2) * ts.ld], v3 = ts.ft2[(size_t)(nb + v) * ts.ld];
            float acc = v1 * w1;
            acc = acc + v2 * w2;
            acc = acc + v3 * w3;
            float o = acc;
            if (ts.out_mode == 0) o = acc > 0.f ? acc : acc * ts.out_slope;
            else if (ts.out_mode == 1) o = 1.f / (1.f + expf(-acc));
            ts.scores[e0 + m] = H > hmax ? 0.f : o;
        }
        __syncthreads();                                        // (the wave's own stores, visible to its own loads)
        cluster_wave_body(s_keys, f, cfg, n_frames, head_off, en_off, head_cam, en_pair, ts.scores, pcap, n_pow2, hmax, persons, n_persons);
        return;
    }
    // every (edge-node, joint) of the batch: the pair's two skeletons, lower camera first
    const size_t i = (size_t)(blockIdx.x - n_frames) * 256 + threadIdx.x;
    if (i >= (size_t)n_en * J) return;
    const int m = (int)(i / J), j = (int)(i - (size_t)m * J);
    const int f = en_frame[m];
    const int hb = head_off[f], H = head_off[f + 1] - hb;
    if (H > hmax) return;
    int la = en_pair[2 * (size_t)m], lb = en_pair[2 * (size_t)m + 1];
    int ca = head_cam[hb + la], cb = head_cam[hb + lb];
    if (ca > cb) {
        const int t = la;
        la = lb;
        lb = t;
        const int c = ca;
        ca = cb;
        cb = c;
    }
    const int ga = hb + la, gb = hb + lb;
    if (!((tri_mask[ga] >> j) & 1u) || !((tri_mask[gb] >> j) & 1u)) return;
    double ua, va, ub, vb;
    dltc::undistort_point(cfg, ca, xy[((size_t)ga * J + j) * 2], xy[((size_t)ga * J + j) * 2 + 1], &ua, &va);
    dltc::undistort_point(cfg, cb, xy[((size_t)gb * J + j) * 2], xy[((size_t)gb * J + j) * 2 + 1], &ub, &vb);
    dltc::dlt_solve(cfg->P[ca], cfg->P[cb], ua, va, ub, vb, pair_pts + (((size_t)ga * hmax + lb) * J + j) * 3);
}

// Can the tail launch serve this context's frames?  (the one-wave clustering kernel: at most 64 skeletons per frame)
bool lat_tail_available(int hmax, size_t keys_per_frame) {
    if (getenv("MPE_CLUSTER_KERNEL") || hmax > 64) return false;
    int n_pow2 = 64;
    while ((size_t)n_pow2 < keys_per_frame) n_pow2 <<= 1;
    return (size_t)n_pow2 * (sizeof(uint64_t) + sizeof(uint32_t)) + 256 * sizeof(int32_t) <= 48 * 1024;
}

hipError_t launch_lat_tail(hipStream_t s, const DevCfg *cfg, const mpe_batch &b, int J, const int32_t *en_pair, const int32_t *en_frame,
                           const float *ft2, int ld, const float *a12, float alpha, float out_slope, int out_mode, const int32_t *node_off,
                           float *scores, int pcap, int hmax, size_t keys_per_frame, int32_t *persons, int32_t *n_persons, double *pair_pts) {
    if (b.n_frames <= 0) return hipSuccess;
    int n_pow2 = 64;
    while ((size_t)n_pow2 < keys_per_frame) n_pow2 <<= 1;
    const size_t shm = (size_t)n_pow2 * (sizeof(uint64_t) + sizeof(uint32_t)) + 256 * sizeof(int32_t);
    TailScores ts{ft2, ld, a12, alpha, out_slope, out_mode, node_off, scores};
    const size_t items = pair_pts ? (size_t)b.n_edge_nodes * J : 0;
    const unsigned grid = (unsigned)(b.n_frames + (items + 255) / 256);
    hipLaunchKernelGGL(k_lat_tail, dim3(grid), dim3(256), shm, s, cfg, b.n_frames, pair_pts ? b.n_edge_nodes : 0, J, b.d_frame_head_off,
                       b.d_frame_en_off, b.d_head_cam, en_pair, en_frame, ts, pcap, n_pow2, hmax, persons, n_persons, b.d_tri_mask, b.d_xy,
                       pair_pts);
    return hipGetLastError();
}

// ---- workgroup variant for large frames (> 64 heads: 5 x 10+, 23 cameras) ----------------
// One workgroup of 1024 threads per frame.  Only the matchings above the threshold are kept
// (compaction) and bitonic-sorted -- in LDS up to CB_KCAP keys, in the frame's global key
// scratch beyond.  The greedy rules run as in k_cluster_wave: 1024 pending matchings are tested
// against the LDS-resident state at once, the first one that passes is applied by its own
// thread, one round per accepted matching.  Components are the human-index groups; their output
// order is the rank of their first node in G's insertion order, computed in parallel; thread 0
// replays _plain_bfs + the CPython set only for groups that hold two heads of one camera.
constexpr int CB_THREADS = 1024;
constexpr int CB_WAVES = CB_THREADS / 64;
constexpr int CB_KCAP = 8192;
constexpr int CB_INF = 0x7FFFFFFF;

__global__ __launch_bounds__(CB_THREADS) void k_cluster_block(const DevCfg *__restrict__ cfg, int n_frames,
                                                              const int32_t *__restrict__ head_off,
                                                              const int32_t *__restrict__ en_off,
                                                              const int32_t *__restrict__ head_cam,
                                                              const int32_t *__restrict__ en_pair,
                                                              const float *__restrict__ scores, int pcap, int hmax,
                                                              int table_cap, uint64_t *__restrict__ keys_all,
                                                              size_t keys_per_frame, int32_t *__restrict__ persons,
                                                              int32_t *__restrict__ n_persons) {
    extern __shared__ uint64_t s_keys[];                      // [CB_KCAP], then the int arrays
    const int f = blockIdx.x;
    if (f >= n_frames) return;
    const int V = cfg->V;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int h0 = head_off[f], H = head_off[f + 1] - h0;
    const int e0 = en_off[f], M = en_off[f + 1] - e0;
    int32_t *out = persons + (size_t)f * pcap * V;
    for (int i = t; i < pcap * V; i += CB_THREADS) out[i] = -1;
    if (M <= 0 || H > hmax || (size_t)M > keys_per_frame) {
        if (t == 0) n_persons[f] = 0;
        return;
    }
    const Work w = carve(reinterpret_cast<int32_t *>(s_keys + CB_KCAP), hmax, table_cap);
    int32_t *s_cam = w.tabB + table_cap;
    int32_t *s_first = s_cam + hmax, *s_gor = s_first + hmax, *s_gcnt = s_gor + hmax;
    int32_t *s_gfirst = s_gcnt + hmax, *s_gnp = s_gfirst + hmax;
    int32_t *s_sc = s_gnp + hmax;                             // scalars
    enum { NVALID = 0, NE, CUR, MFROM, MTO, NNODES, SLOT, NP, NONPLAIN, Q0 };   // Q0 .. Q0 + CB_WAVES - 1
    int32_t *s_rank = w.done, *s_ord = w.order;
    const float thr = cfg->threshold;
    const int min_views = cfg->min_views;
    const int32_t *prs = en_pair + 2 * (size_t)e0;

    if (t < 16 + CB_WAVES) s_sc[t] = 0;
    for (int h = t; h < H; h += CB_THREADS) {
        const int c = head_cam[h0 + h];
        s_cam[h] = c;
        w.linked[h] = (int32_t)(1u << c);
        w.human[h] = -1;
        w.cfh[h] = 0;
        s_first[h] = CB_INF;
        s_gor[h] = 0;
        s_gcnt[h] = 0;
        s_gfirst[h] = CB_INF;
        s_gnp[h] = -1;
    }
    __threadfence_block();                                    // the -1 fill lands before any person row
    __syncthreads();

    // matchings above the threshold: count, then compact (order is irrelevant, the sort follows);
    // first appearance of every head in the edge scan on the way
    {
        int mine = 0;
        for (int m = t; m < M; m += CB_THREADS) {
            mine += scores[e0 + m] > thr ? 1 : 0;
            atomicMin(&s_first[prs[2 * m]], 2 * m);
            atomicMin(&s_first[prs[2 * m + 1]], 2 * m + 1);
        }
        if (mine) atomicAdd(&s_sc[NVALID], mine);
    }
    __syncthreads();
    const int n_valid = s_sc[NVALID];
    int n = 2 * CB_THREADS;
    while (n < n_valid) n <<= 1;
    uint64_t *keys = n <= CB_KCAP ? s_keys : keys_all + (size_t)f * keys_per_frame;
    for (int m = t; m < M; m += CB_THREADS) {
        const uint64_t key = make_key(scores[e0 + m], thr, m);
        if (key != KEY_NONE) keys[atomicAdd(&s_sc[SLOT], 1)] = key;
    }
    for (int i = n_valid + t; i < n; i += CB_THREADS) keys[i] = KEY_NONE;
    // G.add_node order: rank of the first appearance
    for (int h = t; h < H; h += CB_THREADS) {
        const int fv = s_first[h];
        if (fv == CB_INF) continue;
        int rank = 0;
        for (int l = 0; l < H; ++l) rank += s_first[l] < fv ? 1 : 0;
        s_rank[h] = rank;
        s_ord[rank] = h;
        atomicAdd(&s_sc[NNODES], 1);
    }
    __threadfence_block();
    __syncthreads();
    // bitonic network over n keys.  Keys that do not fit the LDS array live in global memory; the
    // compare-exchange passes whose partner distance stays inside a CB_KCAP-key chunk (all but a few)
    // then run on one chunk at a time staged in LDS.  The keys are unique (low bits = matching index),
    // so the sorted order does not depend on how the network is scheduled.
    auto pass = [&](uint64_t *buf, int cnt, int j, int k, int base) {
        for (int x = t; x < cnt / 2; x += CB_THREADS) {
            const int i = ((x & ~(j - 1)) << 1) | (x & (j - 1));
            const int p = i | j;
            const bool up = ((base + i) & k) == 0;
            const uint64_t a = buf[i], b = buf[p];
            if ((a > b) == up) {
                buf[i] = b;
                buf[p] = a;
            }
        }
        __threadfence_block();
        __syncthreads();
    };
    if (n <= CB_KCAP) {
        for (int k = 2; k <= n; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) pass(keys, n, j, k, 0);
    } else {
        for (int k = CB_KCAP; k <= n; k <<= 1) {
            for (int j = k >> 1; j >= CB_KCAP; j >>= 1) pass(keys, n, j, k, 0);       // partners in different chunks
            for (int c0 = 0; c0 < n; c0 += CB_KCAP) {
                for (int i = t; i < CB_KCAP; i += CB_THREADS) s_keys[i] = keys[c0 + i];
                __syncthreads();
                if (k == CB_KCAP) {                                                   // all stages up to the chunk size
                    for (int kk = 2; kk <= CB_KCAP; kk <<= 1)
                        for (int j = kk >> 1; j > 0; j >>= 1) pass(s_keys, CB_KCAP, j, kk, c0);
                } else {
                    for (int j = CB_KCAP >> 1; j > 0; j >>= 1) pass(s_keys, CB_KCAP, j, k, c0);
                }
                for (int i = t; i < CB_KCAP; i += CB_THREADS) keys[c0 + i] = s_keys[i];
                __threadfence_block();
                __syncthreads();
            }
        }
    }
    const int n_nodes = s_sc[NNODES];

    // greedy rules, 1024 pending matchings per chunk, one round per accepted matching
    for (int k0 = 0; k0 < n_valid; k0 += CB_THREADS) {
        const bool valid = k0 + t < n_valid;
        int h1 = 0, h2 = 0;
        if (valid) {
            const int m = (int)(uint32_t)keys[k0 + t];
            h1 = prs[2 * m];
            h2 = prs[2 * m + 1];
        }
        const bool first1 = pair_first_is_h1(h1, h2);
        const int a = first1 ? h1 : h2, b = first1 ? h2 : h1;
        const uint32_t ba = 1u << s_cam[a], bb = 1u << s_cam[b];
        int start = 0;
        while (true) {
            __syncthreads();                                  // state of the previous round is final
            const uint32_t la = (uint32_t)w.linked[a], lb = (uint32_t)w.linked[b];
            const int ha = w.human[a], hb = w.human[b];
            const uint32_t ca = ha >= 0 ? (uint32_t)w.cfh[ha] : 0u, cb = hb >= 0 ? (uint32_t)w.cfh[hb] : 0u;
            const bool rej = (lb & ba) || (la & bb) || (ca & bb) || (cb & ba) || (ha >= 0 && hb >= 0 && (cb & ca));
            const unsigned long long bal = __ballot(valid && t >= start && !rej);
            if (lane == 0) s_sc[Q0 + wave] = bal ? wave * 64 + __builtin_ctzll(bal) : CB_INF;
            __syncthreads();
            int q = CB_INF;
#pragma unroll
            for (int wv = 0; wv < CB_WAVES; ++wv) q = min(q, s_sc[Q0 + wv]);
            if (q == CB_INF) break;
            if (t == q) {
                int mfrom = -1;
                if (ha < 0 && hb < 0) {
                    const int cur = s_sc[CUR];
                    w.human[a] = cur;
                    w.human[b] = cur;
                    w.cfh[cur] = (int32_t)(ba | bb);
                    s_sc[CUR] = cur + 1;
                } else if (ha >= 0 && hb < 0) {
                    w.human[b] = ha;
                    w.cfh[ha] = (int32_t)(ca | bb);
                } else if (hb >= 0 && ha < 0) {
                    w.human[a] = hb;
                    w.cfh[hb] = (int32_t)(cb | ba);
                } else {
                    mfrom = hb;                               // relabelled by everybody below; the absorbed
                    s_sc[MTO] = ha;                           // group's camera list is dropped (:97-102)
                }
                s_sc[MFROM] = mfrom;
                const int ne = s_sc[NE];
                w.ea[ne] = a;
                w.eb[ne] = b;
                s_sc[NE] = ne + 1;
                w.linked[a] = (int32_t)(la | bb);
                w.linked[b] = (int32_t)(lb | ba);
            }
            __syncthreads();
            const int mfrom = s_sc[MFROM];
            if (mfrom >= 0) {
                const int mto = s_sc[MTO];
                for (int h = t; h < H; h += CB_THREADS)
                    if (w.human[h] == mfrom) w.human[h] = mto;
            }
            start = q + 1;
        }
    }
    __syncthreads();

    // components = human-index groups (+ unmatched single heads)
    for (int h = t; h < H; h += CB_THREADS) {
        const int g = w.human[h];
        if (g < 0) continue;
        atomicOr(&s_gor[g], 1 << s_cam[h]);
        atomicAdd(&s_gcnt[g], 1);
        atomicMin(&s_gfirst[g], s_rank[h]);
    }
    __syncthreads();
    const int n_groups = s_sc[CUR];
    const bool singles = min_views <= 1;
    auto emitted_before = [&](int fr) {
        int c = 0;
        for (int g = 0; g < n_groups; ++g) c += (s_gcnt[g] >= min_views && s_gcnt[g] > 0 && s_gfirst[g] < fr) ? 1 : 0;
        if (singles)
            for (int h = 0; h < H; ++h) c += (w.human[h] < 0 && s_first[h] != CB_INF && s_rank[h] < fr) ? 1 : 0;
        return c;
    };
    for (int g = t; g < n_groups; g += CB_THREADS) {
        const int cnt = s_gcnt[g];
        if (cnt <= 0 || cnt < min_views) continue;
        const int idx = emitted_before(s_gfirst[g]);
        atomicAdd(&s_sc[NP], 1);
        if (idx >= pcap) continue;
        s_gnp[g] = idx;
        if (__popc((unsigned)s_gor[g]) != cnt) s_sc[NONPLAIN] = 1;
    }
    if (singles)
        for (int h = t; h < H; h += CB_THREADS) {
            if (w.human[h] >= 0 || s_first[h] == CB_INF) continue;
            const int idx = emitted_before(s_rank[h]);
            atomicAdd(&s_sc[NP], 1);
            if (idx < pcap) out[(size_t)idx * V + s_cam[h]] = h;
        }
    __syncthreads();
    for (int h = t; h < H; h += CB_THREADS) {
        const int g = w.human[h];
        if (g < 0) continue;
        const int idx = s_gnp[g];
        if (idx >= 0 && __popc((unsigned)s_gor[g]) == s_gcnt[g]) out[(size_t)idx * V + s_cam[h]] = h;
    }
    if (t == 0) {
        n_persons[f] = s_sc[NP] < pcap ? s_sc[NP] : pcap;
        if (s_sc[NONPLAIN]) {
            // two heads of one camera in a group: networkx's set iteration order picks the winner
            build_csr(w, H, s_sc[NE]);
            for (int g = 0; g < n_groups; ++g) {
                const int idx = s_gnp[g];
                if (idx < 0 || __popc((unsigned)s_gor[g]) == s_gcnt[g]) continue;
                PySet set{w.tabA, w.tabB, 7, 0};
                bfs_component(w, s_ord[s_gfirst[g]], n_nodes, set);
                for (int i = 0; i <= set.mask; ++i) {
                    const int h = set.tab[i];
                    if (h >= 0) out[(size_t)idx * V + s_cam[h]] = h;
                }
            }
        }
    }
}

// ---- global-scratch variant (frames too large for LDS) ---------------------------------
__global__ __launch_bounds__(64) void k_cluster_big(const DevCfg *__restrict__ cfg, int n_frames,
                                                    const int32_t *__restrict__ head_off,
                                                    const int32_t *__restrict__ en_off,
                                                    const int32_t *__restrict__ head_cam,
                                                    const int32_t *__restrict__ en_pair,
                                                    const float *__restrict__ scores, int pcap, int hmax,
                                                    int table_cap, uint64_t *__restrict__ keys_all,
                                                    size_t keys_per_frame, int32_t *__restrict__ scratch_all,
                                                    size_t scratch_per_frame, int32_t *__restrict__ persons,
                                                    int32_t *__restrict__ n_persons) {
    const int f = blockIdx.x;
    if (f >= n_frames) return;
    const int V = cfg->V;
    const int h0 = head_off[f], H = head_off[f + 1] - h0;
    const int e0 = en_off[f], M = en_off[f + 1] - e0;
    int32_t *out = persons + (size_t)f * pcap * V;
    for (int i = threadIdx.x; i < pcap * V; i += blockDim.x) out[i] = -1;
    if (M <= 0 || H > hmax || (size_t)M > keys_per_frame) {
        if (threadIdx.x == 0) n_persons[f] = 0;
        return;
    }
    uint64_t *keys = keys_all + (size_t)f * keys_per_frame;
    const float thr = cfg->threshold;
    for (int m = threadIdx.x; m < M; m += blockDim.x) keys[m] = make_key(scores[e0 + m], thr, m);
    __syncthreads();
    if (threadIdx.x != 0) return;
    // heapsort ascending
    for (int start = M / 2 - 1; start >= 0; --start) sift_down(keys, start, M - 1);
    for (int end = M - 1; end > 0; --end) {
        const uint64_t t = keys[0];
        keys[0] = keys[end];
        keys[end] = t;
        sift_down(keys, 0, end - 1);
    }
    Work w = carve(scratch_all + (size_t)f * scratch_per_frame, hmax, table_cap);
    const int32_t *prs = en_pair + 2 * (size_t)e0;
    auto pair_of = [&](int m, int &h1, int &h2) {
        h1 = prs[2 * m];
        h2 = prs[2 * m + 1];
    };
    n_persons[f] = greedy_and_components(w, keys, M, pair_of, H, M, head_cam + h0, V, cfg->min_views, pcap, out);
}

hipError_t launch_cluster(hipStream_t s, const DevCfg *cfg, const mpe_batch &b, const int32_t *en_pair,
                          const float *scores, int pcap, int hmax, uint64_t *keys, size_t keys_per_frame,
                          int32_t *scratch, size_t scratch_per_frame, int32_t *persons, int32_t *n_persons) {
    if (b.n_frames <= 0) return hipSuccess;
    const int table_cap = cluster_table_cap(hmax);
    size_t m_cap = keys_per_frame;
    int n_pow2 = 64;
    while ((size_t)n_pow2 < m_cap) n_pow2 <<= 1;
    const size_t shm = (size_t)n_pow2 * (sizeof(uint64_t) + sizeof(uint32_t)) +
                       ((size_t)13 * hmax + 1 + 2 * (size_t)table_cap + hmax) * sizeof(int32_t);
    // kernel choice: per-lane registers when a frame cannot hold more than 64 heads; LDS work
    // arrays up to the LDS budget; global scratch beyond.  MPE_CLUSTER_KERNEL=wave|lds|big
    // overrides (tests run the known answers through every variant).
    const char *force = getenv("MPE_CLUSTER_KERNEL");
    const bool want_wave = force ? !strcmp(force, "wave") : hmax <= 64;
    const size_t shm_wave = (size_t)n_pow2 * (sizeof(uint64_t) + sizeof(uint32_t)) + 256 * sizeof(int32_t);
    if (want_wave && shm_wave <= 96 * 1024) {
        if (shm_wave > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cluster_wave),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_wave);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_cluster_wave, dim3(b.n_frames), dim3(64), shm_wave, s, cfg, b.n_frames, b.d_frame_head_off,
                           b.d_frame_en_off, b.d_head_cam, en_pair, scores, pcap, n_pow2, hmax, persons, n_persons);
        return hipGetLastError();
    }
    const size_t shm_block = (size_t)CB_KCAP * sizeof(uint64_t) +
                             ((size_t)19 * hmax + 1 + 2 * (size_t)table_cap + 16 + CB_WAVES) * sizeof(int32_t);
    const bool want_block = force ? !strcmp(force, "block") : true;
    if (want_block && shm_block <= 128 * 1024) {
        static PerDeviceFlag attr_done;
        if (!attr_done.test()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cluster_block),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return e;
            attr_done.set();
        }
        hipLaunchKernelGGL(k_cluster_block, dim3(b.n_frames), dim3(CB_THREADS), shm_block, s, cfg, b.n_frames,
                           b.d_frame_head_off, b.d_frame_en_off, b.d_head_cam, en_pair, scores, pcap, hmax, table_cap,
                           keys, keys_per_frame, persons, n_persons);
        return hipGetLastError();
    }
    if (shm <= 96 * 1024 && !(force && !strcmp(force, "big"))) {
        if (shm > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cluster_lds),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_cluster_lds, dim3(b.n_frames), dim3(64), shm, s, cfg, b.n_frames, b.d_frame_head_off,
                           b.d_frame_en_off, b.d_head_cam, en_pair, scores, pcap, hmax, table_cap, n_pow2, persons,
                           n_persons);
    } else {
        hipLaunchKernelGGL(k_cluster_big, dim3(b.n_frames), dim3(64), 0, s, cfg, b.n_frames, b.d_frame_head_off,
                           b.d_frame_en_off, b.d_head_cam, en_pair, scores, pcap, hmax, table_cap, keys,
                           keys_per_frame, scratch, scratch_per_frame, persons, n_persons);
    }
    return hipGetLastError();
}

}  // namespace mpe
