// Greedy person clustering, one wavefront per frame (sequential integer logic; lane 0
// replays the reference's algorithm, the other lanes help with the sort).
//
// Restates get_person_proposal_from_network_output (reference
// utils/skeleton_matching_utils.py:12-132) on the implicit topology, bit for bit:
//   * a matching exists for edge-node X=(h1,h2) when score[X] > threshold (:49-55); which of
//     the two heads is `a` follows CPython's iteration order of the 2-element set (:53);
//   * stable sort by score, descending (:60) -> 64-bit keys (inverted score bits, creation
//     index), so ties keep creation order;
//   * camera-uniqueness tests and human-index bookkeeping (:61-108) with camera sets as
//     32-bit masks, including the quirk that merging two humans drops the absorbed group's
//     camera list (:97-102);
//   * connected components in node-insertion order (:117-130) with networkx 3.4.2's
//     _plain_bfs, whose result is a Python set: iteration order of that set decides which
//     head wins when a component holds two heads of one camera, so CPython's set
//     (open addressing, LINEAR_PROBES 9, PERTURB_SHIFT 5, growth x4) is emulated exactly.
#include "mpe_internal.h"

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

size_t cluster_keys_per_frame(int hmax) { return (size_t)hmax * hmax / 2 + 1; }

size_t cluster_scratch_per_frame(int hmax) {
    return (size_t)10 * hmax + 2 * (size_t)cluster_table_cap(hmax) + cluster_keys_per_frame(hmax);
}

__global__ __launch_bounds__(64) void k_cluster(const DevCfg *__restrict__ cfg, int n_frames,
                                                const int32_t *__restrict__ head_off,
                                                const int32_t *__restrict__ en_off,
                                                const int32_t *__restrict__ head_cam,
                                                const int32_t *__restrict__ en_pair, const float *__restrict__ scores,
                                                int pcap, int hmax, int table_cap, uint64_t *__restrict__ keys_all,
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
    if (threadIdx.x != 0) return;
    if (M <= 0 || H > hmax || (size_t)M > keys_per_frame) {
        n_persons[f] = 0;
        return;
    }
    uint64_t *keys = keys_all + (size_t)f * keys_per_frame;
    int32_t *sc = scratch_all + (size_t)f * scratch_per_frame;
    int32_t *seen = sc;
    int32_t *order = seen + hmax;
    int32_t *linked = order + hmax;
    int32_t *human = linked + hmax;
    int32_t *cfh = human + hmax;
    int32_t *ea = cfh + hmax;
    int32_t *eb = ea + hmax;
    int32_t *done = eb + hmax;
    int32_t *lvl = done + hmax;
    int32_t *nxt = lvl + hmax;
    int32_t *tabA = nxt + hmax;
    int32_t *tabB = tabA + table_cap;
    int32_t *ab = tabB + table_cap;
    const int32_t *cam = head_cam + h0;
    const float thr = cfg->threshold;

    for (int h = 0; h < H; ++h) {
        seen[h] = 0;
        human[h] = -1;
        done[h] = 0;
        linked[h] = (int32_t)(1u << cam[h]);
    }
    int n_nodes = 0, nm = 0;
    for (int m = 0; m < M; ++m) {
        const int h1 = en_pair[2 * (size_t)(e0 + m) + 0], h2 = en_pair[2 * (size_t)(e0 + m) + 1];
        if (!seen[h1]) { seen[h1] = 1; order[n_nodes++] = h1; }
        if (!seen[h2]) { seen[h2] = 1; order[n_nodes++] = h2; }
        const float s = scores[e0 + m];
        if (s > thr) {
            const bool first1 = pair_first_is_h1(h1, h2);
            const int a = first1 ? h1 : h2, b = first1 ? h2 : h1;
            ab[nm] = (a << 16) | b;
            keys[nm] = ((uint64_t)(0xFFFFFFFFu - __float_as_uint(s)) << 32) | (uint32_t)nm;
            ++nm;
        }
    }
    // heapsort ascending on (inverted score bits, creation index)
    for (int start = nm / 2 - 1; start >= 0; --start) sift_down(keys, start, nm - 1);
    for (int end = nm - 1; end > 0; --end) {
        const uint64_t t = keys[0];
        keys[0] = keys[end];
        keys[end] = t;
        sift_down(keys, 0, end - 1);
    }
    int cur = 0, ne = 0;
    for (int k = 0; k < nm; ++k) {
        const int idx = (int)(keys[k] & 0xFFFFFFFFu);
        const int a = ab[idx] >> 16, b = ab[idx] & 0xFFFF;
        const uint32_t ba = 1u << cam[a], bb = 1u << cam[b];
        if (((uint32_t)linked[b] & ba) || ((uint32_t)linked[a] & bb)) continue;
        const int ha = human[a], hb = human[b];
        if (ha >= 0 && ((uint32_t)cfh[ha] & bb)) continue;
        if (hb >= 0 && ((uint32_t)cfh[hb] & ba)) continue;
        if (ha < 0 && hb < 0) {
            human[a] = cur;
            human[b] = cur;
            cfh[cur] = (int32_t)(ba | bb);
            ++cur;
        } else if (ha >= 0 && hb < 0) {
            human[b] = ha;
            cfh[ha] = (int32_t)((uint32_t)cfh[ha] | bb);
        } else if (hb >= 0 && ha < 0) {
            human[a] = hb;
            cfh[hb] = (int32_t)((uint32_t)cfh[hb] | ba);
        } else {
            if ((uint32_t)cfh[hb] & (uint32_t)cfh[ha]) continue;
            for (int n = 0; n < H; ++n)
                if (human[n] == hb) human[n] = ha;
            // the absorbed group's camera list is dropped, not merged (reference :97-102)
        }
        ea[ne] = a;
        eb[ne] = b;
        ++ne;
        linked[a] = (int32_t)((uint32_t)linked[a] | bb);
        linked[b] = (int32_t)((uint32_t)linked[b] | ba);
    }
    // connected components in node-insertion order
    int np = 0;
    const int min_views = cfg->min_views;
    for (int oi = 0; oi < n_nodes; ++oi) {
        const int v = order[oi];
        if (done[v]) continue;
        PySet set{tabA, tabB, 7, 0};
        pyset_init(set);
        pyset_add(set, v);
        int nl = 1;
        lvl[0] = v;
        bool full = set.fill == n_nodes;
        int32_t *L = lvl, *N = nxt;
        while (nl > 0 && !full) {
            int nn = 0;
            for (int li = 0; li < nl; ++li) {
                const int x = L[li];
                for (int e = 0; e < ne; ++e) {
                    int w = -1;
                    if (ea[e] == x) w = eb[e];
                    else if (eb[e] == x) w = ea[e];
                    if (w >= 0 && !pyset_contains(set, w)) {
                        pyset_add(set, w);
                        N[nn++] = w;
                    }
                }
                if (set.fill == n_nodes) { full = true; break; }
            }
            int32_t *t = L; L = N; N = t;
            nl = nn;
        }
        const int cnt = set.fill;
        int32_t *pout = (cnt >= min_views && np < pcap) ? out + (size_t)np * V : nullptr;
        for (int i = 0; i <= set.mask; ++i) {
            const int h = set.tab[i];
            if (h < 0) continue;
            done[h] = 1;
            if (pout) pout[cam[h]] = h;
        }
        if (pout) ++np;
    }
    n_persons[f] = np;
}

hipError_t launch_cluster(hipStream_t s, const DevCfg *cfg, const mpe_batch &b, const int32_t *en_pair,
                          const float *scores, int pcap, int hmax, uint64_t *keys, size_t keys_per_frame,
                          int32_t *scratch, size_t scratch_per_frame, int32_t *persons, int32_t *n_persons) {
    if (b.n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_cluster, dim3(b.n_frames), dim3(64), 0, s, cfg, b.n_frames, b.d_frame_head_off,
                       b.d_frame_en_off, b.d_head_cam, en_pair, scores, pcap, hmax, cluster_table_cap(hmax), keys,
                       keys_per_frame, scratch, scratch_per_frame, persons, n_persons);
    return hipGetLastError();
}

}  // namespace mpe
