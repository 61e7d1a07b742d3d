// Frame-JSON -> packed batch, native host code.
//
// The reference's wire format (panoptic_conversor/get_joints_from_panoptic_model_multi.py:
// 231-236, 281, 287, 303-307; consumers test/metrics_from_model.py:118-140, 184) is a list of
// frames; a frame maps camera name -> [ "<JSON text of the skeleton list>", timestamp,
// "no_image", bodies_3D ]; a skeleton maps joint id string -> [id, x, y, valid, prob], with an
// optional "ID" entry.  The skeleton list is JSON inside a JSON string, so the reference
// parses every frame twice in Python (json.load, then json.loads per camera).  Here one pass
// of a small recursive-descent scanner does both levels and writes the structure-of-arrays
// batch of include/mpe.h directly; frames are independent, so they are parsed by a pool of
// threads after a bracket-depth scan has located their extents.
//
// Ordering rules are those of MergedMultipleHumansDataset.load_people_view_graph (reference
// graph_generator.py:573-605), identical to packing.py: cameras in key order, cameras outside
// used_cameras_skeleton_matching skipped, skeletons without a joint key skipped, numbers
// converted with strtod (correctly rounded, like Python's float()).
#include <sched.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mpe.h"

namespace {

struct Head {
    int32_t cam;
    uint32_t joint_mask, tri_mask;
    int32_t skeleton_index;
    double xy[MPE_MAX_JOINTS][2];
    float vp[MPE_MAX_JOINTS][2];
};

struct HeadInfo {
    int32_t cam;
    uint32_t joint_mask, tri_mask;
    int32_t skeleton_index;
};

struct FrameOut {                    // compact per-frame result (J values per head, not MPE_MAX_JOINTS)
    std::vector<HeadInfo> heads;
    std::vector<double> xy;          // [heads][J][2]
    std::vector<float> vp;           // [heads][J][2]
    std::vector<int32_t> slot_cam, slot_n;
    std::vector<int32_t> extent;     // optional (mpe_pack_views_into): [heads][2] byte offsets of each skeleton object in its camera text
    std::string error;
};

struct Cursor {
    const char *p, *end;
    bool inner;        // true while scanning text that sits inside a JSON string (one escape level)
};

// Whitespace.  Inside a JSON string (`inner`) a line break or tab of the inner document arrives as the two-character
// escape \n / \t / \r (a pretty-printed skeleton list, json.dumps(..., indent=1), then dumped again): json.loads of the
// outer document turns it back into whitespace before the inner json.loads sees it, so it is whitespace here too.
inline void skip_ws(Cursor &c) {
    while (c.p < c.end) {
        if (*c.p == ' ' || *c.p == '\n' || *c.p == '\t' || *c.p == '\r') { ++c.p; continue; }
        if (c.inner && *c.p == '\\' && c.p + 1 < c.end && (c.p[1] == 'n' || c.p[1] == 't' || c.p[1] == 'r')) { c.p += 2; continue; }
        break;
    }
}

// In "inner" mode the text is the body of a JSON string: a quote appears as \" and a
// backslash as \\ .  The skeleton lists contain only ASCII keys and numbers, so only these
// two escapes can occur inside them.
inline bool at_quote(const Cursor &c) {
    if (c.inner) return c.p + 1 < c.end && c.p[0] == '\\' && c.p[1] == '"';
    return c.p < c.end && *c.p == '"';
}

inline void eat_quote(Cursor &c) { c.p += c.inner ? 2 : 1; }

bool read_string(Cursor &c, std::string *out) {
    if (!at_quote(c)) return false;
    eat_quote(c);
    if (out) out->clear();
    while (c.p < c.end) {
        if (at_quote(c)) {
            eat_quote(c);
            return true;
        }
        if (!c.inner && *c.p == '\\') {          // escape at the outer level: copy the escaped char
            if (c.p + 1 >= c.end) return false;
            if (out) out->push_back(c.p[1]);
            c.p += 2;
            continue;
        }
        if (c.inner && *c.p == '"') return false;   // the enclosing string ended unexpectedly
        if (out) out->push_back(*c.p);
        ++c.p;
    }
    return false;
}

// A key of the OUTER document (a camera name) with JSON's escapes resolved as json.loads resolves them: \" \\ \/ \b \f \n
// \r \t and \uXXXX (surrogate pairs -> one code point; output UTF-8).  read_string copies the character behind a backslash
// as it stands, which is right for \" and \\ only: "tracker\u0061" must look up `trackera`, not be dropped as unknown.
bool read_key(Cursor &c, std::string *out) {
    if (c.inner || c.p >= c.end || *c.p != '"') return false;
    ++c.p;
    out->clear();
    auto hex4 = [&](const char *q, unsigned *v) {
        unsigned x = 0;
        for (int i = 0; i < 4; ++i) {
            const char h = q[i];
            x <<= 4;
            if (h >= '0' && h <= '9') x |= (unsigned)(h - '0');
            else if (h >= 'a' && h <= 'f') x |= (unsigned)(h - 'a' + 10);
            else if (h >= 'A' && h <= 'F') x |= (unsigned)(h - 'A' + 10);
            else return false;
        }
        *v = x;
        return true;
    };
    while (c.p < c.end) {
        const char ch = *c.p;
        if (ch == '"') { ++c.p; return true; }
        if (ch != '\\') { out->push_back(ch); ++c.p; continue; }
        if (c.p + 1 >= c.end) return false;
        const char e = c.p[1];
        c.p += 2;
        switch (e) {
            case '"': out->push_back('"'); break;
            case '\\': out->push_back('\\'); break;
            case '/': out->push_back('/'); break;
            case 'b': out->push_back('\b'); break;
            case 'f': out->push_back('\f'); break;
            case 'n': out->push_back('\n'); break;
            case 'r': out->push_back('\r'); break;
            case 't': out->push_back('\t'); break;
            case 'u': {
                unsigned cp = 0;
                if (c.end - c.p < 4 || !hex4(c.p, &cp)) return false;
                c.p += 4;
                if (cp >= 0xD800 && cp <= 0xDBFF && c.end - c.p >= 6 && c.p[0] == '\\' && c.p[1] == 'u') {
                    unsigned lo = 0;
                    if (hex4(c.p + 2, &lo) && lo >= 0xDC00 && lo <= 0xDFFF) {
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        c.p += 6;
                    }
                }
                if (cp < 0x80) out->push_back((char)cp);
                else if (cp < 0x800) { out->push_back((char)(0xC0 | (cp >> 6))); out->push_back((char)(0x80 | (cp & 0x3F))); }
                else if (cp < 0x10000) { out->push_back((char)(0xE0 | (cp >> 12))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
                else { out->push_back((char)(0xF0 | (cp >> 18))); out->push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
                break;
            }
            default: return false;                  // not a JSON escape: json.loads rejects it
        }
    }
    return false;
}

bool read_number(Cursor &c, double *v) {
    skip_ws(c);
    // Clinger fast path: <= 15 significant digits and |exp10| <= 22 are exact in binary64
    // arithmetic; everything else goes to glibc's strtod (correctly rounded, lock-free).
    // (libstdc++ 11's from_chars<double> takes the global locale lock on every call.)
    double d = 0;
    const char *e = c.p;
    {
        const char *q = c.p;
        bool neg = false;
        if (q < c.end && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
        uint64_t m = 0;
        int digits = 0, exp10 = 0;
        const char *ds = q;
        bool dropped = false;     // a non-zero digit beyond the 19th was discarded -> inexact
        while (q < c.end && *q >= '0' && *q <= '9') {
            if (digits < 19) { m = m * 10 + (uint64_t)(*q - '0'); if (m) ++digits; } else { ++exp10; dropped = dropped || *q != '0'; }
            ++q;
        }
        bool any = q > ds;
        if (q < c.end && *q == '.') {
            ++q;
            const char *fs = q;
            while (q < c.end && *q >= '0' && *q <= '9') {
                if (digits < 19) { m = m * 10 + (uint64_t)(*q - '0'); if (m) ++digits; --exp10; } else dropped = dropped || *q != '0';
                ++q;
            }
            any = any || q > fs;
        }
        bool ok = any;
        const bool exact_in = !dropped;
        if (ok && q < c.end && (*q == 'e' || *q == 'E')) {
            const char *es = q + 1;
            bool eneg = false;
            if (es < c.end && (*es == '-' || *es == '+')) { eneg = *es == '-'; ++es; }
            int ev = 0;
            const char *e0 = es;
            while (es < c.end && *es >= '0' && *es <= '9' && ev < 10000) { ev = ev * 10 + (*es - '0'); ++es; }
            if (es > e0) { exp10 += eneg ? -ev : ev; q = es; }
        }
        static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                     1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        static const long double p10l[] = {1e0L, 1e1L, 1e2L, 1e3L, 1e4L, 1e5L, 1e6L, 1e7L, 1e8L, 1e9L, 1e10L,
                                           1e11L, 1e12L, 1e13L, 1e14L, 1e15L, 1e16L, 1e17L, 1e18L, 1e19L, 1e20L,
                                           1e21L, 1e22L, 1e23L, 1e24L, 1e25L, 1e26L, 1e27L};
        bool done = false;
        if (ok && exact_in && digits <= 15 && exp10 >= -22 && exp10 <= 22) {
            d = (double)m;
            d = exp10 < 0 ? d / p10[-exp10] : d * p10[exp10];
            done = true;
        } else if (ok && exact_in && exp10 >= -27 && exp10 <= 27 && m != 0) {
            // up to 19 digits: m and 10^|e| are exact in x87 extended precision (64-bit
            // significand), so m*10^e carries ONE rounding there; rounding that again to
            // binary64 is exact unless the extended value sits within 1 ulp of a binary64
            // midpoint -- only then fall back to strtod
            long double r = (long double)m;
            r = exp10 < 0 ? r / p10l[-exp10] : r * p10l[exp10];
            uint64_t mant;
            memcpy(&mant, &r, sizeof mant);
            const uint64_t low = mant & 0x7FFu;
            if (low < 0x3FFu || low > 0x401u) {
                d = (double)r;
                done = d >= 2.3e-308 && d <= 1.7e308;      // normal range only
            }
        }
        if (done) {
            if (neg) d = -d;
            e = q;
        } else if (ok) {
            // the buffer is length-delimited, not NUL-terminated: hand strtod a bounded copy of
            // the token [c.p, q) (q = end of the digits / fraction / exponent scanned above)
            char tok[128];
            std::string big;
            const size_t n = (size_t)(q - c.p);
            const char *src = tok;
            if (n < sizeof tok) {
                memcpy(tok, c.p, n);
                tok[n] = 0;
            } else {
                big.assign(c.p, n);
                src = big.c_str();
            }
            char *e2 = nullptr;
            d = strtod(src, &e2);
            e = c.p + (e2 - src);
        }
    }
    if (e == c.p) {
        // JSON literals that Python's json accepts where a number may stand
        if (c.end - c.p >= 4 && !strncmp(c.p, "true", 4)) { *v = 1; c.p += 4; return true; }
        if (c.end - c.p >= 5 && !strncmp(c.p, "false", 5)) { *v = 0; c.p += 5; return true; }
        if (c.end - c.p >= 3 && !strncmp(c.p, "NaN", 3)) { *v = NAN; c.p += 3; return true; }
        if (c.end - c.p >= 8 && !strncmp(c.p, "Infinity", 8)) { *v = INFINITY; c.p += 8; return true; }
        if (c.end - c.p >= 9 && !strncmp(c.p, "-Infinity", 9)) { *v = -INFINITY; c.p += 9; return true; }
        // null where a number may stand: the Python packer stores it into its float arrays as NaN (numpy's conversion of None)
        if (c.end - c.p >= 4 && !strncmp(c.p, "null", 4)) { *v = NAN; c.p += 4; return true; }
        return false;
    }
    c.p = e;
    *v = d;
    return true;
}

bool skip_value(Cursor &c);

typedef const char *(*find_special_fn)(const char *, const char *);
typedef const char *(*find_quote_fn)(const char *, const char *, const char *);
extern find_special_fn g_find_special;      // next of " { } [ ]          (scalar or AVX2, chosen at start-up)
extern find_quote_fn g_find_quote;          // next quote that closes a string

// Skip one balanced container at the OUTER level (plain JSON text, not the body of a string) without
// descending into it: jump from one structural byte to the next, strings skipped whole, bracket kinds
// checked with a small stack.  This is where the unused ground-truth lists of a frame (about 40 % of
// its bytes) go by.  Containers nested deeper than the stack fall back to the recursive skip.
bool skip_balanced(Cursor &c, bool *too_deep) {
    char stack[64];
    int depth = 0;
    const char *p = c.p;
    *too_deep = false;
    for (;;) {
        p = g_find_special(p, c.end);
        if (p >= c.end) return false;
        const char ch = *p;
        if (ch == '"') {
            const char *hit = g_find_quote(p + 1, c.end, p + 1);
            if (!hit) return false;
            p = hit + 1;
            continue;
        }
        if (ch == '{' || ch == '[') {
            if (depth == 64) { *too_deep = true; return false; }
            stack[depth++] = ch;
        } else {
            if (depth == 0 || stack[depth - 1] != (ch == '}' ? '{' : '[')) return false;
            if (--depth == 0) { c.p = p + 1; return true; }
        }
        ++p;
    }
}

bool skip_container(Cursor &c, char open, char close) {
    if (c.p >= c.end || *c.p != open) return false;
    ++c.p;
    skip_ws(c);
    if (c.p < c.end && *c.p == close) { ++c.p; return true; }
    while (c.p < c.end) {
        skip_ws(c);
        if (open == '{') {
            if (!read_string(c, nullptr)) return false;
            skip_ws(c);
            if (c.p >= c.end || *c.p != ':') return false;
            ++c.p;
        }
        if (!skip_value(c)) return false;
        skip_ws(c);
        if (c.p >= c.end) return false;
        if (*c.p == ',') { ++c.p; continue; }
        if (*c.p == close) { ++c.p; return true; }
        return false;
    }
    return false;
}

bool skip_value(Cursor &c) {
    skip_ws(c);
    if (c.p >= c.end) return false;
    if (at_quote(c)) return read_string(c, nullptr);
    if (!c.inner && (*c.p == '{' || *c.p == '[')) {
        bool too_deep = false;
        if (skip_balanced(c, &too_deep)) return true;
        if (!too_deep) return false;
    }
    if (*c.p == '{') return skip_container(c, '{', '}');
    if (*c.p == '[') return skip_container(c, '[', ']');
    if (c.end - c.p >= 4 && !strncmp(c.p, "null", 4)) { c.p += 4; return true; }
    // numbers and bare literals are skipped lexically (no conversion)
    const char *q = c.p;
    while (q < c.end && *q != ',' && *q != ']' && *q != '}' && *q != ' ' && *q != '\n' && *q != '\\' && *q != '"') ++q;
    if (q == c.p) return false;
    c.p = q;
    return true;
}

// one skeleton: {"5": [5, x, y, valid, prob], ..., "ID": n}
bool parse_skeleton(Cursor &c, int J, Head *h, std::string *err) {
    skip_ws(c);
    if (c.p >= c.end || *c.p != '{') { *err = "skeleton is not an object"; return false; }
    ++c.p;
    h->joint_mask = h->tri_mask = 0;
    memset(h->xy, 0, sizeof h->xy);
    memset(h->vp, 0, sizeof h->vp);
    skip_ws(c);
    if (c.p < c.end && *c.p == '}') { ++c.p; return true; }
    std::string key;
    while (c.p < c.end) {
        skip_ws(c);
        if (!read_string(c, &key)) { *err = "bad joint key"; return false; }
        skip_ws(c);
        if (c.p >= c.end || *c.p != ':') { *err = "missing ':'"; return false; }
        ++c.p;
        skip_ws(c);
        if (key == "ID") {
            if (!skip_value(c)) { *err = "bad ID value"; return false; }
        } else {
            char *e = nullptr;
            const long j = strtol(key.c_str(), &e, 10);
            if (e == key.c_str() || *e != 0 || j < 0 || j >= J) { *err = "joint id '" + key + "' out of range"; return false; }
            if (c.p >= c.end || *c.p != '[') { *err = "joint value is not a list"; return false; }
            ++c.p;
            double v[5] = {0, 0, 0, 0, 0};
            int n = 0;
            bool closed = false;
            while (c.p < c.end) {
                skip_ws(c);
                if (c.p >= c.end) break;
                if (*c.p == ']') { ++c.p; closed = true; break; }
                double d;
                const bool is_null = c.end - c.p >= 4 && !strncmp(c.p, "null", 4);
                if (!read_number(c, &d)) { *err = "bad number in joint"; return false; }
                if (n == 0 && is_null) { *err = "joint id is null"; return false; }     // `values[0] > 0.` raises on None (dataset :75)
                if (n < 5) v[n] = d;
                ++n;
                skip_ws(c);
                if (c.p < c.end && *c.p == ',') ++c.p;
            }
            if (!closed) { *err = "unterminated joint value"; return false; }
            if (n < 5) { *err = "joint value needs 5 numbers"; return false; }
            h->joint_mask |= 1u << j;
            if (v[0] > 0.) h->tri_mask |= 1u << j;
            h->xy[j][0] = v[1];
            h->xy[j][1] = v[2];
            h->vp[j][0] = (float)v[3];
            h->vp[j][1] = (float)v[4];
        }
        skip_ws(c);
        if (c.p >= c.end) break;
        if (*c.p == ',') { ++c.p; continue; }
        if (*c.p == '}') { ++c.p; return true; }
        *err = "bad skeleton object";
        return false;
    }
    *err = "unterminated skeleton";
    return false;
}

// the camera entry's first element: a JSON string holding the skeleton list (or, leniently,
// the list itself)
bool parse_skeleton_list(Cursor &c, int J, int cam, FrameOut *fo, int *n_here, const char *extent_base = nullptr) {
    skip_ws(c);
    Cursor in = c;
    bool quoted = false;
    if (at_quote(c)) {
        if (c.inner) { fo->error = "doubly nested skeleton string"; return false; }
        quoted = true;
        in.p = c.p + 1;
        in.inner = true;
    }
    skip_ws(in);
    if (in.p >= in.end || *in.p != '[') { fo->error = "skeleton list expected"; return false; }
    ++in.p;
    int idx = 0;
    *n_here = 0;
    skip_ws(in);
    if (in.p < in.end && *in.p == ']') {
        ++in.p;
    } else {
        while (in.p < in.end) {
            Head h;
            h.cam = cam;
            h.skeleton_index = idx++;
            skip_ws(in);
            const char *sk_begin = in.p;
            if (!parse_skeleton(in, J, &h, &fo->error)) return false;
            if (h.joint_mask) {
                if (extent_base) {
                    fo->extent.push_back((int32_t)(sk_begin - extent_base));
                    fo->extent.push_back((int32_t)(in.p - extent_base));
                }
                fo->heads.push_back({h.cam, h.joint_mask, h.tri_mask, h.skeleton_index});
                for (int j = 0; j < J; ++j) {
                    fo->xy.push_back(h.xy[j][0]);
                    fo->xy.push_back(h.xy[j][1]);
                    fo->vp.push_back(h.vp[j][0]);
                    fo->vp.push_back(h.vp[j][1]);
                }
                ++*n_here;
            }
            skip_ws(in);
            if (in.p >= in.end) { fo->error = "unterminated skeleton list"; return false; }
            if (*in.p == ',') { ++in.p; continue; }
            if (*in.p == ']') { ++in.p; break; }
            fo->error = "bad skeleton list";
            return false;
        }
    }
    if (quoted) {
        skip_ws(in);
        if (in.p >= in.end || *in.p != '"') { fo->error = "skeleton string not closed"; return false; }
        c.p = in.p + 1;
    } else {
        c.p = in.p;
    }
    return true;
}

bool parse_frame(const char *b, const char *e, const std::vector<std::string> &cams, int J, FrameOut *fo) {
    Cursor c{b, e, false};
    skip_ws(c);
    if (c.p >= c.end || *c.p != '{') { fo->error = "frame is not an object"; return false; }
    ++c.p;
    const int V = (int)cams.size();
    fo->heads.reserve(32);
    fo->xy.reserve((size_t)32 * J * 2);
    fo->vp.reserve((size_t)32 * J * 2);
    fo->slot_cam.reserve(V);
    fo->slot_n.reserve(V);
    skip_ws(c);
    if (c.p < c.end && *c.p == '}') return true;
    std::string key;
    while (c.p < c.end) {
        skip_ws(c);
        if (!read_key(c, &key)) { fo->error = "bad camera key"; return false; }
        skip_ws(c);
        if (c.p >= c.end || *c.p != ':') { fo->error = "missing ':' after camera"; return false; }
        ++c.p;
        int cam = -1;
        for (int i = 0; i < V; ++i)
            if (cams[i] == key) { cam = i; break; }
        skip_ws(c);
        if (cam < 0) {
            if (!skip_value(c)) { fo->error = "bad value of unused camera"; return false; }
        } else {
            if (c.p >= c.end || *c.p != '[') { fo->error = "camera entry is not a list"; return false; }
            ++c.p;
            if ((int)fo->slot_cam.size() >= V) { fo->error = "frame lists more cameras than configured"; return false; }
            int n_here = 0;
            if (!parse_skeleton_list(c, J, cam, fo, &n_here)) return false;
            fo->slot_cam.push_back(cam);
            fo->slot_n.push_back(n_here);
            // remaining elements (timestamp, 'no_image', bodies_3D) are not on the hot path
            bool closed = false;
            while (c.p < c.end) {
                skip_ws(c);
                if (c.p >= c.end) break;
                if (*c.p == ',') { ++c.p; if (!skip_value(c)) { fo->error = "bad camera entry"; return false; } continue; }
                if (*c.p == ']') { ++c.p; closed = true; break; }
                fo->error = "bad camera entry";
                return false;
            }
            if (!closed) { fo->error = "unterminated camera entry"; return false; }
        }
        skip_ws(c);
        if (c.p >= c.end) break;
        if (*c.p == ',') { ++c.p; continue; }
        if (*c.p == '}') return true;
        fo->error = "bad frame object";
        return false;
    }
    fo->error = "unterminated frame";
    return false;
}

// First level only (device-side parse, jsonparse.hip): the camera keys of a frame and, for every camera in
// `cams`, the extent of the STRING that holds its skeleton list; everything else is skipped as parse_frame
// does.  `unsupported` = a shape only the host parser handles (a skeleton list that is not a string, more
// camera entries than configured): the caller then packs the window on the host.
struct StrExtent {
    int cam;
    const char *b, *e;
};
bool index_frame(const char *b, const char *e, const std::vector<std::string> &cams, std::vector<StrExtent> *out, std::string *err,
                 bool *unsupported) {
    Cursor c{b, e, false};
    skip_ws(c);
    if (c.p >= c.end || *c.p != '{') { *err = "frame is not an object"; return false; }
    ++c.p;
    const int V = (int)cams.size();
    skip_ws(c);
    if (c.p < c.end && *c.p == '}') return true;
    std::string key;
    while (c.p < c.end) {
        skip_ws(c);
        if (!read_key(c, &key)) { *err = "bad camera key"; return false; }
        skip_ws(c);
        if (c.p >= c.end || *c.p != ':') { *err = "missing ':' after camera"; return false; }
        ++c.p;
        int cam = -1;
        for (int i = 0; i < V; ++i)
            if (cams[i] == key) { cam = i; break; }
        skip_ws(c);
        if (cam < 0) {
            if (!skip_value(c)) { *err = "bad value of unused camera"; return false; }
        } else {
            if (c.p >= c.end || *c.p != '[') { *err = "camera entry is not a list"; return false; }
            ++c.p;
            if ((int)out->size() >= V) { *unsupported = true; return true; }
            skip_ws(c);
            if (c.p >= c.end || *c.p != '"') { *unsupported = true; return true; }
            const char *hit = g_find_quote(c.p + 1, c.end, c.p + 1);
            if (!hit) { *err = "skeleton string not closed"; return false; }
            out->push_back({cam, c.p + 1, hit});
            c.p = hit + 1;
            bool closed = false;
            while (c.p < c.end) {
                skip_ws(c);
                if (c.p >= c.end) break;
                if (*c.p == ',') { ++c.p; if (!skip_value(c)) { *err = "bad camera entry"; return false; } continue; }
                if (*c.p == ']') { ++c.p; closed = true; break; }
                *err = "bad camera entry";
                return false;
            }
            if (!closed) { *err = "unterminated camera entry"; return false; }
        }
        skip_ws(c);
        if (c.p >= c.end) break;
        if (*c.p == ',') { ++c.p; continue; }
        if (*c.p == '}') return true;
        *err = "bad frame object";
        return false;
    }
    *err = "unterminated frame";
    return false;
}

// ---- byte searches of the frame scanner -------------------------------------------------------
// scalar forms (any CPU) and AVX2 forms (32 bytes per step), chosen once at start-up
static const char *find_special_scalar(const char *p, const char *e) {
    static const struct Special {
        bool is[256] = {};
        Special() { is[(unsigned char)'"'] = is[(unsigned char)'{'] = is[(unsigned char)'}'] = is[(unsigned char)'['] = is[(unsigned char)']'] = true; }
    } special;
    while (p < e && !special.is[(unsigned char)*p]) ++p;      // numbers, commas, blanks: nothing to track
    return p;
}

// first quote at or after q that closes the string (even number of backslashes in front of it, counted
// back to `lo`, the first byte of the string body); nullptr if there is none
static const char *find_closing_quote_scalar(const char *q, const char *e, const char *lo) {
    for (;;) {
        const char *hit = static_cast<const char *>(memchr(q, '"', (size_t)(e - q)));
        if (!hit) return nullptr;
        const char *bs = hit;               // count the backslashes in front of it
        while (bs > lo && bs[-1] == '\\') --bs;
        if (((hit - bs) & 1) == 0) return hit;
        q = hit + 1;
    }
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) static const char *find_special_avx2(const char *p, const char *e) {
    const __m256i q = _mm256_set1_epi8('"'), a = _mm256_set1_epi8('{'), b = _mm256_set1_epi8('}'), c = _mm256_set1_epi8('['),
                  d = _mm256_set1_epi8(']');
    while (p + 32 <= e) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
        const __m256i m = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, q), _mm256_cmpeq_epi8(v, a)),
                                          _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, b), _mm256_cmpeq_epi8(v, c)),
                                                          _mm256_cmpeq_epi8(v, d)));
        const unsigned bits = (unsigned)_mm256_movemask_epi8(m);
        if (bits) return p + __builtin_ctz(bits);
        p += 32;
    }
    return find_special_scalar(p, e);
}

// Inside the camera strings nearly every quote is an escaped one (\"): a quote with exactly one
// backslash in front is skipped in the mask, one with none closes the string, and only quotes behind two
// or more backslashes (rare) are settled by counting the run.
__attribute__((target("avx2"))) static const char *find_closing_quote_avx2(const char *q, const char *e, const char *lo) {
    const __m256i vq = _mm256_set1_epi8('"'), vb = _mm256_set1_epi8('\\');
    // backslashes in the two bytes in front of the block
    uint64_t prev2 = 0;
    if (q - lo >= 1 && q[-1] == '\\') prev2 |= 2;
    if (q - lo >= 2 && q[-2] == '\\') prev2 |= 1;
    while (q + 32 <= e) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q));
        const uint64_t mq = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, vq));
        const uint64_t mb = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, vb));
        const uint64_t MB = (mb << 2) | prev2;             // bit i + 2 = backslash at byte i of the block
        if (mq) {
            const uint64_t esc1 = mq & (MB >> 1);          // a backslash right in front
            const uint64_t esc2 = esc1 & MB;               // and one in front of that: count the run
            uint64_t m = (mq & ~esc1) | esc2;
            while (m) {
                const int i = __builtin_ctzll(m);
                if (!((esc2 >> i) & 1)) return q + i;
                const char *hit = q + i, *bs = hit;
                while (bs > lo && bs[-1] == '\\') --bs;
                if (((hit - bs) & 1) == 0) return hit;
                m &= m - 1;
            }
        }
        prev2 = (mb >> 30) & 3;
        q += 32;
    }
    return find_closing_quote_scalar(q, e, lo);
}
#endif

find_special_fn g_find_special = find_special_scalar;
find_quote_fn g_find_quote = find_closing_quote_scalar;
static const bool g_scan_simd = [] {
#if defined(__x86_64__)
    if (!getenv("MPE_PACK_NO_SIMD") && __builtin_cpu_supports("avx2")) {
        g_find_special = find_special_avx2;
        g_find_quote = find_closing_quote_avx2;
        return true;
    }
#endif
    return false;
}();

// Extents of the elements of the top-level array: a RESUMABLE scan (bracket depth with string
// awareness).  Most of a frame's bytes sit inside the per-camera JSON strings (the skeleton lists
// are JSON inside JSON), so inside a string the scanner jumps from one quote / backslash to the next
// with memchr instead of walking bytes.
struct FrameScanner {
    const char *b = nullptr, *e = nullptr, *p = nullptr;
    int depth = 0;
    bool started = false, finished = false, single = false;
    const char *start = nullptr;
    std::vector<std::pair<const char *, const char *>> ext;
    std::string err;
    // speculative parallel scan (mpe_json_index): `stops` = the '{' bytes later scanners started from, each on the
    // guess that it opens a frame.  Reached with depth 0 the guess was right and this scanner stops there (`met`);
    // passed at any other depth it was wrong and this scanner simply goes on (that scanner's frames are dropped).
    std::vector<const char *> stops;        // guessed frame openings of the later parts, ascending
    size_t si = 0;
    const char *met_at = nullptr;
    bool met = false;

    // a scanner that starts in the middle of the document, at a '{' guessed to open a frame
    void begin_at(const char *b_, const char *frame_open, const char *e_) {
        b = b_;
        p = frame_open;
        e = e_;
        started = true;
    }

    bool begin(const char *b_, const char *e_) {
        b = p = b_;
        e = e_;
        while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
        if (p >= e) { err = "empty document"; return false; }
        if (*p == '{') {            // a single frame
            ext.push_back({p, e});
            finished = single = true;
            return true;
        }
        if (*p != '[') { err = "top level must be a list of frames"; return false; }
        ++p;
        started = true;
        return true;
    }

    // scan on until at least `want` frames are known or the list ends; false on malformed input
    bool extend(size_t want) {
        while (!finished && ext.size() < want) {
            p = g_find_special(p, e);
            if (p >= e) { err = "unterminated top-level list"; return false; }
            const char ch = *p;
            if (ch == '"') {
                // skip the string: next quote that is not escaped
                const char *hit = g_find_quote(p + 1, e, p + 1);
                if (!hit) { err = "unterminated string"; return false; }
                p = hit + 1;
                continue;
            }
            if (ch == '{' || ch == '[') {
                if (depth == 0) {
                    while (si < stops.size() && stops[si] < p) ++si;          // guesses passed at another depth were wrong
                    if (si < stops.size() && stops[si] == p) {               // the frame a later part starts with: handed over
                        met = finished = true;
                        met_at = p;
                        break;
                    }
                    start = p;
                }
                ++depth;
            } else if (ch == '}' || ch == ']') {
                if (depth == 0) {                       // end of the top-level list
                    finished = true;
                    break;
                }
                --depth;
                if (depth == 0) ext.push_back({start, p + 1});
            }
            ++p;
        }
        return true;
    }
};

bool split_frames(const char *b, const char *e, std::vector<std::pair<const char *, const char *>> *out, std::string *err) {
    FrameScanner sc;
    if (!sc.begin(b, e) || !sc.extend((size_t)-1)) {
        *err = sc.err;
        return false;
    }
    *out = std::move(sc.ext);
    return true;
}

}  // namespace

struct mpe_packed {
    int32_t V = 0, J = 0, n_frames = 0;
    std::vector<int32_t> frame_head_off, frame_en_off, slot_cam, slot_n, head_cam, skeleton_index;
    std::vector<uint32_t> joint_mask, tri_mask;
    std::vector<double> xy;
    std::vector<float> vp;
    std::string err;
};

// Frame index of one document, filled by a background thread: the serial scan for the frame extents
// (4-5 GB/s) runs ahead of the windows that are being parsed, so packing a window costs its parse time
// only.  Extents are published in batches; readers wait for the frame they need.
struct mpe_json_index {
    FrameScanner sc;                      // owned by the scan thread once it runs
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::pair<const char *, const char *>> pub;   // published extents (guarded by mu)
    bool done = false, failed = false, stop = false;
    std::string err;
    // Speculative parallel scan.  The serial scan (6-8 GB/s) was the ceiling of the JSON path once the second
    // level moved to the device.  The document is cut into parts of ~16 MB at guessed frame boundaries (the byte
    // pattern that closes a camera entry and a frame and opens the next: `]]}, {"`, else `}, {"`); helper threads take
    // the parts IN ORDER and scan each from its guess as if it were at the top level.  The scanner of the part before
    // VERIFIES the guess when it gets there: arriving at that '{' between frames (depth 0) it stops and the part's
    // frames follow; arriving at any other depth the guess was wrong (e.g. inside the ground-truth lists), the part's
    // frames are dropped and the scanner goes on through its range.  The published sequence is therefore exactly the
    // serial one; a bad guess only costs speed.  Parts are small and taken in order so that the FRONT of the document
    // is known at P times the serial rate too: a streaming reader (Engine.stream_json) consumes windows from the
    // front, and with P contiguous quarters its first quarter arrived at the serial rate (measured: the first 11 of 48
    // windows 6-9 ms apart, the rest 5.65 ms).
    struct Part {
        FrameScanner sc;
        const char *open = nullptr;
        bool ok = true;
        bool fin = false;                     // guarded by pmu
    };
    std::vector<std::unique_ptr<Part>> parts;
    std::vector<std::thread> helpers;
    std::atomic<size_t> next_part{0};
    std::atomic<bool> abort_parts{false};
    std::mutex pmu;
    std::condition_variable pcv;

    bool scanning() {
        std::lock_guard<std::mutex> lk(mu);
        return !done && !failed;
    }

    void plan_parts(const char *b, const char *e, int P) {
        const size_t len = (size_t)(e - b);
        if (P < 2 || len < ((size_t)4 << 20) || sc.single) return;
        size_t chunk = (size_t)16 << 20;
        if (const char *ev = getenv("MPE_SCAN_CHUNK_KB")) {
            const long v = atol(ev);
            if (v >= 64) chunk = (size_t)v << 10;
        }
        size_t n = len / chunk;
        if (n < (size_t)P) n = (size_t)P;
        chunk = len / n;
        std::vector<const char *> opens;
        for (size_t k = 1; k < n; ++k) {
            const char *from = b + chunk * k;
            const size_t reach = chunk / 2;
            const char *open = nullptr;
            const char *hit = static_cast<const char *>(memmem(from, reach + 7 < (size_t)(e - from) ? reach + 7 : (size_t)(e - from), "]]}, {\"", 7));
            if (hit) open = hit + 5;
            else {
                hit = static_cast<const char *>(memmem(from, reach + 5 < (size_t)(e - from) ? reach + 5 : (size_t)(e - from), "}, {\"", 5));
                if (hit) open = hit + 3;
            }
            if (open && (opens.empty() || open > opens.back())) opens.push_back(open);
        }
        sc.stops = opens;
        for (size_t k = 0; k < opens.size(); ++k) {
            parts.emplace_back(new Part());
            Part *pt = parts.back().get();
            pt->open = opens[k];
            pt->sc.begin_at(b, opens[k], e);
            pt->sc.stops.assign(opens.begin() + (long)k + 1, opens.end());
        }
        const int n_helpers = (int)parts.size() < P - 1 ? (int)parts.size() : P - 1;
        for (int h = 0; h < n_helpers; ++h) {
            helpers.emplace_back([this] {
                for (;;) {
                    const size_t k = next_part.fetch_add(1);
                    if (k >= parts.size() || abort_parts.load()) return;
                    Part *pt = parts[k].get();
                    while (!pt->sc.finished && !abort_parts.load()) {
                        if (!pt->sc.extend(pt->sc.ext.size() + 256)) {       // malformed FROM THIS GUESS: decided by the verifier
                            pt->ok = false;
                            break;
                        }
                    }
                    {
                        std::lock_guard<std::mutex> lk(pmu);
                        pt->fin = true;
                    }
                    pcv.notify_all();
                }
            });
        }
    }

    void publish(FrameScanner &from, size_t *sent) {
        std::lock_guard<std::mutex> lk(mu);
        for (; *sent < from.ext.size(); ++*sent) pub.push_back(from.ext[*sent]);
    }

    void run() {
        FrameScanner *cur = &sc;
        size_t sent = 0;
        for (;;) {
            const bool ok = cur->extend(sent + 64);
            publish(*cur, &sent);
            bool all = !ok;
            std::string e2 = ok ? std::string() : cur->err;
            if (ok && cur->finished) {
                all = true;
                if (cur->met) {
                    // reached a later part's guess between two frames: the guess was right, that part's frames follow
                    for (auto &up : parts)
                        if (up->open == cur->met_at) {
                            {
                                std::unique_lock<std::mutex> lk(pmu);
                                pcv.wait(lk, [&] { return up->fin || abort_parts.load(); });
                            }
                            if (abort_parts.load()) return;
                            if (!up->ok) e2 = up->sc.err;              // malformed from a verified boundary on
                            else {
                                cur = &up->sc;
                                sent = 0;
                                all = false;
                            }
                            break;
                        }
                }
            }
            bool quit;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e2.empty()) {
                    failed = true;
                    err = e2;
                }
                if (all) done = true;
                quit = done || stop;
            }
            cv.notify_all();
            if (quit) return;
        }
    }

    // extent of frame idx: 1 = found, 0 = the document has fewer frames, -1 = malformed before it
    int get(size_t idx, std::pair<const char *, const char *> *out) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return pub.size() > idx || done; });
        if (idx < pub.size()) {
            *out = pub[idx];
            return 1;
        }
        return failed ? -1 : 0;
    }

    ~mpe_json_index() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        {
            std::lock_guard<std::mutex> lk(pmu);
            abort_parts = true;
        }
        pcv.notify_all();
        if (th.joinable()) th.join();
        for (auto &h : helpers)
            if (h.joinable()) h.join();
    }
};

static thread_local std::string g_pack_error;

// Worker threads when the caller does not say: the CPUs this process may actually use -- the smaller of
// the hardware threads, the affinity mask and the cgroup CPU quota (a container on a 256-thread host is
// often limited to a few cores; 256 runnable parsers then only starve the scan thread).
static int default_threads();
// threads of the document scan: MPE_SCAN_THREADS, else a quarter of the CPUs this process may use (the rest parse / stage)
static int scan_threads() {
    if (const char *ev = getenv("MPE_SCAN_THREADS")) return atoi(ev) > 0 ? atoi(ev) : 1;
    const int hw = default_threads();
    return hw >= 16 ? 4 : hw >= 6 ? 2 : 1;
}

static int default_threads() {
    static const int n = [] {
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
#ifdef __linux__
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int a = CPU_COUNT(&set);
            if (a > 0 && a < hw) hw = a;
        }
        double quota = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                  // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            long period = 0;
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota = atof(q) / (double)period;
            fclose(f);
        } else {
            long q = -1, per = 0;                                                // cgroup v1
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%ld", &q) != 1) q = -1;
                fclose(g);
            }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(g, "%ld", &per) != 1) per = 0;
                fclose(g);
            }
            if (q > 0 && per > 0) quota = (double)q / (double)per;
        }
        if (quota >= 1.0 && quota < hw) hw = (int)quota;
#endif
        if (const char *e = getenv("MPE_PACK_THREADS")) {
            const int v = atoi(e);
            if (v > 0) hw = v;
        }
        return hw;
    }();
    return n;
}


extern "C" {

const char *mpe_pack_last_error(void) { return g_pack_error.c_str(); }

// parse the selected frames (thread pool); on failure g_pack_error is set
static int parse_selected(const char *json, size_t len, const char *const *camera_names, int32_t n_cameras, int32_t n_joints,
                          int32_t frame_start, int32_t frame_step, int32_t max_frames, int32_t n_threads,
                          std::vector<FrameOut> *fo_out, mpe_json_index *index = nullptr) {
    if ((!json && !index) || !camera_names || n_cameras < 1 || n_cameras > MPE_MAX_CAMERAS || n_joints < 1 || n_joints > MPE_MAX_JOINTS ||
        frame_step < 1 || frame_start < 0) {
        g_pack_error = "mpe_pack_json: bad argument";
        return MPE_ERR_INVALID;
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::string> cams(camera_names, camera_names + n_cameras);
    std::vector<std::pair<const char *, const char *>> sel;
    std::vector<FrameOut> &fo = *fo_out;
    std::atomic<int> next{0};
    std::atomic<bool> failed{false}, malformed{false};
    int B = 0, nt = n_threads > 0 ? n_threads : default_threads();
    auto t1 = t0;
    if (index && max_frames > 0) {
        // frames are parsed as the background scan publishes them; the window ends where the document does
        std::atomic<int> limit{max_frames};
        fo.assign((size_t)max_frames, FrameOut());
        auto work = [&]() {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= limit.load()) return;
                std::pair<const char *, const char *> ext;
                const int r = index->get((size_t)frame_start + (size_t)i * frame_step, &ext);
                if (r <= 0) {
                    if (r < 0) malformed = true;
                    int cur = limit.load();
                    while (i < cur && !limit.compare_exchange_weak(cur, i)) {}
                    return;
                }
                if (!parse_frame(ext.first, ext.second, cams, n_joints, &fo[i])) failed = true;
            }
        };
        if (nt < 1) nt = 1;
        if (nt > max_frames) nt = max_frames;
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (malformed) {
            std::lock_guard<std::mutex> lk(index->mu);
            g_pack_error = index->err;
            return MPE_ERR_INVALID;
        }
        B = limit.load();
        fo.resize((size_t)B);
    } else {
        std::vector<std::pair<const char *, const char *>> own;
        if (index) {
            // every frame of the document: wait for the scan to finish
            std::pair<const char *, const char *> ext;
            if (index->get((size_t)-2, &ext) < 0) {
                std::lock_guard<std::mutex> lk(index->mu);
                g_pack_error = index->err;
                return MPE_ERR_INVALID;
            }
            std::lock_guard<std::mutex> lk(index->mu);
            own = index->pub;
        } else {
            std::string err;
            if (!split_frames(json, json + len, &own, &err)) {
                g_pack_error = err;
                return MPE_ERR_INVALID;
            }
        }
        for (size_t i = (size_t)frame_start; i < own.size(); i += (size_t)frame_step) {
            if (max_frames > 0 && (int32_t)sel.size() >= max_frames) break;
            sel.push_back(own[i]);
        }
        t1 = std::chrono::steady_clock::now();
        B = (int)sel.size();
        fo.assign((size_t)B, FrameOut());
        auto work = [&]() {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= B) return;
                if (!parse_frame(sel[i].first, sel[i].second, cams, n_joints, &fo[i])) failed = true;
            }
        };
        if (nt < 1) nt = 1;
        if (nt > B) nt = B > 0 ? B : 1;
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    if (failed) {
        for (int i = 0; i < B; ++i)
            if (!fo[i].error.empty()) {
                g_pack_error = "frame " + std::to_string(frame_start + i * frame_step) + ": " + fo[i].error;
                break;
            }
        return MPE_ERR_INVALID;
    }
    if (getenv("MPE_PACK_TIMING")) {
        const auto t2 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "mpe_pack_json: split %.1f ms, parse %.1f ms (%d threads)\n", ms(t0, t1), ms(t1, t2), nt);
    }
    return MPE_OK;
}

// per-frame results -> the structure-of-arrays batch in caller-visible arrays
static void assemble(const std::vector<FrameOut> &fo, int n_cameras, int n_joints, int32_t *frame_head_off, int32_t *frame_en_off,
                     int32_t *slot_cam, int32_t *slot_n, int32_t *head_cam, int32_t *skeleton_index, uint32_t *joint_mask,
                     uint32_t *tri_mask, double *xy, float *vp) {
    const int B = (int)fo.size();
    frame_head_off[0] = frame_en_off[0] = 0;
    for (int f = 0; f < B; ++f) {
        long tot = 0, sq = 0;
        for (int s = 0; s < n_cameras; ++s) {
            slot_cam[(size_t)f * n_cameras + s] = -1;
            slot_n[(size_t)f * n_cameras + s] = 0;
        }
        for (size_t s = 0; s < fo[f].slot_cam.size(); ++s) {
            slot_cam[(size_t)f * n_cameras + s] = fo[f].slot_cam[s];
            slot_n[(size_t)f * n_cameras + s] = fo[f].slot_n[s];
            tot += fo[f].slot_n[s];
            sq += (long)fo[f].slot_n[s] * fo[f].slot_n[s];
        }
        frame_head_off[f + 1] = frame_head_off[f] + (int32_t)tot;
        frame_en_off[f + 1] = frame_en_off[f] + (int32_t)((tot * tot - sq) / 2);
    }
    const size_t per = (size_t)n_joints * 2;
    for (int f = 0; f < B; ++f) {
        size_t h = (size_t)frame_head_off[f];
        if (!fo[f].heads.empty()) {
            memcpy(&xy[h * per], fo[f].xy.data(), fo[f].xy.size() * sizeof(double));
            memcpy(&vp[h * per], fo[f].vp.data(), fo[f].vp.size() * sizeof(float));
        }
        for (const HeadInfo &hd : fo[f].heads) {
            head_cam[h] = hd.cam;
            skeleton_index[h] = hd.skeleton_index;
            joint_mask[h] = hd.joint_mask;
            tri_mask[h] = hd.tri_mask;
            ++h;
        }
    }
}

int mpe_pack_json(const char *json, size_t len, const char *const *camera_names, int32_t n_cameras, int32_t n_joints,
                  int32_t frame_start, int32_t frame_step, int32_t max_frames, int32_t n_threads, mpe_packed **out) {
    if (!out) {
        g_pack_error = "mpe_pack_json: bad argument";
        return MPE_ERR_INVALID;
    }
    *out = nullptr;
    std::vector<FrameOut> fo;
    const int rc = parse_selected(json, len, camera_names, n_cameras, n_joints, frame_start, frame_step, max_frames, n_threads, &fo);
    if (rc) return rc;
    mpe_packed *pk = new (std::nothrow) mpe_packed();
    if (!pk) return MPE_ERR_NOMEM;
    const int B = (int)fo.size();
    pk->V = n_cameras;
    pk->J = n_joints;
    pk->n_frames = B;
    size_t H = 0;
    for (const FrameOut &f : fo) H += f.heads.size();
    pk->frame_head_off.assign((size_t)B + 1, 0);
    pk->frame_en_off.assign((size_t)B + 1, 0);
    pk->slot_cam.assign((size_t)B * n_cameras, -1);
    pk->slot_n.assign((size_t)B * n_cameras, 0);
    pk->head_cam.resize(H);
    pk->skeleton_index.resize(H);
    pk->joint_mask.resize(H);
    pk->tri_mask.resize(H);
    pk->xy.assign(H * n_joints * 2, 0.0);
    pk->vp.assign(H * n_joints * 2, 0.f);
    assemble(fo, n_cameras, n_joints, pk->frame_head_off.data(), pk->frame_en_off.data(), pk->slot_cam.data(), pk->slot_n.data(),
             pk->head_cam.data(), pk->skeleton_index.data(), pk->joint_mask.data(), pk->tri_mask.data(), pk->xy.data(),
             pk->vp.data());
    *out = pk;
    return MPE_OK;
}

int mpe_pack_json_into(const char *json, size_t len, const char *const *camera_names, int32_t n_cameras, int32_t n_joints,
                       int32_t frame_start, int32_t frame_step, int32_t max_frames, int32_t n_threads,
                       const mpe_pack_dst *dst, int32_t *n_frames, int32_t *n_heads, int32_t *n_edge_nodes) {
    if (!dst || !n_frames || !n_heads || !n_edge_nodes || !dst->frame_head_off || !dst->frame_en_off || !dst->slot_cam ||
        !dst->slot_n || !dst->head_cam || !dst->skeleton_index || !dst->joint_mask || !dst->tri_mask || !dst->xy || !dst->vp ||
        dst->max_frames < 1 || dst->max_heads < 1) {
        g_pack_error = "mpe_pack_json_into: bad argument";
        return MPE_ERR_INVALID;
    }
    if (max_frames <= 0 || max_frames > dst->max_frames) max_frames = dst->max_frames;
    std::vector<FrameOut> fo;
    const int rc = parse_selected(json, len, camera_names, n_cameras, n_joints, frame_start, frame_step, max_frames, n_threads, &fo);
    if (rc) return rc;
    size_t H = 0;
    for (const FrameOut &f : fo) H += f.heads.size();
    if (H > (size_t)dst->max_heads) {
        g_pack_error = "mpe_pack_json_into: " + std::to_string(H) + " skeletons exceed the destination's max_heads = " +
                       std::to_string(dst->max_heads);
        return MPE_ERR_CAPACITY;
    }
    assemble(fo, n_cameras, n_joints, dst->frame_head_off, dst->frame_en_off, dst->slot_cam, dst->slot_n, dst->head_cam,
             dst->skeleton_index, dst->joint_mask, dst->tri_mask, dst->xy, dst->vp);
    *n_frames = (int32_t)fo.size();
    *n_heads = (int32_t)H;
    *n_edge_nodes = dst->frame_en_off[fo.size()];
    return MPE_OK;
}

// One frame as the reference's per-frame callers hold it (test/metrics_from_model.py:182-199: a dict camera -> [text of the skeleton
// list, timestamp]): the camera texts themselves, no outer document -- the one-frame-per-call mirrors no longer serialise the frame
// again just to have it parsed.  Same grammar, same head order and same numbers as parse_frame gives for the same texts.
int mpe_pack_views_into(const char *const *texts, const size_t *lens, const int32_t *cams, int32_t n_views, int32_t n_cameras,
                        int32_t n_joints, const mpe_pack_dst *dst, int32_t *n_heads, int32_t *n_edge_nodes, int32_t *extents) {
    if (!dst || !n_heads || !n_edge_nodes || !dst->frame_head_off || !dst->frame_en_off || !dst->slot_cam || !dst->slot_n ||
        !dst->head_cam || !dst->skeleton_index || !dst->joint_mask || !dst->tri_mask || !dst->xy || !dst->vp || dst->max_frames < 1 ||
        dst->max_heads < 1 || n_views < 0 || (n_views > 0 && (!texts || !lens || !cams)) || n_cameras < 1 || n_cameras > MPE_MAX_CAMERAS ||
        n_views > n_cameras || n_joints < 1 || n_joints > MPE_MAX_JOINTS) {
        g_pack_error = "mpe_pack_views_into: bad argument";
        return MPE_ERR_INVALID;
    }
    std::vector<FrameOut> fo(1);
    FrameOut &f = fo[0];
    f.heads.reserve(32);
    f.xy.reserve((size_t)32 * n_joints * 2);
    f.vp.reserve((size_t)32 * n_joints * 2);
    for (int i = 0; i < n_views; ++i) {
        if (cams[i] < 0 || cams[i] >= n_cameras || !texts[i]) {
            g_pack_error = "mpe_pack_views_into: bad camera index";
            return MPE_ERR_INVALID;
        }
        Cursor c{texts[i], texts[i] + lens[i], false};
        skip_ws(c);
        int n_here = 0;
        if (c.p < c.end && *c.p == '"') f.error = "skeleton list expected";       // (a string inside the string: not what json.loads would turn into a list)
        if (!f.error.empty() || !parse_skeleton_list(c, n_joints, cams[i], &f, &n_here, extents ? texts[i] : nullptr)) {
            g_pack_error = "view " + std::to_string(i) + ": " + f.error;
            return MPE_ERR_INVALID;
        }
        skip_ws(c);
        if (c.p != c.end) {
            g_pack_error = "view " + std::to_string(i) + ": text behind the skeleton list";
            return MPE_ERR_INVALID;
        }
        f.slot_cam.push_back(cams[i]);
        f.slot_n.push_back(n_here);
    }
    const size_t H = f.heads.size();
    if (H > (size_t)dst->max_heads) {
        g_pack_error = "mpe_pack_views_into: " + std::to_string(H) + " skeletons exceed the destination's max_heads = " + std::to_string(dst->max_heads);
        return MPE_ERR_CAPACITY;
    }
    assemble(fo, n_cameras, n_joints, dst->frame_head_off, dst->frame_en_off, dst->slot_cam, dst->slot_n, dst->head_cam, dst->skeleton_index,
             dst->joint_mask, dst->tri_mask, dst->xy, dst->vp);
    if (extents && !f.extent.empty()) memcpy(extents, f.extent.data(), f.extent.size() * sizeof(int32_t));
    *n_heads = (int32_t)H;
    *n_edge_nodes = dst->frame_en_off[1];
    return MPE_OK;
}

int mpe_json_index_create(const char *json, size_t len, mpe_json_index **out) {
    if (!json || !out) {
        g_pack_error = "mpe_json_index_create: bad argument";
        return MPE_ERR_INVALID;
    }
    *out = nullptr;
    mpe_json_index *ix = new (std::nothrow) mpe_json_index();
    if (!ix) return MPE_ERR_NOMEM;
    if (!ix->sc.begin(json, json + len)) {
        g_pack_error = ix->sc.err;
        delete ix;
        return MPE_ERR_INVALID;
    }
    {
        // scan threads: MPE_SCAN_THREADS, else a quarter of the CPUs this process may use (the rest parse / stage)
        ix->plan_parts(json, json + len, scan_threads());
    }
    ix->th = std::thread([ix] { ix->run(); });
    *out = ix;
    return MPE_OK;
}

void mpe_json_index_free(mpe_json_index *ix) { delete ix; }

int mpe_pack_indexed_into(mpe_json_index *ix, const char *const *camera_names, int32_t n_cameras, int32_t n_joints,
                          int32_t frame_start, int32_t frame_step, int32_t max_frames, int32_t n_threads,
                          const mpe_pack_dst *dst, int32_t *n_frames, int32_t *n_heads, int32_t *n_edge_nodes) {
    if (!ix || !dst || !n_frames || !n_heads || !n_edge_nodes || !dst->frame_head_off || !dst->frame_en_off || !dst->slot_cam ||
        !dst->slot_n || !dst->head_cam || !dst->skeleton_index || !dst->joint_mask || !dst->tri_mask || !dst->xy || !dst->vp ||
        dst->max_frames < 1 || dst->max_heads < 1) {
        g_pack_error = "mpe_pack_indexed_into: bad argument";
        return MPE_ERR_INVALID;
    }
    if (max_frames <= 0 || max_frames > dst->max_frames) max_frames = dst->max_frames;
    std::vector<FrameOut> fo;
    const int rc = parse_selected(nullptr, 0, camera_names, n_cameras, n_joints, frame_start, frame_step, max_frames, n_threads, &fo,
                                  ix);
    if (rc) return rc;
    size_t H = 0;
    for (const FrameOut &f : fo) H += f.heads.size();
    if (H > (size_t)dst->max_heads) {
        g_pack_error = "mpe_pack_indexed_into: " + std::to_string(H) + " skeletons exceed the destination's max_heads = " +
                       std::to_string(dst->max_heads);
        return MPE_ERR_CAPACITY;
    }
    assemble(fo, n_cameras, n_joints, dst->frame_head_off, dst->frame_en_off, dst->slot_cam, dst->slot_n, dst->head_cam,
             dst->skeleton_index, dst->joint_mask, dst->tri_mask, dst->xy, dst->vp);
    *n_frames = (int32_t)fo.size();
    *n_heads = (int32_t)H;
    *n_edge_nodes = dst->frame_en_off[fo.size()];
    return MPE_OK;
}

int mpe_json_stage_window(mpe_json_index *ix, const char *const *camera_names, int32_t n_cameras, int32_t frame_start,
                          int32_t frame_step, int32_t max_frames, int32_t n_threads, char *text_dst, size_t text_cap,
                          mpe_json_entry *entries, int32_t entry_cap, int32_t *frame_entry_off, int32_t *n_frames,
                          int32_t *n_entries, size_t *text_bytes) {
    if (!ix || !camera_names || n_cameras < 1 || n_cameras > MPE_MAX_CAMERAS || frame_start < 0 || frame_step < 1 || max_frames < 1 ||
        !text_dst || !entries || !frame_entry_off || !n_frames || !n_entries || !text_bytes) {
        g_pack_error = "mpe_json_stage_window: bad argument";
        return MPE_ERR_INVALID;
    }
    std::vector<std::string> cams(camera_names, camera_names + n_cameras);
    const auto tt0 = std::chrono::steady_clock::now();
    // the window's frame extents (the background scan publishes them in order)
    std::vector<std::pair<const char *, const char *>> sel;
    for (int i = 0; i < max_frames; ++i) {
        std::pair<const char *, const char *> ext;
        const int r = ix->get((size_t)frame_start + (size_t)i * frame_step, &ext);
        if (r < 0) {
            std::lock_guard<std::mutex> lk(ix->mu);
            g_pack_error = ix->err;
            return MPE_ERR_INVALID;
        }
        if (r == 0) break;
        sel.push_back(ext);
    }
    const int B = (int)sel.size();
    const auto tt1 = std::chrono::steady_clock::now();
    std::vector<std::vector<StrExtent>> per((size_t)B);
    std::vector<std::string> errs((size_t)B);
    std::atomic<int> next{0};
    std::atomic<bool> failed{false}, unsupported{false};
    // staging runs beside the document scan and the caller's own thread: under a cgroup CPU quota the sum must stay below it,
    // or the kernel parks the whole process for the rest of the 100 ms period (measured on a 16-CPU share: 16 staging + 4 scan
    // threads -> nr_throttled 10 in 24 windows, windows of 50 ms; json_inclusive 115k-146k frames/s from run to run)
    int nt = n_threads > 0 ? n_threads : default_threads() - (ix->scanning() ? scan_threads() : 0) - 4;   // the caller's thread, its staging worker, the HIP runtime's
    if (nt < 1) nt = 1;
    if (nt > B) nt = B > 0 ? B : 1;
    auto run = [&](auto &&body) {
        next = 0;
        auto work = [&]() {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= B) return;
                body(i);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    };
    run([&](int i) {
        bool un = false;
        per[(size_t)i].reserve((size_t)n_cameras);
        if (!index_frame(sel[(size_t)i].first, sel[(size_t)i].second, cams, &per[(size_t)i], &errs[(size_t)i], &un)) failed = true;
        if (un) unsupported = true;
    });
    if (failed) {
        for (int i = 0; i < B; ++i)
            if (!errs[(size_t)i].empty()) {
                g_pack_error = "frame " + std::to_string(frame_start + i * frame_step) + ": " + errs[(size_t)i];
                break;
            }
        return MPE_ERR_INVALID;
    }
    if (unsupported) {
        g_pack_error = "mpe_json_stage_window: a frame needs the host parser (skeleton list not a string, or more camera entries than cameras)";
        return MPE_ERR_UNSUPPORTED;
    }
    const auto tt2 = std::chrono::steady_clock::now();
    // text offsets (16-byte aligned strings), entry table
    size_t off = 0;
    int ne = 0;
    std::vector<size_t> first_off((size_t)B);
    for (int i = 0; i < B; ++i) {
        frame_entry_off[i] = ne;
        first_off[(size_t)i] = off;
        for (const StrExtent &x : per[(size_t)i]) {
            const size_t len = (size_t)(x.e - x.b);
            if (ne >= entry_cap || off + len > text_cap || off + len > 0xFFFFFFF0u) {
                g_pack_error = "mpe_json_stage_window: staging buffers too small";
                return MPE_ERR_CAPACITY;
            }
            entries[ne].frame = i;
            entries[ne].cam = x.cam;
            entries[ne].begin = (uint32_t)off;
            entries[ne].end = (uint32_t)(off + len);
            ++ne;
            off = (off + len + 15) & ~(size_t)15;
        }
    }
    frame_entry_off[B] = ne;
    run([&](int i) {
        int k = frame_entry_off[i];
        for (const StrExtent &x : per[(size_t)i]) {
            memcpy(text_dst + entries[k].begin, x.b, (size_t)(x.e - x.b));
            ++k;
        }
    });
    if (getenv("MPE_STAGE_TIMING")) {
        const auto tt3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "stage window: %d threads | frame extents %.2f ms | first level %.2f ms | layout + copy %.2f ms\n", nt, ms(tt0, tt1),
                ms(tt1, tt2), ms(tt2, tt3));
    }
    *n_frames = B;
    *n_entries = ne;
    *text_bytes = off;
    return MPE_OK;
}

int mpe_packed_view(const mpe_packed *pk, mpe_packed_arrays *v) {
    if (!pk || !v) return MPE_ERR_INVALID;
    v->n_frames = pk->n_frames;
    v->n_heads = pk->frame_head_off.back();
    v->n_edge_nodes = pk->frame_en_off.back();
    v->n_cameras = pk->V;
    v->n_joints = pk->J;
    v->frame_head_off = pk->frame_head_off.data();
    v->frame_en_off = pk->frame_en_off.data();
    v->slot_cam = pk->slot_cam.data();
    v->slot_n = pk->slot_n.data();
    v->head_cam = pk->head_cam.data();
    v->skeleton_index = pk->skeleton_index.data();
    v->joint_mask = pk->joint_mask.data();
    v->tri_mask = pk->tri_mask.data();
    v->xy = pk->xy.data();
    v->vp = pk->vp.data();
    return MPE_OK;
}

void mpe_packed_free(mpe_packed *pk) { delete pk; }

}  // extern "C"
