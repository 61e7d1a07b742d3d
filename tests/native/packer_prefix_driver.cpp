// Test driver (CPU, AddressSanitizer build): feeds mpe_pack_json every prefix of a valid
// frame-JSON document from an exactly sized, NOT NUL-terminated heap buffer.  Any read past
// `len` is an ASan heap-buffer-overflow.  Prints "<full-document frames> <prefixes accepted>".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/mpe.h"

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *fh = fopen(argv[1], "rb");
    if (!fh) return 2;
    std::string doc;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fh)) > 0) doc.append(buf, n);
    fclose(fh);
    const int J = atoi(argv[2]);
    const int V = argc - 3;
    const char *const *cams = argv + 3;
    int accepted = 0, full_frames = -1;
    for (size_t len = 0; len <= doc.size(); ++len) {
        char *p = static_cast<char *>(malloc(len ? len : 1));
        memcpy(p, doc.data(), len);
        mpe_packed *pk = nullptr;
        const int rc = mpe_pack_json(p, len, cams, V, J, 0, 1, 0, 1, &pk);
        int n_direct = -1;
        if (rc == MPE_OK) {
            ++accepted;
            mpe_packed_arrays v;
            mpe_packed_view(pk, &v);
            n_direct = v.n_frames;
            if (len == doc.size()) full_frames = v.n_frames;
            mpe_packed_free(pk);
        }
        // the resumable index + caller-array form on the same bytes, in windows of 1 frame
        mpe_json_index *ix = nullptr;
        if (mpe_json_index_create(p, len, &ix) == MPE_OK) {
            static int32_t fho[3], feo[3], sc[2 * MPE_MAX_CAMERAS], sn[2 * MPE_MAX_CAMERAS], hc[256], si[256];
            static uint32_t jm[256], tm[256];
            static double xy[256 * MPE_MAX_JOINTS * 2];
            static float vp[256 * MPE_MAX_JOINTS * 2];
            mpe_pack_dst dst = {2, 256, fho, feo, sc, sn, hc, si, jm, tm, xy, vp};
            int n_windowed = 0, ok = 1;
            for (int start = 0; ok; ++start) {
                int32_t nf = 0, nh = 0, ne = 0;
                if (mpe_pack_indexed_into(ix, cams, V, J, start, 1, 1, 1, &dst, &nf, &nh, &ne) != MPE_OK) { ok = 0; n_windowed = -1; break; }
                if (nf == 0) break;
                n_windowed += nf;
            }
            if (rc == MPE_OK && n_windowed != n_direct) { fprintf(stderr, "index/direct mismatch at len %zu: %d vs %d\n", len, n_windowed, n_direct); return 3; }
            mpe_json_index_free(ix);
        }
        free(p);
    }
    printf("%d %d\n", full_frames, accepted);
    return 0;
}
