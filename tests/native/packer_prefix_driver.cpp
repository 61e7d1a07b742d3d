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
        if (rc == MPE_OK) {
            ++accepted;
            mpe_packed_arrays v;
            mpe_packed_view(pk, &v);
            if (len == doc.size()) full_frames = v.n_frames;
            mpe_packed_free(pk);
        }
        free(p);
    }
    printf("%d %d\n", full_frames, accepted);
    return 0;
}
