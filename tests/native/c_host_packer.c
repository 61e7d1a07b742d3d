/* include/mpe.h from plain C (C99, gcc): the header a cgo / JNI / N-API binding would include, and the host-side packer called through
 * it (no GPU involved): a two-frame wire-format document (panoptic_conversor/get_joints_from_panoptic_model_multi.py:231-236,281,287)
 * -> the arrays of an mpe_batch in the reference's head order (graph_generator.py:573-605).  Prints the counts and a checksum that
 * tests/test_host_logic.py compares with the Python binding's on the same text.
 *   gcc -std=c99 -Wall -Wextra -pedantic -I include tests/native/c_host_packer.c -L 3d_multi_pose_estimator_amd -lmpe_hip -o c_host_packer */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpe.h"

int main(int argc, char **argv) {
    const char *cams[MPE_MAX_CAMERAS];
    char *json = NULL;
    long len = 0;
    int n_cams = 0, n_joints, i;
    FILE *f;
    mpe_packed *pk = NULL;
    mpe_packed_arrays a;
    double sum = 0.0;
    unsigned long masks = 0ul;
    if (argc < 4) {
        fprintf(stderr, "usage: %s document.json n_joints camera...\n", argv[0]);
        return 1;
    }
    n_joints = atoi(argv[2]);
    for (i = 3; i < argc && n_cams < MPE_MAX_CAMERAS; ++i) cams[n_cams++] = argv[i];
    f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    len = ftell(f);
    fseek(f, 0, SEEK_SET);
    json = (char *)malloc((size_t)len + 1);
    if (!json || fread(json, 1, (size_t)len, f) != (size_t)len) return 1;
    fclose(f);
    if (mpe_pack_json(json, (size_t)len, cams, n_cams, n_joints, 0, 1, 0, 1, &pk) != MPE_OK) {
        fprintf(stderr, "mpe_pack_json: %s\n", mpe_pack_last_error());
        return 2;
    }
    if (mpe_packed_view(pk, &a) != MPE_OK) return 2;
    for (i = 0; i < a.n_heads * a.n_joints * 2; ++i) sum += a.xy[i] + (double)a.vp[i];
    for (i = 0; i < a.n_heads; ++i) masks += (unsigned long)a.joint_mask[i] % 1000003ul + (unsigned long)a.head_cam[i];
    printf("%s frames %d heads %d edge_nodes %d sum %.17g masks %lu last_off %d\n", mpe_version(), a.n_frames, a.n_heads, a.n_edge_nodes, sum,
           masks, a.frame_head_off[a.n_frames]);
    mpe_packed_free(pk);
    free(json);
    return 0;
}
