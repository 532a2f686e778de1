/*
 * asan_driver.c -- TEST INFRASTRUCTURE: runs the CPU restatement (stereo_oracle.c) on one pair read from a file, built
 * with -fsanitize=address,undefined (make -C oracle asan).  The reference has latent out-of-row reads
 * (CStereoMatching.cpp:628 refine right-window reads, :492 q[boundary_L+1]) which the restatement emulates on a flat
 * buffer (stereo_oracle.c); a silent out-of-bounds access of the emulation itself would hide here, so the whole pair is
 * run under the sanitizers and its results are compared with the ordinary -O3 build (tests/test_oracle_sanitizers.py).
 *
 * in : int32 W, H, levels, radius, offset, origin_width; double ws, Q[16], R[9], T[3]; image0, image1 (W*H*3 each),
 *      mask0, mask1 (W*H each)
 * out: int32 status; int32 margin[2][6]; int64 n_points, v_top; double disparity0[W*H], disparity1[W*H], xyz[3*n]; u8 bgr[3*n]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "stereo_oracle.h"

static void rd(void *p, size_t n, FILE *f) {
    if (fread(p, 1, n, f) != n) {
        fprintf(stderr, "asan_driver: short read\n");
        exit(3);
    }
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[6];
    rd(hdr, sizeof hdr, f);
    orc_pair_in in;
    memset(&in, 0, sizeof in);
    in.width = hdr[0], in.height = hdr[1], in.pyr_levels = hdr[2], in.radius = hdr[3], in.offset = hdr[4], in.origin_width = hdr[5];
    rd(&in.ws, 8, f);
    rd(in.Q, sizeof in.Q, f);
    rd(in.R_final, sizeof in.R_final, f);
    rd(in.T_final, sizeof in.T_final, f);
    const size_t px = (size_t)in.width * in.height;
    /* exact-size heap blocks: any read or write one byte outside an image, mask or map is reported */
    uint8_t *img[2], *msk[2];
    for (int v = 0; v < 2; v++) {
        img[v] = malloc(px * 3);
        rd(img[v], px * 3, f);
    }
    for (int v = 0; v < 2; v++) {
        msk[v] = malloc(px);
        rd(msk[v], px, f);
    }
    fclose(f);
    for (int v = 0; v < 2; v++) in.image[v] = img[v], in.mask[v] = msk[v];
    orc_pair_out out;
    memset(&out, 0, sizeof out);
    out.disparity[0] = malloc(px * sizeof(double));
    out.disparity[1] = malloc(px * sizeof(double));
    out.max_points = (int64_t)px;
    out.xyz = malloc(px * 3 * sizeof(double));
    out.bgr = malloc(px * 3);
    int32_t st = orc_match_pair(&in, &out);
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(&st, 4, 1, f);
    for (int v = 0; v < 2; v++) {
        int32_t m[6] = {out.margin[v].YL, out.margin[v].YR, out.margin[v].XL, out.margin[v].XR, out.margin[v].width, out.margin[v].height};
        fwrite(m, sizeof m, 1, f);
    }
    fwrite(&out.n_points, 8, 1, f);
    fwrite(&out.v_top, 8, 1, f);
    fwrite(out.disparity[0], sizeof(double), px, f);
    fwrite(out.disparity[1], sizeof(double), px, f);
    const size_t n = st == 0 ? (size_t)out.n_points : 0;
    fwrite(out.xyz, sizeof(double), 3 * n, f);
    fwrite(out.bgr, 1, 3 * n, f);
    fclose(f);
    for (int v = 0; v < 2; v++) free(img[v]), free(msk[v]), free(out.disparity[v]);
    free(out.xyz);
    free(out.bgr);
    return 0;
}
