/* f-2: the precompiled-pattern blob -- the analogue of Precompile.precompile writing <className>.class
 * (needle-compiler/.../precompile/Precompile.java:30-53) -- through the C header alone: compile, serialize to a file,
 * read it back, deserialize, and run the three ops on the DEVICE with the revived pattern; results must equal the
 * original pattern's and the known answers of DFACompilerTest.java:524-540 / readme.md:36-50.  C99, -pedantic. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "needle_hip.h"

#define ROWS 6
#define STRIDE 32

static int fail(const char *what) {
    printf("FAIL %s: %s\n", what, needle_last_error());
    return 1;
}

int main(int argc, char **argv) {
    static const char *text[ROWS] = {"http://www.google.com", "see http://a.b now", "ftp://x", "", "http://", "xhttp://q\n"};
    const char *rx = "http://.+";
    uint16_t regex[16], rows[ROWS][STRIDE];
    uint32_t lengths[ROWS];
    uint64_t bm_a[3] = {0, 0, 0}, bm_b[3] = {0, 0, 0};
    int32_t s_a[ROWS], e_a[ROWS], s_b[ROWS], e_b[ROWS];
    needle_pattern *p = NULL, *q = NULL;
    needle_batch_view v;
    size_t need = 0, got = 0, i;
    unsigned char *blob, *back;
    FILE *f;
    const char *path = argc > 1 ? argv[1] : "/tmp/needle_blob_roundtrip.ndlt";
    int r;

    for (i = 0; i < strlen(rx); ++i) regex[i] = (uint16_t)rx[i];
    if (needle_compile(regex, strlen(rx), 0, &p) != NEEDLE_OK) return fail("compile");
    if (needle_pattern_serialize(p, NULL, 0, &need) != NEEDLE_OK || need == 0) return fail("serialize (size)");
    blob = (unsigned char *)malloc(need);
    if (needle_pattern_serialize(p, blob, need, &need) != NEEDLE_OK) return fail("serialize");
    f = fopen(path, "wb");
    if (!f || fwrite(blob, 1, need, f) != need) return fail("write blob");
    fclose(f);
    back = (unsigned char *)malloc(need);
    f = fopen(path, "rb");
    if (!f || (got = fread(back, 1, need, f)) != need) return fail("read blob");
    fclose(f);
    if (needle_pattern_deserialize(back, got, &q) != NEEDLE_OK) return fail("deserialize");
    back[7] ^= 0x40; /* a damaged blob is refused, not trusted */
    {
        needle_pattern *bad = NULL;
        if (needle_pattern_deserialize(back, got, &bad) == NEEDLE_OK || bad != NULL) return fail("damaged blob accepted");
    }
    if (needle_device_count() < 1) {
        printf("blob round trip ok (%lu bytes); no HIP device: device part skipped\n", (unsigned long)need);
        return 0;
    }
    memset(rows, 0, sizeof(rows));
    for (r = 0; r < ROWS; ++r) {
        lengths[r] = (uint32_t)strlen(text[r]);
        for (i = 0; i < lengths[r]; ++i) rows[r][i] = (uint16_t)(unsigned char)text[r][i];
    }
    memset(&v, 0, sizeof(v));
    v.rows = rows;
    v.char_width = 2;
    v.n_rows = ROWS;
    v.row_stride = STRIDE;
    v.lengths = lengths;
    if (needle_matches_host(p, &v, &bm_a[0]) || needle_contained_in_host(p, &v, &bm_a[1]) || needle_find_host(p, &v, &bm_a[2], s_a, e_a))
        return fail("scan with the compiled pattern");
    if (needle_matches_host(q, &v, &bm_b[0]) || needle_contained_in_host(q, &v, &bm_b[1]) || needle_find_host(q, &v, &bm_b[2], s_b, e_b))
        return fail("scan with the deserialized pattern");
    if (memcmp(bm_a, bm_b, sizeof(bm_a)) || memcmp(s_a, s_b, sizeof(s_a)) || memcmp(e_a, e_b, sizeof(e_a))) {
        printf("FAIL results differ between the compiled and the deserialized pattern\n");
        return 1;
    }
    /* known answers: matches() only rows 0; containedIn / find rows 0, 1, 5 ("http://" alone needs one more char) */
    if (bm_b[0] != 0x01u || bm_b[1] != 0x23u || bm_b[2] != 0x23u) {
        printf("FAIL bitmaps %lx %lx %lx\n", (unsigned long)bm_b[0], (unsigned long)bm_b[1], (unsigned long)bm_b[2]);
        return 1;
    }
    if (s_b[0] != 0 || e_b[0] != 21 || s_b[1] != 4 || e_b[1] != 18 || s_b[2] != -1 || e_b[2] != -1 || s_b[5] != 1 || e_b[5] != 9) {
        printf("FAIL start/end (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", s_b[0], e_b[0], s_b[1], e_b[1], s_b[2], e_b[2], s_b[5], e_b[5]);
        return 1;
    }
    printf("blob round trip ok (%lu bytes); device results identical and as expected\n", (unsigned long)need);
    needle_pattern_destroy(p);
    needle_pattern_destroy(q);
    free(blob);
    free(back);
    return 0;
}
