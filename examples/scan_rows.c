/* A C host of libneedle_hip.so: compile a regex, put a small fixed-stride batch in HBM, run the three ops and
 * find-all (dense slots, then the compact form), print the results.
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/scan_rows.c \
 *       -Lneedle_amd -lneedle_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/needle_amd -o scan_rows
 * (plain C: the HIP runtime is only used for hipMalloc / hipMemcpy). */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "needle_hip.h"

#define ROWS 5
#define STRIDE 32 /* bytes per row: a multiple of 16 */

int main(void) {
    static const char *text[ROWS] = {"order 66 shipped", "no digits here", "", "a1b22c333", "2024-01-31"};
    const uint16_t regex[] = {'[', '0', '-', '9', ']', '+'};
    char host[ROWS][STRIDE];
    uint32_t lengths[ROWS], counts[ROWS];
    uint64_t bitmap = 0;
    int32_t start[ROWS], end[ROWS], all_s[ROWS * 4], all_e[ROWS * 4];
    void *d_rows, *d_len, *d_bm, *d_s, *d_e, *d_cnt, *d_as, *d_ae;
    needle_pattern *p = NULL;
    needle_batch_view v;
    int more = 0, r, k;

    if (needle_device_count() < 1) {
        printf("no HIP device\n");
        return 2;
    }
    if (needle_compile(regex, 6, 0, &p) != NEEDLE_OK) {
        printf("compile failed: %s\n", needle_last_error());
        return 1;
    }
    memset(host, 0, sizeof(host));
    for (r = 0; r < ROWS; ++r) {
        lengths[r] = (uint32_t)strlen(text[r]);
        memcpy(host[r], text[r], lengths[r]);
    }
    hipMalloc(&d_rows, sizeof(host));
    hipMalloc(&d_len, sizeof(lengths));
    hipMalloc(&d_bm, 8);
    hipMalloc(&d_s, sizeof(start));
    hipMalloc(&d_e, sizeof(end));
    hipMalloc(&d_cnt, sizeof(counts));
    hipMalloc(&d_as, sizeof(all_s));
    hipMalloc(&d_ae, sizeof(all_e));
    hipMemcpy(d_rows, host, sizeof(host), hipMemcpyHostToDevice);
    hipMemcpy(d_len, lengths, sizeof(lengths), hipMemcpyHostToDevice);

    memset(&v, 0, sizeof(v));
    v.rows = d_rows;
    v.char_width = 1;
    v.n_rows = ROWS;
    v.row_stride = STRIDE;
    v.lengths = (const uint32_t *)d_len;

    if (needle_contained_in_dev(p, &v, (uint64_t *)d_bm, NULL) != NEEDLE_OK ||
        needle_find_dev(p, &v, (uint64_t *)d_bm, (int32_t *)d_s, (int32_t *)d_e, NULL) != NEEDLE_OK ||
        needle_find_all_dev(p, &v, 4, (uint32_t *)d_cnt, (int32_t *)d_as, (int32_t *)d_ae, &more, NULL) != NEEDLE_OK) {
        printf("scan failed: %s\n", needle_last_error());
        return 1;
    }
    hipDeviceSynchronize();
    hipMemcpy(&bitmap, d_bm, 8, hipMemcpyDeviceToHost);
    hipMemcpy(start, d_s, sizeof(start), hipMemcpyDeviceToHost);
    hipMemcpy(end, d_e, sizeof(end), hipMemcpyDeviceToHost);
    hipMemcpy(counts, d_cnt, sizeof(counts), hipMemcpyDeviceToHost);
    hipMemcpy(all_s, d_as, sizeof(all_s), hipMemcpyDeviceToHost);
    hipMemcpy(all_e, d_ae, sizeof(all_e), hipMemcpyDeviceToHost);
    for (r = 0; r < ROWS; ++r) {
        printf("row %d \"%s\": found=%d first=(%d,%d) all=", r, text[r], (int)((bitmap >> r) & 1), start[r], end[r]);
        for (k = 0; k < (int)counts[r]; ++k) printf("(%d,%d)", all_s[r * 4 + k], all_e[r * 4 + k]);
        printf("\n");
    }
    printf("more=%d\n", more);

    /* the same enumeration in compact form: count pass, prefix sum (here on the host: 5 rows), fill pass */
    {
        uint64_t offsets[ROWS + 1];
        int32_t csr_s[ROWS * 4], csr_e[ROWS * 4];
        void *d_off;
        if (needle_count_matches_dev(p, &v, (uint32_t *)d_cnt, NULL) != NEEDLE_OK) {
            printf("count failed: %s\n", needle_last_error());
            return 1;
        }
        hipDeviceSynchronize();
        hipMemcpy(counts, d_cnt, sizeof(counts), hipMemcpyDeviceToHost);
        offsets[0] = 0;
        for (r = 0; r < ROWS; ++r) offsets[r + 1] = offsets[r] + counts[r];
        hipMalloc(&d_off, sizeof(offsets));
        hipMemcpy(d_off, offsets, sizeof(offsets), hipMemcpyHostToDevice);
        if (needle_find_all_csr_dev(p, &v, (const uint64_t *)d_off, (int32_t *)d_as, (int32_t *)d_ae, &more, NULL) != NEEDLE_OK) {
            printf("csr failed: %s\n", needle_last_error());
            return 1;
        }
        hipMemcpy(csr_s, d_as, sizeof(csr_s), hipMemcpyDeviceToHost);
        hipMemcpy(csr_e, d_ae, sizeof(csr_e), hipMemcpyDeviceToHost);
        printf("csr total=%d more=%d:", (int)offsets[ROWS], more);
        for (k = 0; k < (int)offsets[ROWS]; ++k) printf(" (%d,%d)", csr_s[k], csr_e[k]);
        printf("\n");
    }
    /* find() with a row's start / end as ONE dword, start | end << 16, stored by the scan kernel itself (0xFFFFFFFF = no match) */
    {
        uint32_t se[ROWS];
        void *d_se;
        hipMalloc(&d_se, sizeof(se));
        if (needle_find_packed16_dev(p, &v, (uint64_t *)d_bm, (uint32_t *)d_se, NULL) != NEEDLE_OK) {
            printf("packed find failed: %s\n", needle_last_error());
            return 1;
        }
        hipMemcpy(se, d_se, sizeof(se), hipMemcpyDeviceToHost);
        printf("packed:");
        for (r = 0; r < ROWS; ++r) {
            if (se[r] == 0xFFFFFFFFu) printf(" -");
            else printf(" (%u,%u)", se[r] & 0xFFFFu, se[r] >> 16);
        }
        printf("\n");
        hipFree(d_se);
    }
    /* the library's environment switches, as the library lists them */
    {
        size_t need = 0;
        char head[64];
        if (needle_tuning_info(NULL, 0, &need) != NEEDLE_OK || needle_tuning_info(head, sizeof(head), NULL) != NEEDLE_OK) return 1;
        printf("tuning info: %u bytes, header \"%.33s\"\n", (unsigned)(need > 1000), head);
    }
    needle_pattern_destroy(p);
    return 0;
}
