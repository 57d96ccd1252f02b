/* The boundary is a C ABI: this file is compiled as C (gcc -std=c99 -pedantic) against include/needle_hip.h and
 * linked with libneedle_hip.so by tests/test_c_abi.py.  Everything here runs without a GPU: compile a regex, read
 * the tables back, round-trip the precompiled blob, check the error conventions. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "needle_hip.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            printf("FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, needle_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(void) {
    const uint16_t regex[] = {'[', '0', '-', '9', ']', '+'};
    const uint16_t bad[] = {'(', 'a'};
    needle_pattern *p = NULL, *q = NULL;
    needle_pattern_info info, info2;
    uint8_t *class_map;
    int16_t table[2 * 4];
    uint8_t accepting[2];
    size_t need = 0;
    void *blob;

    CHECK(strstr(needle_version(), "needle_hip") != NULL);
    CHECK(needle_compile(regex, 6, 0, &p) == NEEDLE_OK && p != NULL);
    CHECK(needle_pattern_get_info(p, &info) == NEEDLE_OK);
    CHECK(info.stride == 4 && info.n_states[1] == 2 && info.fixed_len == -1 && info.min_len == 1 && info.max_len == -1);
    class_map = (uint8_t *)malloc(65536);
    CHECK(needle_pattern_get_class_map(p, class_map) == NEEDLE_OK);
    CHECK(class_map['5'] == 2 && class_map['a'] == 1 && class_map[0xFFFF] == 0); /* DigitPlus snapshot */
    CHECK(needle_pattern_get_table(p, 1, table, accepting) == NEEDLE_OK);          /* CONTAINED_IN: "0:1-0,2-1,0-0" */
    CHECK(table[0] == 0 && table[1] == 0 && table[2] == 1 && accepting[0] == 0 && accepting[1] == 1);

    CHECK(needle_pattern_serialize(p, NULL, 0, &need) == NEEDLE_OK && need > 65536);
    blob = malloc(need);
    CHECK(needle_pattern_serialize(p, blob, need, &need) == NEEDLE_OK);
    CHECK(needle_pattern_deserialize(blob, need, &q) == NEEDLE_OK && q != NULL);
    CHECK(needle_pattern_get_info(q, &info2) == NEEDLE_OK && info2.stride == info.stride && info2.n_states[2] == info.n_states[2]);

    CHECK(needle_compile(bad, 2, 0, &q) == NEEDLE_ERR_SYNTAX);          /* PatternSyntaxException */
    CHECK(strlen(needle_last_error()) > 0);
    CHECK(needle_compile(regex, 6, 0x4, &q) == NEEDLE_ERR_INVALID);     /* unknown flag bit: IllegalArgumentException */
    CHECK(needle_compile(regex, 6, 0, NULL) == NEEDLE_ERR_INVALID);
    CHECK(needle_device_count() >= 0);

    needle_pattern_destroy(p);
    free(class_map);
    free(blob);
    printf("c abi ok\n");
    return 0;
}
