/*
 * JNI shim: com.justinblank.strings.gpu.Native -> libneedle_hip.so (include/needle_hip.h).
 * Thin by design: unpack Java arrays / direct buffers, call the C ABI, return its status code.  No exception is
 * raised here; GpuPattern.check() maps status codes to the reference's exception types.
 *
 * NOT COMPILED IN THE BUILD CONTAINER: the image has no JDK (no jni.h).  Build on a box with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude bindings/jni/needle_jni.c \
 *       -Lneedle_amd -lneedle_hip -o libneedle_jni.so
 * The same entry points are exercised without a JVM by tests/ through ctypes (needle_amd/_lib.py).
 */
#include <jni.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include "needle_hip.h"

#define NATIVE(ret, name) JNIEXPORT ret JNICALL Java_com_justinblank_strings_gpu_Native_##name

NATIVE(jstring, lastError)(JNIEnv *env, jclass c) { return (*env)->NewStringUTF(env, needle_last_error()); }
NATIVE(jint, deviceCount)(JNIEnv *env, jclass c) { return needle_device_count(); }

NATIVE(jint, compile)(JNIEnv *env, jclass c, jcharArray regex, jint flags, jlongArray out) {
    jsize n = (*env)->GetArrayLength(env, regex);
    jchar *u = (*env)->GetCharArrayElements(env, regex, NULL);
    needle_pattern *p = NULL;
    int rc = needle_compile((const uint16_t *)u, (size_t)n, flags, &p);
    (*env)->ReleaseCharArrayElements(env, regex, u, JNI_ABORT);
    jlong h = (jlong)(intptr_t)p;
    (*env)->SetLongArrayRegion(env, out, 0, 1, &h);
    return rc;
}

NATIVE(jint, fromTables)(JNIEnv *env, jclass c, jbyteArray classMap, jint stride, jintArray nStates, jintArray maxChar,
                         jobjectArray tables, jobjectArray accepting, jint fixedLen, jlongArray out) {
    /* Everything the C ABI will read is checked against the Java arrays' real lengths first: needle_pattern_from_tables
     * copies n_states * stride shorts, n_states bytes and 65536 class-map bytes from raw pointers. */
    if (!classMap || !nStates || !maxChar || !tables || !accepting || !out) return NEEDLE_ERR_INVALID;
    if ((*env)->GetArrayLength(env, classMap) < 65536 || (*env)->GetArrayLength(env, nStates) < 4 ||
        (*env)->GetArrayLength(env, maxChar) < 4 || (*env)->GetArrayLength(env, tables) < 4 ||
        (*env)->GetArrayLength(env, accepting) < 4 || (*env)->GetArrayLength(env, out) < 1 || stride < 1 || stride > 255)
        return NEEDLE_ERR_INVALID;
    needle_table_desc d;
    memset(&d, 0, sizeof(d));
    jint ns[4], mc[4];
    (*env)->GetIntArrayRegion(env, nStates, 0, 4, ns);
    (*env)->GetIntArrayRegion(env, maxChar, 0, 4, mc);
    if ((*env)->ExceptionCheck(env)) return NEEDLE_ERR_INVALID;
    needle_dfa_desc *ds[4] = {&d.matches, &d.contained_in, &d.forwards, &d.backwards};
    jshortArray ta[4] = {0};
    jbyteArray aa[4] = {0};
    jshort *tp[4] = {0};
    jbyte *ap[4] = {0};
    jbyte *cm = NULL;
    int rc = NEEDLE_OK;
    for (int i = 0; i < 4 && rc == NEEDLE_OK; i++) {
        ta[i] = (jshortArray)(*env)->GetObjectArrayElement(env, tables, i);
        aa[i] = (jbyteArray)(*env)->GetObjectArrayElement(env, accepting, i);
        if (!ta[i] || !aa[i] || ns[i] < 1 || ns[i] > 16383 ||
            (jlong)(*env)->GetArrayLength(env, ta[i]) < (jlong)ns[i] * stride || (*env)->GetArrayLength(env, aa[i]) < ns[i])
            rc = NEEDLE_ERR_INVALID;
    }
    if (rc == NEEDLE_OK && !(cm = (*env)->GetByteArrayElements(env, classMap, NULL))) rc = NEEDLE_ERR_INVALID;
    for (int i = 0; i < 4 && rc == NEEDLE_OK; i++) {
        tp[i] = (*env)->GetShortArrayElements(env, ta[i], NULL);
        ap[i] = (*env)->GetByteArrayElements(env, aa[i], NULL);
        if (!tp[i] || !ap[i]) rc = NEEDLE_ERR_INVALID; /* OutOfMemoryError pending */
        ds[i]->n_states = ns[i];
        ds[i]->max_char = mc[i];
        ds[i]->table = (const int16_t *)tp[i];
        ds[i]->accepting = (const uint8_t *)ap[i];
    }
    needle_pattern *p = NULL;
    if (rc == NEEDLE_OK) {
        d.class_map = (const uint8_t *)cm;
        d.stride = stride;
        d.fixed_len = fixedLen;
        rc = needle_pattern_from_tables(&d, &p); /* copies everything it needs */
    }
    for (int i = 0; i < 4; i++) {
        if (tp[i]) (*env)->ReleaseShortArrayElements(env, ta[i], tp[i], JNI_ABORT);
        if (ap[i]) (*env)->ReleaseByteArrayElements(env, aa[i], ap[i], JNI_ABORT);
    }
    if (cm) (*env)->ReleaseByteArrayElements(env, classMap, cm, JNI_ABORT);
    jlong h = (jlong)(intptr_t)p;
    (*env)->SetLongArrayRegion(env, out, 0, 1, &h);
    return rc;
}

NATIVE(void, destroyPattern)(JNIEnv *env, jclass c, jlong h) { needle_pattern_destroy((needle_pattern *)(intptr_t)h); }

static void view_of(JNIEnv *env, needle_batch_view *v, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths) {
    memset(v, 0, sizeof(*v));
    v->rows = (*env)->GetDirectBufferAddress(env, rows);
    v->char_width = (uint32_t)cw;
    v->n_rows = (uint64_t)n;
    v->row_stride = (uint64_t)stride;
    v->row_len = (uint32_t)rowLen;
    v->lengths = lengths ? (const uint32_t *)(*env)->GetDirectBufferAddress(env, lengths) : NULL;
}

/* n ints / longs fit the Java array (a short array from the caller must not become a write past it) */
static int int_room(JNIEnv *env, jintArray a, jlong n) { return a != NULL && (jlong)(*env)->GetArrayLength(env, a) >= n; }
static int long_room(JNIEnv *env, jlongArray a, jlong n) { return a != NULL && (jlong)(*env)->GetArrayLength(env, a) >= n; }

static jint bitmap_call(JNIEnv *env, int which, jlong h, needle_batch_view *v, jlongArray bitmap, jintArray start, jintArray end) {
    if (!long_room(env, bitmap, ((jlong)v->n_rows + 63) / 64)) return NEEDLE_ERR_INVALID;
    if (which == 2 && (!int_room(env, start, (jlong)v->n_rows) || !int_room(env, end, (jlong)v->n_rows))) return NEEDLE_ERR_INVALID;
    jlong *bm = (*env)->GetLongArrayElements(env, bitmap, NULL);
    jint *s = start ? (*env)->GetIntArrayElements(env, start, NULL) : NULL;
    jint *e = end ? (*env)->GetIntArrayElements(env, end, NULL) : NULL;
    const needle_pattern *p = (const needle_pattern *)(intptr_t)h;
    int rc = which == 0 ? needle_matches_host(p, v, (uint64_t *)bm)
           : which == 1 ? needle_contained_in_host(p, v, (uint64_t *)bm)
                        : needle_find_host(p, v, (uint64_t *)bm, (int32_t *)s, (int32_t *)e);
    (*env)->ReleaseLongArrayElements(env, bitmap, bm, 0);
    if (s) (*env)->ReleaseIntArrayElements(env, start, s, 0);
    if (e) (*env)->ReleaseIntArrayElements(env, end, e, 0);
    return rc;
}

NATIVE(jint, matchesHost)(JNIEnv *env, jclass c, jlong h, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths, jlongArray bitmap) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    return bitmap_call(env, 0, h, &v, bitmap, NULL, NULL);
}
NATIVE(jint, containedInHost)(JNIEnv *env, jclass c, jlong h, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths, jlongArray bitmap) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    return bitmap_call(env, 1, h, &v, bitmap, NULL, NULL);
}
NATIVE(jint, findHost)(JNIEnv *env, jclass c, jlong h, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths, jlongArray bitmap, jintArray start, jintArray end) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    return bitmap_call(env, 2, h, &v, bitmap, start, end);
}

NATIVE(jint, findAllHost)(JNIEnv *env, jclass c, jlong h, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths, jint maxPerRow, jintArray counts, jintArray start, jintArray end, jintArray more) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    if (maxPerRow < 0 || !int_room(env, counts, n) || !int_room(env, start, n * (jlong)maxPerRow) || !int_room(env, end, n * (jlong)maxPerRow)) return NEEDLE_ERR_INVALID;
    jint *cn = (*env)->GetIntArrayElements(env, counts, NULL);
    jint *st = (*env)->GetIntArrayElements(env, start, NULL);
    jint *en = (*env)->GetIntArrayElements(env, end, NULL);
    int m = 0;
    int rc = needle_find_all_host((const needle_pattern *)(intptr_t)h, &v, (uint32_t)maxPerRow, (uint32_t *)cn, (int32_t *)st, (int32_t *)en, &m);
    jint jm = m;
    if (int_room(env, more, 1)) (*env)->SetIntArrayRegion(env, more, 0, 1, &jm);
    (*env)->ReleaseIntArrayElements(env, counts, cn, 0);
    (*env)->ReleaseIntArrayElements(env, start, st, 0);
    (*env)->ReleaseIntArrayElements(env, end, en, 0);
    return rc;
}

NATIVE(jint, findAllPacked16Host)(JNIEnv *env, jclass c, jlong h, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths, jint maxPerRow, jintArray counts, jintArray startEnd, jintArray more) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    if (maxPerRow < 0 || !int_room(env, counts, n) || !int_room(env, startEnd, n * (jlong)maxPerRow)) return NEEDLE_ERR_INVALID;
    jint *cn = (*env)->GetIntArrayElements(env, counts, NULL);
    jint *se = (*env)->GetIntArrayElements(env, startEnd, NULL);
    int m = 0;
    int rc = needle_find_all_packed16_host((const needle_pattern *)(intptr_t)h, &v, (uint32_t)maxPerRow, (uint32_t *)cn, (uint32_t *)se, &m);
    jint mm = m;
    if (int_room(env, more, 1)) (*env)->SetIntArrayRegion(env, more, 0, 1, &mm);
    (*env)->ReleaseIntArrayElements(env, counts, cn, 0);
    (*env)->ReleaseIntArrayElements(env, startEnd, se, 0);
    return rc;
}

/* needle_find_all_csr_host: offsets long[nRows + 1]; start / end int[capacity] (may be null with capacity 0: count only);
 * total long[1]. */
NATIVE(jint, findAllCsrHost)(JNIEnv *env, jclass c, jlong h, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths, jlongArray offsets, jintArray start, jintArray end, jlongArray total) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    if (!offsets || !total || (*env)->GetArrayLength(env, offsets) < n + 1 || (*env)->GetArrayLength(env, total) < 1) return NEEDLE_ERR_INVALID;
    jsize cap = start ? (*env)->GetArrayLength(env, start) : 0;
    if (end && (*env)->GetArrayLength(env, end) < cap) cap = (*env)->GetArrayLength(env, end);
    if (!end) cap = 0;
    jlong *of = (*env)->GetLongArrayElements(env, offsets, NULL);
    jint *st = cap ? (*env)->GetIntArrayElements(env, start, NULL) : NULL;
    jint *en = cap ? (*env)->GetIntArrayElements(env, end, NULL) : NULL;
    uint64_t t = 0;
    int rc = needle_find_all_csr_host((const needle_pattern *)(intptr_t)h, &v, (uint64_t *)of, (int32_t *)st, (int32_t *)en, (uint64_t)cap, &t);
    jlong jt = (jlong)t;
    (*env)->SetLongArrayRegion(env, total, 0, 1, &jt);
    (*env)->ReleaseLongArrayElements(env, offsets, of, 0);
    if (st) (*env)->ReleaseIntArrayElements(env, start, st, 0);
    if (en) (*env)->ReleaseIntArrayElements(env, end, en, 0);
    return rc;
}

/* needle_find_compact_host: bitmap long[ceil(n / 64)]; records int[2 * cap] = needle_match_rec[cap]; nMatched long[1]. */
NATIVE(jint, findCompactHost)(JNIEnv *env, jclass c, jlong h, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths, jlongArray bitmap, jintArray records, jlongArray nMatched) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    if (!bitmap || !nMatched || (*env)->GetArrayLength(env, bitmap) < (n + 63) / 64 || (*env)->GetArrayLength(env, nMatched) < 1) return NEEDLE_ERR_INVALID;
    const jsize cap = records ? (*env)->GetArrayLength(env, records) / 2 : 0;
    jlong *bm = (*env)->GetLongArrayElements(env, bitmap, NULL);
    jint *rec = cap ? (*env)->GetIntArrayElements(env, records, NULL) : NULL;
    uint64_t m = 0;
    int rc = needle_find_compact_host((const needle_pattern *)(intptr_t)h, &v, (uint64_t *)bm, (needle_match_rec *)rec, (uint64_t)cap, &m);
    jlong jm = (jlong)m;
    (*env)->SetLongArrayRegion(env, nMatched, 0, 1, &jm);
    (*env)->ReleaseLongArrayElements(env, bitmap, bm, 0);
    if (rec) (*env)->ReleaseIntArrayElements(env, records, rec, 0);
    return rc;
}

NATIVE(jint, findPacked16Host)(JNIEnv *env, jclass c, jlong h, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths, jlongArray bitmap, jintArray startEnd) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    if (!long_room(env, bitmap, (n + 63) / 64) || !int_room(env, startEnd, n)) return NEEDLE_ERR_INVALID;
    jlong *bm = (*env)->GetLongArrayElements(env, bitmap, NULL);
    jint *se = (*env)->GetIntArrayElements(env, startEnd, NULL);
    int rc = needle_find_packed16_host((const needle_pattern *)(intptr_t)h, &v, (uint64_t *)bm, (uint32_t *)se);
    (*env)->ReleaseLongArrayElements(env, bitmap, bm, 0);
    (*env)->ReleaseIntArrayElements(env, startEnd, se, 0);
    return rc;
}

/* needle_find_packed8_host: rows of at most 256 chars, ONE short per row (start | (end - start) << 8; 0xFFFF: no match, 0xFFFE: (0, 256)) */
NATIVE(jint, findPacked8Host)(JNIEnv *env, jclass c, jlong h, jobject rows, jint cw, jlong n, jlong stride, jint rowLen, jobject lengths, jlongArray bitmap, jshortArray startLen) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    if (!long_room(env, bitmap, (n + 63) / 64) || startLen == NULL || (jlong)(*env)->GetArrayLength(env, startLen) < n) return NEEDLE_ERR_INVALID;
    jlong *bm = (*env)->GetLongArrayElements(env, bitmap, NULL);
    jshort *sl = (*env)->GetShortArrayElements(env, startLen, NULL);
    int rc = needle_find_packed8_host((const needle_pattern *)(intptr_t)h, &v, (uint64_t *)bm, (uint16_t *)sl);
    (*env)->ReleaseLongArrayElements(env, bitmap, bm, 0);
    (*env)->ReleaseShortArrayElements(env, startLen, sl, 0);
    return rc;
}

NATIVE(jstring, tuningInfo)(JNIEnv *env, jclass c) {
    size_t need = 0;
    if (needle_tuning_info(NULL, 0, &need) != NEEDLE_OK || need == 0) return (*env)->NewStringUTF(env, "");
    char *buf = (char *)malloc(need);
    if (!buf) return (*env)->NewStringUTF(env, "");
    (void)needle_tuning_info(buf, need, NULL);
    jstring s = (*env)->NewStringUTF(env, buf);
    free(buf);
    return s;
}

NATIVE(jint, trimScratch)(JNIEnv *env, jclass c, jlong keepBytes) { return needle_trim_scratch(keepBytes < 0 ? 0 : (size_t)keepBytes); }

NATIVE(jstring, prefilterInfo)(JNIEnv *env, jclass c, jlong h, jint which, jintArray info) {
    needle_prefilter_info pi;
    if (needle_pattern_prefilter_info((const needle_pattern *)(intptr_t)h, which, &pi, NULL) != NEEDLE_OK) return (*env)->NewStringUTF(env, needle_last_error());
    if (int_room(env, info, 7)) {
        const jint v[7] = {pi.on, pi.mode, pi.stride, pi.warm, pi.min_len, pi.n_windows, pi.bitmap_bytes};
        (*env)->SetIntArrayRegion(env, info, 0, 7, v);
    }
    return (*env)->NewStringUTF(env, pi.why);
}

/* needle_pattern_set_prefilter / needle_pattern_prefilter_state: the filter's flood watch pinned and reported (include/needle_hip.h).
 * state[0..3] = mode, has_filter, suspended_calls_left, backoff; rate[0] = last candidates per KiB; counts[0..1] = launches, suspended calls. */
NATIVE(jint, setPrefilter)(JNIEnv *env, jclass c, jlong h, jint mode) { return needle_pattern_set_prefilter((needle_pattern *)(intptr_t)h, mode); }

NATIVE(jint, prefilterState)(JNIEnv *env, jclass c, jlong h, jint which, jintArray state, jfloatArray rate, jlongArray counts) {
    needle_prefilter_state st;
    const int rc = needle_pattern_prefilter_state((const needle_pattern *)(intptr_t)h, which, &st);
    if (rc != NEEDLE_OK) return rc;
    if (int_room(env, state, 4)) {
        const jint v[4] = {st.mode, st.has_filter, st.suspended_calls_left, st.backoff};
        (*env)->SetIntArrayRegion(env, state, 0, 4, v);
    }
    if (rate && (*env)->GetArrayLength(env, rate) >= 1) {
        const jfloat r = st.last_candidates_per_kib;
        (*env)->SetFloatArrayRegion(env, rate, 0, 1, &r);
    }
    if (counts && (*env)->GetArrayLength(env, counts) >= 2) {
        const jlong v[2] = {(jlong)st.filter_launches, (jlong)st.suspended_calls};
        (*env)->SetLongArrayRegion(env, counts, 0, 2, v);
    }
    return NEEDLE_OK;
}

/* needle_pattern_utf16_route: out[0] = page (-1: none), out[1] = substitute byte */
NATIVE(jint, utf16Route)(JNIEnv *env, jclass c, jlong h, jintArray out) {
    int32_t page = -1, sub = 0;
    const int rc = needle_pattern_utf16_route((const needle_pattern *)(intptr_t)h, &page, &sub);
    if (rc != NEEDLE_OK) return rc;
    if (int_room(env, out, 2)) {
        const jint v[2] = {page, sub};
        (*env)->SetIntArrayRegion(env, out, 0, 2, v);
    }
    return NEEDLE_OK;
}

NATIVE(jint, packedHost)(JNIEnv *env, jclass c, jlong h, jint op, jcharArray data, jlongArray offsets, jlongArray bitmap, jintArray start, jintArray end) {
    needle_packed_view v;
    memset(&v, 0, sizeof(v));
    jsize n1 = (*env)->GetArrayLength(env, offsets);
    jchar *d = (*env)->GetCharArrayElements(env, data, NULL);
    jlong *o = (*env)->GetLongArrayElements(env, offsets, NULL);
    jlong *bm = (*env)->GetLongArrayElements(env, bitmap, NULL);
    jint *st = start ? (*env)->GetIntArrayElements(env, start, NULL) : NULL;
    jint *en = end ? (*env)->GetIntArrayElements(env, end, NULL) : NULL;
    v.data = d;
    v.char_width = 2;
    v.n_rows = (uint64_t)(n1 - 1);
    v.offsets = (const uint64_t *)o;
    const needle_pattern *p = (const needle_pattern *)(intptr_t)h;
    int rc = op == 0   ? needle_matches_packed_host(p, &v, (uint64_t *)bm)
             : op == 1 ? needle_contained_in_packed_host(p, &v, (uint64_t *)bm)
                       : needle_find_packed_host(p, &v, (uint64_t *)bm, (int32_t *)st, (int32_t *)en);
    (*env)->ReleaseCharArrayElements(env, data, d, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, offsets, o, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, bitmap, bm, 0);
    if (st) (*env)->ReleaseIntArrayElements(env, start, st, 0);
    if (en) (*env)->ReleaseIntArrayElements(env, end, en, 0);
    return rc;
}

NATIVE(jbyteArray, serialize)(JNIEnv *env, jclass c, jlong h) {
    size_t need = 0;
    const needle_pattern *p = (const needle_pattern *)(intptr_t)h;
    if (needle_pattern_serialize(p, NULL, 0, &need) != NEEDLE_OK || need > 0x7FFFFFFF) return NULL;
    jbyteArray out = (*env)->NewByteArray(env, (jsize)need);
    if (!out) return NULL;
    jbyte *b = (*env)->GetByteArrayElements(env, out, NULL);
    if (!b) return NULL;
    int rc = needle_pattern_serialize(p, b, need, &need);
    (*env)->ReleaseByteArrayElements(env, out, b, 0);
    return rc == NEEDLE_OK ? out : NULL;
}

NATIVE(jint, deserialize)(JNIEnv *env, jclass c, jbyteArray blob, jlongArray out) {
    if (!blob || !out || (*env)->GetArrayLength(env, out) < 1) return NEEDLE_ERR_INVALID;
    jsize n = (*env)->GetArrayLength(env, blob);
    jbyte *b = (*env)->GetByteArrayElements(env, blob, NULL);
    if (!b) return NEEDLE_ERR_INVALID;
    needle_pattern *p = NULL;
    int rc = needle_pattern_deserialize(b, (size_t)n, &p); /* validates every length and value itself */
    (*env)->ReleaseByteArrayElements(env, blob, b, JNI_ABORT);
    jlong h = (jlong)(intptr_t)p;
    (*env)->SetLongArrayRegion(env, out, 0, 1, &h);
    return rc;
}

NATIVE(jint, multiCreate)(JNIEnv *env, jclass c, jintArray devices, jint flags, jlongArray out) {
    if (!devices || !out || (*env)->GetArrayLength(env, out) < 1) return NEEDLE_ERR_INVALID;
    jsize n = (*env)->GetArrayLength(env, devices);
    jint *d = (*env)->GetIntArrayElements(env, devices, NULL);
    if (!d) return NEEDLE_ERR_INVALID;
    needle_multi *m = NULL;
    int rc = needle_multi_create((const int *)d, (int)n, (unsigned)flags, &m);
    (*env)->ReleaseIntArrayElements(env, devices, d, JNI_ABORT);
    jlong h = (jlong)(intptr_t)m;
    (*env)->SetLongArrayRegion(env, out, 0, 1, &h);
    return rc;
}
NATIVE(void, multiDestroy)(JNIEnv *env, jclass c, jlong m) { needle_multi_destroy((needle_multi *)(intptr_t)m); }

NATIVE(jint, scanHostMulti)(JNIEnv *env, jclass c, jlong m, jlong h, jint op, jobject rows, jint cw, jlong n, jlong stride, jint rowLen,
                            jobject lengths, jlongArray bitmap, jintArray start, jintArray end) {
    needle_batch_view v;
    view_of(env, &v, rows, cw, n, stride, rowLen, lengths);
    if (!bitmap || (*env)->GetArrayLength(env, bitmap) < (jsize)((n + 63) / 64)) return NEEDLE_ERR_INVALID;
    if (op == 2 && (!start || !end || (*env)->GetArrayLength(env, start) < n || (*env)->GetArrayLength(env, end) < n)) return NEEDLE_ERR_INVALID;
    jlong *bm = (*env)->GetLongArrayElements(env, bitmap, NULL);
    jint *s = (op == 2) ? (*env)->GetIntArrayElements(env, start, NULL) : NULL;
    jint *e = (op == 2) ? (*env)->GetIntArrayElements(env, end, NULL) : NULL;
    int rc = needle_scan_host_multi((needle_multi *)(intptr_t)m, (const needle_pattern *)(intptr_t)h, op, &v, (uint64_t *)bm, (int32_t *)s, (int32_t *)e);
    (*env)->ReleaseLongArrayElements(env, bitmap, bm, 0);
    if (s) (*env)->ReleaseIntArrayElements(env, start, s, 0);
    if (e) (*env)->ReleaseIntArrayElements(env, end, e, 0);
    return rc;
}

NATIVE(jint, matcherCreate)(JNIEnv *env, jclass c, jlong pattern, jcharArray s, jlongArray out) {
    jsize n = (*env)->GetArrayLength(env, s);
    jchar *u = (*env)->GetCharArrayElements(env, s, NULL);
    needle_matcher *m = NULL;
    int rc = needle_matcher_create((const needle_pattern *)(intptr_t)pattern, (const uint16_t *)u, (size_t)n, &m);
    (*env)->ReleaseCharArrayElements(env, s, u, JNI_ABORT);
    jlong h = (jlong)(intptr_t)m;
    (*env)->SetLongArrayRegion(env, out, 0, 1, &h);
    return rc;
}
NATIVE(void, matcherDestroy)(JNIEnv *env, jclass c, jlong m) { needle_matcher_destroy((needle_matcher *)(intptr_t)m); }

#define BOOL_CALL(jname, cname)                                                        \
    NATIVE(jint, jname)(JNIEnv *env, jclass c, jlong m, jintArray r) {                 \
        int v = 0;                                                                     \
        int rc = cname((needle_matcher *)(intptr_t)m, &v);                             \
        jint jv = v;                                                                   \
        (*env)->SetIntArrayRegion(env, r, 0, 1, &jv);                                  \
        return rc;                                                                     \
    }
BOOL_CALL(matcherMatches, needle_matcher_matches)
BOOL_CALL(matcherContainedIn, needle_matcher_contained_in)
BOOL_CALL(matcherFind, needle_matcher_find)

NATIVE(jint, matcherFindRange)(JNIEnv *env, jclass c, jlong m, jint from, jint to, jintArray r) {
    int v = 0;
    int rc = needle_matcher_find_range((needle_matcher *)(intptr_t)m, from, to, &v);
    jint jv = v;
    (*env)->SetIntArrayRegion(env, r, 0, 1, &jv);
    return rc;
}
NATIVE(jint, matcherStart)(JNIEnv *env, jclass c, jlong m) { return needle_matcher_start((const needle_matcher *)(intptr_t)m); }
NATIVE(jint, matcherEnd)(JNIEnv *env, jclass c, jlong m) { return needle_matcher_end((const needle_matcher *)(intptr_t)m); }
