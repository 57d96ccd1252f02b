package com.justinblank.strings.gpu;

/**
 * JNI surface of libneedle_hip.so (include/needle_hip.h).  One-to-one with the C ABI; every method returns the
 * library's status code except where noted, and the shim (bindings/jni/needle_jni.c) never throws -- exceptions
 * are raised on the Java side from the status (see GpuPattern.check).
 *
 * NOT COMPILED IN THE BUILD CONTAINER (no JDK, no jni.h); shipped as source for a maintainer with a JDK.
 */
final class Native {
    static {
        System.loadLibrary("needle_jni"); // links libneedle_hip.so
    }

    private Native() {
    }

    static native String lastError();

    static native int deviceCount();

    /** needle_compile: regex as UTF-16 code units (String.toCharArray()). handleOut[0] receives the pattern handle. */
    static native int compile(char[] regex, int flags, long[] handleOut);

    /**
     * needle_pattern_from_tables: what DFATableEmitter extracted from needle's own four DFAs.
     * classMap: 65536 bytes (BYTE_CLASSES[0..65535]); tables[i]: flat short[nStates[i] * stride], -1 = none;
     * accepting[i]: byte[nStates[i]]; order of i: matches, containedIn, forwards, backwards.
     */
    static native int fromTables(byte[] classMap, int stride, int[] nStates, int[] maxChar, short[][] tables,
                                 byte[][] accepting, int fixedLen, long[] handleOut);

    static native void destroyPattern(long handle);

    /**
     * Host-buffer batches.  rows: direct ByteBuffer of nRows * rowStride chars (charWidth 1 or 2, native order);
     * lengths: direct buffer of nRows ints or null; bitmap: long[(nRows + 63) / 64]; start/end: int[nRows].
     */
    static native int matchesHost(long handle, java.nio.ByteBuffer rows, int charWidth, long nRows, long rowStride,
                                  int rowLen, java.nio.ByteBuffer lengths, long[] bitmap);

    static native int containedInHost(long handle, java.nio.ByteBuffer rows, int charWidth, long nRows,
                                      long rowStride, int rowLen, java.nio.ByteBuffer lengths, long[] bitmap);

    static native int findHost(long handle, java.nio.ByteBuffer rows, int charWidth, long nRows, long rowStride,
                               int rowLen, java.nio.ByteBuffer lengths, long[] bitmap, int[] start, int[] end);

    /**
     * Packed host batches (needle_*_packed_host): the UTF-16 code units of all haystacks back to back + offsets[n + 1]
     * (what a String[] flattens to).  op: 0 matches, 1 containedIn, 2 find (start/end may be null otherwise).
     */
    static native int packedHost(long handle, int op, char[] data, long[] offsets, long[] bitmap, int[] start, int[] end);

    /** needle_find_all_host: counts int[nRows]; start / end int[nRows * maxPerRow]; more int[1]. */
    static native int findAllHost(long handle, java.nio.ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen,
                                  java.nio.ByteBuffer lengths, int maxPerRow, int[] counts, int[] start, int[] end, int[] more);

    /** needle_find_all_packed16_host: counts int[nRows]; startEnd int[nRows * maxPerRow], each match start | end << 16. */
    static native int findAllPacked16Host(long handle, java.nio.ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen,
                                          java.nio.ByteBuffer lengths, int maxPerRow, int[] counts, int[] startEnd, int[] more);

    /** needle_find_all_csr_host: offsets long[nRows + 1]; start / end int[capacity] (null: count only); total long[1]. */
    static native int findAllCsrHost(long handle, java.nio.ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen,
                                     java.nio.ByteBuffer lengths, long[] offsets, int[] start, int[] end, long[] total);

    /**
     * needle_find_compact_host: find() with the MATCHED rows only, in row order -- records[2 * k] = row, records[2 * k + 1] =
     * start | end << 16 (needle_match_rec as two ints); nMatched long[1] = the number of matched rows (may exceed the room).
     */
    static native int findCompactHost(long handle, java.nio.ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen,
                                      java.nio.ByteBuffer lengths, long[] bitmap, int[] records, long[] nMatched);

    /**
     * needle_find_packed16_host: find() with start / end of every row as ONE int, start | end << 16 (0xFFFFFFFF = no match; rows of
     * at most 65 534 chars): the scan kernel stores that form itself, 4 result bytes per row on the device and over PCIe.
     */
    static native int findPacked16Host(long handle, java.nio.ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen,
                                       java.nio.ByteBuffer lengths, long[] bitmap, int[] startEnd);

    /**
     * needle_find_packed8_host (round 6): rows of at most 256 chars, ONE short per row -- start | (end - start) << 8; 0xFFFF = no match,
     * 0xFFFE = the match (0, 256): 2 result bytes per row over PCIe.
     */
    static native int findPacked8Host(long handle, java.nio.ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen,
                                      java.nio.ByteBuffer lengths, long[] bitmap, short[] startLen);

    /** needle_tuning_info: the library's NEEDLE_* environment switches, tab-separated lines (name, default, current, scope, effect). */
    static native String tuningInfo();

    /** needle_trim_scratch: hand the library's free scratch memory beyond keepBytes back to the driver. */
    static native int trimScratch(long keepBytes);

    /**
     * needle_pattern_prefilter_info: whether containedIn (which = 1) / find (which = 2) of this pattern run behind the n-gram
     * candidate filter on batches of 8-bit rows -- info[0] on, [1] kernel mode, [2] stride, [3] run-up, [4] shortest match,
     * [5] windows, [6] bitmap bytes; returns the reason when there is no filter ("" otherwise).
     */
    static native String prefilterInfo(long handle, int which, int[] info);

    /** needle_pattern_set_prefilter: 0 auto (the flood watch decides), 1 on (never suspended), 2 off (the ordinary kernels). */
    static native int setPrefilter(long handle, int mode);

    /**
     * needle_pattern_prefilter_state (which: 1 containedIn, 2 find): state[0..3] = mode, has_filter, suspended calls left, backoff;
     * rate[0] = candidates per KiB of the last evaluation; counts[0..1] = filter launches, suspended calls.
     */
    static native int prefilterState(long handle, int which, int[] state, float[] rate, long[] counts);

    /** needle_pattern_utf16_route: out[0] = the pattern's one page of the BMP (-1: it spans several), out[1] = the substitute byte. */
    static native int utf16Route(long handle, int[] out);

    /** needle_pattern_serialize / needle_pattern_deserialize: the precompiled-pattern blob (Precompile's analogue). */
    static native byte[] serialize(long handle);

    static native int deserialize(byte[] blob, long[] handleOut);

    /**
     * Row sharding over several GPUs of one node from this JVM (needle_multi_*): devices[0] is the root the results are
     * gathered to (one group of RCCL send / receive pairs over xGMI).  scanHostMulti splits a host batch into contiguous
     * row blocks on 64-row boundaries, one per device; op: 0 matches, 1 containedIn, 2 find.
     */
    static native int multiCreate(int[] devices, int flags, long[] handleOut);

    static native void multiDestroy(long multi);

    static native int scanHostMulti(long multi, long pattern, int op, java.nio.ByteBuffer rows, int charWidth, long nRows,
                                    long rowStride, int rowLen, java.nio.ByteBuffer lengths, long[] bitmap, int[] start, int[] end);

    // one Matcher (reference cursor semantics)
    static native int matcherCreate(long pattern, char[] s, long[] handleOut);

    static native void matcherDestroy(long matcher);

    static native int matcherMatches(long matcher, int[] result);

    static native int matcherContainedIn(long matcher, int[] result);

    static native int matcherFind(long matcher, int[] result);

    static native int matcherFindRange(long matcher, int from, int to, int[] result);

    static native int matcherStart(long matcher);

    static native int matcherEnd(long matcher);
}
