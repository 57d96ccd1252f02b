package com.justinblank.strings.gpu;

import com.justinblank.strings.Matcher;
import com.justinblank.strings.Pattern;
import com.justinblank.strings.PatternClassCompilationException;
import com.justinblank.strings.PatternSyntaxException;

import java.nio.ByteBuffer;

/**
 * Drop-in implementation of com.justinblank.strings.Pattern (needle-types/.../Pattern.java:3-34) whose matchers
 * run on the MI355X kernels, plus the batch entry points the reference lacks.  Immutable and shareable like the
 * reference's generated Pattern classes (DFACompiler.createPatternClass, DFACompiler.java:96-111).
 *
 * NOT COMPILED IN THE BUILD CONTAINER (no JDK); shipped as source.
 */
public final class GpuPattern implements Pattern, AutoCloseable {
    private long handle;

    private GpuPattern(long handle) {
        this.handle = handle;
    }

    /** The library builds the automata itself (needle_compile). */
    public static GpuPattern compile(String regex, int flags) {
        if ((flags & ~ALL_FLAGS) != 0) {
            throw new IllegalArgumentException("Unknown flag bits"); // CompilerOptions.java:9-16
        }
        long[] h = new long[1];
        check(Native.compile(regex.toCharArray(), flags, h), regex);
        return new GpuPattern(h[0]);
    }

    /** needle's own DFACompiler built the automata; DFATableEmitter flattened them (see DFATableEmitter). */
    public static GpuPattern fromTables(byte[] classMap, int stride, int[] nStates, int[] maxChar, short[][] tables,
                                 byte[][] accepting, int fixedLen) {
        long[] h = new long[1];
        check(Native.fromTables(classMap, stride, nStates, maxChar, tables, accepting, fixedLen, h), null);
        return new GpuPattern(h[0]);
    }

    /** The precompiled-pattern blob -- what Precompile writes as a .class file in the reference (Precompile.java:30-53). */
    public byte[] toBytes() {
        byte[] blob = Native.serialize(handle);
        if (blob == null) {
            throw new IllegalStateException(Native.lastError());
        }
        return blob;
    }

    public static GpuPattern fromBytes(byte[] blob) {
        long[] h = new long[1];
        check(Native.deserialize(blob, h), null);
        return new GpuPattern(h[0]);
    }

    /**
     * A single haystack: one 1-row batch on the GPU per call (tens of microseconds of launch latency) -- correct, and
     * orders of magnitude slower than the reference's generated class.  Single strings belong on the JVM path; the GPU
     * earns its keep on the *Batch entry points.
     */
    @Override
    public Matcher matcher(String s) {
        return new GpuMatcher(handle, s);
    }

    /**
     * op 0 matches, 1 containedIn, 2 find over a host batch split across several GPUs of the node (row blocks on 64-row
     * boundaries; the results are gathered to devices[0] over xGMI and downloaded).  The handle is reusable.
     */
    public long[] scanBatchMulti(GpuDevices devices, int op, ByteBuffer rows, int charWidth, long nRows, long rowStride,
                                 int rowLen, ByteBuffer lengths, int[] start, int[] end) {
        long[] bitmap = new long[(int) ((nRows + 63) / 64)];
        check(Native.scanHostMulti(devices.handle(), handle, op, rows, charWidth, nRows, rowStride, rowLen, lengths, bitmap,
                start, end), null);
        return bitmap;
    }

    /** bit (r &amp; 63) of word (r &gt;&gt; 6) = containedIn() of row r. */
    public long[] containedInBatch(ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen,
                                   ByteBuffer lengths) {
        long[] bitmap = new long[(int) ((nRows + 63) / 64)];
        check(Native.containedInHost(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, bitmap), null);
        return bitmap;
    }

    public long[] matchesBatch(ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen,
                               ByteBuffer lengths) {
        long[] bitmap = new long[(int) ((nRows + 63) / 64)];
        check(Native.matchesHost(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, bitmap), null);
        return bitmap;
    }

    /** start/end receive the first find() of every row; unmatched rows get -1/-1. */
    public long[] findBatch(ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen,
                            ByteBuffer lengths, int[] start, int[] end) {
        long[] bitmap = new long[(int) ((nRows + 63) / 64)];
        check(Native.findHost(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, bitmap, start, end), null);
        return bitmap;
    }

    /**
     * Every non-overlapping match of every row, as repeated {@code matcher.find()} calls would report them: up to
     * {@code maxPerRow} per row.  counts[r] matches of row r are at start/end[r * maxPerRow + k].  Returns true when
     * some row had more matches than slots.
     */
    public boolean findAllBatch(ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen, ByteBuffer lengths,
                                int maxPerRow, int[] counts, int[] start, int[] end) {
        int[] more = new int[1];
        check(Native.findAllHost(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, maxPerRow, counts, start, end, more), null);
        return more[0] != 0;
    }

    /**
     * {@link #findAllBatch} with each match as one int, {@code start | end << 16} (rows of at most 65 535 chars): half the result
     * bytes on the device and over PCIe (needle_find_all_packed16_host).
     */
    public boolean findAllBatchPacked(ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen, ByteBuffer lengths,
                                      int maxPerRow, int[] counts, int[] startEnd) {
        int[] more = new int[1];
        check(Native.findAllPacked16Host(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, maxPerRow, counts, startEnd, more), null);
        return more[0] != 0;
    }

    /** Result of {@link #findAllCompact}: row r's matches are start/end[offsets[r] .. offsets[r + 1]). */
    public static final class Matches {
        public final long[] offsets;
        public final int[] start;
        public final int[] end;

        Matches(long[] offsets, int[] start, int[] end) {
            this.offsets = offsets;
            this.start = start;
            this.end = end;
        }
    }

    /**
     * Every non-overlapping match of every row in compact form (no per-row limit): a counting pass sizes the result,
     * a second call files it.
     */
    public Matches findAllCompact(ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen, ByteBuffer lengths) {
        long[] offsets = new long[(int) nRows + 1];
        long[] total = new long[1];
        check(Native.findAllCsrHost(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, offsets, null, null, total), null);
        int[] start = new int[(int) total[0]];
        int[] end = new int[(int) total[0]];
        if (total[0] > 0) {
            check(Native.findAllCsrHost(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, offsets, start, end, total), null);
        }
        return new Matches(offsets, start, end);
    }

    /** Result of {@link #findCompact}: the matched rows in row order; match k is row[k], [start[k], end[k]). */
    public static final class MatchedRows {
        public final long[] bitmap;
        public final int[] row;
        public final int[] start;
        public final int[] end;

        MatchedRows(long[] bitmap, int[] row, int[] start, int[] end) {
            this.bitmap = bitmap;
            this.row = row;
            this.start = start;
            this.end = end;
        }
    }

    /**
     * find() of every row, reporting the MATCHED rows only (what Matcher.find() + start() + end() give per haystack,
     * DFAClassBuilder.java:625-667): 1 bit per row + 8 bytes per matched row come back from the device instead of 8 bytes per
     * row.  Rows of at most 65 534 chars.
     */
    public MatchedRows findCompact(ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen, ByteBuffer lengths) {
        long[] bitmap = new long[(int) ((nRows + 63) / 64)];
        long[] n = new long[1];
        int[] rec = new int[2 * (int) nRows];
        check(Native.findCompactHost(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, bitmap, rec, n), null);
        int m = (int) n[0];
        int[] row = new int[m], start = new int[m], end = new int[m];
        for (int k = 0; k < m; k++) {
            row[k] = rec[2 * k];
            start[k] = rec[2 * k + 1] & 0xFFFF;
            end[k] = rec[2 * k + 1] >>> 16;
        }
        return new MatchedRows(bitmap, row, start, end);
    }

    /**
     * find() with start() / end() of every row as ONE int, {@code start | end << 16} ({@code -1}: no match; rows of at most 65 534
     * chars): needle_find_packed16_host -- the scan kernel stores that form itself.  Returns the match bitmap.
     */
    public long[] findBatchPacked(ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen, ByteBuffer lengths, int[] startEnd) {
        long[] bitmap = new long[(int) ((nRows + 63) / 64)];
        check(Native.findPacked16Host(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, bitmap, startEnd), null);
        return bitmap;
    }

    /**
     * find() on rows of at most 256 chars with every row's result as ONE short (needle_find_packed8_host): decode with
     * {@link #start8(short)} / {@link #end8(short)}.  Returns the match bitmap.
     */
    public long[] findBatchPacked8(ByteBuffer rows, int charWidth, long nRows, long rowStride, int rowLen, ByteBuffer lengths, short[] startLen) {
        long[] bitmap = new long[(int) ((nRows + 63) / 64)];
        check(Native.findPacked8Host(handle, rows, charWidth, nRows, rowStride, rowLen, lengths, bitmap, startLen), null);
        return bitmap;
    }

    /** start() of a findBatchPacked8 entry (-1: no match): Matcher.start(), DFAClassBuilder.java:660-667. */
    public static int start8(short e) {
        int x = e & 0xFFFF;
        return x == 0xFFFF ? -1 : x == 0xFFFE ? 0 : (x & 0xFF);
    }

    /** end() of a findBatchPacked8 entry (-1: no match). */
    public static int end8(short e) {
        int x = e & 0xFFFF;
        return x == 0xFFFF ? -1 : x == 0xFFFE ? 256 : (x & 0xFF) + (x >> 8);
    }

    /**
     * Whether find() of this pattern runs behind the n-gram candidate filter on batches of 8-bit rows (the table-level form of the
     * reference's prefix / first-byte narrowing, DFAClassBuilder.java:365-376, :420-426); the reason when it does not.
     */
    public String prefilterWhyNot() {
        int[] info = new int[7];
        String why = Native.prefilterInfo(handle, 2, info);
        return info[0] != 0 ? "" : why;
    }

    public static final int PREFILTER_AUTO = 0, PREFILTER_ON = 1, PREFILTER_OFF = 2;

    /**
     * Pins the n-gram filter's flood watch (needle_pattern_set_prefilter): ON = the filter kernel whenever the batch shape allows it,
     * never suspended; OFF = the ordinary scan kernels; AUTO = the watch decides (default).  Answers are the same in all three; pin
     * the mode for work captured into a HIP graph.
     */
    public void setPrefilter(int mode) {
        check(Native.setPrefilter(handle, mode), null);
    }

    /** Calls of find() batches that will still take the ordinary kernel because the last evaluated text flooded the filter (0: none). */
    public int prefilterSuspendedCallsLeft() {
        int[] st = new int[4];
        check(Native.prefilterState(handle, 2, st, new float[1], new long[2]), null);
        return st[2];
    }

    /**
     * The page of the BMP this pattern lives on (0: ASCII / Latin-1, 4: Cyrillic, ...), or -1 when it spans several: String batches
     * (UTF-16) of one-page dictionaries run behind the n-gram filter of that page's byte program (needle_pattern_utf16_route).
     */
    public int utf16Page() {
        int[] out = new int[2];
        check(Native.utf16Route(handle, out), null);
        return out[0];
    }

    /** The library's NEEDLE_* environment switches (needle_tuning_info). */
    public static String tuningInfo() {
        return Native.tuningInfo();
    }

    /**
     * find() over an array of haystacks: the strings are flattened to one char buffer + offsets (no per-string
     * Matcher objects, SURVEY.md s8 a9) and cross the boundary once.  Returns the match bitmap.
     */
    public long[] findBatch(String[] haystacks, int[] start, int[] end) {
        long[] offsets = new long[haystacks.length + 1];
        for (int i = 0; i < haystacks.length; i++) {
            offsets[i + 1] = offsets[i] + haystacks[i].length();
        }
        char[] data = new char[(int) offsets[haystacks.length]];
        for (int i = 0; i < haystacks.length; i++) {
            haystacks[i].getChars(0, haystacks[i].length(), data, (int) offsets[i]);
        }
        long[] bitmap = new long[(haystacks.length + 63) / 64];
        check(Native.packedHost(handle, 2, data, offsets, bitmap, start, end), null);
        return bitmap;
    }

    @Override
    public void close() {
        if (handle != 0) {
            Native.destroyPattern(handle);
            handle = 0;
        }
    }

    /** Status code -> the exception the reference would have thrown at the same point. */
    static void check(int status, String regex) {
        switch (status) {
            case 0:
                return;
            case 1:
                throw new IllegalArgumentException(Native.lastError());
            case 2:
                throw new PatternSyntaxException(Native.lastError()); // RegexParser.java:86-98
            case 3:
            case 4:
                throw new PatternClassCompilationException(
                        "Failed to compile pattern from regex '" + regex + "': " + Native.lastError(), null);
            default:
                throw new IllegalStateException(Native.lastError()); // device errors
        }
    }
}
