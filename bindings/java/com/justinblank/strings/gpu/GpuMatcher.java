package com.justinblank.strings.gpu;

import com.justinblank.strings.Matcher;

/**
 * com.justinblank.strings.Matcher (needle-types/.../Matcher.java:6-26) over one native matcher object, which keeps
 * the reference's cursor fields (nextStart / start / end, DFAClassBuilder.addFields :688-699).  Not thread safe,
 * like the generated matchers.  NOT COMPILED IN THE BUILD CONTAINER (no JDK); shipped as source.
 */
final class GpuMatcher implements Matcher, AutoCloseable {
    private long handle;

    GpuMatcher(long pattern, String s) {
        long[] h = new long[1];
        GpuPattern.check(Native.matcherCreate(pattern, s.toCharArray(), h), null);
        handle = h[0];
    }

    private boolean call(int status, int[] r) {
        GpuPattern.check(status, null);
        return r[0] != 0;
    }

    @Override
    public boolean matches() {
        int[] r = new int[1];
        return call(Native.matcherMatches(handle, r), r);
    }

    @Override
    public boolean containedIn() {
        int[] r = new int[1];
        return call(Native.matcherContainedIn(handle, r), r);
    }

    @Override
    public boolean find() {
        int[] r = new int[1];
        return call(Native.matcherFind(handle, r), r);
    }

    @Override
    public boolean find(int start, int end) {
        int[] r = new int[1];
        return call(Native.matcherFindRange(handle, start, end, r), r);
    }

    @Override
    public int start() {
        return Native.matcherStart(handle);
    }

    @Override
    public int end() {
        return Native.matcherEnd(handle);
    }

    @Override
    public void close() {
        if (handle != 0) {
            Native.matcherDestroy(handle);
            handle = 0;
        }
    }
}
