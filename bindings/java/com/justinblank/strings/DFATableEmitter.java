package com.justinblank.strings;

import com.justinblank.strings.gpu.GpuPattern;

import java.util.Arrays;

/**
 * Replaces DFAClassBuilder + mako for the GPU path: DFACompiler still parses the regex and builds the four DFAs
 * (DFACompiler.java:48-63); instead of generating a JVM class this flattens them to exactly the data that class
 * would have carried -- BYTE_CLASSES, STATES_MATCHES / CONTAINEDIN / FORWARDS / BACKWARDS, accepting sets, maxChar
 * constants, the fixed-length rule -- and hands them to libneedle_hip.so.
 *
 * Lives in package com.justinblank.strings because DFA, ByteClasses, FindMethodSpec and DFAStateTransitions are
 * package-private.  The call to add in DFACompiler is shown in INTEGRATION.md.
 * NOT COMPILED IN THE BUILD CONTAINER (no JDK); shipped as source.
 */
final class DFATableEmitter {

    private DFATableEmitter() {
    }

    static GpuPatternHolder emit(DFA dfa, DFA containedInDFA, DFA dfaReversed, DFA dfaSearch,
                                 Factorization factorization) {
        // byte classes come from the search DFA and are applied to all four (DFAClassBuilder.java:66-76)
        ByteClasses bc = dfaSearch.byteClasses().orElseThrow(() ->
                new PatternClassCompilationException("no table form: more than 255 char classes", null));
        int stride = effectiveByteClassCount(bc.byteClassCount);
        byte[] classMap = Arrays.copyOf(bc.ranges, 65536);
        DFA[] all = {dfa, containedInDFA, dfaSearch, dfaReversed}; // matches, containedIn, forwards, backwards
        int[] nStates = new int[4];
        int[] maxChar = new int[4];
        short[][] tables = new short[4][];
        byte[][] accepting = new byte[4][];
        for (int i = 0; i < 4; i++) {
            DFA d = all[i];
            nStates[i] = d.statesCount();
            maxChar[i] = d.maxChar();
            short[] t = new short[nStates[i] * stride];
            Arrays.fill(t, (short) -1); // populateByteClassArrays, DFAClassBuilder.java:317-333
            byte[] acc = new byte[nStates[i]];
            for (DFA state : d.allStates()) {
                int s = state.getStateNumber();
                acc[s] = (byte) (state.isAccepting() ? 1 : 0);
                boolean[] seen = new boolean[256];
                for (var tr : state.getTransitions()) { // DFAStateTransitions.buildByteClassString :30-62
                    for (int c = tr.getLeft().getStart(); c <= tr.getLeft().getEnd(); c++) {
                        int k = bc.ranges[c] & 0xFF;
                        if (!seen[k]) {
                            seen[k] = true;
                            t[s * stride + k] = (short) tr.getRight().getStateNumber();
                        }
                    }
                }
            }
            tables[i] = t;
            accepting[i] = acc;
        }
        int fixedLen = factorization.canOnlyHaveOneLength() ? factorization.getMinLength() : -1;
        return new GpuPatternHolder(GpuPattern.fromTables(classMap, stride, nStates, maxChar, tables,
                accepting, fixedLen));
    }

    /** DFAClassBuilder.getEffectiveByteClassCount :240-253 */
    static int effectiveByteClassCount(int c) {
        if (c > CompilationPolicy.THRESHOLD_TO_ROUND_UP_ALL_BYTECLASSES || c < 3) {
            return c;
        } else if (c < 4) {
            return 4;
        } else if (c < 8) {
            return 8;
        } else if (c < 16) {
            return 16;
        }
        return c;
    }

    /** Thin holder so that DFACompiler can return a Pattern. */
    static final class GpuPatternHolder {
        final GpuPattern pattern;

        GpuPatternHolder(GpuPattern pattern) {
            this.pattern = pattern;
        }
    }
}
