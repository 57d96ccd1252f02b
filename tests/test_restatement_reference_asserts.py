"""Table-level known answers of the reference's own unit tests, checked on the Python restatement of the compile
pipeline (oracle/needle_compile.py) -- the generator that tests/test_compile_vs_python_restatement.py holds the
product's C++ generator to:
  NFAToDFACompilerTest.java:12-35   raw subset-construction state counts per conversion mode for (AB){1,2}
  DFATest.java:186-250,299-352      char-class partitions (DFA.byteClasses) of BASIC DFAs
CPU only."""
from oracle import needle_compile as nc


def program(regex, flags=0):
    return nc.build_program(nc.Parser(regex, flags).parse(), bool(flags & nc.LEFTMOST_LONGEST))


def basic_dfa(regex, flags=0):
    """DFA.createDFA(regex, BASIC, flags) = NFAToDFACompiler.compile: _compile, pruneDeadStates, minimizeDFA."""
    return nc.minimize(nc.prune_dead(nc.subset_construction(program(regex, flags), nc.BASIC)))


def test_raw_state_counts_per_conversion_mode():
    prog = program("(AB){1,2}")
    assert len(nc.subset_construction(prog, nc.BASIC).accepting) == 7
    assert len(nc.subset_construction(prog, nc.CONTAINED).accepting) == 3
    assert len(nc.subset_construction(prog, nc.SEARCH).accepting) == 6


def classes(regex, flags=0):
    cmap, _count = nc.byte_classes(basic_dfa(regex, flags))
    return cmap


def test_byte_classes_literal():
    c = classes("abc")
    assert all(c[i] == 0 for i in range(ord("a")))
    assert (c[ord("a")], c[ord("b")], c[ord("c")]) == (1, 2, 3)
    assert all(c[i] == 0 for i in range(ord("d"), 65535))


def test_byte_classes_two_disconnected_ranges_followed_by_literal():
    c = classes("[A-Za-z]+ab")
    assert all(c[i] == 0 for i in range(ord("A")))
    assert c[ord("A")] == 1 and c[ord("Z")] == 1 and c[ord("a")] == 2 and c[ord("b")] == 3
    assert all(c[i] == 1 for i in range(ord("c"), ord("z") + 1))
    assert all(c[i] == 0 for i in range(ord("z") + 1, 65535))


def test_byte_classes_with_dot_under_dotall():
    c = classes("[A-Za-z]+.b", nc.DOTALL)
    assert all(c[i] == 1 for i in range(ord("A")))
    assert c[ord("A")] == 2 and c[ord("Z")] == 2 and c[ord("a")] == 2 and c[ord("b")] == 3
    assert all(c[i] == 2 for i in range(ord("c"), ord("z") + 1))
    assert all(c[i] == 1 for i in range(ord("z") + 1, 65535))


def test_byte_classes_url_under_dotall():
    c = classes("http://.+", nc.DOTALL)
    assert all(c[i] == 1 for i in range(ord("/")))
    assert c[ord("/")] == 2
    assert all(c[i] == 1 for i in range(ord("0"), ord(":")))
    assert c[ord(":")] == 3


def test_byte_classes_h_colon_dot_plus():
    c = classes("h:.+")
    assert (c[0], c[ord(":")], c[ord(";")], c[ord("h")], c[ord("i")]) == (1, 2, 1, 3, 1)


def test_byte_classes_holmes_near_watson_style_union():
    c = classes("Hol.{0,2}Wat|Wat.{0,2}Hol")
    want = {"\0": 1, "H": 2, "I": 1, "W": 3, "X": 1, "a": 4, "b": 1, "l": 5, "m": 1, "o": 6, "p": 1, "t": 7}
    for ch, k in want.items():
        assert c[ord(ch)] == k, ch
