"""TEST INFRASTRUCTURE -- a second, independent restatement of the reference's table PRODUCERS.

Regex -> AST -> Thompson program with priorities -> subset construction (BASIC / CONTAINED_IN / DFA_SEARCH) ->
dead-state pruning -> minimisation -> char-class partition -> the four tables a generated class holds, written in
plain Python straight from the reference sources, so that the product's C++ generator (needle_amd/csrc/
needle_regex.cpp) can be compared against it on arbitrary regexes (tests/test_compile_vs_python_restatement.py) --
tables, state numbering included.  Slow by design (pure-Python loops): cold path, small cases.

Reference files followed (needle-compiler/src/main/java/com/justinblank/strings/):
  RegexParser.java:100-275,297-370,372-529,586-760   RegexAST/{Union,Concatenation,LiteralNode,...}.java
  RegexInstrBuilder.java:28-209   NFA.java:225-266   NFAToDFACompiler.java:34-178   StateSet.java:16-53
  CharRange.java:81-157   DFA.java:63-83,384-398,438-566,745-792   MinimizeDFA.java:18-217
  DFAStateTransitions.java:30-62   DFAClassBuilder.java:240-253   DFACompiler.java:45-83

java.util.HashSet<Integer> iteration order is observable (NFAToDFACompiler.getEpsilonClosure keeps the FIRST
priority among equal distances), so JavaIntHashSet below models java.util.HashMap's table literally: power-of-two
bucket array, per-bucket insertion order, resize at 0.75 load with order-preserving splits (tree bins never form for
these keys)."""

DOTALL, CASE_INSENSITIVE, UNICODE_CASE, UNICODE_CHARACTER_CLASS, LEFTMOST_LONGEST = 0x20, 0x02, 0x40, 0x100, 0x800000


class PatternSyntaxError(Exception):
    pass


class Unsupported(Exception):
    pass


# ---------------------------------------------------------------------------------------------- java.util.HashSet<Integer>
class JavaIntHashSet:
    def __init__(self):
        self.table = None
        self.size = 0

    def _index(self, k, n):
        h = k ^ (k >> 16)
        return h & (n - 1)

    def __contains__(self, k):
        return self.table is not None and k in self.table[self._index(k, len(self.table))]

    def add(self, k):
        if self.table is None:
            self.table = [[] for _ in range(16)]
        b = self.table[self._index(k, len(self.table))]
        if k in b:
            return False
        b.append(k)
        self.size += 1
        if self.size > len(self.table) * 3 // 4:
            old = self.table
            self.table = [[] for _ in range(len(old) * 2)]
            for bucket in old:  # split preserves relative order
                for x in bucket:
                    self.table[self._index(x, len(self.table))].append(x)
        return True

    def remove(self, k):
        b = self.table[self._index(k, len(self.table))]
        b.remove(k)
        self.size -= 1

    def __iter__(self):
        if self.table is None:
            return iter(())
        return iter([x for bucket in self.table for x in bucket])

    def __len__(self):
        return self.size


# ---------------------------------------------------------------------------------------------- AST
class Node:
    pass


class Lit(Node):
    def __init__(self, s):
        self.s = s  # mutable: the parser appends in place


class Rng(Node):
    def __init__(self, a, b):
        if a > b:
            raise PatternSyntaxError("range start > end")
        self.a, self.b = a, b


class Cat(Node):
    def __init__(self, head, tail):
        self.head, self.tail = head, tail


class Alt(Node):
    def __init__(self, left, right, prio):
        self.left, self.right, self.prio = left, right, prio


class Star(Node):
    def __init__(self, node):
        self.node = node


class Counted(Node):
    def __init__(self, node, lo, hi):
        if lo > hi:
            raise PatternSyntaxError("bad repetition range")
        self.node, self.lo, self.hi = node, lo, hi


class LParen(Node):
    pass


def node_eq(x, y):
    if x is None:
        raise PatternSyntaxError("null union branch")
    if y is None:
        return False
    if isinstance(x, Lit):
        return isinstance(y, Lit) and x.s == y.s
    return x is y


def union(l, r, prio):  # Union.of
    if node_eq(l, r):
        return l
    if isinstance(l, Alt) and (node_eq(l.left, r) or node_eq(l.right, r)):
        return l
    if isinstance(r, Alt) and (node_eq(r.left, l) or node_eq(r.right, l)):
        return r
    return Alt(l, r, prio)


def concat_static(h, t):  # Concatenation.concatenate
    if isinstance(h, Lit) and isinstance(t, Lit):
        h.s += t.s
        return h
    if isinstance(h, Lit) and isinstance(t, Cat) and isinstance(t.head, Lit):
        h.s += t.head.s
        return Cat(h, t.tail)
    return Cat(h, t)


def of_chars(cs):
    cs = sorted(cs)
    if len(cs) == 1:
        return Rng(cs[0], cs[0])
    u, start = None, 0
    for i in range(len(cs)):
        if i == len(cs) - 1 or cs[i] + 1 != cs[i + 1]:
            r = Rng(cs[start], cs[i])
            u = r if u is None else union(u, r, False)
            start = i + 1
    return u


def complement(ranges):
    ranges = sorted(ranges)
    out, last = [], None
    for a, b in ranges:
        if last is None:
            out.append(Rng(0, (a - 1) & 0xFFFF))
        else:
            lo, hi = (last[1] + 1) & 0xFFFF, (a - 1) & 0xFFFF
            if lo <= hi:
                out.append(Rng(lo, hi))
        last = (a, b)
    out.append(Rng((last[1] + 1) & 0xFFFF, 0xFFFF))
    u = union(out[0], out[1], False)
    for n in out[2:]:
        u = union(u, n, False)
    return u


def no_open_union(n):
    """'a|' leaves Union(a, null) on the stack; the reference dereferences the null (NullPointerException out of
    DFACompiler.compile).  Reported here as "unsupported", like the product does."""
    if n is None:
        raise Unsupported("union with an empty branch")
    if isinstance(n, Cat):
        no_open_union(n.head), no_open_union(n.tail)
    elif isinstance(n, Alt):
        no_open_union(n.left), no_open_union(n.right)
    elif isinstance(n, (Star, Counted)):
        no_open_union(n.node)


def min_len(n):
    if isinstance(n, Lit):
        return len(n.s)
    if isinstance(n, Rng):
        return 1
    if isinstance(n, Cat):
        return min_len(n.head) + min_len(n.tail)
    if isinstance(n, Alt):
        return min(min_len(n.left), min_len(n.right))
    if isinstance(n, Star):
        return 0
    return n.lo * min_len(n.node)


def max_len(n):
    if isinstance(n, Lit):
        return len(n.s)
    if isinstance(n, Rng):
        return 1
    if isinstance(n, Cat):
        a, b = max_len(n.head), max_len(n.tail)
        return None if a is None or b is None else a + b
    if isinstance(n, Alt):
        a, b = max_len(n.left), max_len(n.right)
        return None if a is None or b is None else max(a, b)
    if isinstance(n, Star):
        return None
    a = max_len(n.node)
    return None if a is None else a * n.hi


def reverse(n):
    if isinstance(n, Lit):
        return Lit(n.s[::-1])
    if isinstance(n, Rng):
        return n
    if isinstance(n, Cat):
        return Cat(reverse(n.tail), reverse(n.head))
    if isinstance(n, Alt):
        return union(reverse(n.left), reverse(n.right), False)
    if isinstance(n, Star):
        return Star(reverse(n.node))
    return Counted(reverse(n.node), n.lo, n.hi)


HSPACE = [0x20, 0x09, 0xA0, 0x1680, 0x180E] + list(range(0x2000, 0x200B)) + [0x202F, 0x205F, 0x3000]
VSPACE = [0x0A, 0x0B, 0x0C, 0x0D, 0x85, 0x2028, 0x2029]
SPACE = [0x20, 0x09, 0x0A, 0x0B, 0x0C, 0x0D]


class Parser:
    def __init__(self, regex, flags):
        self.re = regex
        self.i = 0
        self.dotall = bool(flags & DOTALL)
        self.ci = bool(flags & CASE_INSENSITIVE)
        self.ucc = bool(flags & UNICODE_CHARACTER_CLASS)
        self.uci = True if self.ucc else bool(flags & UNICODE_CASE)
        self.stack = []

    def err(self, m):
        return PatternSyntaxError(m)

    def take(self):
        if self.i >= len(self.re):
            raise self.err("unexpected end")
        c = self.re[self.i]
        self.i += 1
        return c

    def peek(self, c):
        return self.i < len(self.re) and self.re[self.i] == c

    def pop(self):
        if not self.stack:
            raise self.err("empty stack")
        return self.stack.pop()

    def concat(self, nxt, node):  # RegexParser.concatenate
        if isinstance(nxt, Lit) and isinstance(node, Lit):
            nxt.s += node.s
            return nxt
        if isinstance(nxt, LParen) or isinstance(node, LParen):
            raise self.err("paren")
        return concat_static(nxt, node)

    def no_lazy(self):
        if self.peek("?") or self.peek("+"):
            raise self.err("reluctant / possessive quantifiers are not supported")

    def case_variants(self, c):
        o = ord(c)
        if o >= 128:
            if o in (0x212A, 0x17F, 0x130, 0x131) or o > 0xBF:
                raise Unsupported("UNICODE_CASE folding of non-ASCII characters")
            return [o]
        lo = o + 32 if 65 <= o <= 90 else o
        if not 97 <= lo <= 122:
            return [o]
        out = [lo - 32, lo]
        if lo == ord("k"):
            out.append(0x212A)
        if lo == ord("s"):
            out.append(0x17F)
        if lo == ord("i"):
            out += [0x130, 0x131]
        return out

    def parse(self):
        st = self.stack
        while self.i < len(self.re):
            c = self.take()
            if c == ".":
                st.append(Rng(0, 0xFFFF) if self.dotall else union(Rng(0, 9), union(Rng(0xB, 0xC), Rng(0xE, 0xFFFF), False), False))
            elif c in "^$":
                raise self.err("anchors are not supported")
            elif c == "(":
                st.append(LParen())
                if self.re.startswith("?:", self.i):
                    self.i += 2
                elif self.re.startswith("?<", self.i):
                    g = self.i + 2
                    while g < len(self.re):
                        ch = self.re[g]
                        if ch.isascii() and ch.isalnum():
                            g += 1
                        elif ch == ">":
                            self.i = g + 1
                            break
                        else:
                            raise self.err("group name")
            elif c == "{":
                if not st:
                    raise self.err("'{' with nothing before it")
                lo = self.number()
                nx = self.take()
                if nx == "}":
                    st.append(Counted(self.pop(), lo, lo))
                    self.no_lazy()
                    continue
                if nx != ",":
                    raise self.err("expected ','")
                hi = self.number()
                st.append(Counted(self.pop(), lo, hi))
                if self.take() != "}":
                    raise self.err("unclosed brackets")
                self.no_lazy()
            elif c == "?":
                if not st:
                    raise self.err("'?' with nothing before it")
                self.no_lazy()
                st.append(Counted(self.pop(), 0, 1))
            elif c == "[":
                n = self.char_set()
                if n is not None:
                    st.append(n)
            elif c == "+":
                if not st:
                    raise self.err("'+' with nothing before it")
                self.no_lazy()
                last = self.pop()
                st.append(self.concat(last, Star(last)))
            elif c == "*":
                if not st:
                    raise self.err("'*' with nothing before it")
                self.no_lazy()
                st.append(Star(self.pop()))
            elif c == "|":
                if not st:
                    raise self.err("'|' with nothing before it")
                self.collapse_literals()
                st.append(union(self.pop(), None, True))
            elif c == "\\":
                st.append(self.escape())
            elif c == ")":
                self.collapse_paren()
            else:
                self.literal_char(c)
        if not st:
            return Lit("")
        node = self.pop()
        if isinstance(node, LParen):
            raise self.err("unbalanced '('")
        while st:
            nxt = self.pop()
            if isinstance(nxt, Alt) and nxt.right is None:
                node = union(nxt.left, node, True)
            elif isinstance(nxt, Lit) and isinstance(node, Lit):
                node = Lit(nxt.s + node.s)
            elif isinstance(nxt, LParen):
                raise self.err("unbalanced '('")
            else:
                node = self.concat(nxt, node)
        return node

    def literal_char(self, c):
        st = self.stack
        if not self.ci:
            st.append(Lit(c))
        elif self.uci:
            vs = self.case_variants(c)
            if len(vs) > 1:
                u = None
                for v in vs:
                    u = Lit(chr(v)) if u is None else union(u, Lit(chr(v)), False)
                st.append(u)
            else:
                st.append(Lit(c))
        elif "A" <= c <= "Z":
            st.append(union(Lit(c), Lit(chr(ord(c) + 32)), False))
        elif "a" <= c <= "z":
            st.append(union(Lit(c), Lit(chr(ord(c) - 32)), False))
        else:
            st.append(Lit(c))

    def number(self):
        s = self.i
        while self.i < len(self.re):
            if not "0" <= self.re[self.i] <= "9":
                if self.i == s or self.i - s > 9:
                    raise self.err("expected number")
                return int(self.re[s:self.i])
            self.i += 1
        raise self.err("expected number")

    def collapse_literals(self):
        st = self.stack
        last = self.pop()
        while st:
            prev = st[-1]
            if isinstance(prev, LParen):
                break
            st.pop()
            if isinstance(prev, Alt):
                last = union(prev.left, last, True) if prev.right is None else concat_static(prev, last)
            else:
                last = self.concat(prev, last)
        st.append(last)

    def collapse_paren(self):
        st = self.stack
        if not st:
            raise self.err("unbalanced ')'")
        node = None
        while True:
            if not st:
                raise self.err("unbalanced ')'")
            if isinstance(st[-1], LParen):
                break
            prev = st.pop()
            if node is None:
                node = prev
            elif isinstance(prev, Alt):
                if prev.left is not None and prev.right is not None:
                    node = concat_static(prev, node)
                    continue
                if not st:
                    raise self.err("'|' with nothing before it")
                if isinstance(st[-1], LParen):
                    st.pop()
                    st.append(union(prev.left, node, True))
                    return
                node = union(prev.left, node, True)
            else:
                node = self.concat(prev, node)
            if not st:
                raise self.err("unbalanced ')'")
        st.pop()
        st.append(Lit("") if node is None else node)

    def escape(self):
        if self.i >= len(self.re):
            raise self.err("dangling backslash")
        c = self.take()
        if c == "a":
            return Rng(7, 7)
        if c in "ABbcGpZz":
            raise self.err("escape not supported")
        ucc = self.ucc
        if c == "d":
            if ucc:
                raise Unsupported("\\d under UNICODE_CHARACTER_CLASS")
            return Rng(48, 57)
        if c == "D":
            if ucc:
                raise Unsupported("\\D under UNICODE_CHARACTER_CLASS")
            return complement([(48, 57)])
        if c == "e":
            return Rng(0x1B, 0x1B)
        if c == "f":
            return Rng(0xC, 0xC)
        if c == "H":
            return complement([(x, x) for x in HSPACE])
        if c == "h":
            return of_chars(HSPACE)
        if c == "n":
            return Rng(10, 10)
        if c == "r":
            return Rng(13, 13)
        if c == "s":
            if ucc:
                raise Unsupported("\\s under UNICODE_CHARACTER_CLASS")
            return of_chars(SPACE)
        if c == "S":
            if ucc:
                raise Unsupported("\\S under UNICODE_CHARACTER_CLASS")
            return complement([(x, x) for x in SPACE])
        if c == "t":
            return Rng(9, 9)
        if c == "w":
            if ucc:
                raise Unsupported("\\w under UNICODE_CHARACTER_CLASS")
            return union(Rng(48, 57), union(Rng(95, 95), union(Rng(97, 122), Rng(65, 90), False), False), False)
        if c == "W":
            if ucc:
                raise Unsupported("\\W under UNICODE_CHARACTER_CLASS")
            return complement([(48, 57), (95, 95), (97, 122), (65, 90)])
        if c == "x":
            digs = ""
            while len(digs) < 2 and self.i < len(self.re) and self.re[self.i] in "0123456789ABCDEF":
                digs += self.take()
            if len(digs) != 2:
                raise self.err("wrong number of hex chars")
            return Rng(int(digs, 16), int(digs, 16))
        if c == "V":
            return complement([(x, x) for x in VSPACE])
        if c == "v":
            return of_chars(VSPACE)
        if c == "0":
            digs = ""
            while len(digs) < 3 and self.i < len(self.re) and "0" <= self.re[self.i] <= "7":
                if len(digs) == 2 and digs[0] > "3":
                    break
                digs += self.take()
            if not digs:
                raise self.err("illegal octal escape")
            return Rng(int(digs, 8), int(digs, 8))
        if c in "\\[|()$*?+{:^.":
            return Rng(ord(c), ord(c))
        if "1" <= c <= "9":
            raise self.err("backreferences are not supported")
        if c < "A" or "Z" < c < "a" or c > "z":
            return Rng(ord(c), ord(c))
        raise self.err("unrecognized escape")

    @staticmethod
    def with_alternate(node, alt):
        if node is not None:
            return union(alt, node, False) if alt is not None else node
        return alt

    def build_set_node(self, ranges, neg):
        if not ranges:
            return None
        if len(ranges) == 1:
            return complement([ranges[0]]) if neg else Rng(*ranges[0])
        rs = sorted(ranges)
        merged, cur = [], rs[0]
        for r in rs[1:]:  # CharRange.compact: only exactly adjacent ranges merge
            if (cur[1] + 1) & 0xFFFF == r[0]:
                cur = (cur[0], r[1])
            else:
                merged.append(cur)
                cur = r
        merged.append(cur)
        if len(merged) == 1:
            return complement([merged[0]]) if neg else Rng(*merged[0])
        if neg:
            return complement(merged)
        n = union(Rng(*merged[0]), Rng(*merged[1]), False)
        for r in merged[2:]:
            n = union(n, Rng(*r), False)
        return n

    def char_set(self):
        ranges, last, start, alt, neg = [], None, self.i, None, False

        def add(r):
            if r not in ranges:
                ranges.append(r)

        while self.i < len(self.re):
            c = self.take()
            if c == "^" and self.i == start + 1:
                neg = True
            elif c == "]":
                if last is not None:
                    add((last, last))
                return self.with_alternate(self.build_set_node(ranges, neg), alt)
            elif c == "-":
                if self.i == len(self.re):
                    raise self.err("unterminated character range")
                if self.peek("]"):
                    if last is not None:
                        add((last, last))
                    last = ord("-")
                    continue
                if last is None:
                    last = ord(c)
                    continue
                nx = self.take()
                if nx == "\\" and (self.peek("[") or self.peek("]") or self.peek("\\")):
                    nx = self.take()
                nx = ord(nx)
                if nx < last:
                    raise self.err("range start must be <= end")
                if self.ci:
                    if self.uci:
                        vs = set()
                        for rc in range(last, nx + 1):
                            vs.update(self.case_variants(chr(rc)))
                        for v in sorted(vs):
                            add((v, v))
                    elif nx < 65 or 122 < last:
                        add((last, nx))
                    else:
                        us, ue = max(last, 65), min(nx, 90)
                        ls, le = max(last, 97), min(nx, 122)
                        if us <= ue:
                            add((us, ue))
                            add((us + 32, ue + 32))
                        if ls <= le:
                            add((ls, le))
                            add((ls - 32, le - 32))
                        add((last, nx))
                else:
                    add((last, nx))
                last = None
            elif c == "[":
                if ranges:
                    rn = None
                    for r in sorted(ranges):
                        q = Rng(*r)
                        rn = q if rn is None else union(rn, q, False)
                    ranges.clear()
                    alt = self.with_alternate(rn, alt)
                inner = self.char_set()
                if inner is None:
                    raise self.err("unbalanced [")
                if self.peek("]"):
                    self.take()
                    return self.with_alternate(inner, alt)
                alt = self.with_alternate(inner, alt)
            elif c == "\\":
                if self.peek("[") or self.peek("]") or self.peek("\\"):
                    n = ord(self.take())
                    add((n, n))
                    last = n
                else:
                    alt = self.escape()
            else:
                if last is not None:
                    add((last, last))
                last = ord(c)
        raise self.err("unmatched [")


# ---------------------------------------------------------------------------------------------- Thompson program
CHAR, JUMP, SPLIT, MATCH = 0, 1, 2, 3


class Instr:
    __slots__ = ("op", "a", "b", "target", "targets", "prio")

    def __init__(self, op, a=0, b=0, target=-1, targets=(), prio=0):
        self.op, self.a, self.b, self.target, self.targets, self.prio = op, a, b, target, list(targets), max(prio, 0)


def build_program(ast, leftmost_longest):
    prog = []
    state = {"maxp": 1}

    def emit(n):
        if isinstance(n, Cat):
            emit(n.head)
            emit(n.tail)
        elif isinstance(n, Star):
            si = len(prog)
            prog.append(None)
            emit(n.node)
            prog.append(Instr(JUMP, target=si, prio=state["maxp"]))
            if not leftmost_longest:
                state["maxp"] += 1
            prog[si] = Instr(SPLIT, targets=[si + 1, len(prog)], prio=state["maxp"])
        elif isinstance(n, Counted):
            for _ in range(n.lo):
                emit(n.node)
            sw = []
            for _ in range(n.lo, n.hi):
                sw.append(len(prog))
                prog.append(None)
                emit(n.node)
            fin = len(prog)
            for s in sw:
                prog[s] = Instr(SPLIT, targets=[s + 1, fin], prio=state["maxp"])
        elif isinstance(n, Alt):
            if n.right is None:
                raise Unsupported("union with an empty branch")
            si = len(prog)
            if not isinstance(n.left, Alt):
                prog.append(None)
            first_t = len(prog)
            first_p = state["maxp"]
            emit(n.left)
            if not leftmost_longest and n.prio:
                state["maxp"] += 1
            fj = len(prog)
            prog.append(None)
            second_t = len(prog)
            emit(n.right)
            if not leftmost_longest and n.prio:
                state["maxp"] += 1
            prog[fj] = Instr(JUMP, target=len(prog), prio=first_p)
            ts = []
            if prog[first_t].op == SPLIT:
                ts += prog[first_t].targets
            else:
                ts.append(first_t)
            if second_t < len(prog) and prog[second_t].op == SPLIT:
                ts += prog[second_t].targets
            else:
                ts.append(second_t)
            prog[si] = Instr(SPLIT, targets=ts, prio=first_p)
        elif isinstance(n, Rng):
            prog.append(Instr(CHAR, n.a, n.b, prio=state["maxp"]))
        elif isinstance(n, Lit):
            for ch in n.s:
                prog.append(Instr(CHAR, ord(ch), ord(ch), prio=state["maxp"]))
        else:
            raise Unsupported("unexpected node")

    emit(ast)
    mi = len(prog)
    mp = 2 ** 31 - 1
    for ins in prog:
        if ins.op == JUMP and ins.target == mi:
            mp = min(mp, ins.prio)
        elif ins.op == SPLIT and mi in ins.targets:
            mp = min(mp, ins.prio)
    prog.append(Instr(MATCH, prio=mp))

    def resolve(j):
        r, t = -1, j
        while prog[t].op == JUMP:
            r = prog[t].target
            t = r
        return r

    for ins in prog:
        if ins.op == JUMP:
            r = resolve(ins.target)
            if r != -1:
                ins.target = r
        elif ins.op == SPLIT:
            ins.targets = [t if resolve(t) == -1 else resolve(t) for t in ins.targets]
    return prog


# ---------------------------------------------------------------------------------------------- subset construction
BASIC, CONTAINED, SEARCH = 0, 1, 2


class StateSet:
    def __init__(self):
        self.data = {}  # state -> (distance, priority)
        self.states = JavaIntHashSet()
        self.seen_accepting = False

    def add(self, s, dist, prio):
        cur = self.data.get(s)
        if cur is None or cur[0] < dist:
            self.data[s] = (dist, prio)
        return self.states.add(s)

    def prune(self, acc, boundary, prio):
        removed = False
        for s in list(self.states):
            if s == acc:
                continue
            d, p = self.data[s]
            if d < boundary or prio < p:
                self.states.remove(s)
                del self.data[s]
                removed = True
        return removed

    def key(self):
        return frozenset(self.data)


def minimal_covering(ranges):
    if len(ranges) < 2:
        return list(ranges)
    ranges = sorted(ranges, key=lambda r: r[0])  # stable, by start only
    out, last_s, last_e = [], -1, -1
    for i, cur in enumerate(ranges):
        while last_e < cur[1]:
            s, e = cur
            if last_s >= s:
                s = last_s + 1
            if last_e >= s:
                s = last_e + 1
            for nx in ranges[i + 1:]:
                if s < nx[0] <= e:
                    e = nx[0] - 1
                if s <= nx[1] <= e:
                    e = nx[1]
            last_s, last_e = s, e
            out.append((s, e))
    return sorted(out, key=lambda r: r[0])


def cover_all(ranges):
    if not ranges:
        return [(0, 0xFFFF)]
    out, cur = [], None
    for r in ranges:
        if cur is None:
            if r[0] > 0:
                out.append((0, r[0] - 1))
        elif r[0] > cur[1] + 1:
            out.append((cur[1] + 1, r[0] - 1))
        out.append(r)
        cur = r
    if cur[1] < 0xFFFF:
        out.append((cur[1] + 1, 0xFFFF))
    return out


class Dfa:
    def __init__(self):
        self.accepting = []
        self.trans = []  # per state: list of [start, end, target], sorted by start

    def new_state(self, acc):
        self.accepting.append(acc)
        self.trans.append([])
        return len(self.accepting) - 1

    def add_transition(self, s, rng, target):
        for t in self.trans[s]:
            if (t[0], t[1]) == rng:
                return
            if t[1] + 1 == rng[0] and t[2] == target:
                t[1] = rng[1]
                return
        self.trans[s].append([rng[0], rng[1], target])
        self.trans[s].sort(key=lambda t: t[0])


def subset_construction(prog, mode):
    match_state = len(prog) - 1
    closure_cache = {}

    def nfa_closure(state):
        if state not in closure_cache:
            seen, closure, queue = set(), JavaIntHashSet(), [state]
            qi = 0
            while qi < len(queue):
                nx = queue[qi]
                qi += 1
                seen.add(nx)
                ins = prog[nx]
                if ins.op == SPLIT:
                    queue += [t for t in ins.targets if t not in seen]
                elif ins.op == JUMP:
                    if ins.target not in seen:
                        queue.append(ins.target)
                else:
                    closure.add(nx)
            closure_cache[state] = list(closure)
        return closure_cache[state]

    def eps_closure(states):
        c = StateSet()
        for s in states.states:
            prio = prog[s].prio
            for e in nfa_closure(s):
                if e == match_state:
                    c.seen_accepting = True
                c.add(e, states.data[e][0] if e in states.data else states.data[s][0], prio)
        c.seen_accepting = c.seen_accepting or states.seen_accepting
        return c

    dfa = Dfa()
    stored = {}

    def lookup(ss):
        for seen_acc, idx in stored.get(ss.key(), ()):
            if len(ss.states) == 1 or ss.seen_accepting == seen_acc:
                return idx
        return None

    def store(ss, idx):
        stored.setdefault(ss.key(), []).append((ss.seen_accepting, idx))

    init = StateSet()
    init.add(0, 0, 1)
    root = eps_closure(init)
    dfa.new_state(match_state in root.data)
    root.seen_accepting = dfa.accepting[0]
    store(root, 0)
    pending = [root]
    while pending:
        cur = pending.pop()
        idx = lookup(cur)
        ec = eps_closure(cur)
        accepting = ec.seen_accepting
        if accepting and mode == CONTAINED:
            continue
        if mode == CONTAINED or (not accepting and mode == SEARCH):
            ec.add(0, 0, 1)
        order = list(ec.states)
        crs = [(prog[s].a, prog[s].b) for s in order if prog[s].op == CHAR]
        for rng in cover_all(minimal_covering(crs)):
            tr = StateSet()
            for s in order:
                ins = prog[s]
                if ins.op == CHAR and ins.a <= rng[0] <= ins.b:
                    tr.add(s + 1, ec.data[s][0] + 1, ins.prio)
            post = eps_closure(tr)
            if not post.seen_accepting:
                post.seen_accepting = ec.seen_accepting or match_state in post.data
            if post.seen_accepting and mode != BASIC:
                while match_state in post.data and post.prune(match_state, *post.data[match_state]):
                    pass
            if not post.seen_accepting and mode != BASIC:
                post.add(0, 0, 1)
                post = eps_closure(post)
            target = lookup(post)
            if target is None:
                target = dfa.new_state(match_state in post.data)
                store(post, target)
                pending.append(post)
            dfa.add_transition(idx, rng, target)
    return dfa


def prune_dead(d):
    n = len(d.accepting)
    live = {0} | {i for i in range(n) if d.accepting[i]}
    changed = True
    while changed:
        changed = False
        for i in range(n):
            if i not in live and any(t[2] in live for t in d.trans[i]):
                live.add(i)
                changed = True
    renum = {old: new for new, old in enumerate(sorted(live))}
    out = Dfa()
    for old in sorted(live):
        out.new_state(d.accepting[old])
    for old in sorted(live):
        out.trans[renum[old]] = [[t[0], t[1], renum[t[2]]] for t in d.trans[old] if t[2] in live]
    return out


def minimize(d):
    n = len(d.accepting)
    block = {}
    ids = {}
    for i in range(n):
        key = (d.accepting[i], len(d.trans[i]), sum(t[0] for t in d.trans[i]))
        block[i] = ids.setdefault(key, len(ids))
    while True:
        ids = {}
        nb = {}
        for i in range(n):
            sig = (block[i],) + tuple((t[0], t[1], block[t[2]]) for t in d.trans[i])
            nb[i] = ids.setdefault(sig, len(ids))
        stable = len(ids) == len(set(block.values()))
        block = nb
        if stable:
            break
    order, rep = {}, []
    for i in range(n):
        for s in [i] + [t[2] for t in d.trans[i]]:
            if block[s] not in order:
                order[block[s]] = len(rep)
                rep.append(s)
    out = Dfa()
    for s in rep:
        out.new_state(d.accepting[s])
    for b, s in enumerate(rep):
        for t in d.trans[s]:
            out.add_transition(b, (t[0], t[1]), order[block[t[2]]])
    return out


def byte_classes(d):
    all_r = sorted({(t[0], t[1]) for tr in d.trans for t in tr})
    distinct, nxt = [], 0
    done = False
    for i, left in enumerate(all_r):
        if done:
            break
        nxt = max(nxt, left[0])
        while nxt <= left[1]:
            end = left[1]
            for right in all_r[i + 1:]:
                if right[1] < nxt:
                    continue
                if right[0] > end:
                    break
                end = min(end, right[1]) if nxt >= right[0] else right[0] - 1
            distinct.append((nxt, end))
            if end == 0xFFFF:
                done = True
                break
            nxt = end + 1
    groups = {}
    for r in distinct:
        sig = frozenset((s, t[2]) for s, tr in enumerate(d.trans) for t in tr if t[0] <= r[1] and t[1] >= r[0])
        groups.setdefault(sig, []).append(r)
    gl = sorted(sorted(g) for g in groups.values())
    cmap = [0] * 65536
    cls = 1
    for g in gl:
        for a, b in g:
            for c in range(a, min(b + 1, 65535)):
                cmap[c] = cls
        cls += 1
        if cls > 255:
            return None, 0
    return cmap, cls & 0xFF


def effective_count(c):
    if c > 16 or c < 3:
        return c
    return 4 if c < 4 else 8 if c < 8 else 16 if c < 16 else c


def table_of(d, cmap, stride):
    n = len(d.accepting)
    table = [-1] * (n * stride)
    max_char = 0
    for s in range(n):
        seen = set()
        for a, b, tgt in d.trans[s]:
            max_char = max(max_char, a, b)
            c = a
            while c <= b:
                k = cmap[c]
                if k not in seen:
                    seen.add(k)
                    table[s * stride + k] = tgt
                # skip to the end of this class run inside the range
                e = c
                while e < b and cmap[e + 1] == k:
                    e += 1
                c = e + 1
    return table, max_char


def compile_regex(regex, flags=0):
    """-> dict(class_map, stride, fixed_len, min_len, max_len, dfas={matches,contained_in,forwards,backwards:
    dict(n_states, table, accepting, max_char)})"""
    ast = Parser(regex, flags).parse()
    no_open_union(ast)
    mn, mx = min_len(ast), max_len(ast)
    lml = bool(flags & LEFTMOST_LONGEST)
    fwd = build_program(ast, lml)
    rev = build_program(reverse(ast), lml)

    def build(prog, mode):
        return minimize(prune_dead(subset_construction(prog, mode)))

    dfas = {"matches": build(fwd, BASIC), "contained_in": build(fwd, CONTAINED), "backwards": build(rev, BASIC),
            "forwards": build(fwd, SEARCH)}
    for d in dfas.values():
        if len(d.accepting) > 16383:
            raise Unsupported("more than 16383 states")
    cmap, count = byte_classes(dfas["forwards"])
    if cmap is None:
        raise Unsupported("more than 255 char classes")
    stride = effective_count(count)
    out = {"class_map": cmap, "stride": stride, "min_len": mn, "max_len": -1 if mx is None else mx,
           "fixed_len": mn if mx is not None and mn == mx else -1, "dfas": {}}
    for name, d in dfas.items():
        table, max_char = table_of(d, cmap, stride)
        out["dfas"][name] = {"n_states": len(d.accepting), "table": table, "max_char": max_char,
                             "accepting": [i for i, a in enumerate(d.accepting) if a]}
    return out
