"""TEST INFRASTRUCTURE -- not product code.

A minimal JVM class-file reader + bytecode interpreter, just big enough to EXECUTE the reference's
own compiled matchers: the 12 generated classes committed under
/root/reference/needle-compiler/src/test/resources/snapshots/*.class (written by
needle-compiler/src/test/java/com/justinblank/strings/SnapshotTests.java:72-79 through
DFACompiler.compileToBytes, needle-compiler/src/main/java/com/justinblank/strings/DFACompiler.java:45-74).

There is no JVM in this image, so this is the only way to run the reference's real output.  It is
used ONLY in this container (the class files live under /root/reference and never travel) by
tests/golden/gen_snapshot_vectors.py, which records inputs -> outputs as fixtures under
tests/golden/.  Nothing in the product path, the GPU tests, smoke() or bench.py imports this file.

Supported: the opcode subset those classes use (aload/iload/istore/astore, iconst/bipush/sipush/ldc,
get/putstatic, get/putfield, baload/saload/bastore/sastore, newarray, dup, iadd/isub/imul/ixor,
if_icmp*, if*, goto, ireturn/return/areturn, invokevirtual/special/static, iinc, pop, i2c/i2b/i2s,
tableswitch/lookupswitch) and the 12 external methods they call (String.charAt/length/indexOf/
startsWith, Math.min/max, Arrays.fill, Object.<init>, and ByteClassUtil.fillBytes /
fillMultipleByteClassesFromString[UsingShorts]_singleArray whose semantics are restated from
needle-types/src/main/java/com/justinblank/strings/ByteClassUtil.java:44-120).
"""
import struct


class ClassFile:
    def __init__(self, data: bytes):
        self.data = data
        self.pos = 0
        magic, self.minor, self.major = self._u4(), self._u2(), self._u2()
        assert magic == 0xCAFEBABE
        n = self._u2()
        self.cp = [None] * n
        i = 1
        while i < n:
            tag = self._u1()
            if tag == 1:
                ln = self._u2()
                raw = self.data[self.pos:self.pos + ln]
                self.pos += ln
                self.cp[i] = ("Utf8", _decode_mutf8(raw))
            elif tag == 3:
                self.cp[i] = ("Integer", struct.unpack(">i", self._take(4))[0])
            elif tag == 4:
                self.cp[i] = ("Float", struct.unpack(">f", self._take(4))[0])
            elif tag == 5:
                self.cp[i] = ("Long", struct.unpack(">q", self._take(8))[0])
                i += 1
            elif tag == 6:
                self.cp[i] = ("Double", struct.unpack(">d", self._take(8))[0])
                i += 1
            elif tag == 7:
                self.cp[i] = ("Class", self._u2())
            elif tag == 8:
                self.cp[i] = ("String", self._u2())
            elif tag in (9, 10, 11):
                self.cp[i] = ({9: "Fieldref", 10: "Methodref", 11: "InterfaceMethodref"}[tag], self._u2(), self._u2())
            elif tag == 12:
                self.cp[i] = ("NameAndType", self._u2(), self._u2())
            elif tag == 15:
                self.cp[i] = ("MethodHandle", self._u1(), self._u2())
            elif tag == 16:
                self.cp[i] = ("MethodType", self._u2())
            elif tag == 18:
                self.cp[i] = ("InvokeDynamic", self._u2(), self._u2())
            else:
                raise ValueError("cp tag %d" % tag)
            i += 1
        self.access = self._u2()
        self.this_class = self.class_name(self._u2())
        self.super_class = self.class_name(self._u2())
        self.interfaces = [self.class_name(self._u2()) for _ in range(self._u2())]
        self.fields = []
        for _ in range(self._u2()):
            acc, name, desc = self._u2(), self.utf8(self._u2()), self.utf8(self._u2())
            attrs = self._attrs()
            const = None
            if "ConstantValue" in attrs:
                idx = struct.unpack(">H", attrs["ConstantValue"])[0]
                const = self.const(idx)
            self.fields.append(dict(access=acc, name=name, desc=desc, const=const))
        self.methods = {}
        for _ in range(self._u2()):
            acc, name, desc = self._u2(), self.utf8(self._u2()), self.utf8(self._u2())
            attrs = self._attrs()
            code = None
            if "Code" in attrs:
                a = attrs["Code"]
                max_stack, max_locals, clen = struct.unpack(">HHI", a[:8])
                code = dict(max_stack=max_stack, max_locals=max_locals, code=a[8:8 + clen])
            self.methods[(name, desc)] = dict(access=acc, name=name, desc=desc, code=code)

    # -- raw readers
    def _take(self, n):
        b = self.data[self.pos:self.pos + n]
        self.pos += n
        return b

    def _u1(self):
        return self._take(1)[0]

    def _u2(self):
        return struct.unpack(">H", self._take(2))[0]

    def _u4(self):
        return struct.unpack(">I", self._take(4))[0]

    def _attrs(self):
        out = {}
        for _ in range(self._u2()):
            name = self.utf8(self._u2())
            ln = self._u4()
            out[name] = self._take(ln)
        return out

    # -- constant pool helpers
    def utf8(self, i):
        e = self.cp[i]
        assert e[0] == "Utf8"
        return e[1]

    def class_name(self, i):
        return self.utf8(self.cp[i][1]) if i else None

    def const(self, i):
        e = self.cp[i]
        if e[0] in ("Integer", "Float", "Long", "Double"):
            return e[1]
        if e[0] == "String":
            return self.utf8(e[1])
        raise ValueError(e)

    def member(self, i):
        e = self.cp[i]
        cls = self.class_name(e[1])
        nt = self.cp[e[2]]
        return cls, self.utf8(nt[1]), self.utf8(nt[2])

    def strings(self):
        """All CONSTANT_String values (the table strings live here)."""
        return [self.utf8(e[1]) for e in self.cp if e and e[0] == "String"]


def _decode_mutf8(raw: bytes) -> str:
    """Java modified UTF-8 -> str of UTF-16 code units (one Python char per code unit)."""
    out = []
    i, n = 0, len(raw)
    while i < n:
        b = raw[i]
        if b < 0x80:
            out.append(chr(b)); i += 1
        elif (b & 0xE0) == 0xC0:
            out.append(chr(((b & 0x1F) << 6) | (raw[i + 1] & 0x3F))); i += 2
        else:
            out.append(chr(((b & 0x0F) << 12) | ((raw[i + 1] & 0x3F) << 6) | (raw[i + 2] & 0x3F))); i += 3
    return "".join(out)


def _i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def _i8(x):
    x &= 0xFF
    return x - 256 if x & 0x80 else x


def _i16(x):
    x &= 0xFFFF
    return x - 65536 if x & 0x8000 else x


class JavaThrow(Exception):
    pass


class JObject:
    def __init__(self, cls):
        self.cls = cls
        self.fields = {}


def _count_args(desc):
    args = []
    i = 1
    while desc[i] != ")":
        start = i
        while desc[i] == "[":
            i += 1
        if desc[i] == "L":
            i = desc.index(";", i)
        i += 1
        args.append(desc[start:i])
    return args, desc[i + 1:]


# ByteClassUtil natives, restated from needle-types/.../ByteClassUtil.java:44-120
def _fill_bytes(arr, state, start, end):
    for i in range(start, end + 1):
        arr[i] = state


def _fill_from_string(arr, length, s, narrow):
    for state_string in s.split(";"):
        if state_string == "":
            continue
        parts = state_string.split(":")
        if len(parts) != 2:
            raise JavaThrow("IllegalArgumentException: malformed " + state_string)
        state = int(parts[0], 16)
        for comp in parts[1].split(","):
            pieces = comp.split("-")
            if len(pieces) != 2:
                raise JavaThrow("IllegalArgumentException: malformed " + comp)
            bc, target = int(pieces[0], 16), int(pieces[1], 16)
            off = state * length + bc
            if off < 0 or off >= len(arr):
                raise JavaThrow("ArrayIndexOutOfBoundsException %d" % off)
            arr[off] = narrow(target)


class Machine:
    """Loads one generated matcher class and runs its methods."""

    def __init__(self, path):
        with open(path, "rb") as f:
            self.cf = ClassFile(f.read())
        self.statics = {}
        for fld in self.cf.fields:
            if fld["access"] & 0x0008:  # static
                self.statics[fld["name"]] = fld["const"] if fld["const"] is not None else 0
        self.steps = 0
        if ("<clinit>", "()V") in self.cf.methods:
            self.invoke("<clinit>", "()V", [])

    def new_matcher(self, s: str):
        obj = JObject(self.cf.this_class)
        for fld in self.cf.fields:
            if not (fld["access"] & 0x0008):
                obj.fields[fld["name"]] = 0
        self.invoke("<init>", "(Ljava/lang/String;)V", [obj, s])
        # instance-field initial values are ConstantValue-less in generated code; the generator's
        # field defaults (start/end = -1, DFAClassBuilder.java:692-693) are applied by mako in
        # <init>; whatever the bytecode does is what we run.
        return obj

    def call(self, obj, name, desc, *args):
        return self.invoke(name, desc, [obj] + list(args))

    # -- externals
    def _external(self, cls, name, desc, args):
        if cls == "java/lang/Object" and name == "<init>":
            return None
        if cls == "java/lang/String":
            s = args[0]
            if name == "charAt":
                i = args[1]
                if i < 0 or i >= len(s):
                    raise JavaThrow("StringIndexOutOfBoundsException %d (len %d)" % (i, len(s)))
                return ord(s[i])
            if name == "length":
                return len(s)
            if name == "indexOf" and desc == "(Ljava/lang/String;I)I":
                frm = max(args[2], 0)
                return s.find(args[1], frm) if frm <= len(s) else (len(s) if args[1] == "" else -1)
            if name == "startsWith" and desc == "(Ljava/lang/String;)Z":
                return 1 if s.startswith(args[1]) else 0
        if cls == "java/lang/Math":
            if name == "min":
                return min(args[0], args[1])
            if name == "max":
                return max(args[0], args[1])
        if cls == "java/util/Arrays" and name == "fill":
            arr, v = args
            for i in range(len(arr)):
                arr[i] = v
            return None
        if cls == "com/justinblank/strings/ByteClassUtil":
            if name == "fillBytes":
                _fill_bytes(*args)
                return None
            if name == "fillMultipleByteClassesFromString_singleArray":
                _fill_from_string(args[0], args[1], args[2], _i8)
                return None
            if name == "fillMultipleByteClassesFromStringUsingShorts_singleArray":
                _fill_from_string(args[0], args[1], args[2], _i16)
                return None
        raise NotImplementedError("%s.%s%s" % (cls, name, desc))

    def invoke(self, name, desc, args):
        m = self.cf.methods[(name, desc)]
        code = m["code"]["code"]
        loc = [0] * max(m["code"]["max_locals"], len(args) + 1)
        # all generated-method parameters are category-1 (int / reference)
        for i, a in enumerate(args):
            loc[i] = a
        st = []
        pc = 0
        cf = self.cf

        def s2(p):
            return struct.unpack(">h", code[p:p + 2])[0]

        def u2(p):
            return struct.unpack(">H", code[p:p + 2])[0]

        while True:
            self.steps += 1
            op = code[pc]
            if op == 0x00:  # nop
                pc += 1
            elif op == 0x01:  # aconst_null
                st.append(None); pc += 1
            elif 0x02 <= op <= 0x08:  # iconst_m1..5
                st.append(op - 3); pc += 1
            elif op == 0x10:  # bipush
                st.append(_i8(code[pc + 1])); pc += 2
            elif op == 0x11:  # sipush
                st.append(s2(pc + 1)); pc += 3
            elif op == 0x12:  # ldc
                st.append(cf.const(code[pc + 1])); pc += 2
            elif op == 0x13:  # ldc_w
                st.append(cf.const(u2(pc + 1))); pc += 3
            elif op in (0x15, 0x19):  # iload, aload
                st.append(loc[code[pc + 1]]); pc += 2
            elif 0x1A <= op <= 0x1D:  # iload_n
                st.append(loc[op - 0x1A]); pc += 1
            elif 0x2A <= op <= 0x2D:  # aload_n
                st.append(loc[op - 0x2A]); pc += 1
            elif op in (0x33, 0x35, 0x34, 0x2E):  # baload, saload, caload, iaload
                i = st.pop(); arr = st.pop()
                if i < 0 or i >= len(arr):
                    raise JavaThrow("ArrayIndexOutOfBoundsException %d (len %d)" % (i, len(arr)))
                v = arr[i]
                st.append(int(v)); pc += 1
            elif op in (0x36, 0x3A):  # istore, astore
                loc[code[pc + 1]] = st.pop(); pc += 2
            elif 0x3B <= op <= 0x3E:  # istore_n
                loc[op - 0x3B] = st.pop(); pc += 1
            elif 0x4B <= op <= 0x4E:  # astore_n
                loc[op - 0x4B] = st.pop(); pc += 1
            elif op in (0x54, 0x56, 0x4F, 0x55):  # bastore, sastore, iastore, castore
                v = st.pop(); i = st.pop(); arr = st.pop()
                if i < 0 or i >= len(arr):
                    raise JavaThrow("ArrayIndexOutOfBoundsException %d" % i)
                if op == 0x54:
                    v = _i8(v) if not isinstance(arr, _BoolArray) else (v & 1)
                elif op == 0x56:
                    v = _i16(v)
                elif op == 0x55:
                    v &= 0xFFFF
                arr[i] = v; pc += 1
            elif op == 0x57:  # pop
                st.pop(); pc += 1
            elif op == 0x59:  # dup
                st.append(st[-1]); pc += 1
            elif op == 0x60:
                b = st.pop(); a = st.pop(); st.append(_i32(a + b)); pc += 1
            elif op == 0x64:
                b = st.pop(); a = st.pop(); st.append(_i32(a - b)); pc += 1
            elif op == 0x68:
                b = st.pop(); a = st.pop(); st.append(_i32(a * b)); pc += 1
            elif op == 0x6C:  # idiv
                b = st.pop(); a = st.pop()
                if b == 0:
                    raise JavaThrow("ArithmeticException")
                q = abs(a) // abs(b)
                st.append(_i32(q if (a < 0) == (b < 0) else -q)); pc += 1
            elif op == 0x7E:
                b = st.pop(); a = st.pop(); st.append(_i32(a & b)); pc += 1
            elif op == 0x80:
                b = st.pop(); a = st.pop(); st.append(_i32(a | b)); pc += 1
            elif op == 0x82:
                b = st.pop(); a = st.pop(); st.append(_i32(a ^ b)); pc += 1
            elif op == 0x84:  # iinc
                loc[code[pc + 1]] = _i32(loc[code[pc + 1]] + _i8(code[pc + 2])); pc += 3
            elif op == 0x91:
                st.append(_i8(st.pop())); pc += 1
            elif op == 0x92:
                st.append(st.pop() & 0xFFFF); pc += 1
            elif op == 0x93:
                st.append(_i16(st.pop())); pc += 1
            elif 0x99 <= op <= 0x9E:  # ifeq..ifle
                v = st.pop()
                t = (v == 0, v != 0, v < 0, v >= 0, v > 0, v <= 0)[op - 0x99]
                pc = pc + s2(pc + 1) if t else pc + 3
            elif 0x9F <= op <= 0xA4:  # if_icmp*
                b = st.pop(); a = st.pop()
                t = (a == b, a != b, a < b, a >= b, a > b, a <= b)[op - 0x9F]
                pc = pc + s2(pc + 1) if t else pc + 3
            elif op in (0xA5, 0xA6):  # if_acmpeq/ne
                b = st.pop(); a = st.pop()
                t = (a is b) if op == 0xA5 else (a is not b)
                pc = pc + s2(pc + 1) if t else pc + 3
            elif op == 0xA7:  # goto
                pc = pc + s2(pc + 1)
            elif op == 0xAA:  # tableswitch
                base = pc
                p = (pc + 4) & ~3
                default, low, high = struct.unpack(">iii", code[p:p + 12])
                v = st.pop()
                if v < low or v > high:
                    pc = base + default
                else:
                    pc = base + struct.unpack(">i", code[p + 12 + 4 * (v - low):p + 16 + 4 * (v - low)])[0]
            elif op == 0xAB:  # lookupswitch
                base = pc
                p = (pc + 4) & ~3
                default, npairs = struct.unpack(">ii", code[p:p + 8])
                v = st.pop()
                tgt = default
                for k in range(npairs):
                    m_, off = struct.unpack(">ii", code[p + 8 + 8 * k:p + 16 + 8 * k])
                    if m_ == v:
                        tgt = off
                        break
                pc = base + tgt
            elif op in (0xAC, 0xB0):  # ireturn, areturn
                return st.pop()
            elif op == 0xB1:
                return None
            elif op == 0xB2:  # getstatic
                cls, nm, _ = cf.member(u2(pc + 1))
                assert cls == cf.this_class, cls
                st.append(self.statics[nm]); pc += 3
            elif op == 0xB3:  # putstatic
                cls, nm, _ = cf.member(u2(pc + 1))
                assert cls == cf.this_class, cls
                self.statics[nm] = st.pop(); pc += 3
            elif op == 0xB4:  # getfield
                _, nm, _ = cf.member(u2(pc + 1))
                st.append(st.pop().fields[nm]); pc += 3
            elif op == 0xB5:  # putfield
                _, nm, _ = cf.member(u2(pc + 1))
                v = st.pop(); o = st.pop(); o.fields[nm] = v; pc += 3
            elif op in (0xB6, 0xB7, 0xB8):  # invokevirtual/special/static
                cls, nm, d = cf.member(u2(pc + 1))
                argt, ret = _count_args(d)
                n = len(argt) + (0 if op == 0xB8 else 1)
                a = st[len(st) - n:] if n else []
                del st[len(st) - n:]
                if cls == cf.this_class:
                    r = self.invoke(nm, d, a)
                else:
                    r = self._external(cls, nm, d, a)
                if ret != "V":
                    st.append(r)
                pc += 3
            elif op == 0xBC:  # newarray
                n = st.pop()
                t = code[pc + 1]
                st.append(_BoolArray(n) if t == 4 else [0] * n); pc += 2
            elif op == 0xBE:  # arraylength
                st.append(len(st.pop())); pc += 1
            else:
                raise NotImplementedError("opcode 0x%02x at %d in %s%s" % (op, pc, name, desc))


def instruction_starts(code: bytes):
    """pcs of instruction boundaries (linear sweep) for the opcode subset above."""
    pcs = []
    pc = 0
    while pc < len(code):
        pcs.append(pc)
        op = code[pc]
        if op in (0x10, 0x12, 0x15, 0x16, 0x17, 0x18, 0x19, 0x36, 0x37, 0x38, 0x39, 0x3A, 0xBC, 0xA9):
            pc += 2
        elif op in (0x11, 0x13, 0x14, 0x84, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xBB, 0xBD, 0xC0, 0xC1) \
                or 0x99 <= op <= 0xA8 or op in (0xC6, 0xC7):
            pc += 3
        elif op == 0xB9 or op == 0xBA or op in (0xC8, 0xC9):
            pc += 5
        elif op == 0xAA:
            p = (pc + 4) & ~3
            low, high = struct.unpack(">ii", code[p + 4:p + 12])
            pc = p + 12 + 4 * (high - low + 1)
        elif op == 0xAB:
            p = (pc + 4) & ~3
            npairs = struct.unpack(">i", code[p + 4:p + 8])[0]
            pc = p + 8 + 8 * npairs
        else:
            pc += 1
    return pcs


class _BoolArray(list):
    def __init__(self, n):
        super().__init__([0] * n)


class SnapshotMatcher:
    """Python-facing wrapper with the reference Matcher interface (needle-types/.../Matcher.java:6-26)."""

    def __init__(self, machine: Machine, s: str):
        self.m = machine
        self.o = machine.new_matcher(s)

    def matches(self):
        return bool(self.m.call(self.o, "matches", "()Z"))

    def containedIn(self):
        return bool(self.m.call(self.o, "containedIn", "()Z"))

    def find(self, *a):
        if a:
            return bool(self.m.call(self.o, "find", "(II)Z", *a))
        return bool(self.m.call(self.o, "find", "()Z"))

    def start(self):
        return self.m.call(self.o, "start", "()I")

    def end(self):
        return self.m.call(self.o, "end", "()I")
