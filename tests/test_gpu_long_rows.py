"""Few, long rows: packed-mode automata switch to the stripe path (function composition across
4 KiB stripes, intra-row parallelism); everything else stays on the one-row-per-lane kernels.  Both must agree with
the oracle bit for bit: verdicts, find() end = lastMatch (possibly megabytes into the row, possibly a match that
itself spans many stripes) and start."""
import numpy as np
import pytest

from test_gpu_configs import compiled


def gpu_all(p, rows, lens):
    import torch
    from needle_amd.pattern import unpack_bitmap
    t = torch.from_numpy(rows.view(np.int16) if rows.dtype == np.uint16 else rows).cuda()
    tl = None if lens is None else torch.from_numpy(lens.astype(np.int32)).cuda()
    n = rows.shape[0]
    m = unpack_bitmap(p.matches_batch(t, tl), n)
    c = unpack_bitmap(p.contained_in_batch(t, tl), n)
    fw, fs, fe = p.find_batch(t, tl)
    return m, c, unpack_bitmap(fw, n), fs.cpu().numpy(), fe.cpu().numpy()


def check(p, o, rows, lens):
    m, c, f, fs, fe = gpu_all(p, rows, lens)
    L = None if lens is None else lens.astype(np.uint32)
    assert (m == o.batch_matches(rows, L, threads=4)).all()
    assert (c == o.batch_contained_in(rows, L, threads=4)).all()
    of, os_, oe = o.batch_find(rows, L, threads=4)
    assert (f == of).all()
    assert (fe == oe).all(), (fe, oe)
    assert (fs == os_).all(), (fs, os_)


CASES = [
    # regex, char width, noise alphabet, what gets planted
    ("[0-9]+", 1, "abcdefghij klmnop", "0123456789"),
    ("a.", 1, "xyz\n ", "ab"),
    ("a.c", 1, "xyz\n ", "abc"),  # 5 states: one too many for packed functions -> speculative stripes
    ("ε|λ", 2, "abc xyz", "ελ"),
    ("[α-ω]{3}[α-ω]*", 2, "abc xyz—", "αβγδω"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("regex,cw,noise,plant", CASES)
def test_long_rows_take_the_stripe_path_and_match_the_oracle(regex, cw, noise, plant):
    p, o = compiled(regex)
    # packed mode (<= 4 reference states + the sink): the function-composition stripe path; else speculative stripes
    assert (p.info()["kernel_mode"]["forwards"] == 0) == (regex != "a.c")
    rng = np.random.default_rng(5)
    dtype = np.uint8 if cw == 1 else np.uint16
    n, stride = 7, 300_000  # stride not a multiple of the 4 KiB stripe; 8-bit: 74 stripes per row
    stride -= stride % (16 // cw)
    noise_a, plant_a = np.array([ord(ch) for ch in noise]), np.array([ord(ch) for ch in plant])
    rows = rng.choice(noise_a, (n, stride)).astype(dtype)
    # row 0: no match at all; row 1: one short match deep inside; row 2: match in the very last chars; row 3: a run that
    # spans several stripes (find's end far from its start); row 4: match at 0; row 5: many matches; row 6: all plant
    three = np.resize(plant_a, 3)
    rows[1, 200_123:200_126] = three
    rows[2, stride - 3:] = three
    rows[3, 50_000:50_000 + 3 * 4096 // cw + 77] = rng.choice(plant_a, 3 * 4096 // cw + 77)
    rows[4, 0:3] = three
    rows[5, rng.integers(0, stride, 2000)] = rng.choice(plant_a, 2000)
    rows[6, :] = rng.choice(plant_a, stride)
    check(p, o, rows, None)
    # ragged: lengths cut rows inside a stripe, at stripe boundaries, to zero and to one char
    lens = np.array([stride, 200_125, stride - 1, 50_000 + 4096 // cw, 0, 1, 8192 // cw], dtype=np.int64)
    check(p, o, rows, lens)


@pytest.mark.gpu
def test_one_very_long_row():
    import torch
    from needle_amd.pattern import unpack_bitmap
    p, o = compiled("[0-9]+")
    n_chars = 64 * 1024 * 1024
    row = np.full((1, n_chars), ord("x"), dtype=np.uint8)
    t = torch.from_numpy(row).cuda()
    assert not unpack_bitmap(p.contained_in_batch(t), 1)[0]
    row[0, n_chars - 5: n_chars - 2] = [ord("4"), ord("2"), ord("7")]
    t = torch.from_numpy(row).cuda()
    assert unpack_bitmap(p.contained_in_batch(t), 1)[0]
    fw, fs, fe = p.find_batch(t)
    assert unpack_bitmap(fw, 1)[0] and (int(fs[0]), int(fe[0])) == (n_chars - 5, n_chars - 2)
    assert o.find(row[0].tobytes().decode("latin-1")) == (True, n_chars - 5, n_chars - 2)


@pytest.mark.gpu
def test_stripe_path_equals_lane_path_on_ordinary_batches(monkeypatch):
    """The two work splits are interchangeable: forcing the stripe path on a short-row batch (NEEDLE_LONG_ROWS is read
    once per process, so the forced run happens in a subprocess) gives the same bits as the default path."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler, unpack_bitmap
p = DFACompiler.compile("[0-9]+", "d")
rows = W.digits_batch(torch, 0, 3000, 256, device="cuda")
lens = torch.from_numpy((np.arange(3000) * 2654435761 % 257).astype(np.int32)).cuda()
c = unpack_bitmap(p.contained_in_batch(rows, lens), 3000)
m = unpack_bitmap(p.matches_batch(rows, lens), 3000)
fw, fs, fe = p.find_batch(rows, lens)
np.save(sys.argv[1], np.concatenate([c, m, unpack_bitmap(fw, 3000), fs.cpu().numpy(), fe.cpu().numpy()]))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for force in ("0", "1"):
        path = "/tmp/needle_long_rows_%s.npy" % force
        env = dict(os.environ, NEEDLE_LONG_ROWS=force)
        subprocess.check_call([sys.executable, "-c", code, path], cwd=root, env=env)
        out.append(np.load(path))
    assert (out[0] == out[1]).all()


TABLE_CASES = [
    # regex, char width, noise alphabet, planted strings
    ("Sherlock|Holmes|Watson|Irene|Adler|John|Baker", 1, "abcdefgh SHWIAJB\n", ["Sherlock", "Holmes", "Baker", "Watso"]),
    ("[a-zA-Z]+ing", 1, "abging .,\n", ["running", "ing", "xing "]),
    ("http://.+", 1, "htp:/ wxyz\n", ["http://a.b/c", "http:/", "http://"]),
    ("(foo|bar)[a-z]{3,5}baz", 1, "fobarz xy\n", ["fooabcbaz", "barxyzzzbaz", "foobaz"]),
    ("Holmes.{1,10}Watson|Watson.{1,10}Holmes", 2, "HolmesWat \n", ["Holmes and Watson", "Watson, Holmes", "HolmesWatson"]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("regex,cw,noise,plants", TABLE_CASES)
def test_long_rows_of_table_mode_automata_take_speculative_stripes(regex, cw, noise, plants):
    """Automata too big for function composition: every 4 KiB stripe is scanned from the start state, then re-walked
    from its true entry state until the two runs meet.  Bit-exact with the oracle's sequential walk -- matches that
    straddle stripe boundaries, rows that die early, ragged lengths."""
    p, o = compiled(regex)
    assert p.info()["kernel_mode"]["forwards"] != 0
    rng = np.random.default_rng(8)
    dtype = np.uint8 if cw == 1 else np.uint16
    n, stride = 6, 512 * 1024 // cw
    noise_a = np.array([ord(ch) for ch in noise])
    rows = rng.choice(noise_a, (n, stride)).astype(dtype)

    def plant(r, at, s):
        rows[r, at:at + len(s)] = [ord(ch) for ch in s]

    per = 4096 // cw
    plant(1, 300_000 // cw, plants[0])
    plant(2, 7 * per - 3, plants[0])            # straddles a stripe boundary
    plant(2, 9 * per - 1, plants[1])
    plant(3, stride - len(plants[0]), plants[0])  # at the very end
    for at in rng.integers(0, stride - 40, 300):
        plant(4, int(at), plants[int(at) % len(plants)])
    rows[5, :] = rng.choice(noise_a[:3], stride)  # row 5: a tiny alphabet (long partial matches)
    check(p, o, rows, None)
    lens = np.array([stride, 300_000 // cw + 3, 7 * per + 2, stride - 1, 9 * per, 0], dtype=np.int64)
    check(p, o, rows, lens)


@pytest.mark.gpu
def test_matches_on_long_rows_of_a_table_mode_automaton():
    """matches() is no search: a stripe entered in the middle of the language is nowhere near the start state, and the
    speculative run usually dies at once -- the fix-up then simply IS the true run of that stripe (stripe-parallel
    all the same).  Whole-row matches of 512 KiB, a row spoilt by one char deep inside, ragged lengths."""
    regex = "((ab|cd|ef|gh|ij|kl)+ )+"
    p, o = compiled(regex)
    assert p.info()["kernel_mode"]["matches"] != 0
    rng = np.random.default_rng(21)
    pairs = ["ab", "cd", "ef", "gh", "ij", "kl"]
    n, stride = 5, 512 * 1024
    rows = np.zeros((n, stride), dtype=np.uint8)
    for r in range(n):
        out = []
        size = 0
        while size < stride:
            word = "".join(rng.choice(pairs, int(rng.integers(1, 5)))) + " "
            out.append(word)
            size += len(word)
        text = "".join(out)[:stride]
        rows[r] = np.frombuffer(text.encode("latin-1"), dtype=np.uint8)
    lens = np.array([stride, stride, stride, 300_001, 0], dtype=np.int64)
    for r in range(n):  # make every row end on a word boundary inside its length
        end = int(lens[r])
        while end > 0 and rows[r, end - 1] != 32:
            end -= 1
        lens[r] = end
    rows[2, 222_222] = ord("z")  # spoilt
    m, c, f, fs, fe = gpu_all(p, rows, lens)
    want = o.batch_matches(rows, lens.astype(np.uint32), threads=4)
    assert (m == want).all()
    assert want.tolist() == [True, True, False, True, False]


@pytest.mark.gpu
def test_an_automaton_that_never_resynchronises_falls_back_to_the_lane_walk():
    """`q.*z` under DOTALL: once a `q` is seen the true run sits in a state no run started later in the row ever reaches,
    so every fix-up round settles just one more stripe; after the bounded number of rounds the rows are walked one lane
    each.  Same bits either way."""
    from needle_amd.pattern import DOTALL
    p, o = compiled("q.*z#[0-9a-f]{6}", DOTALL)
    assert p.info()["kernel_mode"]["forwards"] != 0
    rng = np.random.default_rng(4)
    n, stride = 4, 512 * 1024
    rows = rng.choice(np.array([ord(c) for c in "abcdefgh \n"]), (n, stride)).astype(np.uint8)
    rows[0, 10] = ord("q")                                # q, then never a z: no match, 128 unsettled stripes
    rows[1, 10] = ord("q")
    rows[1, 400_000:400_008] = [ord(c) for c in "z#00ff00"]   # q ... z#00ff00 far away
    rows[2, 300_000:300_008] = [ord(c) for c in "z#00ff00"]   # no q at all
    rows[3, 7] = ord("q")
    rows[3, 4096 * 3 - 2:4096 * 3 + 6] = [ord(c) for c in "z#abcdef"]  # straddles a stripe boundary
    check(p, o, rows, None)


@pytest.mark.gpu
def test_stripe_find_without_candidate_stripes_in_a_child():
    """find() on the function-composition stripe path walks, a second time, only each row's LAST stripe that passes through an
    accepting state from its true entry state (the default; the tests above).  NEEDLE_STRIPE_CAND=0 -- every stripe walked
    again, round 2's form -- must give the same (oracle-checked) results; the switch is read once per process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_long_rows as T
for case in T.CASES:
    if case[0] != "a.c":
        T.test_long_rows_take_the_stripe_path_and_match_the_oracle.__wrapped__(*case) if hasattr(T.test_long_rows_take_the_stripe_path_and_match_the_oracle, "__wrapped__") else T.test_long_rows_take_the_stripe_path_and_match_the_oracle(*case)
print("STRIPE-ALL-OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, NEEDLE_STRIPE_CAND="0"), capture_output=True, text=True, timeout=900)
    assert "STRIPE-ALL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
