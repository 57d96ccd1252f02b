"""One-off wider run of tests/test_gpu_fuzz.py's differential check: python scripts/fuzz_campaign.py <first_seed> <n_seeds>"""
import sys, traceback
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_fuzz as T
a, n = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, a + n):
    try:
        T.test_random_regexes_on_random_haystacks(seed)
    except AssertionError as e:
        tb = traceback.format_exc()
        bad += 1
        print("SEED", seed, "FAILED:", tb[-600:])
    except Exception:
        bad += 1
        print("SEED", seed, "ERROR"); traceback.print_exc()
print("campaign done: %d seeds, %d failures" % (n, bad))
