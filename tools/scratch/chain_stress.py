"""Random call sequences of tests/test_early_solution.py over many more seeds (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_early_solution as t

def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    for seed in range(lo, hi):
        try:
            t.test_random_call_sequences_keep_the_chain(seed)
        except Exception as e:  # noqa: BLE001
            bad.append((seed, repr(e)[:300]))
    print(f"seeds {lo} .. {hi - 1}: {len(bad)} failures", bad)

if __name__ == "__main__":
    main()
