"""Extended randomised parity run (not collected by pytest): the bodies of test_random_configurations and
test_random_gather_and_tail over many more seeds than the suite carries.  Usage (GPU box):
    python tests/fuzz_extended.py FIRST LAST [LOGFILE]
Prints one line per failing seed and a summary; progress goes to LOGFILE every 10 seeds."""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import test_gpu_parity as T  # noqa: E402


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    log = open(sys.argv[3], "a") if len(sys.argv) > 3 else sys.stdout
    fails = []
    for seed in range(first, last):
        for fn in (T.test_random_configurations, T.test_random_gather_and_tail):
            try:
                fn(seed)
            except Exception as e:  # noqa: BLE001
                fails.append((fn.__name__, seed))
                print(f"FAIL {fn.__name__}({seed}): {type(e).__name__}: {str(e)[:300]}", file=log, flush=True)
                if not isinstance(e, AssertionError):
                    traceback.print_exc(file=log)
        if seed % 10 == 0:
            print(f"seed {seed} done, {len(fails)} failures so far", file=log, flush=True)
    print(f"fuzz {first}..{last}: {len(fails)} failures {fails}", file=log, flush=True)
    print(f"fuzz {first}..{last}: {len(fails)} failures {fails}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
