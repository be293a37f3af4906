"""scratch: one seed of a fuzz body, printing the assertion text"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_parity as T
name, seed = sys.argv[1], int(sys.argv[2])
try:
    getattr(T, name)(seed)
    print(name, seed, "ok")
except AssertionError as e:
    print(name, seed, "FAIL", e)
