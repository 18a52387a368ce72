"""bench.py's secondary.end_to_end case (struct entry points on the 128^3 matrix, built-in ordering) several times in one process:
    [SF_TRACE=1] python tools/end_to_end_timing.py [repeats]"""
import importlib, json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    r = bench.end_to_end_case(sf, np, 128)
    print(json.dumps({k: r[k] for k in ("analyze_s", "factorize_first_call_s", "validate_s", "cleanup_s", "end_to_end_s")}), flush=True)
