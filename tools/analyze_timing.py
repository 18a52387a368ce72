"""Host-side analysis timing (SF_TRACE phases) for the 128^3 Laplacian with 1 .. 16 analysis threads:
    python tools/analyze_timing.py [grid]"""
import importlib, os, subprocess, sys
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import time
    sys.path.insert(0, os.getcwd())
    sf = importlib.import_module("sparse-matrix-factorization-library_amd")
    g = int(sys.argv[1])
    n, Cp, Ci, Cx = sf.gen.laplacian_lower(g, g, g)
    perm = sf.grid_nd_perm(g, g, g, 3, 1)
    for _ in range(2):
        t = time.time(); sym = sf.analyze(n, Cp, Ci, Cx, perm, sf.REFERENCE_SLOT_1GPU); print("analyze %.3f s" % (time.time() - t), flush=True)
else:
    g = sys.argv[1] if len(sys.argv) > 1 else "128"
    for T in (1, 2, 4, 8, 16):
        env = dict(os.environ, SF_TRACE="1", SF_ANALYZE_THREADS=str(T))
        print("== SF_ANALYZE_THREADS =", T, flush=True)
        subprocess.run([sys.executable, __file__, g, "child"], env=env)
