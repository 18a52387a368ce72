"""time of the built-in nested dissection (sf_graph_nd_perm) on the 128^3 grid for a given SF_ANALYZE_THREADS (read once per process)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, Cp, Ci, Cx = sf.gen.laplacian_lower(N, N, N)
ts = []
for _ in range(3):
    t = time.time(); p = sf.graph_nd_perm(n, Cp, Ci); ts.append(time.time() - t)
print("SF_ANALYZE_THREADS=%s graph_nd_perm %d^3: %s s" % (os.environ.get("SF_ANALYZE_THREADS", "default"), N, " ".join("%.3f" % t for t in ts)), flush=True)
