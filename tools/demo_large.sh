#!/bin/bash
# End-to-end stage times of the demo program on a large matrix: bash tools/demo_large.sh [grid=128]
cd "$(dirname "$0")/.."
G=${1:-128}
python - $G <<'PY'
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
g = int(sys.argv[1])
n, Cp, Ci, Cx = sf.gen.laplacian_lower(g, g, g)
cols = np.repeat(np.arange(n), np.diff(Cp))
t = time.time()
with open("/tmp/lap3d_big.mtx", "w") as f:
    f.write("%%MatrixMarket matrix coordinate real symmetric\n")
    f.write("%d %d %d\n" % (n, n, len(Ci)))
    np.savetxt(f, np.column_stack([Ci + 1, cols + 1, Cx]), fmt="%d %d %.17g")
print("wrote /tmp/lap3d_big.mtx: n = %d, %d entries, %.1f s" % (n, len(Ci), time.time() - t), flush=True)
PY
SF_TRACE=1 ./sparse-matrix-factorization-library_amd/sf_demo /tmp/lap3d_big.mtx
rm -f /tmp/lap3d_big.mtx
