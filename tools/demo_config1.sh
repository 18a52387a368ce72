#!/bin/bash
# BASELINE config 1 through the reference's own entry point: writes the 2-D 5-point Laplacian 100 x 100 (and a 3-D 32^3 one) as
# MatrixMarket files and runs the demo program on them.   bash tools/demo_config1.sh
cd "$(dirname "$0")/.."
python - <<'PY'
import importlib, sys, os
sys.path.insert(0, os.getcwd())
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
n, Cp, Ci, Cx = sf.gen.laplacian_lower(100, 100); sf.gen.write_matrix_market("/tmp/lap2d_100x100.mtx", n, Cp, Ci, Cx, True)
n, Cp, Ci, Cx = sf.gen.laplacian_lower(32, 32, 32); sf.gen.write_matrix_market("/tmp/lap3d_32.mtx", n, Cp, Ci, Cx, True)
PY
./sparse-matrix-factorization-library_amd/sf_demo /tmp/lap2d_100x100.mtx /tmp/lap3d_32.mtx
