import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["SF_TRACE"]="1"
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
g=128
n, Cp, Ci, Cx = sf.gen.laplacian_lower(g, g, g)
perm = sf.grid_nd_perm(g,g,g,3,1)
sym = sf.analyze(n, Cp, Ci, Cx, perm, sf.REFERENCE_SLOT_1GPU)
for _ in range(2):
    t=time.time(); plan = sf.CholPlan(sym); print("plan_create %.3f s" % (time.time()-t), flush=True); plan.close()
