#!/usr/bin/env python3
"""Is the slow mode of the struct entry point (0.55 s -> 0.9 s after a few calls) the kernel's automatic NUMA balancing at work on
the 30 GB of pageable Lsx?  Prints /proc/vmstat deltas (numa_hint_faults, numa_pages_migrated, pgfault, thp_*) per call.

    python tools/struct_numa_probe.py [grid=128] [calls=8]
"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
KEYS = ("numa_hint_faults", "numa_hint_faults_local", "numa_pages_migrated", "numa_pte_updates", "pgfault", "pgmigrate_success",
        "thp_fault_alloc", "thp_collapse_alloc", "thp_split_page", "compact_stall")


def vmstat():
    d = {}
    for line in open("/proc/vmstat"):
        k, v = line.split()
        if k in KEYS:
            d[k] = int(v)
    return d


for f in ("/proc/sys/kernel/numa_balancing", "/sys/kernel/mm/transparent_hugepage/enabled", "/sys/kernel/mm/transparent_hugepage/defrag"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, e)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, Cp, Ci, Cx = sf.gen.laplacian_lower(N, N, N)
common = sf.CommonInfo(dev_slot_size=sf.REFERENCE_SLOT_1GPU)
mi = sf.MatrixInfo()
mi.set_csc(n, Cp, Ci, Cx)
mi.set_perm(sf.grid_nd_perm(N, N, N, 3, 1))
mi.analyze(common)
for k in range(calls):
    v0 = vmstat(); t0 = time.perf_counter(); mi.factorize(common); dt = time.perf_counter() - t0; v1 = vmstat()
    print(f"call {k}: {1e3 * dt:.1f} ms  " + "  ".join(f"{key}+{v1[key] - v0[key]}" for key in KEYS if key in v1 and v1[key] != v0[key]), flush=True)
    if len(sys.argv) > 3:
        time.sleep(float(sys.argv[3]))
mi.cleanup(); common.close()
