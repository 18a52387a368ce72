#!/bin/bash
# Round-end evidence for profiles/: bash tools/collect_profiles.sh TAG   (GPU box, repo root; writes gpurun_out/TAG/)
set -e
TAG=${1:-r02_c}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
# 1. the driver's default command
python bench.py > $O/128cubed_bench.json 2> $O/bench.err
# 2. kernel stats of 4 factorizations (1 warm-up + 2 timed + 1 event-profiled) + solve
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --grid 128 --steps 2 --warmup 1 --cpu-grid 0 --no-pcie --no-secondary > $O/128cubed_bench_under_rocprof.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/128cubed_kernel_stats.csv
rm -rf $O/stats
# 3. PMC passes (separate, no tracing flags), ONE factorization each
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"; do
  d=$O/pmc_$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $d -- python3 bench.py --grid 128 --steps 1 --warmup 0 --cpu-grid 0 --no-roofline --no-pcie --no-secondary > /dev/null 2> $d.err
done
python tools/summarize_pmc.py --factorizations 1 --what "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE TCC_EA0_ATOMIC_sum; SQ_* GRBM_GUI_ACTIVE; one pass per directory, no tracing flags) on python3 bench.py --grid 128 --steps 1 --warmup 0 --cpu-grid 0 --no-roofline --no-pcie --no-secondary (ONE factorization + one device solve), summed per kernel. FETCH_SIZE/WRITE_SIZE in KiB; fetch bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction, guide + tools/fetch_calib.hip)." $O/pmc_traffic_128cubed.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_WAVE_CYCLES > /dev/null
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_WAVE_CYCLES
# 4. LU (config 5) and config 3 lines on their own
python bench.py --method lu --cpu-grid 56 --no-pcie > $O/lu_config5_79cubed_bench.json 2>> $O/bench.err
python bench.py --workload stencil2d --cpu-grid 0 --no-pcie > $O/config3_stencil2d_1000_bench.json 2>> $O/bench.err
ls -la $O
