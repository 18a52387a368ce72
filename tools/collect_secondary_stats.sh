#!/bin/bash
# rocprofv3 kernel stats of the secondary configurations (config 3, config 5): bash tools/collect_secondary_stats.sh TAG
set -e
TAG=${1:-r02_e}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s3 -- python3 bench.py --workload stencil2d --steps 2 --warmup 1 --cpu-grid 0 --no-pcie > $O/config3_bench_under_rocprof.json 2> $O/s3.err
cp $(ls $O/s3/*/*kernel_stats.csv | head -1) $O/config3_stencil2d_1000_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -- python3 bench.py --method lu --steps 2 --warmup 1 --cpu-grid 0 --no-pcie > $O/lu_config5_bench_under_rocprof.json 2> $O/s5.err
cp $(ls $O/s5/*/*kernel_stats.csv | head -1) $O/lu_config5_79cubed_kernel_stats.csv
rm -rf $O/s3 $O/s5
ls -la $O
