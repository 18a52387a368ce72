#!/bin/bash
# SF_FUSE_MAX sweep (steps of up to that many workgroups run as one fused k_step launch): config 3 and 128^3
cd "$(dirname "$0")/../.."
for fm in 8192 16384 32768 65536 1000000; do
  a=$(SF_FUSE_MAX=$fm python bench.py --workload stencil2d --cpu-grid 0 --no-pcie --steps 5 --warmup 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  b=$(SF_FUSE_MAX=$fm python bench.py --cpu-grid 0 --no-pcie --no-secondary --steps 4 --warmup 1 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "SF_FUSE_MAX=$fm config3 $a   128^3 $b"
done
