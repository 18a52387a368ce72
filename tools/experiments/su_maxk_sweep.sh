for k in 64 96 128 192 256; do echo "SU_MAXK $k"; SF_SU_MAXK=$k python bench.py --steps 3 --warmup 1 --cpu-grid 0 --no-pcie --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], 'gemm1', r['kernel_ms'], 'small', r['small_update_ms'], r['flops_update_small'])"; done
