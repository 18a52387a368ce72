for w in 4 6 8; do
  SF_DL_WORKERS=$w python bench.py --cpu-grid 0 --no-secondary --steps 1 --warmup 1 --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); pc=d['config']['pcie_inclusive']
print('workers $w', 'first', pc['struct_first_call_ms'], 'second', pc['struct_second_call_ms'], 'plan first', pc['plan_first_call_ms'], 'plan second', pc['plan_second_call_ms'])"
done
