for gm in 4 2 8 4 6 3; do
  COGV_GEMM_GROUP_M=$gm python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-second-dtype 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group_m=$gm', round(d['value']), 'tok/s', round(d['ms_per_step'],1), 'ms')"
done
