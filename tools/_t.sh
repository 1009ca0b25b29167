for rep in 1 2; do for cfg in c3 c4; do for v in "CTI_GEMM_RS=0 CTI_GEMM16_SMALL=0" "CTI_GEMM_RS=0 CTI_GEMM16_SMALL=1" "CTI_GEMM_RS=1 CTI_GEMM16_SMALL=1"; do
  env $v python bench.py --config $cfg --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg $v', round(d['value']), 'samples/s', round(d['ms_per_step']*1e3,1), 'us')"
done; done; done
