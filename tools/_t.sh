timeout 600 python -m pytest tests/test_fusions_gpu.py -m gpu -x -q -k "without_a_barrier" 2>&1 | tail -4
for rep in 1 2; do for v in 0 1; do
  CTI_BL_KS_FORM=$v python bench.py --config c4 --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 CTI_BL_KS_FORM=$v', round(d['value']), 'samples/s', round(d['ms_per_step']*1e3,1), 'us')"
done; done
