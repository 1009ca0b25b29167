timeout 1500 python -m pytest tests/test_unrolled_ops_gpu.py tests/test_fusions_gpu.py tests/test_models_gpu.py tests/test_gemm16_gpu.py tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for v in 1 0; do
  CTI_BAN_KCONCAT=$v python bench.py --config c4 --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 CTI_BAN_KCONCAT=$v', round(d['value']), 'samples/s', round(d['ms_per_step']*1e3,1), 'us', d.get('oracle_check') or d.get('check') or '')"
done; done
python bench.py --config c3 --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c3', round(d['value']), 'samples/s', round(d['ms_per_step']*1e3,1), 'us')"
