timeout 1500 python -m pytest tests/test_fusions_gpu.py tests/test_models_gpu.py tests/test_bf16_io_gpu.py tests/test_gemm16_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for cfg in c3 c4; do
  python bench.py --config $cfg --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', round(d['value']), 'samples/s', round(d['ms_per_step']*1e3,1), 'us')"
done; done
