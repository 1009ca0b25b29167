python bench.py --mode train --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train', round(d['value']), d['unit'], round(d['ms_per_step'],3), 'ms')"
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-fp32-exact --no-subrecords 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', round(d['value']), d['unit'], round(d['ms_per_step'],3), 'ms')"
