for rep in 1 2; do for v in 0 1; do
  CTI_SIBLINGS_MAIN_AUX=$v timeout 300 python bench.py --config c4 --steps 100 --warmup 20 2>&1 | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); print('c4 CTI_SIBLINGS_MAIN_AUX=$v', round(d['value']), 'samples/s', round(d['ms_per_step']*1e3,1), 'us')
except Exception as e: print('c4 CTI_SIBLINGS_MAIN_AUX=$v FAILED', t[-300:])"
done; done
