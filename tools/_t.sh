for rep in 1 2; do for v in 1000000 4096 2048; do
  CTI_SKINNY_WIDE_K=$v python bench.py --config c4 --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 CTI_SKINNY_WIDE_K=$v', round(d['value']), 'samples/s', round(d['ms_per_step']*1e3,1), 'us')"
done; done
