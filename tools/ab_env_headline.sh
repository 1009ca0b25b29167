# A/B of an environment knob on the headline step (one box): bash tools/ab_env_headline.sh VAR val1 val2 ... (each value twice, interleaved)
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $var=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-exact --no-subrecords 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$var=$v', round(d['value']), 'samples/s', round(d['ms_per_step'],3), 'ms; mode-3', round(d['roofline']['launch_ms'],3), 'ms')"
done; done
