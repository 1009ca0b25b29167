run() { python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-fp32-exact 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ks={r['kernel'][:28]: round(r['ms'],3) for r in d['roofline_kernels']['kernels']}
print('ms_per_step %.4f  samples/s %.0f  %s' % (d['ms_per_step'], d['value'], ks))"; }
for i in 1 2; do
  echo -n "qrows=1 "; CTI_F6_QROWS=1 run
  echo -n "qrows=0 "; CTI_F6_QROWS=0 run
done
