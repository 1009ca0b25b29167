#!/bin/bash
# end-of-round check at HEAD: smoke, the whole GPU suite, the default bench line (as the driver runs it, and with its default step count)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
python bench.py > $O/bench_f16f6.log 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.log 2>> $O/bench.err; echo "bench (driver form) rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -2 $O/smoke.log; tail -2 $O/pytest_gpu.log; for f in bench_f16f6 bench_driver_form; do tail -1 $O/$f.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$f', round(d['value']), round(d['ms_per_step'], 4), 'frac', round(r['frac'], 4), 'traffic_source', r.get('traffic_source', '')[:40], 'c3/c4', {k: round(v['value']) for k, v in d.get('configs', {}).items()})"; done
