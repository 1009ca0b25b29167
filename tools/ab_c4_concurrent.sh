cd $GRAFT_REPO_ROOT
for i in 1 2; do
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['value'], d['ms_per_step'], d.get('parity'))"
python bench.py --config c4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('concurrent', d['value'], d['ms_per_step'], d.get('parity'))"
done
