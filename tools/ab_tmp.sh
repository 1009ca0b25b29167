python -m pytest tests -m gpu -x -q -k "c2 or f16f6 or guard or mbuild" 2>&1 | tail -2
CTI_NO_AUX_STREAM=1 bash tools/prof_stats.sh 2>&1 | grep "uild_mfma_f6\|samples/sec" | cut -c1-200
for i in 1 2 3; do echo "bench: $(python bench.py --no-cpu-baseline --no-fp32-exact --no-subrecords --steps 200 2>/dev/null | tail -1 | python3 -c 'import sys,json; b=json.loads(sys.stdin.read()); print(b["value"], b["ms_per_step"])')"; done
