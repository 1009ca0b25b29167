for i in 1 2; do
for k in 1 0; do echo "no_mfma=$k: $(CTI_NO_MBUILD_BWD_MFMA=$k python bench.py --mode train --steps 100 2>/dev/null | tail -1 | python3 -c 'import sys,json; b=json.loads(sys.stdin.read()); print(b["value"], b["ms_per_step"])')"; done; done
bash tools/prof_train.sh 2>&1 | grep -i "mbuild\|total kernel"
