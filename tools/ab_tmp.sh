python -m pytest tests/test_edge_gpu.py -m gpu -x -q -k "rank_net" 2>&1 | tail -3
for i in 1 2; do for k in 1 0; do echo "no_rn_mfma=$k: $(CTI_NO_RANKNETS_MFMA=$k python bench.py --mode train --steps 100 2>/dev/null | tail -1 | python3 -c 'import sys,json; b=json.loads(sys.stdin.read()); print(b["value"], b["ms_per_step"])')"; done; done
bash tools/prof_train.sh 2>&1 | grep -i "rn_\|total kernel"
