V=iccv19_vqa-cti_amd/lib/variants
CTI_HIP_LIB=$PWD/$V/libcti_hip_k2.so python -m pytest tests -m gpu -x -q -k "gru or GRU or language or models" 2>&1 | tail -2
for r in 1 2; do for v in k1 k2 k4n2 k2n3; do echo "$v c3 $(CTI_HIP_LIB=$PWD/$V/libcti_hip_$v.so python bench.py --config c3 --steps 300 2>/dev/null | tail -1 | python3 -c 'import sys,json; b=json.loads(sys.stdin.read()); print(b["value"], b["ms_per_step"])')"; done; done
for v in k1 k2 k4n2; do echo "$v c4 $(CTI_HIP_LIB=$PWD/$V/libcti_hip_$v.so python bench.py --config c4 --steps 100 2>/dev/null | tail -1 | python3 -c 'import sys,json; b=json.loads(sys.stdin.read()); print(b["value"], b["ms_per_step"])')"; done
