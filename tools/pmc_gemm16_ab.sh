cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in $VARS; do
  export CTI_HIP_LIB=$R/iccv19_vqa-cti_amd/lib/variants/libcti_hip_$v.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_$v -- python3 $R/tools/pmc_gemm16_one.py 9216 3072 2048 > /dev/null 2>&1
  f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
  echo "== $v $f"
  python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm16' in r['Kernel_Name']: d[r['Counter_Name']]['v'].append(float(r['Counter_Value']))
for k,v in d.items(): print(k, sum(v['v'])/len(v['v']), len(v['v']))
PY
done
