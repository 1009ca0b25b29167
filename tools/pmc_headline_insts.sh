# instruction-class counters of the headline step's kernels (one rocprofv3 --pmc pass; per kernel: mean per launch): bash tools/pmc_headline_insts.sh [variant]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$1" ] && export CTI_HIP_LIB=$R/iccv19_vqa-cti_amd/lib/variants/libcti_hip_$1.so
rm -rf /tmp/pmc_h
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_h -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fp32-exact --no-subrecords > /dev/null 2>&1
f=$(find /tmp/pmc_h -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1].get('SQ_WAVE_CYCLES',[0]))):
    n=len(v['SQ_WAVE_CYCLES'])
    print(k, n, {c: round(sum(x)/len(x)/1e6,2) for c,x in v.items()})
PY
