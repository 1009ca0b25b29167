# kernel time of the few-answer core per library variant: bash tools/ab_core_small.sh <variant> ...
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in "$@"; do
  export CTI_HIP_LIB=$R/iccv19_vqa-cti_amd/lib/variants/libcti_hip_$v.so
  rm -rf /tmp/pcs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pcs -o m -- python3 $R/tools/bench_core_small.py bf16 > /dev/null 2>&1
  f=$(find /tmp/pcs -name "*kernel_stats.csv" | head -1)
  python3 - "$v" "$f" <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[2])):
    if 'core_small' in r[0]: print(sys.argv[1], r[0][30:75], 'calls', r[1], 'avg_us %.1f' % (float(r[3])/1e3), 'min %.1f max %.1f' % (float(r[5])/1e3 if len(r)>5 else 0, float(r[6])/1e3 if len(r)>6 else 0))
PY
done
