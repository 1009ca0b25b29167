cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/s3 && mkdir -p gpurun_out/s3 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s3 -o fwd -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-exact --no-subrecords > gpurun_out/s3/log 2>&1; tail -1 gpurun_out/s3/log | cut -c1-160; python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/s3/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(f)))[1:10]:
    print(r[0][30:110].ljust(82), r[1], "avg %.3f ms min %.3f max %.3f"%(float(r[3])/1e6, float(r[5])/1e6, float(r[6])/1e6))
PY
