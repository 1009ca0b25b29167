cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pt; mkdir -p gpurun_out/pt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pt -o m -- python3 bench.py --mode train --steps 20 --warmup 5 --no-graph > gpurun_out/pt/log 2>&1
tail -1 gpurun_out/pt/log | cut -c1-300
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/pt/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.reader(open(f)))[1:]
tot=sum(float(r[2]) for r in rows)
print("total kernel ms over run", tot/1e6)
for r in rows[:22]:
    print(r[0][:110].ljust(112), r[1], "tot %.2f ms avg %.1f us %s%%"%(float(r[2])/1e6, float(r[3])/1e3, r[4]))
PY
