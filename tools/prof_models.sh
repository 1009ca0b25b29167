# rocprofv3 kernel stats of the c3 / c4 full-model forwards (eager launches: a hipGraph replay hides the kernel names from the trace);
# summaries -> gpurun_out/pc_<cfg>/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in c4 c3; do rm -rf gpurun_out/pc_$c; mkdir -p gpurun_out/pc_$c
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc_$c -o m -- python3 bench.py --config $c --steps 30 --warmup 5 --no-graph > gpurun_out/pc_$c/log 2>&1
tail -1 gpurun_out/pc_$c/log | cut -c1-200
python3 - $c <<'PY' | tee gpurun_out/pc_$c/summary.txt
import csv,glob,sys
f=glob.glob('gpurun_out/pc_%s/**/*kernel_stats.csv'%sys.argv[1],recursive=True)[0]
rows=list(csv.reader(open(f)))[1:]
setup=sum(float(r[2]) for r in rows if 'copyBuffer' in r[0])     # the model's .to(device): host-to-device staging copies during set-up, not the forward
tot=sum(float(r[2]) for r in rows)-setup
print("total kernel ms over run (without the set-up copies, %.2f ms)" % (setup/1e6), tot/1e6, "per forward (35 forwards + 3 warm-ups of the capture path)", tot/1e6/35)
for r in rows[:40]:
    print(r[0][:110].ljust(112), r[1], "tot %.2f ms avg %.1f us %s%%"%(float(r[2])/1e6, float(r[3])/1e3, r[4]))
PY
done
