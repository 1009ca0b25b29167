# ordered kernel names of the LAST eager forward of a model config (rocprofv3 --kernel-trace): tools/kernel_sequence.sh c3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
c=${1:-c3}; rm -rf gpurun_out/ks_$c; mkdir -p gpurun_out/ks_$c
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ks_$c -o m -- python3 bench.py --config $c --steps 3 --warmup 2 --no-graph --no-subrecords > gpurun_out/ks_$c/log 2>&1
python3 - $c <<'PY' | tee gpurun_out/ks_$c/sequence.txt
import csv,glob,sys
f=glob.glob('gpurun_out/ks_%s/**/*kernel_trace.csv'%sys.argv[1],recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
# last forward = from the last embedding_fwd_kernel group back
idx=[i for i,n in enumerate(names) if 'embedding_fwd' in n]
start=idx[-2] if len(idx)>=2 and idx[-1]-idx[-2]<40 else idx[-1]
# find the start of the last forward: the first embedding kernel of the last group
per=[i for i in idx]
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start-3:]:
    print("%9.1f us  %7.1f us  q%-3s %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id','?'),r['Kernel_Name'][:120]))
PY
rm -rf gpurun_out/ks_$c/*/ 2>/dev/null; find gpurun_out/ks_$c -name '*.csv' -delete
