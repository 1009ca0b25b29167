cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in c3 c4; do rm -rf gpurun_out/pc_$c; mkdir -p gpurun_out/pc_$c
CTI_BENCH_SERIAL_MODELS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pc_$c -o m -- python3 bench.py --config $c --steps 12 --warmup 3 --no-graph > gpurun_out/pc_$c/log 2>&1
done
