#!/bin/bash
# Copy the summaries of gpurun_out/evidence (written by tools/collect_evidence.sh on the GPU box) into profiles/ (tracked).
set -eu
cd "$(dirname "$0")/.."
E=gpurun_out/evidence
cp $E/pytest_gpu.log profiles/r01_pytest_gpu.log
for f in bf16x3 fp32 bf16 train; do tail -1 $E/bench_$f.log > profiles/r01_bench_$f.json; done
cp $E/stats/fwd_kernel_stats.csv profiles/r01_rocprof_kernel_stats_bf16x3.csv
cat $E/model_fwd.jsonl $E/model_train.jsonl $E/model_fwd_bf16.jsonl $E/model_train_bf16.jsonl > profiles/r01_model_bench.jsonl
cp $E/hbm_kernels.jsonl profiles/r01_hbm_kernels.jsonl
python tools/pmc_summary.py profiles/r01_pmc_summary.json $E profiles/r01_rocprof_kernel_stats_bf16x3.csv > /dev/null
python - <<'PY'
import json
d = json.load(open('profiles/r01_pmc_summary.json'))
k = [v for n, v in d['kernels'].items() if '<3, 2,' in n][0]
out = {"kernel": "gemm_planes_kernel<terms=3, epi=INTERLEAVE2, tile 256x256, 4-slot ring> (mode-3 GEMM of TCNet.forward)",
       "hbm_bytes_per_launch": k['hbm_read_bytes_corrected'] + k['hbm_write_bytes'], "read_bytes": k['hbm_read_bytes_corrected'],
       "write_bytes": k['hbm_write_bytes'], "l2_hit_rate": k.get('l2_hit_rate'), "effective_clock_ghz": k.get('effective_clock_ghz'),
       "source": "profiles/r01_pmc_summary.json (separate rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline`, B=256, bf16x3; FETCH_SIZE doubled per MI355X_MICROARCH.md)"}
json.dump(out, open('profiles/core_traffic.json', 'w'), indent=1)
for f in ('bf16x3', 'fp32', 'bf16', 'train'):
    b = json.loads(open('profiles/r01_bench_%s.json' % f).read())
    print(f, round(b['value'], 1), round(b['ms_per_step'], 3), (b.get('roofline') or {}).get('launch_ms'))
print(out['hbm_bytes_per_launch'], out['effective_clock_ghz'])
PY
tail -2 profiles/r01_pytest_gpu.log
