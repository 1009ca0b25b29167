#!/bin/bash
# Copy the summaries of gpurun_out/evidence (written by tools/collect_evidence.sh on the GPU box) into profiles/ (tracked).
set -eu
cd "$(dirname "$0")/.."
E=gpurun_out/evidence; R=${1:-r06}
cp $E/pytest_gpu.log profiles/${R}_pytest_gpu.log
for f in f16f6 bf16x3 fp32 bf16 train train_rccl_world1 c3 c4 c4_serial rccl_world1; do [ -s $E/bench_$f.log ] && tail -1 $E/bench_$f.log > profiles/${R}_bench_$f.json; done
cp $E/stats/fwd_kernel_stats.csv profiles/${R}_rocprof_kernel_stats_f16f6.csv
cat $E/model_fwd.jsonl $E/model_train.jsonl $E/model_fwd_bf16.jsonl > profiles/${R}_model_bench.jsonl 2>/dev/null || true
for f in graph_train.jsonl hbm_kernels.jsonl mode3_f16f6_vs_bf16x3.json mb_f16f6.txt mb_issue.txt f16f6_ksweep.txt aside_f16f6.jsonl model_c3_kernel_stats.txt model_c4_kernel_stats.txt model_c3_timeline.txt model_c4_timeline.txt gemm16_vs_vendor.json step_timeline.txt accuracy_envelope.txt; do [ -s $E/$f ] && cp $E/$f profiles/${R}_$f; done
cp $E/pmc_summary.json profiles/${R}_pmc_summary.json
cp $E/g16/pmc_summary_gemm16.json profiles/${R}_pmc_summary_gemm16.json 2>/dev/null || true
cp $E/g16stats/g_kernel_stats.csv profiles/${R}_rocprof_kernel_stats_gemm16.csv 2>/dev/null || true
python - "$R" <<'PY'
import json, sys
R = sys.argv[1]
d = json.load(open('profiles/%s_pmc_summary.json' % R))
k = [v for n, v in d['kernels'].items() if n.startswith('gemm_f16f6_kernel<2')][0]
out = {"kernel": "gemm_f16f6_kernel<epi=INTERLEAVE2, tile 256x192, 4-slot ring of 32-deep K blocks, continuous stream> (mode-3 GEMM of TCNet.forward, f16f6 mode)",
       "variant": "gemm_f16f6_kernel<epi=INTERLEAVE2, tile 256x192>", "batch": 256,
       "hbm_bytes_per_launch": k['hbm_read_bytes_corrected'] + k['hbm_write_bytes'], "read_bytes": k['hbm_read_bytes_corrected'],
       "write_bytes": k['hbm_write_bytes'], "l2_hit_rate": k.get('l2_hit_rate'), "effective_clock_ghz": k.get('effective_clock_ghz'),
       "mfma_busy_frac": (k['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (k['_dur_us'] * 1e3 * k.get('effective_clock_ghz', 2.0))) if 'SQ_VALU_MFMA_BUSY_CYCLES' in k and '_dur_us' in k else None,
       "source": "profiles/%s_pmc_summary.json (separate rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-exact`, B=256, f16f6; FETCH_SIZE doubled per MI355X_MICROARCH.md)" % R}
json.dump(out, open('profiles/core_traffic.json', 'w'), indent=1)
for f in ('f16f6', 'bf16x3', 'fp32', 'bf16', 'train', 'c3', 'c4'):
    try:
        b = json.loads(open('profiles/%s_bench_%s.json' % (R, f)).read())
        print(f, round(b['value'], 1), round(b['ms_per_step'], 3), (b.get('roofline') or {}).get('launch_ms'))
    except Exception as e:
        print(f, "missing", e)
print(out['hbm_bytes_per_launch'], out['effective_clock_ghz'], out['mfma_busy_frac'])
PY
tail -2 profiles/${R}_pytest_gpu.log
