#!/bin/bash
# Round evidence on the GPU box: tests, bench lines, rocprofv3 kernel stats and the separate PMC passes of the same bench command.
# Usage (through gpurun): bash tools/collect_evidence.sh [all|quick]  -> everything under gpurun_out/evidence/
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E=gpurun_out/evidence; rm -rf $E; mkdir -p $E
python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1; tail -2 $E/pytest_gpu.log
python bench.py > $E/bench_f16f6.log 2>&1; tail -1 $E/bench_f16f6.log | cut -c1-200
python bench.py --precision bf16x3 --no-cpu-baseline --no-subrecords > $E/bench_bf16x3.log 2>&1
python bench.py --precision fp32 --no-cpu-baseline --no-subrecords > $E/bench_fp32.log 2>&1
python bench.py --precision bf16 --no-cpu-baseline --no-fp32-exact --no-subrecords > $E/bench_bf16.log 2>&1
python bench.py --mode train > $E/bench_train.log 2>&1; tail -1 $E/bench_train.log | cut -c1-200
CTI_BENCH_FORCE_DIST=1 python bench.py --mode train > $E/bench_train_rccl_world1.log 2>&1; tail -1 $E/bench_train_rccl_world1.log | cut -c1-200   # two graphs around an eager RCCL all-reduce
python bench.py --config c3 > $E/bench_c3.log 2>&1; python bench.py --config c4 > $E/bench_c4.log 2>&1
CTI_BENCH_SERIAL_MODELS=1 python bench.py --config c4 > $E/bench_c4_serial.log 2>&1            # the two models one after the other (what rounds 1-3 timed)
CTI_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-fp32-exact --no-subrecords --steps 10 > $E/bench_rccl_world1.log 2>&1; tail -1 $E/bench_rccl_world1.log | cut -c1-120
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats -o fwd -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-exact --no-subrecords > $E/rocprof_stats.log 2>&1
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-exact --no-subrecords"
# one small counter group per pass: FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md counter budget); a group the
# hardware cannot collect makes rocprofv3 abort and then hang, so every pass is bounded by `timeout`
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $E/pmc$i -o p -- $B > $E/pmc$i.log 2>&1 || echo "pmc pass $i ($grp) failed/timeout"
done
# the plain-bf16 projection GEMM (cti_gemm16.hip) on its own: kernel stats + matrix-pipe busy cycles + clock, separate passes
G="python3 tools/bench_gemm16.py 20"
timeout 300 $G > $E/gemm16_vs_vendor.json 2>$E/gemm16.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/g16stats -o g -- $G > $E/g16stats.log 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $E/g16pmc$i -o p -- $G > $E/g16pmc$i.log 2>&1 || echo "gemm16 pmc pass $i ($grp) failed/timeout"
done
mkdir -p $E/g16; python tools/pmc_summary.py $E/g16/pmc_summary_gemm16.json $E $E/g16stats/g_kernel_stats.csv --only gemm16 > $E/g16/pmc_summary_gemm16.log 2>&1
if [ "${1:-all}" != "quick" ]; then
bash tools/trace_models.sh > $E/trace_models.log 2>&1
python tools/print_forward_timeline.py gpurun_out/pc_c3/m_kernel_trace.csv > $E/model_c3_timeline.txt 2>&1; python tools/print_forward_timeline.py gpurun_out/pc_c4/m_kernel_trace.csv > $E/model_c4_timeline.txt 2>&1
timeout 300 python tools/bench_model.py --steps 50 > $E/model_fwd.jsonl 2>$E/model_fwd.err
timeout 300 python tools/bench_model.py --train --steps 50 > $E/model_train.jsonl 2>$E/model_train.err
timeout 300 python tools/bench_model.py --precision bf16 --steps 50 > $E/model_fwd_bf16.jsonl 2>/dev/null
timeout 300 python tools/graph_train.py ffoe_cti 30 > $E/graph_train.jsonl 2>/dev/null; timeout 300 python tools/graph_train.py ffoe_ban 30 >> $E/graph_train.jsonl 2>/dev/null
timeout 300 python tools/bench_pools.py 30 2>/dev/null | grep kernel > $E/hbm_kernels.jsonl
timeout 300 python tools/bench_f16f6.py 256 20 > $E/mode3_f16f6_vs_bf16x3.json 2>/dev/null
timeout 120 ./tools/mb/mb_f16f6 > $E/mb_f16f6.txt 2>&1
timeout 120 ./tools/mb/mb_issue > $E/mb_issue.txt 2>&1
timeout 300 python tools/f16f6_ksweep.py > $E/f16f6_ksweep.txt 2>/dev/null
timeout 300 python tools/bench_f16f6_aside.py 2>/dev/null | grep '^{' > $E/aside_f16f6.jsonl
bash tools/prof_models.sh > $E/prof_models.log 2>&1; cp gpurun_out/pc_c3/summary.txt $E/model_c3_kernel_stats.txt; cp gpurun_out/pc_c4/summary.txt $E/model_c4_kernel_stats.txt
fi
bash tools/trace_step.sh > $E/trace_step.log 2>&1; cp gpurun_out/step_trace/timeline.txt $E/step_timeline.txt        # kernel timeline of one configs[1] step
python -m pytest tests/test_accuracy_envelope_gpu.py -q -m gpu -s > $E/accuracy_envelope.txt 2>&1                      # the guard's calibration: error and estimate per case
# summaries on the box; the raw traces (100+ MB) stay there: gpurun merges at most 64 MiB back
python tools/pmc_summary.py $E/pmc_summary.json $E $E/stats/fwd_kernel_stats.csv > $E/pmc_summary.log 2>&1
find $E -name "*kernel_trace.csv" -delete; find $E -name "*counter_collection.csv" -delete; find gpurun_out/pc_c3 gpurun_out/pc_c4 -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $E gpurun_out
