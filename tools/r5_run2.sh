#!/bin/bash
# round 5, GPU call 2: guard estimate with whole-call maxima (threshold calibration), a-side chain ablation (256 x 64 tiles, no stores)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_2; mkdir -p $O
python -m pytest tests/test_accuracy_envelope_gpu.py -q -s -m gpu > $O/envelope.log 2>&1; echo "envelope rc=$?" >> $O/summary.txt
python tools/rho_of_bench_inputs.py > $O/rho_bench.txt 2>&1
( echo "# rank-net shape (512 x 801024 x 512)"; python tools/tune_f16f6_planes.py run 4; echo "# Tucker shape (512 x 801024 x 300)"; CTI_TUNE_K=300 python tools/tune_f16f6_planes.py run 4 ) > $O/chain_ablation.txt 2>&1
grep -v Warn $O/envelope.log | tail -40; tail -4 $O/rho_bench.txt; cat $O/chain_ablation.txt
