python tools/bench_gru.py 30 2>&1 | grep '^{'
