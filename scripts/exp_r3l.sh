#!/bin/bash
MB_WIDE_VERBOSE=1 python scripts/bench_onetape.py 20 64 2000 v 2>&1 | grep -E "exact closure|wide \(max\)|viterbi|max\) program" | cut -c1-250
for b in 256 512 1024 2048 4096 8192; do echo "budget $b"; MB_WIDE_VITERBI_CHAIN_BUDGET=$b python scripts/bench_onetape.py 20 64 2000 v 2>&1 | grep -E "viterbi" | cut -c1-200; done
MB_WIDE_VITERBI_CHAINS=0 python scripts/bench_onetape.py 20 64 2000 v 2>&1 | grep -E "viterbi fill" | cut -c1-200
python scripts/bench_onetape.py 20 256 2000 v 2>&1 | grep -E "viterbi fill" | cut -c1-200
python -m pytest tests -m gpu -q -k "one_tape or config5 or hmmer or randomised" 2>&1 | tail -3
bash scripts/profile_modes.sh r03 counts4 viterbi4 > gpurun_out/profile_modes_r03.log 2>&1; tail -2 gpurun_out/profile_modes_r03.log | cut -c1-300
