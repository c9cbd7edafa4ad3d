#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4j
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "count or tiled or medium or envelope or composed or baseline_configs or traceback or viterbi or 128_step or rolling or pipelined" > gpurun_out/r4j/pytest.log 2>&1; tail -4 gpurun_out/r4j/pytest.log
for nb in 1 0; do
  MB_JIT_NEIGHBOUR_SYNC=$nb python scripts/mode_probe.py psw2dna 64 487 2000 2>&1 | tail -1
  MB_JIT_NEIGHBOUR_SYNC=$nb python scripts/mode_probe.py c4b 32 487 3000 roll,vit,cnt 2>&1 | tail -1
done
MB_JIT_NEIGHBOUR_SYNC=1 python scripts/mode_probe.py psw2dna 21 487 10000 cnt 2>&1 | tail -1
MB_JIT_NEIGHBOUR_SYNC=1 python scripts/mode_probe.py psw2dna 256 487 10000 vit 2>&1 | tail -1
