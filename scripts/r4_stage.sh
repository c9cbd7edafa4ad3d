#!/bin/bash
# round 4: stage-wise load batching of the specialised tile kernels, A/B on every mode of the headline machine and on C4b
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "count or tiled or medium or envelope or fuzz or composed or baseline_configs or traceback or viterbi" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
for sl in 1 0; do
  MB_JIT_STAGE_LOADS=$sl python scripts/mode_probe.py psw2dna 64 487 2000 2>&1 | tail -1
done
MB_MEDIUM_COUNT_FLAT=0 python scripts/mode_probe.py psw2dna 64 487 2000 cnt 2>&1 | tail -1
MB_JIT_STAGE_MAXLOADS=16 python scripts/mode_probe.py psw2dna 64 487 2000 2>&1 | tail -1
MB_JIT_STAGE_MAXLOADS=64 python scripts/mode_probe.py psw2dna 64 487 2000 2>&1 | tail -1
python scripts/mode_probe.py psw2dna 21 487 10000 cnt 2>&1 | tail -1
python scripts/mode_probe.py psw2dna 256 487 10000 fwd 2>&1 | tail -1
for sl in 1 0; do
  MB_JIT_STAGE_LOADS=$sl python scripts/mode_probe.py c4b 32 487 3000 2>&1 | tail -1
done
