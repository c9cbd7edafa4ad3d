#!/bin/bash
# round 4: the flat count sweep of the tiled family against the levelled one (parity tests first, then the probe under both)
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "count or tiled or medium or envelope or fuzz or composed or baseline_configs" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
for flat in 1 0; do
  MB_MEDIUM_COUNT_FLAT=$flat python scripts/count_probe.py 2>&1 | tail -1
  MB_MEDIUM_COUNT_FLAT=$flat python scripts/count_probe.py 21 487 10000 2>&1 | tail -1
done
MB_MEDIUM_COUNT_G=4 python scripts/count_probe.py 2>&1 | tail -1
MB_MEDIUM_COUNT_G=1 python scripts/count_probe.py 2>&1 | tail -1
MB_MEDIUM_TS=128 python scripts/count_probe.py 2>&1 | tail -1
MB_MEDIUM_STREAMS=1 python scripts/count_probe.py 2>&1 | tail -1
