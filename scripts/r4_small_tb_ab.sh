#!/bin/bash
# A/B on one box: the small family's traceback with the window fetched by 16-byte loads (the library as built) and by single dwords
# (the kernel file as it was before, scripts/micro/mb_small_kernels_r4_before.hip.txt, rebuilt here)
echo "== 16-byte loads"; python scripts/r4_small_tb_ab.py
cp machineboss_amd/csrc/mb_small_kernels.hip /tmp/new_kernels.hip
cp scripts/micro/mb_small_kernels_r4_before.hip.txt machineboss_amd/csrc/mb_small_kernels.hip
python -m machineboss_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
echo "== single dwords (before)"; python scripts/r4_small_tb_ab.py
cp /tmp/new_kernels.hip machineboss_amd/csrc/mb_small_kernels.hip
