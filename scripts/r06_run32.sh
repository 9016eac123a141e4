#!/bin/bash
# round 6, run 32: Downsample second form at C = 64 with 4 x 16 instead of 8 x 16 output pixels per workgroup (49 KB patch: three workgroups per CU instead of one)
O=gpurun_out; mkdir -p $O
for i in 1 2; do echo "=== 8 x 16 (shipped)"; python scripts/ubench_down.py 2>/dev/null | grep "128x128x64\|total"; echo "=== 4 x 16"; UFORMER_HIP_LIB=$PWD/ab/c64small/libuformer_hip.so python scripts/ubench_down.py 2>/dev/null | grep "128x128x64\|total"; done | tee $O/r06_run32_c64.txt
echo "=== batch 8 / 32"; for b in 8 32; do python scripts/ubench_down.py --batch $b 2>/dev/null | grep "128x128x64"; UFORMER_HIP_LIB=$PWD/ab/c64small/libuformer_hip.so python scripts/ubench_down.py --batch $b 2>/dev/null | grep "128x128x64"; done | tee -a $O/r06_run32_c64.txt
