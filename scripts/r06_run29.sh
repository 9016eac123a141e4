#!/bin/bash
# round 6, run 29: timing experiment (results wrong by construction): the Downsample second form reading its weights as if they were fragment-major
O=gpurun_out; mkdir -p $O
(echo "=== row-major weights (shipped)"; UF_VARIANT="down=2" python scripts/ubench_down.py 2>/dev/null; echo "=== fragment-major addressing (timing only)"; UF_VARIANT="down=2" UFORMER_HIP_LIB=$PWD/ab/fmtiming/libuformer_hip.so python scripts/ubench_down.py 2>/dev/null
 echo "=== batch 32"; UF_VARIANT="down=2" python scripts/ubench_down.py --batch 32 2>/dev/null;  UF_VARIANT="down=2" UFORMER_HIP_LIB=$PWD/ab/fmtiming/libuformer_hip.so python scripts/ubench_down.py --batch 32 2>/dev/null) | tee $O/r06_run29_fm.txt
