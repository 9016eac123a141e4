#!/bin/bash
# round 5, run 5: small batches -- stages with fewer windows than CUs on the 3-kernel attention path (UF_UNFUSE_BELOW=n), whose GEMMs tile over the chip
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for n in 0 20 70 130 300; do echo "UF_UNFUSE_BELOW=$n"; UF_UNFUSE_BELOW=$n python scripts/r05_smallbatch.py 2>&1 | grep "^batch"; done | tee $O/r05_run5_unfuse.txt
