#!/bin/bash
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_bwd.py -m gpu -q -k "downsample_input_gradient" 2>&1 | tail -30) > $O/r06_run39_pytest.txt; tail -4 $O/r06_run39_pytest.txt
(timeout 900 python -m pytest tests/test_gpu_traj.py -m gpu -q 2>&1 | tail -60) > $O/r06_run39_traj.txt; tail -4 $O/r06_run39_traj.txt
(UF_VARIANT="downdx=1" timeout 900 python -m pytest tests/test_gpu_traj.py -m gpu -q 2>&1 | tail -4) | tee $O/r06_run39_traj_old.txt
