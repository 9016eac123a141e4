"""`python bench.py --gpus N` without a torchrun environment re-runs itself as N ranks under torch.distributed.run (VERDICT r04 "next" 4):
the launcher command line, the decision when to self-launch, and the clear error on a box with fewer GPUs -- all on CPU."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_launcher_argv_is_one_process_per_gpu_on_loopback():
    argv = bench.launcher_argv(8, 29555, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[argv.index("--master-port") + 1] == "29555"
    i = argv.index(os.path.join(REPO, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]          # the user's own flags, verbatim


def test_self_launch_only_without_a_torchrun_environment():
    assert bench.needs_self_launch(8, {})
    assert bench.needs_self_launch(2, {"WORLD_SIZE": "1"})
    assert not bench.needs_self_launch(1, {})
    assert not bench.needs_self_launch(8, {"RANK": "3", "WORLD_SIZE": "8"})            # already a rank of torch.distributed.run
    assert not bench.needs_self_launch(8, {"WORLD_SIZE": "8"})


def test_too_few_gpus_is_a_clear_error_not_a_hang():
    with pytest.raises(SystemExit) as e:
        bench.self_launch(8, ["--gpus", "8"], device_count=1)
    assert "--gpus 8" in str(e.value) and "1 GPU" in str(e.value)


def test_free_port_is_bindable():
    p = bench.free_port()
    assert 1024 < p < 65536
