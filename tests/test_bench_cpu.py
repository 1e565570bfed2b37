"""CPU tier: bench.py's own launcher.  `python bench.py --gpus N` with no WORLD_SIZE must start N ranks as a child
`torch.distributed.run` (the parent touches neither torch nor HIP); without a GPU every rank fails loudly - there is no CPU
fallback - and the launcher hands the failure back as its exit code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_starts_the_ranks_and_fails_loudly_without_gpu(ensure_built):
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present: the launcher is exercised by the gpu tier / the driver")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "no HIP device visible" in out.stderr and "no CPU fallback" in out.stderr
    # both ranks were started by the child launcher
    assert out.stderr.count("bench.py needs an MI355X") >= 2 or "local_rank: 1" in out.stderr or "rank: 1" in out.stderr
    assert out.stdout.strip() == ""          # no JSON line from a run that measured nothing


def test_single_gpu_invocation_does_not_spawn(ensure_built):
    """--gpus 1 runs in-process (and fails loudly here: no device)."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "no HIP device visible" in out.stderr and "torch.distributed.run" not in out.stderr
