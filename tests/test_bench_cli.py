"""bench.py's launch contract: a line never claims more GPUs than ranks that ran.
CPU-only: the refusal paths end before any GPU call."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_n_without_launcher_refuses_when_the_node_has_fewer_gpus():
    sys.path.insert(0, ROOT)
    import bench
    if bench.visible_gpus() >= 2:
        import pytest
        pytest.skip("node has the GPUs: the spawn path would really run")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2
    assert "n_gpus" not in r.stdout          # no JSON line at all
    assert "refusing" in r.stderr


def test_gpus_must_equal_the_launchers_world_size():
    r = _run(["--gpus", "8"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "n_gpus" not in r.stdout and "refusing" in r.stderr
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"})
    assert r.returncode == 2 and r.stdout.strip() == ""


def test_gpus_below_one_is_an_argument_error():
    r = _run(["--gpus", "0"])
    assert r.returncode != 0 and r.stdout.strip() == ""
