"""bench.py's own N-rank launcher (`python bench.py --gpus N` without torchrun): the parent starts the ranks before
touching any GPU API, rank 0's JSON line comes through, ranks never share a device. CPU only (gloo, world 2)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    env.update(extra)
    return env


@pytest.mark.timeout(300)
def test_launcher_world2_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher"], env=_clean_env(PV_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0 only
    out = json.loads(lines[0])
    assert out["selftest"] == "launcher" and out["n_gpus"] == 2 and out["ranks_joined"] == 2 and out["ok"] is True
    assert out["counts"] == [6, 5]            # 11 items dealt i % 2


@pytest.mark.timeout(120)
def test_more_ranks_than_devices_is_refused():
    """--gpus N beyond the visible device count exits non-zero with a clear message and runs nothing (here: 0 or 1 device)"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                       capture_output=True, text=True, timeout=110)
    assert r.returncode != 0
    assert "ranks never share a GPU" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.timeout(120)
def test_world_size_must_match_gpus():
    """under torchrun's environment --gpus must equal WORLD_SIZE (no silent single-rank run)"""
    env = _clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode != 0 and "one rank per GPU" in r.stderr
