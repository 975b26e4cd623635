"""bench.py's own launcher: `python bench.py --gpus N` without a process-group environment must
start N ranks itself.  Checked here without a GPU through --rendezvous-only (the ranks form the
gloo group, sum their ranks, rank 0 prints one JSON line)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _json_lines(text):
    return [json.loads(line) for line in text.splitlines() if line.startswith("{")]


def test_gpus_2_starts_two_ranks():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=_clean_env(RT_DIST_BACKEND="gloo"),
                         capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr
    lines = _json_lines(res.stdout)
    assert len(lines) == 1                      # rank 0 alone prints
    assert lines[0]["n_gpus"] == 2 and lines[0]["rank_sum"] == 1


def test_gpus_1_stays_one_process():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--rendezvous-only"], env=_clean_env(),
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert _json_lines(res.stdout) == [{"rendezvous_only": True, "n_gpus": 1, "rank_sum": 0}]


def test_launcher_environment_is_respected():
    """Started by an external launcher (WORLD_SIZE set): no second generation of children, and a
    --gpus that contradicts the launcher is an error, not a silent single-GPU run."""
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"],
                         env=_clean_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE=3" in res.stderr


def test_a_failing_rank_fails_the_launcher():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"],
                         env=_clean_env(RT_DIST_BACKEND="no-such-backend"), capture_output=True, text=True, timeout=240)
    assert res.returncode != 0
    assert _json_lines(res.stdout) == []
