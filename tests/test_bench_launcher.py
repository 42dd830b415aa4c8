"""`python bench.py --gpus N` must really run N ranks (the flag used to be parsed and ignored).  The
launcher is exercised here on CPU ranks (gloo) through bench.py's --selftest-cpu mode, whose per-rank tables come from
the host emulation of the lane code (tests/emu): two ranks over two shards must report n_gpus = 2 and exactly the
counts one rank reports over the whole range."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, expect_ok=True):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=600)
    if expect_ok:
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0
        return json.loads(lines[0])
    return p


def test_gpus_flag_starts_that_many_ranks():
    two = _run(["--gpus", "2", "--selftest-cpu", "--reads", "1500"])
    one = _run(["--gpus", "1", "--selftest-cpu", "--reads", "3000"])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["valid"] is False  # a plumbing test, never a measurement
    assert two["outcomes"] == one["outcomes"]
    assert two["outcomes"]["total_reads"] == 3000
    assert two["table_sum"] == one["table_sum"] == one["outcomes"]["matched"]


def test_gpus_flag_disagreeing_with_world_size_is_an_error():
    p = _run(["--gpus", "2", "--selftest-cpu"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"},
             expect_ok=False)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
