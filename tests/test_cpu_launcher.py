"""CPU: gsaj.launcher -- what `python bench.py --gpus N` uses to start its N ranks -- and bench.py's refusal to run with a
process group that is not the one --gpus asks for.  No GPU: the children are trivial Python commands, bench.py is stopped by its
own checks before it touches a device."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-slam-analytica_jacobian_amd"))
from gsaj import launcher  # noqa: E402


def test_ranks_get_the_launcher_environment_and_rank0_stdout_is_relayed(tmp_path, capfd):
    code = ("import os; r = os.environ['RANK']; open(os.path.join(%r, 'rank' + r), 'w').write(' '.join(os.environ[k] for k in "
            "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'))); print('line of rank ' + r)" % str(tmp_path))
    assert launcher.launch_ranks(3, ["-c", code]) == 0
    seen = [open(tmp_path / ("rank%d" % r)).read().split() for r in range(3)]
    assert [s[:4] for s in seen] == [[str(r), str(r), "3", "127.0.0.1"] for r in range(3)]
    assert len({s[4] for s in seen}) == 1 and int(seen[0][4]) > 0  # one rendezvous port for all
    out = capfd.readouterr().out
    assert "line of rank 0" in out and "line of rank 1" not in out and "line of rank 2" not in out


def test_a_failing_rank_fails_the_job_and_the_others_are_stopped():
    code = "import os, sys, time; r = int(os.environ['RANK']); time.sleep(0 if r == 1 else 30); sys.exit(7 if r == 1 else 0)"
    import time
    t0 = time.monotonic()
    assert launcher.launch_ranks(3, ["-c", code]) == 7
    assert time.monotonic() - t0 < 20  # the ranks that would have waited were terminated
    assert launcher.launch_ranks(2, ["-c", "import time; time.sleep(30)"], timeout=0.5) == 124
    with pytest.raises(ValueError):
        launcher.launch_ranks(0, ["-c", "pass"])


def test_world_mismatch_is_named():
    assert launcher.world_mismatch(2, {}) is None                                  # no launcher: bench.py starts the ranks itself
    assert launcher.world_mismatch(2, {"WORLD_SIZE": "2", "RANK": "1"}) is None
    assert "WORLD_SIZE=1" in launcher.world_mismatch(2, {"WORLD_SIZE": "1", "RANK": "0"})
    assert "WORLD_SIZE=4" in launcher.world_mismatch(1, {"WORLD_SIZE": "4", "RANK": "0"})
    assert "RANK=5" in launcher.world_mismatch(2, {"WORLD_SIZE": "2", "RANK": "5"})


def _bench(args, env_extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_bench_refuses_a_world_that_is_not_the_gpus_asked_for():
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "refusing to run" in r.stderr and r.stdout.strip() == ""
    r = _bench(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "refusing to run" in r.stderr
    r = _bench(["--gpus", "8", "--scaling", "strong", "--views", "12"], {})
    assert r.returncode == 2 and "multiple of --gpus" in r.stderr   # argparse error, before anything is started


def test_bench_starts_its_ranks_itself_and_reports_their_failure():
    """`python bench.py --gpus 2` with no launcher in front: two ranks are started; in this container they have no GPU and stop at
    bench.py's own assertion -- which must come back as a non-zero exit code of the parent, with nothing on stdout."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs an MI355X" in r.stderr or "no GPU" in r.stderr
