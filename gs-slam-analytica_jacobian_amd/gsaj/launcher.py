"""One process per GPU, started from a plain `python bench.py --gpus N` (no torch.distributed.run in front).

The parent never touches a GPU (it imports neither torch.cuda state nor the HIP library): it only picks a rendezvous port,
starts N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- the environment
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would give them -- relays rank 0's standard output as its own and
returns a non-zero exit code if any rank failed.  No reference counterpart: the reference is single-process per role
(slam.py:103-110 starts its front end, back end and GUI with torch.multiprocessing, each on the one GPU).
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_launcher(env=None):
    """True inside a rank started by torch.distributed.run or by launch_ranks()."""
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def world_mismatch(gpus, env=None):
    """An error string if the process group the environment describes is not the `--gpus` asked for, else None.
    (`--gpus N > 1` with no launcher environment is not a mismatch: the caller starts the ranks itself.)"""
    env = os.environ if env is None else env
    if not under_launcher(env):
        return None
    world = int(env["WORLD_SIZE"])
    if world != gpus:
        return "--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (gpus, world)
    rank = int(env["RANK"])
    if not 0 <= rank < world:
        return "RANK=%d outside [0, %d)" % (rank, world)
    return None


def launch_ranks(n, argv, extra_env=None, python=None, timeout=None):
    """Start `python argv...` n times, one rank each; rank 0's stdout is passed through, the other ranks' stdout is dropped
    (every rank's stderr is passed through).  Returns the exit code for the parent: 0 only if every rank returned 0, otherwise
    the first non-zero code by rank order (a rank killed by a signal counts as 128 + signal).  If a rank fails the others are
    terminated (a collective would otherwise wait for it until its own timeout)."""
    if n < 1:
        raise ValueError("launch_ranks: n must be >= 1, got %r" % (n,))
    port = free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between processes on this driver)
        if extra_env:
            env.update(extra_env)
        out = None if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([python or sys.executable] + list(argv), env=env, stdout=out))

    deadline = None if timeout is None else time.monotonic() + timeout
    pending, first_bad = set(range(n)), None
    try:
        while pending:
            for i in sorted(pending):
                rc = procs[i].poll()
                if rc is None:
                    continue
                pending.discard(i)
                if rc != 0 and first_bad is None:
                    first_bad = 128 - rc if rc < 0 else rc
                    for q in procs:  # the others would wait for the failed rank in their next collective
                        if q.poll() is None:
                            q.terminate()
            if pending:
                if deadline is not None and time.monotonic() > deadline:
                    first_bad = first_bad or 124
                    for q in procs:
                        if q.poll() is None:
                            q.kill()
                    break
                time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
        for q in procs:
            q.wait()
    return first_bad or 0
