"""Start one process per GPU on this node and wait for them.

Utterances are independent (every decode starts from [(0.0, [BOS], "")],
scripts/ns_utils.py:214), so a multi-GPU run is N copies of the same program,
each told its rank through the usual torch.distributed environment variables
(RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  The parent never
touches a GPU and never replaces itself with another program: it only starts
children, relays rank 0's standard output and returns the worst exit status.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # RCCL shares buffers between the ranks of a node through IPC handles; hosts whose driver supports only dmabuf
    # handles fail in the legacy mode ("hipIpcGetMemHandle: invalid argument").  A value already set wins.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(cmd: Sequence[str], world: int, *, timeout: Optional[float] = None, env: Optional[dict] = None,
                poll_s: float = 0.2) -> int:
    """Run `cmd` once per rank (world copies) and return the largest exit status.

    Rank 0 inherits this process' stdout (so a JSON line it prints is this program's output); the other
    ranks' stdout goes to stderr.  When one rank fails or the time is up, the ranks still running are
    terminated (by their own process ids) so that nobody waits in a collective forever.
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(list(cmd), env=rank_env(r, world, port, env),
                                      stdout=None if r == 0 else sys.stderr))
    t0 = time.monotonic()
    worst = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [c for c in codes if c not in (None, 0)]
            if failed or all(c is not None for c in codes):
                worst = max([abs(c) for c in failed], default=0)
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                worst = 124
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return worst
