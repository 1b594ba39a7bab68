import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-rank GPU tests fork their ranks from a forkserver started HERE, before any test has initialised the GPU in
    # this process: the children are fresh processes, and nothing that has touched the GPU is ever exec'ed over.
    import multiprocessing as mp
    from multiprocessing import forkserver
    try:
        mp.get_context("forkserver")
        forkserver.ensure_running()
    except Exception:        # no forkserver on this platform: the tests that need it skip
        pass


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_loader():
    return golden


def run_ranks(target, world_size, args, timeout=600):
    """Runs `target(rank, world_size, port, *args)` in `world_size` processes forked from the forkserver; returns
    their exit codes."""
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("forkserver")
    procs = [ctx.Process(target=target, args=(r, world_size, port) + tuple(args)) for r in range(world_size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
    codes = []
    for p in procs:
        if p.is_alive():
            p.kill()             # the exact processes started above
            p.join()
            codes.append("timeout")
        else:
            codes.append(p.exitcode)
    return codes
