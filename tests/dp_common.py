"""Shared by the data-parallel test workers: a stage log and a watchdog, so that a rank that stalls names the stage it
reached and the frame it is stuck in (the test captures the ranks' output)."""
import faulthandler
import os
import sys
import time

_T0 = time.time()


def stage(msg):
    """One flushed, timestamped line per stage; re-arms the watchdog (all threads' tracebacks after 60 s without a new stage)."""
    print(f'[rank {os.environ.get("RANK", "?")} +{time.time() - _T0:6.2f}s] {msg}', file=sys.stderr, flush=True)
    faulthandler.dump_traceback_later(60, repeat=False, file=sys.stderr, exit=False)


def done():
    faulthandler.cancel_dump_traceback_later()
