"""Host-side waits on device work.

`hipEventSynchronize` / `hipStreamSynchronize` spin briefly and then BLOCK on the completion interrupt.  Where interrupt delivery is
slow (observed on this pool's virtualised hosts in round 3: a 7 ms wait intermittently took 67-76 ms, and the streaming config C5
went from 266 ms to exactly 600 ms per batch with the same binaries, profiles/r3e_c5_sched_probe.log) every wait of the generate /
stream loop can cost tens of milliseconds.  These helpers POLL the event instead (hipEventQuery reads the completion signal; no
interrupt involved), yielding the GIL between polls so that worker threads (the Exp(1) feeder of the unseeded host mode) keep running.
CTTS_SPIN_WAIT=0 restores the blocking calls.

The spin is BOUNDED: a wait that has polled for `SPIN_BUDGET_S` (default 20 ms -- longer than any decode chunk, shorter than a batch's
acoustic decode) backs off to short timed sleeps, so a long wait (a waveform the side stream is still decoding, several ranks or
serving threads sharing one CPU quota) does not burn a core per waiter.
"""
from __future__ import annotations

import os
import time

import torch

SPIN = os.environ.get("CTTS_SPIN_WAIT", "1") != "0"
SPIN_BUDGET_S = float(os.environ.get("CTTS_SPIN_BUDGET_MS", "20")) * 1e-3


def wait_event(ev) -> None:
    """returns once everything recorded before `ev` has completed"""
    if not SPIN:
        ev.synchronize()
        return
    t0 = time.perf_counter()
    while not ev.query():
        if time.perf_counter() - t0 < SPIN_BUDGET_S:
            time.sleep(0)     # release the GIL; no timed sleep (a 50 us sleep would already be 10 % of a decode step)
        else:
            time.sleep(2e-4)  # a long wait: stop burning the core (the quota is shared with the feeder thread and the other ranks)


def wait_stream(stream: "torch.cuda.Stream") -> None:
    """returns once everything enqueued on `stream` so far has completed"""
    if not SPIN:
        stream.synchronize()
        return
    ev = torch.cuda.Event()
    ev.record(stream)
    wait_event(ev)
