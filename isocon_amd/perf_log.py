"""Per-call performance record of the hot-path wrappers (auxiliary subsystems: the reference has none).

ISOCON_PERF_LOG=<path> appends one JSON line per device-backed call: which public function, how many sequences / pairs, wall
seconds, and the kernels' statistics block where the call has one.  Unset: nothing is recorded, nothing is paid."""
from __future__ import annotations

import json
import os
import time


class call(object):
    def __init__(self, what, **info):
        self.path = os.environ.get("ISOCON_PERF_LOG")
        self.rec = dict(call=what, **info) if self.path else None

    def __enter__(self):
        self.t0 = time.perf_counter()
        return self

    def add(self, **info):
        if self.rec is not None:
            self.rec.update(info)

    def __exit__(self, exc_type, exc, tb):
        if self.rec is not None:
            self.rec["wall_s"] = time.perf_counter() - self.t0
            self.rec["ok"] = exc_type is None
            try:
                with open(self.path, "a") as f:
                    f.write(json.dumps(self.rec, default=float) + "\n")
            except OSError:
                pass
        return False
