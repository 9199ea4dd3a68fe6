"""Per-stage GPU timing with events on the current stream (bench.py's tracker-level numbers).  `mark(name)` closes the stage that
started at the previous mark; stages that end in a host read-back (NMS count, .cpu()) therefore include that wait, as they do
in the reference loop."""
import time

import torch


class StageTimer:
    def __init__(self):
        self.events, self.host = [], []

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.events.append(("", e))
        self.host.append(("", time.perf_counter()))

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.events.append((name, e))
        self.host.append((name, time.perf_counter()))

    def summary(self, per=1):
        """{stage: GPU ms between its two marks}, {stage: host ms}; both divided by `per` (frames)"""
        torch.cuda.synchronize()
        gpu, host = {}, {}
        for (n0, e0), (n1, e1) in zip(self.events[:-1], self.events[1:]):
            if n1:
                gpu[n1] = gpu.get(n1, 0.0) + e0.elapsed_time(e1) / per
        for (n0, t0), (n1, t1) in zip(self.host[:-1], self.host[1:]):
            if n1:
                host[n1] = host.get(n1, 0.0) + 1e3 * (t1 - t0) / per
        self.events, self.host = [], []
        return gpu, host


class NoTimer:
    def start(self):
        pass

    def mark(self, name):
        pass
