"""Speculative-decoding metrics from the device counters the acceptance
kernels maintain (accepted / emitted / draft tokens).

Definitions follow vLLM's `spec_decode.metrics` (not in the reference tree):
  draft_acceptance_rate = accepted / draft
  system_efficiency     = emitted / (steps * (k + 1))
  accept_len            = accepted / steps            (steps = draft / k)
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Optional


@dataclass
class SpecDecodeWorkerMetrics:
    draft_acceptance_rate: float
    system_efficiency: float
    accept_len: float
    draft_tokens: int
    emitted_tokens: int
    accepted_tokens: int
    num_spec_tokens: int

    @staticmethod
    def from_counts(accepted: int, emitted: int, draft: int, k: int) -> "SpecDecodeWorkerMetrics":
        steps = draft // k if k else 0
        return SpecDecodeWorkerMetrics(
            draft_acceptance_rate=accepted / draft if draft else float("nan"),
            system_efficiency=emitted / (steps * (k + 1)) if steps else float("nan"),
            accept_len=accepted / steps if steps else float("nan"),
            draft_tokens=draft, emitted_tokens=emitted, accepted_tokens=accepted, num_spec_tokens=k)


class AsyncMetricsCollector:
    """Reads the engine's counters at most once per interval without stalling a step: when the
    interval has elapsed the counters are copied to pinned host memory on a side stream
    (engine.metrics_begin), and a later call hands them out once the copy has landed
    (engine.metrics_poll), as upstream's collector of the same name does.  force=True, or an
    engine without the two calls, reads synchronously."""

    def __init__(self, engine, collect_interval_s: float = 5.0, timer=time.monotonic):
        self.engine = engine
        self.interval = collect_interval_s
        self._timer = timer
        self._last = timer()
        self._copying = False

    def _from(self, m: dict) -> SpecDecodeWorkerMetrics:
        return SpecDecodeWorkerMetrics.from_counts(m["accepted"], m["emitted"], m["draft"],
                                                   self.engine.k)

    def maybe_collect(self, force: bool = False) -> Optional[SpecDecodeWorkerMetrics]:
        now = self._timer()
        if force or not hasattr(self.engine, "metrics_begin"):
            if not force and now - self._last < self.interval:
                return None
            self._last, self._copying = now, False
            return self._from(self.engine.metrics())
        if self._copying:
            m = self.engine.metrics_poll()
            if m is None:
                return None
            self._copying = False
            return self._from(m)
        if now - self._last >= self.interval:
            self._last, self._copying = now, True
            self.engine.metrics_begin()
        return None
