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
    """Reads the engine's counters at most once per interval; reading costs a
    small device-to-host copy, so it stays off the per-step path."""

    def __init__(self, engine, collect_interval_s: float = 5.0, timer=time.monotonic):
        self.engine = engine
        self.interval = collect_interval_s
        self._timer = timer
        self._last = timer()

    def maybe_collect(self, force: bool = False) -> Optional[SpecDecodeWorkerMetrics]:
        now = self._timer()
        if not force and now - self._last < self.interval:
            return None
        self._last = now
        m = self.engine.metrics()
        return SpecDecodeWorkerMetrics.from_counts(m["accepted"], m["emitted"], m["draft"],
                                                   self.engine.k)
