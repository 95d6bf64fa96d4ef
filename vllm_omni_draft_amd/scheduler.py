"""Continuous-batching loop around a SpecDecodeWorker: the caller side of the worker.

Each step follows the engine-core contract the reference documents
(/root/reference/src/doc/engine_core.md:117-122: schedule -> execute_model ->
update_from_output): running requests take one speculative step (1..k+1 tokens each), waiting
requests are admitted into free sequence slots and prefilled, and requests that reached a stop
condition are finished (their slot and cache pages are released at the start of the next step).
Host logic only; every token comes from the worker.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field
from typing import Deque, Dict, List, Optional, Sequence

from .spec_decode_worker import ExecuteModelRequest, SequenceGroupMetadata


@dataclass
class Request:
    request_id: str
    prompt_token_ids: List[int]
    max_tokens: int = 16
    eos_token_id: Optional[int] = None        # None: never stop on a token (ignore_eos)
    multi_modal_data: Optional[dict] = None   # see SequenceGroupMetadata.multi_modal_data
    output_token_ids: List[int] = field(default_factory=list)
    finish_reason: Optional[str] = None       # "stop" (eos), "length" (max_tokens or model length)

    @property
    def finished(self) -> bool:
        return self.finish_reason is not None


class SpecDecodeScheduler:
    def __init__(self, worker, max_num_seqs: int, max_model_len: int):
        """max_model_len bounds prompt + generated tokens of a request (the engine keeps the
        k + 1 tokens of slack a last step may overshoot by)."""
        if max_num_seqs < 1 or max_model_len < 2:
            raise ValueError("need max_num_seqs >= 1 and max_model_len >= 2")
        self.worker = worker
        self.max_num_seqs, self.max_model_len = max_num_seqs, max_model_len
        self.waiting: Deque[Request] = deque()
        self.running: Dict[str, Request] = {}
        self._to_free: List[str] = []
        self.steps = 0

    @classmethod
    def for_worker(cls, worker) -> "SpecDecodeScheduler":
        eng = worker.engine
        return cls(worker, eng.cfg.max_batch, eng.cfg.max_seq_len)

    def add_request(self, req: Request) -> None:
        n = len(req.prompt_token_ids)
        if n < 1 or n + 1 > self.max_model_len:
            raise ValueError(f"prompt of {n} tokens does not fit max_model_len {self.max_model_len}")
        if req.max_tokens < 1:
            raise ValueError("max_tokens must be at least 1")
        if req.request_id in self.running or any(w.request_id == req.request_id for w in self.waiting):
            raise KeyError(f"request {req.request_id} is already queued")
        self.waiting.append(req)

    def has_unfinished(self) -> bool:
        return bool(self.waiting or self.running)

    def _update(self, req: Request, new: Sequence[int]) -> None:
        """Appends a step's tokens up to the first stop condition."""
        for t in new:
            req.output_token_ids.append(int(t))
            if req.eos_token_id is not None and t == req.eos_token_id:
                req.finish_reason = "stop"
            elif (len(req.output_token_ids) >= req.max_tokens or
                  len(req.prompt_token_ids) + len(req.output_token_ids) >= self.max_model_len):
                req.finish_reason = "length"
            if req.finished:
                return

    def step(self) -> List[Request]:
        """One engine step; returns the requests that finished in it."""
        meta = [SequenceGroupMetadata(rid, False) for rid in self.running]
        admitted: List[Request] = []
        while self.waiting and len(self.running) + len(admitted) < self.max_num_seqs:
            req = self.waiting.popleft()
            admitted.append(req)
            meta.append(SequenceGroupMetadata(req.request_id, True, list(req.prompt_token_ids),
                                              multi_modal_data=req.multi_modal_data))
        if not meta and not self._to_free:
            return []
        out = self.worker.execute_model(ExecuteModelRequest(meta, finished_requests_ids=self._to_free))
        self._to_free = []
        self.steps += 1
        for req in admitted:
            self.running[req.request_id] = req
        done = []
        for rid, toks in zip(out.req_ids, out.sampled_token_ids):
            req = self.running[rid]
            self._update(req, toks)
            if req.finished:
                done.append(req)
        for req in done:
            del self.running[req.request_id]
            self._to_free.append(req.request_id)
        return done

    def drain(self) -> None:
        """Releases the slots of requests that finished in the last step."""
        if self._to_free:
            self.step()

    def generate(self, prompts: Sequence[Sequence[int]], max_tokens: int = 16,
                 eos_token_id: Optional[int] = None) -> List[Request]:
        """Runs the prompts to completion (continuous batching over max_num_seqs slots) and
        returns their requests in the order given."""
        reqs = [Request(f"gen-{self.steps}-{i}", list(p), max_tokens, eos_token_id)
                for i, p in enumerate(prompts)]
        for r in reqs:
            self.add_request(r)
        while self.has_unfinished():
            self.step()
        self.drain()
        return reqs
