"""The speculative-decoding plugin surface: SpecDecodeWorker, Proposer, Scorer,
RejectionSampler and their data carriers.

Names and call shapes follow vLLM's V0 `vllm.spec_decode` package (the
surface the task names); that package is not part of the reference tree
(/root/reference/pyproject.toml:31 only depends on `vllm`), so the shapes are
restated here.  The worker plugs into an engine the way the reference's custom
workers do: an executor sets `parallel_config.worker_cls` to this class
(/root/reference/src/qwen_image_gen/executor.py:14-26), the engine constructs
it with `(vllm_config, local_rank, rank, distributed_init_method,
is_driver_worker)` (/root/reference/src/qwen_image_gen/worker.py:19-29), calls
`load_model()` and then `execute_model(...)` once per step
(/root/reference/src/doc/engine_core.md:117-122).  `execute_model` returns one
variable-length token list per request — `ModelRunnerOutput.sampled_token_ids`
(/root/reference/src/doc/engine_core.md:323-324).

Everything heavy happens in SpecEngine (engine.py) on the device; these classes
only sequence it and hand out views of its buffers.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import Dict, FrozenSet, List, Optional, Sequence, Set

import torch
from torch import nn

from . import _rt, ops
from .engine import EngineConfig, SpecEngine
from .metrics import AsyncMetricsCollector, SpecDecodeWorkerMetrics
from .weights import ModelWeights


# --------------------------------------------------------------------------- carriers
@dataclass
class SamplingParams:
    # None: the worker's temperature.  A sampling worker (SpecDecodeConfig.temperature > 0) serves
    # every request at its own temperature, greedy requests (0) included: the step's softmax
    # kernels read one temperature per sequence.  A greedy worker has no sampling buffers and
    # refuses a request that asks to sample.
    temperature: Optional[float] = None
    # Stop conditions belong to the caller's scheduler (scheduler.SpecDecodeScheduler enforces
    # max_tokens and the end-of-sequence token); the worker emits tokens until a request is
    # listed as finished, like a V1 model runner.
    max_tokens: int = 16
    # The request's own random stream (sampling): every sequence draws its uniforms from
    # hash(seed, step, index) on the device, so a seeded request samples the same tokens alone
    # and among any batch mates.  None: a stream derived from SpecDecodeConfig.seed and the order
    # of admission.  Greedy workers ignore it.
    seed: Optional[int] = None
    # Filters of the distribution a sampling request draws from (0 / -1, 1.0, 0.0: off); applied
    # to the target's probabilities before acceptance, in upstream's order top-k, min-p, top-p.
    # Greedy decoding is unaffected by them (the most likely token always survives).
    top_k: int = 0
    top_p: float = 1.0
    min_p: float = 0.0
    ignore_eos: bool = True
    # Logit masks, applied to the target's logits before every argmax / softmax of the request
    # (sd_add_logit_bias): additive values per token, a whitelist (everything else is banned),
    # banned single tokens (upstream's bad_words of one token each).
    logit_bias: Optional[Dict[int, float]] = None
    allowed_token_ids: Optional[List[int]] = None
    bad_token_ids: Optional[List[int]] = None
    # None: no log-probabilities.  n >= 0: ModelRunnerOutput.logprobs carries, for every token
    # the request emits, its log-probability under the target's raw logits (no temperature, no
    # filters: upstream's default "raw_logprobs"), its rank, and the n most likely tokens.
    logprobs: Optional[int] = None


@dataclass
class SequenceGroupMetadata:
    """One request in a step: a new prompt (is_prompt) or a running sequence."""
    request_id: str
    is_prompt: bool
    prompt_token_ids: List[int] = field(default_factory=list)
    sampling_params: SamplingParams = field(default_factory=SamplingParams)
    # Prompts with images (a target with mrope_section): {"image_grid_thw": [(t, h, w), ...]}
    # plus either "pixel_values" (patch rows for SpecDecodeConfig.vision, image processor
    # layout) or "image_embeds" (the vision tower's output rows); the prompt holds one run of
    # placeholder tokens per image.
    multi_modal_data: Optional[dict] = None
    # Chunked prefill: how many prompt tokens this step may compute (None: the whole prompt in
    # this step).  A prompt that is not finished emits nothing; it is continued by naming the
    # request again (is_prompt False) with the next piece's size.
    num_prompt_tokens: Optional[int] = None
    # A running request as a V1 scheduler describes it (CachedRequestData): whether it returns
    # from a preemption (its cache was dropped, the scheduler recomputes it) and how many of its
    # tokens the scheduler counts as computed.  None: not a V1 step (num_prompt_tokens, if set,
    # is then the caller's own piece size).
    resumed_from_preemption: Optional[bool] = None
    num_computed_tokens: Optional[int] = None
    # Set by the worker on the prompt-shaped request that rebuilds a parked sequence: the
    # position of the first token the sequence emitted (the length of its original prompt).
    # Tokens from there on are recomputed the way decode passes computed them.
    decoded_from: Optional[int] = None


@dataclass
class ExecuteModelRequest:
    seq_group_metadata_list: List[SequenceGroupMetadata]
    num_lookahead_slots: int = 0
    finished_requests_ids: List[str] = field(default_factory=list)
    # request_id -> proposed continuation (at most k tokens) made outside this worker, for a
    # worker built with SpecDecodeConfig.external_proposals (no draft model): verified this step
    external_proposals: Optional[Dict[str, List[int]]] = None


@dataclass
class SpeculativeProposals:
    """Draft proposals of one step (device tensors; views of engine buffers)."""
    proposal_token_ids: torch.Tensor            # [B, k] int64
    proposal_probs: Optional[torch.Tensor]      # [B, k, V] float32 (None under greedy decoding)
    proposal_lens: torch.Tensor                 # [B] int32
    no_proposals: bool = False


@dataclass
class SpeculativeScores:
    """Target scores of the k proposals plus the bonus position."""
    probs: Optional[torch.Tensor]               # [B, k+1, V] float32 (None under greedy decoding)
    token_ids: torch.Tensor                     # [B, k+1] int64: target argmax / bonus sample
    logits: torch.Tensor                        # [B, k+1, V] float32
    hidden_states: Optional[torch.Tensor] = None


@dataclass
class LogprobsLists:
    """ModelRunnerOutput.logprobs (/root/reference/src/doc/engine_core.md:326-329): one row per
    emitted token, requests in req_ids order, a request's tokens in order; rows
    cu_num_generated_tokens[i] .. cu_num_generated_tokens[i + 1] - 1 belong to request i (a
    speculative step emits a variable number).  Column 0 is the emitted token, columns 1.. the
    most likely tokens, largest first."""
    logprob_token_ids: List[List[int]]
    logprobs: List[List[float]]
    sampled_token_ranks: List[int]
    cu_num_generated_tokens: List[int]


@dataclass
class ModelRunnerOutput:
    """What a V1 worker returns from execute_model: the fields of
    /root/reference/src/doc/engine_core.md:316-340, in that order.  This worker fills the
    request index, the variable-length token lists and, when a request of the step asks for
    them, the emitted tokens' log-probabilities (LogprobsLists); it computes no prompt
    log-probabilities, pooled outputs or KV-connector output (None / empty, as a model runner
    that was not asked for them returns)."""
    req_ids: List[str]
    req_id_to_index: Dict[str, int]
    sampled_token_ids: List[List[int]]
    logprobs: Optional[LogprobsLists] = None
    prompt_logprobs_dict: Dict[str, Optional[object]] = field(default_factory=dict)
    pooler_output: List[Optional[torch.Tensor]] = field(default_factory=list)
    kv_connector_output: Optional[object] = None
    num_nans_in_logits: Optional[Dict[str, int]] = None
    # not a V1 field: acceptance counters of this worker, when a collection interval elapsed
    spec_decode_metrics: Optional[SpecDecodeWorkerMetrics] = None


def _bias_of(sp: "SamplingParams") -> Optional[dict]:
    """The engine's logit-mask spec of a request, or None."""
    spec = {k: v for k, v in (("logit_bias", sp.logit_bias), ("allowed_token_ids", sp.allowed_token_ids),
                              ("bad_token_ids", sp.bad_token_ids)) if v}
    return spec or None


def _get(obj, name, default=None):
    """Field `name` of a dataclass-like object or of a dict."""
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def request_from_scheduler_output(so) -> "ExecuteModelRequest":
    """ExecuteModelRequest from a V1 SchedulerOutput (duck-typed on the fields transcribed at
    /root/reference/src/doc/engine_core.md:280-313): `scheduled_new_reqs` become prompts
    (req_id, prompt_token_ids, sampling_params, and mm_inputs / mm_kwargs when they carry
    image_grid_thw), `scheduled_cached_reqs` running requests (one object with `req_ids`, or a
    list of objects with `req_id`: both layouts exist upstream), `finished_req_ids` is passed
    through, `scheduled_spec_decode_tokens` (req_id -> proposed token ids) become the step's
    external proposals.  Running requests come first, in the scheduler's order."""
    # what a V1 scheduler may ask of a model runner and this worker does not do is refused
    # here, before anything runs, rather than served wrongly
    if _get(so, "grammar_bitmask", None) is not None or _get(so, "structured_output_request_ids", None):
        raise ValueError("structured output (a grammar bitmask over the logits) is not supported")
    if _get(so, "kv_connector_metadata", None) is not None:
        raise ValueError("KV-cache connectors (transfers to or from another worker) are not supported")
    new = list(_get(so, "scheduled_new_reqs", None) or [])
    cached = _get(so, "scheduled_cached_reqs", None)
    # per running request: resumed_from_preemption and num_computed_tokens (upstream's
    # CachedRequestData carries them as parallel lists; the older layout as fields per request)
    if cached is None:
        cached_ids: List[str] = []
        flags: List[bool] = []
        computed: List[Optional[int]] = []
    elif _get(cached, "req_ids", None) is not None:
        cached_ids = list(_get(cached, "req_ids"))
        fl = list(_get(cached, "resumed_from_preemption", None) or [])
        nc = list(_get(cached, "num_computed_tokens", None) or [])
        flags = [bool(fl[i]) if i < len(fl) else False for i in range(len(cached_ids))]
        computed = [int(nc[i]) if i < len(nc) else None for i in range(len(cached_ids))]
    else:
        cached_ids = [_get(c, "req_id") for c in cached]
        flags = [bool(_get(c, "resumed_from_preemption", False)) for c in cached]
        computed = [_get(c, "num_computed_tokens", None) for c in cached]
    n_tok = _get(so, "num_scheduled_tokens", None) or {}
    # (for a running request the count is 1 + its speculative tokens and means nothing here; for
    # a prompt still being prefilled, or a request recomputed after a preemption, it is the next
    # piece, which the worker picks up)
    metas = [SequenceGroupMetadata(rid, False, num_prompt_tokens=n_tok.get(rid),
                                   resumed_from_preemption=flags[i], num_computed_tokens=computed[i])
             for i, rid in enumerate(cached_ids)]
    for r in new:
        rid = _get(r, "req_id")
        prompt = list(_get(r, "prompt_token_ids", None) or [])
        piece = None
        if rid in n_tok and n_tok[rid] != len(prompt):
            if not 1 <= n_tok[rid] < len(prompt):
                raise ValueError(f"request {rid}: {n_tok[rid]} tokens scheduled for a prompt of {len(prompt)}")
            piece = int(n_tok[rid])          # chunked prefill: the first piece
        if _get(r, "num_computed_tokens", 0):
            raise ValueError(f"request {rid}: prefix-cached prompt tokens are not supported")
        sp = _get(r, "sampling_params", None)
        for name, off in (("prompt_logprobs", None), ("repetition_penalty", 1.0),
                          ("presence_penalty", 0.0), ("frequency_penalty", 0.0)):
            val = _get(sp, name, off) if sp is not None else off
            if val is not None and val != off and val != [] and val != {}:
                raise ValueError(f"request {rid}: sampling_params.{name}={val!r} is not supported")
        # bad words: only as token ids, one token each (a longer one would need the history)
        bad_ids = _get(sp, "bad_words_token_ids", None) or _get(sp, "_bad_words_token_ids", None) or []
        if _get(sp, "bad_words", None) and not bad_ids:
            raise ValueError(f"request {rid}: sampling_params.bad_words as text is not supported "
                             "(pass bad_words_token_ids)")
        if any(len(seq) != 1 for seq in bad_ids):
            raise ValueError(f"request {rid}: bad words of several tokens are not supported")
        if _get(r, "pooling_params", None) is not None:
            raise ValueError(f"request {rid}: pooling requests are not supported")
        if _get(r, "lora_request", None) is not None:
            raise ValueError(f"request {rid}: LoRA adapters are not supported")
        params = SamplingParams(temperature=_get(sp, "temperature", None),
                                max_tokens=_get(sp, "max_tokens", 16) or 16,
                                seed=_get(sp, "seed", None),
                                top_k=_get(sp, "top_k", 0) or 0, top_p=_get(sp, "top_p", 1.0) or 1.0,
                                min_p=_get(sp, "min_p", 0.0) or 0.0,
                                ignore_eos=bool(_get(sp, "ignore_eos", True)),
                                logit_bias={int(t): float(v) for t, v in
                                            (_get(sp, "logit_bias", None) or {}).items()} or None,
                                allowed_token_ids=list(_get(sp, "allowed_token_ids", None) or []) or None,
                                bad_token_ids=[int(seq[0]) for seq in bad_ids] or None,
                                logprobs=_get(sp, "logprobs", None)) if sp is not None \
            else SamplingParams()
        mm = None
        for cand in (_get(r, "mm_inputs", None), _get(r, "mm_kwargs", None)):
            items = cand if isinstance(cand, (list, tuple)) else [cand]
            for item in items:
                if item is not None and _get(item, "image_grid_thw", None) is not None:
                    mm = item if isinstance(item, dict) else {
                        k: _get(item, k) for k in ("image_grid_thw", "pixel_values", "image_embeds")
                        if _get(item, k, None) is not None}
        metas.append(SequenceGroupMetadata(rid, True, prompt, params, multi_modal_data=mm,
                                           num_prompt_tokens=piece))
    spec = _get(so, "scheduled_spec_decode_tokens", None) or {}
    return ExecuteModelRequest(metas, finished_requests_ids=list(_get(so, "finished_req_ids", None) or []),
                               external_proposals={rid: list(t) for rid, t in spec.items()} or None)


# --------------------------------------------------------------------------- components
class SpeculativeProposer(ABC):
    @abstractmethod
    def get_spec_proposals(self, execute_model_req: Optional[ExecuteModelRequest],
                           seq_ids_with_bonus_token_in_last_step: Set[int]) -> SpeculativeProposals:
        ...


class SpeculativeScorer(ABC):
    @abstractmethod
    def score_proposals(self, execute_model_req: Optional[ExecuteModelRequest],
                        proposals: SpeculativeProposals) -> SpeculativeScores:
        ...


class MultiStepWorker:
    """The draft model: k sequential single-token passes with on-device
    sampling and KV append, no host synchronisation in between."""

    def __init__(self, engine: SpecEngine):
        self.engine = engine

    def sampler_output(self, execute_model_req: Optional[ExecuteModelRequest], sample_len: int,
                       seq_ids_with_bonus_token_in_last_step: Set[int]) -> None:
        if sample_len != self.engine.k:
            raise ValueError(f"this engine proposes k={self.engine.k} tokens, asked for {sample_len}")
        # A sequence whose previous step accepted every proposal lacks the draft KV of
        # its last proposal.  The first draft pass always feeds the last two tokens,
        # so no per-sequence expansion is needed and the set can be ignored.
        with _rt.stream_ctx(self.engine.stream):
            self.engine._enqueue_propose()


class Top1Proposer(SpeculativeProposer):
    """Proposes the draft model's single best (or sampled) continuation."""

    def __init__(self, worker: MultiStepWorker):
        self.worker = worker

    def get_spec_proposals(self, execute_model_req=None,
                           seq_ids_with_bonus_token_in_last_step=frozenset()) -> SpeculativeProposals:
        eng = self.worker.engine
        B, k = eng.active, eng.k
        self.worker.sampler_output(execute_model_req, k, set(seq_ids_with_bonus_token_in_last_step))
        lens = torch.full((B,), k, dtype=torch.int32, device=eng.device)
        return SpeculativeProposals(
            proposal_token_ids=eng.draft_ids[:B],
            proposal_probs=eng.d_probs[:B] if eng.sampling else None,
            proposal_lens=lens, no_proposals=(B == 0))


class HostProposer(SpeculativeProposer):
    """Proposals the engine holds before the step starts, made on the host: it uploaded them in
    prepare_step; this hands out the views.  Sequences without a proposal (proposal_lens 0) get
    one token from the target as in plain decoding."""

    def __init__(self, engine: SpecEngine):
        self.engine = engine

    def get_spec_proposals(self, execute_model_req=None,
                           seq_ids_with_bonus_token_in_last_step=frozenset()) -> SpeculativeProposals:
        eng = self.engine
        B = eng.active
        with _rt.stream_ctx(eng.stream):
            eng._enqueue_lookup_proposals()
        return SpeculativeProposals(
            proposal_token_ids=eng.draft_ids[:B],
            proposal_probs=eng.d_probs[:B] if eng.sampling else None,
            proposal_lens=eng.prop_lens[:B], no_proposals=(B == 0))


class NGramProposer(HostProposer):
    """Prompt-lookup proposals: no draft model; the continuation of the most recent earlier
    occurrence of a sequence's last n tokens (upstream's NGramWorker role), found by the engine
    on the host in prepare_step."""


class ExternalProposer(HostProposer):
    """Proposals that arrive with the step (ExecuteModelRequest.external_proposals, or a V1
    scheduler output's scheduled_spec_decode_tokens)."""


class MQAScorer(SpeculativeScorer):
    """Scores all k+1 positions of every sequence in ONE target pass: the k+1
    query tokens of a sequence share each KV page read (no batch expansion)."""

    def __init__(self, engine: SpecEngine):
        self.engine = engine

    def _take_proposals(self, proposals: SpeculativeProposals) -> None:
        """Copies proposals that were not made by this engine's own proposer (their token ids
        are not the engine's buffer) into the token window and the proposal lengths."""
        eng = self.engine
        B, k = eng.active, eng.k
        ids = proposals.proposal_token_ids
        if ids.data_ptr() == eng.draft_ids.data_ptr():
            return                                  # the engine's own proposals: already in place
        if not getattr(eng, "host_props", False):
            raise ValueError("this engine proposes with its draft model; proposals made elsewhere "
                             "need a worker built with external_proposals (no draft model)")
        if tuple(ids.shape) != (B, k) or proposals.proposal_lens.numel() != B:
            raise ValueError(f"proposals must be [{B}, {k}] token ids with {B} lengths")
        if eng.sampling and proposals.proposal_probs is None:
            raise ValueError("sampling needs the proposals' probability rows (one-hot for "
                             "deterministic proposers)")
        eng.ext_props[:B].copy_(ids.to(device=eng.device, dtype=torch.int32))
        eng.prop_lens[:B].copy_(proposals.proposal_lens.to(device=eng.device, dtype=torch.int32)
                                .clamp_(0, k))
        ops.set_proposals(eng.ext_props[:B], eng.tok_buf[:B], k)
        ops.gather_proposals(eng.tok_buf[:B], eng.draft_ids[:B], k)
        if eng.sampling:
            eng.d_probs[:B].copy_(proposals.proposal_probs.to(device=eng.device, dtype=torch.float32))

    def score_proposals(self, execute_model_req=None, proposals=None) -> SpeculativeScores:
        eng = self.engine
        B, k = eng.active, eng.k
        with _rt.stream_ctx(eng.stream):
            if proposals is not None:
                self._take_proposals(proposals)
            eng._enqueue_score()
        logits = eng.t_logits[:B * (k + 1)].view(B, k + 1, eng.V)
        if eng.sampling:
            token_ids = eng.target_ids[:B]
            with _rt.stream_ctx(eng.stream):
                ops.argmax_rows(eng.t_logits[:B * (k + 1)], out=eng.target_ids.view(-1))
                token_ids[:, k] = eng.bonus_ids[:B]
            return SpeculativeScores(probs=eng.t_probs[:B], token_ids=token_ids, logits=logits)
        return SpeculativeScores(probs=None, token_ids=eng.target_ids[:B], logits=logits)


# Same results as upstream's batch-expansion scorer, without expanding the batch.
BatchExpansionTop1Scorer = MQAScorer


class RejectionSampler(nn.Module):
    """Modified rejection sampling (Leviathan et al. 2023, Alg. 1) as one HIP
    kernel per batch.  Call shape of vLLM's `RejectionSampler.forward`; uniforms
    come from a device generator (per-sequence generators via `seeded_seqs`)."""

    def __init__(self, seed: int = 0):
        super().__init__()
        self._seed = seed
        self._gen: Optional[torch.Generator] = None
        self.num_accepted_tokens = 0
        self.num_emitted_tokens = 0
        self.num_draft_tokens = 0
        self._counters: Optional[torch.Tensor] = None

    def _uniform(self, shape, device, seeded_seqs, row_of_seq):
        if self._gen is None:
            self._gen = torch.Generator(device=device).manual_seed(self._seed)
        u = torch.rand(shape, generator=self._gen, device=device)
        for b, g in (seeded_seqs or {}).items():
            u[row_of_seq(b)] = torch.rand(shape[1:], generator=g, device=device)
        return u

    def forward(self, target_with_bonus_probs: torch.Tensor, bonus_token_ids: torch.Tensor,
                draft_probs: torch.Tensor, draft_token_ids: torch.Tensor,
                seeded_seqs: Optional[Dict[int, torch.Generator]] = None,
                uniforms: Optional[tuple] = None) -> torch.Tensor:
        """Returns [B, k+1] int64: the accepted prefix, one recovered / bonus
        token, then -1.  `uniforms=(u_accept [B,k], u_recover [B,k+1])` makes the
        call deterministic (tests)."""
        B, k1, V = target_with_bonus_probs.shape
        k = k1 - 1
        dev = target_with_bonus_probs.device
        if uniforms is None:
            ident = lambda b: b
            uniforms = (self._uniform((B, k), dev, seeded_seqs, ident),
                        self._uniform((B, k + 1), dev, seeded_seqs, ident))
        if self._counters is None or self._counters.device != dev:
            self._counters = torch.zeros(3, dtype=torch.int64, device=dev)
        out = ops.rejection_sample(target_with_bonus_probs.contiguous(), bonus_token_ids.contiguous(),
                                   draft_probs.contiguous(), draft_token_ids.contiguous(),
                                   uniforms[0].contiguous(), uniforms[1].contiguous(), None,
                                   self._counters)
        return out

    def sync_counters(self) -> None:
        if self._counters is not None:
            a, e, d = self._counters.tolist()
            self.num_accepted_tokens, self.num_emitted_tokens, self.num_draft_tokens = a, e, d


class TypicalAcceptanceSampler(nn.Module):
    """Typical acceptance (Cai et al. 2024): a proposal stands if the target gives it more than
    min(posterior_threshold, posterior_alpha * exp(-entropy)).  Call shape and counters of
    vLLM's `TypicalAcceptanceSampler.forward`; `draft_probs` is accepted and unused."""

    def __init__(self, posterior_threshold: float = 0.09, posterior_alpha: float = 0.3):
        super().__init__()
        self.posterior_threshold, self.posterior_alpha = posterior_threshold, posterior_alpha
        self.num_accepted_tokens = 0
        self.num_emitted_tokens = 0
        self.num_draft_tokens = 0
        self._counters: Optional[torch.Tensor] = None

    def forward(self, target_with_bonus_probs: torch.Tensor, bonus_token_ids: torch.Tensor,
                draft_probs: Optional[torch.Tensor], draft_token_ids: torch.Tensor) -> torch.Tensor:
        dev = target_with_bonus_probs.device
        if self._counters is None or self._counters.device != dev:
            self._counters = torch.zeros(3, dtype=torch.int64, device=dev)
        return ops.typical_acceptance(target_with_bonus_probs.contiguous(),
                                      bonus_token_ids.contiguous(), draft_token_ids.contiguous(),
                                      self.posterior_threshold, self.posterior_alpha, None,
                                      self._counters)

    def sync_counters(self) -> None:
        if self._counters is not None:
            a, e, d = self._counters.tolist()
            self.num_accepted_tokens, self.num_emitted_tokens, self.num_draft_tokens = a, e, d


# --------------------------------------------------------------------------- worker
@dataclass
class SpecDecodeConfig:
    target: ModelWeights
    draft: Optional[ModelWeights] = None
    num_speculative_tokens: int = 5
    # without a draft model: prompt-lookup proposals from n-grams of these sizes
    ngram_prompt_lookup_max: int = 0
    ngram_prompt_lookup_min: int = 1
    # without a draft model and without prompt lookup: proposals come with every step
    # (ExecuteModelRequest.external_proposals / scheduled_spec_decode_tokens) or from the
    # `proposer` object given to the worker; the engine only verifies
    external_proposals: bool = False
    # multi-candidate verification of host-made proposals (prompt lookup / external): up to
    # this many continuations per sequence are merged into a token tree and scored by one
    # verify pass (greedy, or under sampling with the rejection sampler); see
    # EngineConfig.tree_candidates
    speculative_tree_candidates: int = 1
    speculative_tree_nodes: int = 0
    # > 1: the draft model proposes a tree of this many branches (EngineConfig.draft_tree_branches)
    speculative_draft_tree_branches: int = 0
    max_num_seqs: int = 1
    max_model_len: int = 2048
    temperature: float = 0.0
    seed: int = 0
    split_tokens: int = 256
    use_graph: bool = True
    rows: str = "auto"             # kernel family, see EngineConfig.rows
    draft_rows: str = "auto"       # the draft's kernel family, see EngineConfig.draft_rows
    speculative_disable_by_batch_size: int = 0   # 0 = speculate at every batch size
    # "always", or "auto": the engine times its own steps and speculates in a step only while
    # that yields more tokens per second than the plain target step at the current batch size
    # (EngineConfig.spec_policy); speculative_disable_by_batch_size stays a hard upper bound.
    speculative_policy: str = "always"
    # With quantization="fp8": the wide kernel family (prompt chunks, large batches) also
    # quantises its activations to fp8, per token row, and multiplies on the fp8 matrix
    # instruction of gfx950 (W8A8; EngineConfig.fp8_activations).
    fp8_activations: bool = False
    spec_decoding_acceptance_method: str = "rejection_sampler"  # or "typical_acceptance_sampler"
    typical_acceptance_sampler_posterior_threshold: float = 0.09
    typical_acceptance_sampler_posterior_alpha: float = 0.3
    # Tensor-parallel verify: the target is sharded over this many ranks (one process per
    # GPU), the draft stays whole on each.  `target` is then either this rank's shard or a
    # (ModelConfig, state_dict) pair that load_model shards itself.
    tensor_parallel_size: int = 1
    # how the ranks' partial sums are combined: "rccl" (the library's RCCL group), or "ipc":
    # one-shot all-reduce / all-gather kernels over peer-mapped device memory (the ranks of one
    # node; every rank reads every peer once, in rank order: suited to the rows x hidden
    # messages of decode and verify passes, bit-identical on all ranks)
    tensor_parallel_collective: str = "rccl"
    # under tensor parallelism each rank holds vocab / tp rows of the LM head and the ranks
    # exchange slice maxima (greedy) or gather rows (sampling) instead of replicating the head
    vocab_parallel_lm_head: bool = True
    # replay the step as a graph under tensor parallelism too (the RCCL calls are captured with
    # the kernels).  Verified with a group of one rank only, hence opt-in.
    tensor_parallel_graph: bool = False
    # 1: the draft is whole on every rank (no communication in its passes); equal to
    # tensor_parallel_size: the draft is sharded like the target (upstream's
    # speculative_draft_tensor_parallel_size).  `draft` is then a (ModelConfig, state_dict)
    # pair that load_model shards, or this rank's shard.
    draft_tensor_parallel_size: int = 1
    # prompts with images: the vision tower (vision.VisionTower) that turns pixel patches into
    # input rows of the target, and the tokenizer's placeholder token
    vision: Optional[object] = None
    image_token_id: int = 151655
    # Weight-only fp8 (e4m3, one f32 scale per output row, quantised at load) for a model given
    # as a (ModelConfig, state_dict) pair.  "fp8" on the draft alone halves the bytes its k
    # passes stream while the emitted tokens stay the bf16 target's own stream (greedy) or
    # distribution (sampling): only the acceptance rate can move.  "fp8" on the target changes
    # the target model itself.  A ModelWeights object loaded with fp8=True needs no option.
    quantization: Optional[str] = None
    draft_quantization: Optional[str] = None
    # Token ids (a multiple of 16 of them, e.g. the most frequent tokens of the deployment's
    # traffic) that the draft's LM head covers; None = the whole vocabulary.  The draft given as
    # (ModelConfig, state_dict) is loaded with a head of those rows only: each draft pass streams
    # that fraction of the head, the draft can only propose those tokens, the target still emits
    # any token (greedy decoding only).
    draft_vocab_subset: Optional[Sequence[int]] = None
    # The draft's passes of up to 4 rows with every layer as three launches instead of five
    # (EngineConfig.draft_fused_layers; measured no faster on MI355X, hence off).  The draft's
    # rows then differ from the five-launch layer's in the last bits; what is emitted does not.
    draft_fused_layers: bool = False


@dataclass
class _StepPlan:
    """What one execute_model call does with each of its requests (made and validated by
    SpecDecodeWorker._plan_step before anything runs)."""
    req: ExecuteModelRequest
    metas: List[SequenceGroupMetadata]          # the step's requests, returning ones rebuilt as prompts
    running: List[SequenceGroupMetadata]        # get a speculative step
    continued: List[SequenceGroupMetadata]      # next piece of a prompt that is being prefilled
    pieces: List[SequenceGroupMetadata]         # first piece of a prompt
    prompts: List[SequenceGroupMetadata]        # whole prompts
    finished: List[str]                         # running requests that end
    dropped: List[str]                          # half-prefilled requests that end
    parked_now: List[str]                       # running requests the step left out (or restarts)
    resumed: Set[str]                           # requests the scheduler recomputes in this step
    mm_inputs: Dict[str, dict]
    reprefilled: List[str] = field(default_factory=list)   # half-prefilled requests that start over
    asked: List[int] = field(default_factory=list)   # logprobs counts asked for by the step's requests


class SpecDecodeWorker:
    """Worker that runs draft-k / verify / accept for every running sequence.

    Construction mirrors a vLLM worker; `vllm_config` may be a SpecDecodeConfig
    (stand-alone use) or any object with a `spec_decode_config` attribute holding
    one (when an engine builds the worker from its own config object)."""

    def __init__(self, vllm_config, local_rank: int = 0, rank: int = 0,
                 distributed_init_method: Optional[str] = None, is_driver_worker: bool = True,
                 engine: Optional[SpecEngine] = None,
                 proposer: Optional[SpeculativeProposer] = None):
        """proposer: a SpeculativeProposer made elsewhere (any object with get_spec_proposals);
        needs external_proposals.  Its proposals are verified by the scorer every step."""
        cfg = getattr(vllm_config, "spec_decode_config", vllm_config)
        self.config: SpecDecodeConfig = cfg
        self.local_rank, self.rank = local_rank, rank
        self.distributed_init_method = distributed_init_method
        self.is_driver_worker = is_driver_worker
        self.engine: Optional[SpecEngine] = engine
        self.proposer_worker: Optional[MultiStepWorker] = None
        self.proposer: Optional[SpeculativeProposer] = None
        self._foreign_proposer = proposer
        self.scorer: Optional[SpeculativeScorer] = None
        self.metrics: Optional[AsyncMetricsCollector] = None
        self._slot_of: Dict[str, int] = {}
        self._req_at: List[str] = []
        self._prefill_of: Dict[str, int] = {}     # request id -> engine handle of a prompt in pieces
        self._prefill_left: Dict[str, int] = {}   # request id -> prompt tokens not yet computed
        self._logprobs_of: Dict[str, int] = {}    # request id -> SamplingParams.logprobs, if asked
        # what it takes to rebuild a sequence that a step left out (a V1 scheduler preempts a
        # request by not scheduling it; it comes back among the cached requests later): the
        # request's tokens so far (prompt + emitted), its parameters and multimodal inputs
        self._tokens: Dict[str, List[int]] = {}
        self._request: Dict[str, SequenceGroupMetadata] = {}
        self._parked: Set[str] = set()             # requests without a slot, waiting to be named again
        self._park_count: Dict[str, int] = {}      # request id -> times it was parked
        if engine is not None:
            self._wire()

    # -- life cycle (vLLM worker protocol) ----------------------------------------
    def init_device(self) -> None:
        if _rt.on_gpu():
            if not torch.cuda.is_available():
                raise RuntimeError("SpecDecodeWorker needs a GPU: the GPU library has no CPU path "
                                   "(the CPU build of the library is chosen with SD_AMD_BACKEND=cpu)")
            torch.cuda.set_device(self.local_rank)
        tp = self.config.tensor_parallel_size
        if tp > 1:
            from . import parallel
            if not torch.distributed.is_initialized():
                raise RuntimeError("tensor_parallel_size > 1 needs an initialised torch.distributed "
                                   "group (parallel.init_from_env) to exchange the RCCL id")
            if torch.distributed.get_world_size() != tp:
                raise ValueError("tensor_parallel_size must equal the world size")
            kind = self.config.tensor_parallel_collective
            if kind == "rccl":
                self.tp_group = parallel.TensorParallelGroup(self.rank, tp)
            elif kind == "ipc":
                tgt = self.config.target
                mcfg = tgt[0] if isinstance(tgt, tuple) else tgt.cfg
                # the largest messages: a 256-row prefill pass's hidden states, or (sampling) the
                # gathered logit slices of a verify pass
                rows = max(16, self.config.max_num_seqs * (self.config.num_speculative_tokens + 1))
                need = max(256 * mcfg.hidden * 2, rows * (-(-mcfg.vocab // tp)) * 4)
                self.tp_group = parallel.IpcTensorParallelGroup(self.rank, tp,
                                                                max_message_bytes=min(need, 16 << 20))
            else:
                raise ValueError(f"tensor_parallel_collective must be 'rccl' or 'ipc', not {kind!r}")

    def load_model(self) -> None:
        c = self.config
        target, tp = c.target, c.tensor_parallel_size
        for name, q, given in (("quantization", c.quantization, c.target),
                               ("draft_quantization", c.draft_quantization, c.draft)):
            if q not in (None, "fp8"):
                raise ValueError(f"{name} must be None or 'fp8', not {q!r}")
            if q and given is None:
                raise ValueError(f"{name} without a model")
            if q and tp > 1:
                raise ValueError(f"{name}: fp8 weights are not sharded (tensor_parallel_size > 1)")
            if q and not isinstance(given, tuple) and not getattr(given, "fp8", False):
                raise ValueError(f"{name}={q!r} but the weights object was loaded without fp8; "
                                 "pass (ModelConfig, state_dict) or load it with fp8=True")
        if isinstance(target, tuple):
            from . import weights as W
            cfg, state = target
            if tp > 1:
                vp = c.vocab_parallel_lm_head and cfg.vocab % (16 * tp) == 0
                cfg, state = W.shard_state_dict(cfg, state, tp, self.rank, vocab_parallel=vp)
            target = W.from_state_dict(cfg, state, _rt.device(self.local_rank),
                                       fp8=c.quantization == "fp8")
        draft = c.draft
        if c.draft_tensor_parallel_size not in (1, tp):
            raise ValueError("draft_tensor_parallel_size must be 1 or tensor_parallel_size")
        draft_tp = tp > 1 and c.draft_tensor_parallel_size == tp
        if isinstance(draft, tuple):
            from . import weights as W
            dcfg, dstate = draft
            if draft_tp:   # head split like the target's
                dcfg, dstate = W.shard_state_dict(dcfg, dstate, tp, self.rank,
                                                  vocab_parallel=target.cfg.lm_vocab > 0)
            draft = W.from_state_dict(dcfg, dstate, _rt.device(self.local_rank),
                                      fp8=c.draft_quantization == "fp8",
                                      head_subset=c.draft_vocab_subset)
        elif c.draft_vocab_subset is not None and getattr(draft, "head_map", None) is None:
            raise ValueError("draft_vocab_subset needs the draft as (ModelConfig, state_dict), or a "
                             "weights object loaded with head_subset")
        self.engine = SpecEngine(target, draft, EngineConfig(
            k=c.num_speculative_tokens, max_batch=c.max_num_seqs, max_seq_len=c.max_model_len,
            split_tokens=c.split_tokens, temperature=c.temperature, seed=c.seed,
            use_graph=c.use_graph and (tp == 1 or c.tensor_parallel_graph), rows=c.rows,
            draft_rows=c.draft_rows,
            ngram_max=c.ngram_prompt_lookup_max, ngram_min=c.ngram_prompt_lookup_min,
            external_proposals=c.external_proposals,
            tree_candidates=c.speculative_tree_candidates, tree_nodes=c.speculative_tree_nodes,
            draft_tree_branches=c.speculative_draft_tree_branches,
            disable_by_batch_size=c.speculative_disable_by_batch_size,
            spec_policy=c.speculative_policy, fp8_activations=c.fp8_activations,
            acceptance_method=c.spec_decoding_acceptance_method,
            posterior_threshold=c.typical_acceptance_sampler_posterior_threshold,
            posterior_alpha=c.typical_acceptance_sampler_posterior_alpha,
            draft_fused_layers=c.draft_fused_layers),
            tp_group=getattr(self, "tp_group", None), draft_tp=draft_tp)
        if tp > 1:
            # the ranks' kernels wait for each other with bounded waits: no rank may start a pass
            # while a peer is still loading its shard
            _rt.synchronize(_rt.device(self.local_rank))
            torch.distributed.barrier()
        self._wire()

    def _wire(self) -> None:
        if self._foreign_proposer is not None:
            if not getattr(self.engine, "external", False):
                raise ValueError("a proposer object needs external_proposals (and no draft model)")
            self.proposer_worker = None
            self.proposer = self._foreign_proposer
        elif getattr(self.engine, "external", False):
            self.proposer_worker = None
            self.proposer = ExternalProposer(self.engine)
        elif getattr(self.engine, "ngram", False):
            self.proposer_worker = None
            self.proposer = NGramProposer(self.engine)
        else:
            self.proposer_worker = MultiStepWorker(self.engine)
            self.proposer = Top1Proposer(self.proposer_worker)
        self.scorer = MQAScorer(self.engine)
        self.metrics = AsyncMetricsCollector(self.engine)

    def determine_num_available_blocks(self):
        """(gpu_blocks, cpu_blocks): pages each model's cache holds; caches are
        sized at construction for max_num_seqs x max_model_len."""
        return self.engine.t_kv.num_pages, 0

    def initialize_cache(self, num_gpu_blocks: int, num_cpu_blocks: int = 0) -> None:
        if num_gpu_blocks > self.engine.t_kv.num_pages:
            raise ValueError("the cache was sized at load time; cannot grow it")

    # -- one step -----------------------------------------------------------------
    def execute_model(self, scheduler_output, delay_capture: bool = False) -> ModelRunnerOutput:
        """Runs one step: running sequences get a speculative step (1..k+1 tokens each), new
        prompts are prefilled in both models and get their first token.

        scheduler_output: a V1 SchedulerOutput as the reference's workers receive it
        (/root/reference/src/qwen_image_gen/worker.py:47-51; fields at
        /root/reference/src/doc/engine_core.md:280-313, read by request_from_scheduler_output),
        or the V0-shaped ExecuteModelRequest of this module.  delay_capture is accepted for
        signature compatibility: step graphs are captured on first use of a batch shape anyway.

        Every request of the step is checked before anything runs (_plan_step): a refused step
        leaves the running sequences where they were (no token is produced and then dropped)."""
        req = scheduler_output
        if not isinstance(req, ExecuteModelRequest):
            if _get(req, "scheduled_new_reqs", None) is None and _get(req, "scheduled_cached_reqs", None) is None:
                raise TypeError("execute_model takes a SchedulerOutput (scheduled_new_reqs / "
                                "scheduled_cached_reqs / finished_req_ids) or an ExecuteModelRequest")
            req = request_from_scheduler_output(req)
        plan = self._plan_step(req)
        self._retire(plan)
        out: Dict[str, List[int]] = {}
        lp: Dict[str, List[tuple]] = {}
        self._step_running(plan, out, lp)
        self._admit_prompts(plan, out, lp)
        self._admit_pieces(plan, out, lp)
        return self._collect(plan, out, lp)

    def _rebuilt(self, g: SequenceGroupMetadata, leaving: bool = False) -> SequenceGroupMetadata:
        """The prompt-shaped request that rebuilds a parked sequence named again by the step:
        prompt + emitted tokens, prefilled whole (it then emits a token in this step) or, when
        the scheduler recomputes it from position 0 in pieces, in those pieces.  The tokens it
        emitted are recomputed by decode-shaped passes (decoded_from), so a greedy stream goes
        on bit for bit as if the sequence had never left."""
        rid = g.request_id
        first, toks = self._request[rid], self._tokens[rid]
        piece = g.num_prompt_tokens
        if g.resumed_from_preemption is None:
            pass                                 # not a V1 step: the caller's own piece size
        elif g.resumed_from_preemption:
            if g.num_computed_tokens:
                raise ValueError(f"request {rid} resumes with {g.num_computed_tokens} prefix-cached "
                                 "tokens: not supported (this worker recomputes a resumed request "
                                 "from its first token)")
        else:
            # The scheduler only left the request out of a step (token budget): its count is
            # 1 + speculative tokens, not a piece of a recomputation.
            piece = None
        sp = first.sampling_params
        if sp.seed is not None:      # a stream of its own for every incarnation
            n_parked = self._park_count.get(rid, 0) + (1 if leaving else 0)   # (leaving: parked by this step)
            sp = SamplingParams(**{**vars(sp), "seed": (sp.seed + 0x9E3779B1 * n_parked) & 0x7FFFFFFFFFFFFFFF})
        return SequenceGroupMetadata(
            rid, True, list(toks), sp, multi_modal_data=first.multi_modal_data,
            num_prompt_tokens=piece if piece is not None and 1 <= piece < len(toks) else None,
            decoded_from=len(first.prompt_token_ids))

    def _plan_step(self, req: ExecuteModelRequest) -> "_StepPlan":
        """Sorts the step's requests into what happens to them and validates all of it; nothing
        runs here."""
        eng = self.engine
        metas = req.seq_group_metadata_list
        # A running request that the step does not name (and does not finish) is parked: its
        # slot is freed, its tokens are kept.  When it is named again it is rebuilt from them
        # and goes on where it stopped: a V1 scheduler preempts and resumes requests this way.
        named = {g.request_id for g in metas}
        parked_now = [r for r in self._req_at if r not in named and r not in req.finished_requests_ids]
        resumed = {g.request_id for g in metas if not g.is_prompt and g.request_id in self._parked}
        # A scheduler may also preempt a request and resume it without a step in between that
        # leaves it out (nothing else was running, so there was no step): the flag says that it
        # recomputes the request from its first token.  The sequence leaves its slot now and is
        # rebuilt like any other returning request.
        restarted = [g.request_id for g in metas
                     if not g.is_prompt and g.resumed_from_preemption and g.request_id in self._slot_of
                     and g.request_id not in req.finished_requests_ids]
        parked_now += restarted
        # ... or one whose prompt was still arriving in pieces: its prefill starts over
        reprefilled = [g.request_id for g in metas
                       if not g.is_prompt and g.resumed_from_preemption and g.request_id in self._prefill_of
                       and g.request_id not in req.finished_requests_ids]
        resumed |= set(restarted) | set(reprefilled)
        if resumed:
            metas = [self._rebuilt(g, leaving=g.request_id in restarted) if g.request_id in resumed else g
                     for g in metas]
        # prompts that arrive in pieces (chunked prefill): first pieces and continuations
        continued = [g for g in metas if not g.is_prompt and g.request_id in self._prefill_of]
        running = [g for g in metas if not g.is_prompt and g.request_id not in self._prefill_of]
        pieces = [g for g in metas if g.is_prompt and g.num_prompt_tokens is not None
                  and g.num_prompt_tokens < len(g.prompt_token_ids)]
        prompts = [g for g in metas if g.is_prompt and g not in pieces]
        finished = [r for r in dict.fromkeys(req.finished_requests_ids) if r in self._slot_of]
        dropped = [r for r in dict.fromkeys(req.finished_requests_ids) if r in self._prefill_of]
        leaving = set(restarted) | set(reprefilled)      # (their slots are free before anything is admitted)

        have = getattr(getattr(eng, "cfg", None), "temperature", None)
        max_len = getattr(getattr(eng, "cfg", None), "max_seq_len", None)
        seen = set()
        for g in metas:
            if g.request_id in seen:
                raise KeyError(f"request {g.request_id} is listed twice")
            seen.add(g.request_id)
        if (pieces or continued) and not hasattr(eng, "prefill_begin"):
            raise ValueError("this engine prefills a prompt in one step (no chunked prefill)")
        for g in pieces:
            if g.multi_modal_data:
                raise ValueError(f"request {g.request_id}: a prompt with images is prefilled in one step")
            if g.num_prompt_tokens < 1:
                raise ValueError(f"request {g.request_id}: a piece of {g.num_prompt_tokens} prompt tokens")
        for g in continued:
            left = self._prefill_left[g.request_id]
            if g.num_prompt_tokens is None or not 1 <= g.num_prompt_tokens <= left:
                raise ValueError(f"request {g.request_id}: {left} prompt tokens remain, the step "
                                 f"schedules {g.num_prompt_tokens}")
            if g.request_id in dropped:
                raise ValueError(f"request {g.request_id} is continued and listed as finished")
        for g in prompts + pieces:
            sp = g.sampling_params
            if sp.temperature is not None and have is not None:
                t = float(sp.temperature)
                if not t >= 0.0:
                    raise ValueError(f"request {g.request_id}: temperature {sp.temperature} < 0")
                if t != float(have) and not getattr(eng, "sampling", False):
                    raise ValueError(f"request {g.request_id} asks for temperature {sp.temperature}; "
                                     f"this worker decodes greedily (build it with a temperature > 0 "
                                     f"to serve sampling and greedy requests together)")
            if sp.seed is not None and not isinstance(sp.seed, int):
                raise ValueError(f"request {g.request_id}: seed must be an integer")
            if _bias_of(sp) is not None:
                if not hasattr(eng, "_bias_row"):
                    raise ValueError("this engine applies no logit masks")
                eng._bias_row(_bias_of(sp))      # raises on tokens outside the vocabulary
            if sp.logprobs is not None:
                if not isinstance(sp.logprobs, int) or not 0 <= sp.logprobs <= ops.LOGPROBS_TOP_MAX:
                    raise ValueError(f"request {g.request_id}: logprobs {sp.logprobs!r} outside "
                                     f"0..{ops.LOGPROBS_TOP_MAX}")
                if not hasattr(eng, "logprobs_top"):
                    raise ValueError("this engine computes no log-probabilities")
            if not 0.0 < float(sp.top_p) <= 1.0 or not 0.0 <= float(sp.min_p) <= 1.0:
                raise ValueError(f"request {g.request_id}: top_p {sp.top_p} must be in (0, 1] and "
                                 f"min_p {sp.min_p} in [0, 1]")
            if g.request_id in self._parked and g.request_id not in resumed:
                raise KeyError(f"request {g.request_id} is parked on this worker; finish it first")
            if g.request_id not in leaving and (
                    (g.request_id in self._slot_of and g.request_id not in finished) or (
                        g.request_id in self._prefill_of and g.request_id not in dropped)):
                raise KeyError(f"request {g.request_id} is already running")
            n = len(g.prompt_token_ids)
            if n < 1 or (max_len is not None and n + 1 > max_len):
                raise ValueError(f"request {g.request_id}: prompt of {n} tokens outside 1..{max_len}")
        unknown = [g.request_id for g in running if g.request_id not in self._slot_of]
        if unknown:
            raise KeyError(f"requests {unknown} are not running on this worker")
        gone = [g.request_id for g in running if g.request_id in finished]
        if gone:
            raise ValueError(f"requests {gone} are listed as running and as finished")
        n_after = eng.active - len(finished) - len(parked_now)
        if len(running) != n_after:      # (cannot happen: every slot is named, finished or parked)
            raise ValueError(f"{n_after} sequences are running but the step names {len(running)}")
        if parked_now and not all(r in self._tokens for r in parked_now):
            raise ValueError(f"requests {parked_now} are running but not named by the step")
        n_prefilling = len(self._prefill_of) - len(dropped) - len(reprefilled)
        if n_after + n_prefilling + len(prompts) + len(pieces) > eng.cfg.max_batch:
            raise RuntimeError("no free sequence slot (max_num_seqs reached)")
        if req.external_proposals is not None and not getattr(eng, "external", False):
            raise ValueError("the step carries proposals (external_proposals / "
                             "scheduled_spec_decode_tokens) but this worker makes its own")
        # multimodal inputs (vision tower included) are prepared before the step as well
        mm_inputs = {g.request_id: self._multimodal_inputs(g) for g in prompts}
        return _StepPlan(req, metas, running, continued, pieces, prompts, finished, dropped,
                         parked_now, resumed, mm_inputs, reprefilled)

    def _retire(self, plan: "_StepPlan") -> None:
        """Requests that leave: finished ones (running, parked or half prefilled) and the
        running ones the step left out."""
        eng = self.engine
        for rid in dict.fromkeys(plan.req.finished_requests_ids):
            if rid in self._parked:          # finished while parked: nothing left but its log
                self._forget(rid)
        for rid in plan.finished:
            self._finish(rid)
        for rid in plan.parked_now:
            self._park(rid)
        self._parked -= plan.resumed
        for rid in plan.dropped:
            eng.prefill_abort(self._prefill_of.pop(rid))
            self._prefill_left.pop(rid, None)
            self._forget(rid)
        for rid in plan.reprefilled:         # starts over in this step: its token log stays
            eng.prefill_abort(self._prefill_of.pop(rid))
            self._prefill_left.pop(rid, None)
        # log-probabilities: computed for the whole step as soon as one of its requests asks
        for g in plan.prompts + plan.pieces:
            if g.sampling_params.logprobs is not None:
                self._logprobs_of[g.request_id] = int(g.sampling_params.logprobs)
        plan.asked = [self._logprobs_of[g.request_id] for g in plan.metas if g.request_id in self._logprobs_of]
        if hasattr(eng, "logprobs_top"):
            eng.logprobs_top = max(plan.asked) if plan.asked else None

    def _step_running(self, plan: "_StepPlan", out, lp) -> None:
        eng = self.engine
        if not plan.running:
            return
        ext = plan.req.external_proposals
        if ext is not None:
            eng.set_external_proposals([ext.get(r, []) for r in self._req_at])
        emitted = self._run_speculative_decoding_step(plan.req)
        for slot, toks in enumerate(emitted):
            out[self._req_at[slot]] = toks
            if plan.asked:
                lp[self._req_at[slot]] = eng.last_logprobs[slot]

    def _admit_prompts(self, plan: "_StepPlan", out, lp) -> None:
        """Whole prompts: text prompts that arrive together are prefilled in the same passes
        (one synchronisation for the group); a prompt with images goes alone."""
        eng, prompts, mm_inputs = self.engine, plan.prompts, plan.mm_inputs
        seeded = bool(getattr(eng, "sampling", False))    # greedy engines have no use for seeds
        i = 0
        while i < len(prompts):
            j = i
            while (j < len(prompts) and not mm_inputs[prompts[j].request_id]
                   and prompts[j].decoded_from is None and hasattr(eng, "add_requests")):
                j += 1
            if j - i >= 2:
                group = prompts[i:j]
                kw = dict(seeds=[g.sampling_params.seed for g in group],
                          temperatures=[g.sampling_params.temperature for g in group],
                          filters=[dict(top_k=g.sampling_params.top_k, top_p=g.sampling_params.top_p,
                                        min_p=g.sampling_params.min_p) for g in group]) if seeded else {}
                if any(_bias_of(g.sampling_params) for g in group):
                    kw["biases"] = [_bias_of(g.sampling_params) for g in group]
                firsts = eng.add_requests([g.prompt_token_ids for g in group], **kw)
            else:
                group = prompts[i:i + 1]
                sp0 = group[0].sampling_params
                kw = dict(seed=sp0.seed, temperature=sp0.temperature, top_k=sp0.top_k, top_p=sp0.top_p,
                          min_p=sp0.min_p) if seeded else {}
                if _bias_of(sp0):
                    kw["bias"] = _bias_of(sp0)
                if group[0].decoded_from is not None:
                    kw["decoded_from"] = group[0].decoded_from
                firsts = [eng.add_request(eng.active, group[0].prompt_token_ids,
                                          **mm_inputs[group[0].request_id], **kw)]
            for n, (g, first) in enumerate(zip(group, firsts)):
                self._slot_of[g.request_id] = len(self._req_at)
                self._req_at.append(g.request_id)
                out[g.request_id] = [first]
                if plan.asked:
                    lp[g.request_id] = eng.last_logprobs[n]
            i += len(group)

    def _admit_pieces(self, plan: "_StepPlan", out, lp) -> None:
        """Prompts in pieces: a piece emits nothing; the last one admits the sequence and emits
        its first token."""
        eng = self.engine
        seeded = bool(getattr(eng, "sampling", False))
        for g in plan.pieces:
            sp = g.sampling_params
            kw = dict(seed=sp.seed, temperature=sp.temperature, top_k=sp.top_k, top_p=sp.top_p,
                      min_p=sp.min_p) if seeded else {}
            if _bias_of(sp):
                kw["bias"] = _bias_of(sp)
            if g.decoded_from is not None:
                kw["decoded_from"] = g.decoded_from
            self._prefill_of[g.request_id] = eng.prefill_begin(g.prompt_token_ids, **kw)
            self._prefill_left[g.request_id] = len(g.prompt_token_ids)
        for g in plan.pieces + plan.continued:
            first = eng.prefill_advance(self._prefill_of[g.request_id], g.num_prompt_tokens)
            self._prefill_left[g.request_id] -= g.num_prompt_tokens
            out[g.request_id] = []
            if first is not None:
                del self._prefill_of[g.request_id], self._prefill_left[g.request_id]
                self._slot_of[g.request_id] = len(self._req_at)
                self._req_at.append(g.request_id)
                out[g.request_id] = [first]
                if plan.asked:
                    lp[g.request_id] = eng.last_logprobs[0]

    def _collect(self, plan: "_StepPlan", out, lp) -> ModelRunnerOutput:
        """The step's ModelRunnerOutput; every request's token log grows by what it emitted."""
        ids = [g.request_id for g in plan.metas]
        for g in plan.prompts + plan.pieces:
            if g.request_id not in self._tokens:
                self._request[g.request_id] = g
                self._tokens[g.request_id] = list(g.prompt_token_ids)
        for r in ids:
            if r in self._tokens:
                self._tokens[r].extend(out[r])
        logprobs = None
        if plan.asked:
            rows = [e for r in ids for e in lp.get(r, [])]
            cu = [0]
            for r in ids:
                cu.append(cu[-1] + len(lp.get(r, [])))
            logprobs = LogprobsLists([e[0] for e in rows], [e[1] for e in rows], [e[2] for e in rows], cu)
        return ModelRunnerOutput(
            req_ids=ids, req_id_to_index={r: i for i, r in enumerate(ids)},
            sampled_token_ids=[out[r] for r in ids], logprobs=logprobs,
            prompt_logprobs_dict={r: None for r in ids}, pooler_output=[None] * len(ids),
            spec_decode_metrics=self.metrics.maybe_collect() if self.metrics else None)

    def _multimodal_inputs(self, g: SequenceGroupMetadata):
        """add_request's input_embeds and rope_positions of a prompt with images; nothing for
        plain text."""
        mm = g.multi_modal_data
        if not mm:
            return {}
        from . import multimodal
        c = self.config
        grids = [tuple(int(v) for v in row) for row in mm["image_grid_thw"]]
        rows = mm.get("image_embeds")
        if rows is None:
            if c.vision is None:
                raise ValueError("pixel_values need SpecDecodeConfig.vision (a vision tower)")
            rows = c.vision.forward(mm["pixel_values"], grids)
        embeds, pos = multimodal.prepare_prompt(
            g.prompt_token_ids, self.engine.target.weights, rows, grids, image_token_id=c.image_token_id,
            spatial_merge=c.vision.cfg.spatial_merge_size if c.vision else 2)
        return {"input_embeds": embeds, "rope_positions": pos}

    def _run_speculative_decoding_step(self, req: ExecuteModelRequest) -> List[List[int]]:
        """propose -> score -> verify, as one replayed graph when enabled."""
        eng = self.engine
        if self._foreign_proposer is None and (eng.cfg.use_graph or eng._spec_disabled()
                                               or getattr(eng, "_auto", False) or getattr(eng, "tree", False)):
            # (a policy engine decides inside the step whether to speculate; a tree step is one
            # unit: its candidates are no [B, k] proposal for the scorer's interface)
            return eng.step()
        eng.prepare_step()
        proposals = self.proposer.get_spec_proposals(req, frozenset())
        scores = self.scorer.score_proposals(req, proposals)
        self._verify_tokens(proposals, scores)
        return eng.finish_step()

    def _verify_tokens(self, proposals: SpeculativeProposals, scores: SpeculativeScores) -> None:
        with _rt.stream_ctx(self.engine.stream):
            self.engine._enqueue_accept()

    def _forget(self, request_id: str) -> None:
        for d in (self._logprobs_of, self._tokens, self._request, self._park_count):
            d.pop(request_id, None)
        self._parked.discard(request_id)

    def _park(self, request_id: str) -> None:
        """Frees the slot of a running request the step left out; its token log stays."""
        self._parked.add(request_id)
        self._park_count[request_id] = self._park_count.get(request_id, 0) + 1
        self._release_slot(request_id)

    def _finish(self, request_id: str) -> None:
        self._forget(request_id)
        self._release_slot(request_id)

    def _release_slot(self, request_id: str) -> None:
        slot = self._slot_of.pop(request_id, None)
        if slot is None:
            return
        last = len(self._req_at) - 1
        self.engine.remove_request(slot)
        if slot != last:
            moved = self._req_at[last]
            self._req_at[slot] = moved
            self._slot_of[moved] = slot
        self._req_at.pop()
