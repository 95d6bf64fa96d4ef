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

from . import ops
from .engine import EngineConfig, SpecEngine
from .metrics import AsyncMetricsCollector, SpecDecodeWorkerMetrics
from .weights import ModelWeights


# --------------------------------------------------------------------------- carriers
@dataclass
class SamplingParams:
    # None: the worker's temperature.  One temperature serves the whole batch (the step is one
    # captured graph); a request that asks for another one is refused, not silently served.
    temperature: Optional[float] = None
    max_tokens: int = 16
    seed: Optional[int] = None
    ignore_eos: bool = True


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


@dataclass
class ExecuteModelRequest:
    seq_group_metadata_list: List[SequenceGroupMetadata]
    num_lookahead_slots: int = 0
    finished_requests_ids: List[str] = field(default_factory=list)


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
class ModelRunnerOutput:
    req_ids: List[str]
    req_id_to_index: Dict[str, int]
    sampled_token_ids: List[List[int]]
    spec_decode_metrics: Optional[SpecDecodeWorkerMetrics] = None


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
        with torch.cuda.stream(self.engine.stream):
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


class NGramProposer(SpeculativeProposer):
    """Prompt-lookup proposals: no draft model; the continuation of the most recent earlier
    occurrence of a sequence's last n tokens (upstream's NGramWorker role).  The engine
    computes them on the host in prepare_step; sequences without a match propose nothing
    (proposal_lens 0) and get one token from the target as in plain decoding."""

    def __init__(self, engine: SpecEngine):
        self.engine = engine

    def get_spec_proposals(self, execute_model_req=None,
                           seq_ids_with_bonus_token_in_last_step=frozenset()) -> SpeculativeProposals:
        eng = self.engine
        B = eng.active
        with torch.cuda.stream(eng.stream):
            eng._enqueue_lookup_proposals()
        return SpeculativeProposals(
            proposal_token_ids=eng.draft_ids[:B],
            proposal_probs=eng.d_probs[:B] if eng.sampling else None,
            proposal_lens=eng.prop_lens[:B], no_proposals=(B == 0))


class MQAScorer(SpeculativeScorer):
    """Scores all k+1 positions of every sequence in ONE target pass: the k+1
    query tokens of a sequence share each KV page read (no batch expansion)."""

    def __init__(self, engine: SpecEngine):
        self.engine = engine

    def score_proposals(self, execute_model_req=None, proposals=None) -> SpeculativeScores:
        eng = self.engine
        B, k = eng.active, eng.k
        with torch.cuda.stream(eng.stream):
            eng._enqueue_score()
        logits = eng.t_logits[:B * (k + 1)].view(B, k + 1, eng.V)
        if eng.sampling:
            token_ids = eng.target_ids[:B]
            with torch.cuda.stream(eng.stream):
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
    max_num_seqs: int = 1
    max_model_len: int = 2048
    temperature: float = 0.0
    seed: int = 0
    split_tokens: int = 256
    use_graph: bool = True
    rows: str = "auto"             # kernel family, see EngineConfig.rows
    draft_rows: str = "auto"       # the draft's kernel family, see EngineConfig.draft_rows
    speculative_disable_by_batch_size: int = 0   # 0 = speculate at every batch size
    spec_decoding_acceptance_method: str = "rejection_sampler"  # or "typical_acceptance_sampler"
    typical_acceptance_sampler_posterior_threshold: float = 0.09
    typical_acceptance_sampler_posterior_alpha: float = 0.3
    # Tensor-parallel verify: the target is sharded over this many ranks (one process per
    # GPU), the draft stays whole on each.  `target` is then either this rank's shard or a
    # (ModelConfig, state_dict) pair that load_model shards itself.
    tensor_parallel_size: int = 1
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


class SpecDecodeWorker:
    """Worker that runs draft-k / verify / accept for every running sequence.

    Construction mirrors a vLLM worker; `vllm_config` may be a SpecDecodeConfig
    (stand-alone use) or any object with a `spec_decode_config` attribute holding
    one (when an engine builds the worker from its own config object)."""

    def __init__(self, vllm_config, local_rank: int = 0, rank: int = 0,
                 distributed_init_method: Optional[str] = None, is_driver_worker: bool = True,
                 engine: Optional[SpecEngine] = None):
        cfg = getattr(vllm_config, "spec_decode_config", vllm_config)
        self.config: SpecDecodeConfig = cfg
        self.local_rank, self.rank = local_rank, rank
        self.distributed_init_method = distributed_init_method
        self.is_driver_worker = is_driver_worker
        self.engine: Optional[SpecEngine] = engine
        self.proposer_worker: Optional[MultiStepWorker] = None
        self.proposer: Optional[SpeculativeProposer] = None
        self.scorer: Optional[SpeculativeScorer] = None
        self.metrics: Optional[AsyncMetricsCollector] = None
        self._slot_of: Dict[str, int] = {}
        self._req_at: List[str] = []
        if engine is not None:
            self._wire()

    # -- life cycle (vLLM worker protocol) ----------------------------------------
    def init_device(self) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("SpecDecodeWorker needs a GPU: there is no CPU path")
        torch.cuda.set_device(self.local_rank)
        tp = self.config.tensor_parallel_size
        if tp > 1:
            from . import parallel
            if not torch.distributed.is_initialized():
                raise RuntimeError("tensor_parallel_size > 1 needs an initialised torch.distributed "
                                   "group (parallel.init_from_env) to exchange the RCCL id")
            if torch.distributed.get_world_size() != tp:
                raise ValueError("tensor_parallel_size must equal the world size")
            self.tp_group = parallel.TensorParallelGroup(self.rank, tp)

    def load_model(self) -> None:
        c = self.config
        target, tp = c.target, c.tensor_parallel_size
        if isinstance(target, tuple):
            from . import weights as W
            cfg, state = target
            if tp > 1:
                vp = c.vocab_parallel_lm_head and cfg.vocab % (16 * tp) == 0
                cfg, state = W.shard_state_dict(cfg, state, tp, self.rank, vocab_parallel=vp)
            target = W.from_state_dict(cfg, state, torch.device("cuda", self.local_rank))
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
            draft = W.from_state_dict(dcfg, dstate, torch.device("cuda", self.local_rank))
        self.engine = SpecEngine(target, draft, EngineConfig(
            k=c.num_speculative_tokens, max_batch=c.max_num_seqs, max_seq_len=c.max_model_len,
            split_tokens=c.split_tokens, temperature=c.temperature, seed=c.seed,
            use_graph=c.use_graph and (tp == 1 or c.tensor_parallel_graph), rows=c.rows,
            draft_rows=c.draft_rows,
            ngram_max=c.ngram_prompt_lookup_max, ngram_min=c.ngram_prompt_lookup_min,
            disable_by_batch_size=c.speculative_disable_by_batch_size,
            acceptance_method=c.spec_decoding_acceptance_method,
            posterior_threshold=c.typical_acceptance_sampler_posterior_threshold,
            posterior_alpha=c.typical_acceptance_sampler_posterior_alpha),
            tp_group=getattr(self, "tp_group", None), draft_tp=draft_tp)
        self._wire()

    def _wire(self) -> None:
        if getattr(self.engine, "ngram", False):
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
    def execute_model(self, execute_model_req: ExecuteModelRequest) -> ModelRunnerOutput:
        """Runs one step: running sequences get a speculative step (1..k+1 tokens
        each), new prompts are prefilled in both models and get their first token."""
        eng = self.engine
        for rid in execute_model_req.finished_requests_ids:
            self._finish(rid)
        running = [g for g in execute_model_req.seq_group_metadata_list if not g.is_prompt]
        prompts = [g for g in execute_model_req.seq_group_metadata_list if g.is_prompt]
        have = getattr(getattr(eng, "cfg", None), "temperature", None)
        for g in prompts:   # checked before anything runs
            want = g.sampling_params.temperature
            if want is not None and have is not None and float(want) != float(have):
                raise ValueError(f"request {g.request_id} asks for temperature {want}; this worker "
                                 f"runs every sequence at {have}")
        unknown = [g.request_id for g in running if g.request_id not in self._slot_of]
        if unknown:
            raise KeyError(f"requests {unknown} are not running on this worker")
        if len(running) != eng.active:
            raise ValueError(f"{eng.active} sequences are running but the step names {len(running)}: "
                             "list every running request or finish it")
        out: Dict[str, List[int]] = {}
        if running:
            emitted = self._run_speculative_decoding_step(execute_model_req)
            for slot, toks in enumerate(emitted):
                out[self._req_at[slot]] = toks
        for g in prompts:
            if g.request_id in self._slot_of:
                raise KeyError(f"request {g.request_id} is already running")
            if eng.active >= eng.cfg.max_batch:
                raise RuntimeError("no free sequence slot (max_num_seqs reached)")
            slot = eng.active
            first = eng.add_request(slot, g.prompt_token_ids, **self._multimodal_inputs(g))
            self._slot_of[g.request_id] = slot
            self._req_at.append(g.request_id)
            out[g.request_id] = [first]
        ids = [g.request_id for g in execute_model_req.seq_group_metadata_list]
        return ModelRunnerOutput(
            req_ids=ids, req_id_to_index={r: i for i, r in enumerate(ids)},
            sampled_token_ids=[out[r] for r in ids],
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
        if eng.cfg.use_graph or eng._spec_disabled():
            return eng.step()
        eng.prepare_step()
        proposals = self.proposer.get_spec_proposals(req, frozenset())
        scores = self.scorer.score_proposals(req, proposals)
        self._verify_tokens(proposals, scores)
        return eng.finish_step()

    def _verify_tokens(self, proposals: SpeculativeProposals, scores: SpeculativeScores) -> None:
        with torch.cuda.stream(self.engine.stream):
            self.engine._enqueue_accept()

    def _finish(self, request_id: str) -> None:
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
