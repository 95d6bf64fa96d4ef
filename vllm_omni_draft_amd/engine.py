"""SpecEngine: the device-resident speculative-decoding loop of one GPU.

Holds the target and draft DecoderModels, their paged KV caches and the
per-sequence state (sequence lengths, the token window {t[n-2], t[n-1],
d_1..d_k}, generated tokens, acceptance counters) as device tensors, and
enqueues one whole speculative step — k draft passes, one (k+1)-token target
verify pass, acceptance, state update — without any host synchronisation.  A
step of fixed shape is captured into a hipGraph and replayed.

State invariant between steps, for a sequence of n tokens: both caches hold
K/V of tokens 0..n-2 (the draft's may lack token n-2 after a step in which
every proposal was accepted; its first pass therefore always feeds the last
TWO tokens, which is idempotent when it did not).

This is the machinery behind the SpecDecodeWorker / Proposer / Scorer surface
in spec_decode_worker.py (vLLM-V0 names; the reference only documents the
V1 step contract they plug into, /root/reference/src/doc/engine_core.md:117-122).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib, _rt, ops
from .kv_cache import PagedKVCache
from .model_runner import MAX_ROWS, DecoderModel
from .weights import ModelWeights


@dataclass
class EngineConfig:
    k: int = 5                     # proposals per step; 0 = plain autoregressive decoding
    max_batch: int = 1
    max_seq_len: int = 2048        # prompt + generated tokens (+ k+1 slack is added)
    split_tokens: int = 256        # attention keys per workgroup, target model
    draft_split_tokens: int = 0    # same for the draft model (0 = like the target)
    temperature: float = 0.0       # 0 = greedy
    seed: int = 0
    use_graph: bool = True
    gen_capacity: int = 4096       # generated tokens kept on the device per sequence
    # Kernel family of the linear layers: "narrow" (<= 16 token rows per pass, larger batches
    # run as sub-batches), "wide" (<= 256 rows per pass) or "auto" (wide when a verify pass of
    # the whole batch exceeds 16 rows).  The families round differently, so an autoregressive
    # engine that is to reproduce a speculative engine's stream must be given the same one.
    rows: str = "auto"
    # Kernel family of the draft model's passes: "narrow", "wide", "same" as the target's, or
    # "auto": narrow up to 32 sequences (beyond 16 as two sub-batches on parallel streams, see
    # draft_lanes), else the target's.  The draft's rounding never shows in the output (only
    # in what gets proposed), so its family may differ from the target's; measured with a
    # 0.5B draft the narrow draft makes the step 18 % shorter at 8 sequences, 6 % at 24 and
    # 1 % at 32 (26 % longer there with the sub-batches one after the other).
    draft_rows: str = "auto"
    # Sub-batches of a narrow draft pass (batches beyond 16 sequences) run on this many HIP
    # streams at once (forked and joined inside the step, also under graph capture): the small
    # kernels of a small draft model leave most of the chip idle, so two sub-batches overlap
    # almost completely.  1 = one after the other.
    draft_lanes: int = 4
    # Prompt-lookup proposals instead of a draft model (k > 0, draft = None): the last
    # ngram_max..ngram_min tokens of a sequence are searched in its own history and the k
    # tokens after the most recent match are proposed (none if nothing matches).
    ngram_max: int = 0
    ngram_min: int = 1
    # Proposals made outside the engine (k > 0, draft = None, ngram_max = 0): the caller hands
    # every step up to k tokens per sequence (set_external_proposals / step(proposals=...)),
    # the engine only verifies them.  A sequence without proposals gets one target token.
    external_proposals: bool = False
    # With more than this many running sequences a step does not speculate (0 = always
    # speculate): one target token per sequence, and one draft pass that only keeps the
    # draft's KV cache in step (upstream's speculative_disable_by_batch_size).
    disable_by_batch_size: int = 0
    # "always": every step speculates (unless disable_by_batch_size says no).  "auto": the engine
    # times its own steps per batch size in both modes (wall clock from one step's start to the
    # next one's, so the host's share counts; intervals with an admission, a removal or a graph
    # capture in them are left out) and speculates in a step only while tokens per second of
    # speculative steps (emitted per step / step time, running means) beat the plain target step's; the other mode is probed for policy_probe_steps
    # steps every policy_probe_every steps, so a change in acceptance or context is noticed.
    # Steps that do not speculate then leave the draft out altogether: its cache is brought up
    # to date by one wide pass over the missing tokens every few steps and before the next
    # speculative step.  Greedy streams do not depend on the choice; a seeded sampling stream
    # does (the modes draw different numbers of uniforms).  Draft-model proposals only (no
    # trees, no host-made proposals), one rank only.
    spec_policy: str = "always"
    policy_probe_every: int = 96
    policy_probe_steps: int = 2
    # Acceptance rule under sampling (temperature > 0): "rejection_sampler" (lossless,
    # Leviathan et al.) or "typical_acceptance_sampler" (a proposal stands if the target gives
    # it more than min(posterior_threshold, posterior_alpha * exp(-entropy)); the first
    # rejected position emits the target's most likely token).  Greedy decoding matches tokens.
    acceptance_method: str = "rejection_sampler"
    posterior_threshold: float = 0.09
    posterior_alpha: float = 0.3
    # Multi-candidate verification (host-made proposals only: prompt lookup or external).
    # Up to tree_candidates continuations per sequence are merged into a token tree of at most
    # tree_nodes nodes (0: tree_candidates * k, capped by 64 / (Hq/Hkv) - 1) and scored by ONE
    # verify pass.  Greedy: the longest root path that matches the target's own tokens is
    # accepted.  Sampling: multi-step speculative sampling with the candidates as point-mass
    # proposals (sd_tree_accept_sample).  1 = a single chain of k proposals.
    tree_candidates: int = 1
    tree_nodes: int = 0
    # > 1: the DRAFT MODEL grows the tree.  Its first pass's b most likely tokens become b
    # branches under the root; every branch is continued greedily to depth k by tree-masked
    # draft passes (all branches of a sequence in one pass per level), and one target pass
    # verifies the b*k nodes (greedy walk, or sampling with the nodes as point-mass proposals).
    # (b*k + 1) * (query heads per KV head) <= 64 on the target.
    draft_tree_branches: int = 0
    # Draft passes of <= 4 rows with two linear layers per launch (DecoderModel fused_pairs):
    # same proposals bit for bit, three launches per layer instead of five.  Off by default:
    # measured slower than the five launches on MI355X.
    draft_fused_pairs: bool = False
    # Draft passes of <= 4 rows with every layer as THREE launches (attention + output projection
    # in one, gate/up + down in one; DecoderModel fused_layers).  The draft's rows then differ
    # from the five-launch layer's in the last bits (and from run to run: float atomics), which
    # changes at most what is proposed, never what is emitted.  Off by default: measured on
    # MI355X the three launches take as long as the five they replace (a 1B draft's layer:
    # 11.5 vs 11.2 us and 23.7 vs 23.1 us for the two pairs).
    draft_fused_layers: bool = False
    # MiB of weights (output projection, then gate/up) that the idle compute units of each
    # layer's attention launch read into L2 ahead of the launches that stream them, for the
    # target's and the draft's narrow-family passes.  Same bits either way.
    target_read_ahead_mb: int = 0
    draft_read_ahead_mb: int = 0
    # fp8 models only: passes of the wide family (prompt chunks; every pass of a wide engine)
    # quantise their activations to fp8 per token row and multiply on the fp8 matrix
    # instruction (W8A8).  Applies to target and draft alike where their weights are fp8.
    fp8_activations: bool = False


def acceptance_pattern(accept_prob: float, k: int, length: int = 16, rotate: int = 0) -> List[int]:
    """A cyclic schedule of accepted proposals per step, `length` entries in 0..k, whose counts
    follow the distribution independent acceptances of probability accept_prob give
    (P(a) = p^a (1 - p) for a < k, p^k for a = k; largest-remainder rounding), spread evenly
    over the cycle.  Deterministic: the mean accept-len of a run over whole cycles is the
    pattern's mean."""
    p = min(max(float(accept_prob), 0.0), 1.0)
    probs = [p ** a * (1.0 - p) for a in range(k)] + [p ** k]
    want = [q * length for q in probs]
    counts = [int(w) for w in want]
    for a in sorted(range(k + 1), key=lambda a: want[a] - counts[a], reverse=True)[:length - sum(counts)]:
        counts[a] += 1
    # spread every value evenly over the cycle: value a sits at fractions (j + 0.5) / count
    slots = sorted((((j + 0.5) / counts[a], -a), a) for a in range(k + 1) for j in range(counts[a]))
    pat = [a for _, a in slots]
    r = rotate % length
    return pat[r:] + pat[:r]


class SpecEngine:
    def __init__(self, target: ModelWeights, draft: Optional[ModelWeights], cfg: EngineConfig,
                 tp_group=None, draft_tp: bool = False):
        """tp_group: the target is this rank's tensor-parallel shard (its two row-parallel
        projections per layer are summed over the group inside the native forward).  The
        draft is whole on every rank, or with draft_tp this rank's shard over the same group
        (a draft that is not much smaller than the target's share: replicated, its passes
        would stream more bytes per GPU than the sharded verify pass); its LM head may then be
        split by vocabulary rows like the target's.  Every rank runs the same steps on the
        same inputs."""
        if draft_tp and tp_group is None:
            raise ValueError("draft_tp needs a tp_group")
        d_sub = draft is not None and getattr(draft, "head_map", None) is not None
        if d_sub and (draft_tp or cfg.temperature > 0.0):
            raise ValueError("a draft head over a vocabulary subset proposes greedily and is not "
                             "sharded (temperature 0, no draft_tp)")
        if getattr(target, "head_map", None) is not None:
            raise ValueError("the target's LM head must cover the whole vocabulary")
        if draft is not None and draft.cfg.lm_vocab and not draft_tp and not d_sub:
            raise ValueError("a vocabulary-parallel draft head needs draft_tp")
        # a verify pass scores k+1 rows of a sequence in one workgroup column of the attention
        # kernel: (k+1) * (query heads per KV head) <= 64; the fused acceptance walks <= 16 rows
        k_max = min(15, 64 // (target.cfg.n_heads // target.cfg.n_kv_heads) - 1)
        if cfg.k < 0 or cfg.k > k_max:
            raise ValueError(f"k must be in 0..{k_max} for this target")
        if cfg.acceptance_method not in ("rejection_sampler", "typical_acceptance_sampler"):
            raise ValueError(f"unknown acceptance_method {cfg.acceptance_method!r}")
        self.typical = cfg.acceptance_method == "typical_acceptance_sampler"
        self.ngram = cfg.k > 0 and draft is None and cfg.ngram_max > 0
        self.external = cfg.k > 0 and draft is None and cfg.external_proposals and not self.ngram
        self.host_props = self.ngram or self.external     # proposals are uploaded by the host
        if cfg.k > 0 and draft is None and not self.host_props:
            raise ValueError("k > 0 needs a draft model (or ngram_max > 0 for prompt lookup, or "
                             "external_proposals)")
        if self.ngram and not 1 <= cfg.ngram_min <= cfg.ngram_max:
            raise ValueError("need 1 <= ngram_min <= ngram_max")
        if draft is not None and draft.cfg.vocab != target.cfg.vocab:
            raise ValueError("draft and target must share a vocabulary "
                             f"({draft.cfg.vocab} vs {target.cfg.vocab})")
        self.cfg = cfg
        self.k = cfg.k
        dev = target.embed.device
        self.device = dev
        B, k = cfg.max_batch, cfg.k
        self.draft_tree = int(cfg.draft_tree_branches) if cfg.draft_tree_branches > 1 else 0
        if self.draft_tree:
            if draft is None or cfg.k < 1 or cfg.tree_candidates > 1:
                raise ValueError("draft_tree_branches needs a draft model, k >= 1 and tree_candidates == 1")
            if self.draft_tree > ops.TOPK_TOKENS_MAX:
                raise ValueError(f"draft_tree_branches must be at most {ops.TOPK_TOKENS_MAX}")
            if (draft.cfg.arch != "llama" or draft.cfg.mrope_section or d_sub or draft_tp
                    or tp_group is not None or draft.cfg.lm_vocab):
                raise ValueError("a draft-grown tree needs a rotary draft without mrope_section, with a "
                                 "whole-vocabulary head, on one rank")
        self.tree = cfg.tree_candidates > 1 or bool(self.draft_tree)
        self.tree_nodes = 0
        if self.tree:
            if not self.host_props and not self.draft_tree:
                raise ValueError("tree_candidates > 1 needs host-made proposals (prompt lookup or "
                                 "external_proposals) and no draft model")
            if cfg.temperature > 0.0 and cfg.acceptance_method != "rejection_sampler":
                raise ValueError("token trees under sampling use the rejection sampler "
                                 "(typical acceptance walks chains only)")
            if target.cfg.arch != "llama" or target.cfg.mrope_section:
                raise ValueError("token trees need a rotary target without mrope_section")
            cap = min(64 // (target.cfg.n_heads // target.cfg.n_kv_heads) - 1, 63)
            if self.draft_tree:
                self.tree_nodes = self.draft_tree * k
                if self.tree_nodes > cap:
                    raise ValueError(f"{self.draft_tree} branches of depth {k} are {self.tree_nodes} tree "
                                     f"nodes; this target verifies at most {cap}")
            else:
                self.tree_nodes = min(cfg.tree_nodes or cfg.tree_candidates * k, cap)
            if self.tree_nodes < k:
                raise ValueError(f"a tree of {self.tree_nodes} nodes cannot hold a chain of k={k}")
        q_max = max(k, self.tree_nodes) + 1           # rows of a verify pass per sequence
        self._slack = q_max + 1                       # cache slots a step may write past the last token
        self.max_ctx = cfg.max_seq_len + self._slack
        if cfg.rows not in ("auto", "narrow", "wide"):
            raise ValueError("rows must be 'auto', 'narrow' or 'wide'")
        self.wide = cfg.rows == "wide" or (cfg.rows == "auto" and B * q_max > MAX_ROWS)
        self.target = DecoderModel(target, self.max_ctx, cfg.split_tokens, tp_group=tp_group,
                                   wide=self.wide, read_ahead_mb=cfg.target_read_ahead_mb,
                                   fp8_activations=cfg.fp8_activations and bool(getattr(target, "fp8", False)))
        if cfg.draft_rows not in ("auto", "same", "narrow", "wide"):
            raise ValueError("draft_rows must be 'auto', 'same', 'narrow' or 'wide'")
        if cfg.draft_rows == "auto":
            draft_wide = self.wide and B > 2 * MAX_ROWS
        else:
            draft_wide = self.wide if cfg.draft_rows == "same" else cfg.draft_rows == "wide"
        self.draft = (DecoderModel(draft, self.max_ctx, cfg.draft_split_tokens or cfg.split_tokens,
                                   wide=draft_wide, tp_group=tp_group if draft_tp else None,
                                   fused_pairs=cfg.draft_fused_pairs,
                                   fused_layers=bool(cfg.draft_fused_layers) and not cfg.draft_fused_pairs
                                   and not draft_tp, read_ahead_mb=cfg.draft_read_ahead_mb,
                                   fp8_activations=cfg.fp8_activations and bool(getattr(draft, "fp8", False)))
                      if k > 0 and draft is not None else None)
        self.max_rows = self.target.max_rows
        if self.draft_tree:
            rows = self.draft_tree * (k - 1)          # the widest level pass, per sequence
            if rows > self.draft.max_rows or rows * (draft.cfg.n_heads // draft.cfg.n_kv_heads) > 64:
                raise ValueError(f"the draft's level passes would carry {rows} rows per sequence: more "
                                 "than its kernels take (draft_rows='wide' lifts the first bound)")
        self._lanes: List[torch.cuda.Stream] = []
        self._lane_ws: List[torch.Tensor] = []
        # (collectives of one communicator must stay on one stream: no lanes for a sharded draft)
        if (self.draft is not None and not self.draft.wide and cfg.draft_lanes > 1 and not draft_tp
                and _rt.on_gpu()):
            n_sub = -(-2 * B // self.draft.max_rows)      # the first pass carries 2 tokens per sequence
            n = min(cfg.draft_lanes, n_sub)
            if n > 1:
                self._lanes = [torch.cuda.Stream(device=dev) for _ in range(n - 1)]
                self._lane_ws = [self.draft.workspace] + [torch.empty_like(self.draft.workspace)
                                                          for _ in range(n - 1)]
        self.V = target.cfg.vocab
        pages = PagedKVCache.pages_for(target.cfg, B, self.max_ctx)
        self.t_kv = PagedKVCache(target.cfg, pages, B, self.max_ctx, dev)
        self.d_kv = (PagedKVCache(draft.cfg, pages, B, self.max_ctx, dev)
                     if self.draft is not None else None)
        self._pending_props: Optional[List[List[int]]] = None
        # prompts that are being prefilled piece by piece (prefill_begin / prefill_advance):
        # handle -> {prompt, done, slot, sampling arguments}; they hold batch slots behind the
        # active sequences, taken from the top of the batch
        self._prefilling: dict = {}
        self._next_handle = 1
        if self.ngram:
            self.history: List[List[int]] = [[] for _ in range(B)]   # prompt + emitted tokens
        if self.host_props:
            self.ext_props = torch.zeros(B, k, **dict(dtype=torch.int32, device=dev))
            self.prop_lens = torch.zeros(B, **dict(dtype=torch.int32, device=dev))
            self._props_host = _rt.pin(torch.zeros(B, k, dtype=torch.int32))
            self._lens_host = _rt.pin(torch.zeros(B, dtype=torch.int32))

        i32 = dict(dtype=torch.int32, device=dev)
        self.seq_lens = torch.zeros(B, **i32)
        self.tok_buf = torch.zeros(B, k + 2, **i32)
        # every token a sequence can generate fits (a multi-step read-back slices this buffer)
        self.gen_tokens = torch.zeros(B, max(cfg.gen_capacity, self.max_ctx), **i32)
        self.gen_counts = torch.zeros(B, **i32)
        self.counters = torch.zeros(3, dtype=torch.int64, device=dev)
        self.zero_lens = torch.zeros(B, **i32)
        self.prompt_buf = torch.zeros(max(cfg.max_seq_len, 8), **i32)
        self._prompt_rows: Optional[torch.Tensor] = None   # [max_batch, max_seq_len], batched prefill
        self.t_logits = torch.empty(max(MAX_ROWS, B * q_max), self.V,
                                    dtype=torch.float32, device=dev)
        self.d_logits = torch.empty(max(MAX_ROWS, 2 * B, B * self.draft_tree * (k - 1)), self.V,
                                    dtype=torch.float32, device=dev)
        self.ids_tmp = torch.zeros(max(B, 1), dtype=torch.int64, device=dev)
        self.target_ids = torch.zeros(B, k + 1, dtype=torch.int64, device=dev)
        self.draft_ids = torch.zeros(B, max(k, 1), dtype=torch.int64, device=dev)
        self.out_ids = torch.full((B, k + 1), -1, dtype=torch.int64, device=dev)
        n_rows = max(MAX_ROWS, B * q_max)
        if self.tree:
            q = self.tree_nodes + 1
            # tokens, parents, depths, lengths and ancestor masks share one buffer: one upload
            # per step.  Row 0 of a tree is the root (the last emitted token, written on the device).
            n_i32 = 2 * ((3 * B * q + B + 1) // 2)        # even: the 64-bit masks follow
            self._tree_dev = torch.zeros(n_i32 + 2 * B * q, **i32)
            self._tree_host = _rt.pin(torch.zeros(n_i32 + 2 * B * q, dtype=torch.int32))

            def views(buf):
                return (buf[:B * q].view(B, q), buf[B * q:2 * B * q].view(B, q),
                        buf[2 * B * q:3 * B * q].view(B, q), buf[3 * B * q:3 * B * q + B],
                        buf[n_i32:].view(torch.int64).view(B, q))
            (self.tree_tok, self.tree_parent, self.tree_depth, self.tree_len,
             self.tree_mask) = views(self._tree_dev)
            self._tree_np = [v.numpy() for v in views(self._tree_host)]
            self.tree_rope = torch.zeros(B * q, **i32)
            self.tree_ids = torch.zeros(B * q, dtype=torch.int64, device=dev)
            self.tree_path = torch.zeros(B, k, **i32)
            self.tree_npath = torch.zeros(B, **i32)
            if self.draft_tree:
                # The topology never changes: row 1 + (d-1)*b + c is branch c at depth d, its
                # parent the same branch one level up (the root for d = 1).  Uploaded once; the
                # tokens are written on the device by the draft's passes.
                nb = self.draft_tree
                h_tok, h_par, h_dep, h_len, h_mask = self._tree_np
                masks = [1]
                for r in range(1, q):
                    par = 0 if r <= nb else r - nb
                    h_par[:, r], h_dep[:, r] = par, (r - 1) // nb + 1
                    masks.append(masks[par] | (1 << r))
                h_len[:] = q
                h_mask[:] = [m - (1 << 64) if m >= (1 << 63) else m for m in masks]
                self._tree_dev.copy_(self._tree_host)
                # The draft's pass of level d covers tree rows 1..d*b: its ancestor masks are the
                # tree's without the root bit (the root is context there), its rotary positions
                # the rows' (copied into a buffer of the pass's own row stride in every step).
                self.dtree_mask, self.dtree_rope = [None], [None]
                for d in range(1, k):
                    m = torch.tensor([masks[t + 1] >> 1 for t in range(d * nb)], dtype=torch.int64)
                    self.dtree_mask.append(m.repeat(B).contiguous().to(dev))
                    self.dtree_rope.append(torch.zeros(B, d * nb, **i32))
        self.part_val = torch.empty(n_rows, ops.ARGMAX_PARTS, dtype=torch.float32, device=dev)
        self.part_idx = torch.empty(n_rows, ops.ARGMAX_PARTS, dtype=torch.int32, device=dev)
        # Vocabulary-parallel target head (tensor parallelism): a pass leaves this rank's
        # slice of the logits in t_out; greedy steps exchange slice maxima (2 x 256 bytes per
        # row and rank), sampling steps gather whole rows into t_logits.
        self.tp = tp_group
        self._tp_timeouts_seen = 0
        self.vp = tp_group is not None and target.cfg.lm_vocab > 0
        self.t_out = self.t_logits
        if self.vp:
            n_tp, Vl = tp_group.world_size, target.cfg.lm_vocab
            if Vl * n_tp != self.V:
                raise ValueError("the ranks' LM head slices do not add up to the vocabulary")
            self.t_out = torch.empty(n_rows, Vl, dtype=torch.float32, device=dev)
            self.vp_send = torch.empty(2 * n_rows * ops.ARGMAX_PARTS, dtype=torch.float32, device=dev)
            self.vp_recv = torch.empty(n_tp * self.vp_send.numel(), dtype=torch.float32, device=dev)
            self.vp_rows = torch.empty(n_tp * n_rows * Vl, dtype=torch.float32, device=dev)
        # the same for a sharded draft with a vocabulary-parallel head (buffers shared with the
        # target's exchange: everything runs on one stream)
        # a draft head over a vocabulary subset: its row indices become token ids after every
        # proposal (sd_map_tokens)
        self.d_map = draft.head_map if d_sub else None
        self.d_vp = self.draft is not None and draft.cfg.lm_vocab > 0 and not d_sub
        self.d_out = self.d_logits
        if d_sub:
            self.d_out = torch.empty(self.d_logits.shape[0], draft.cfg.lm_vocab, dtype=torch.float32,
                                     device=dev)
        if self.d_vp:
            if not self.vp or draft.cfg.lm_vocab != target.cfg.lm_vocab:
                raise ValueError("a vocabulary-parallel draft head needs the target's head split "
                                 "the same way")
            self.d_out = torch.empty(self.d_logits.shape[0], draft.cfg.lm_vocab, dtype=torch.float32,
                                     device=dev)
        # Step record {n_emitted, seq_len, tokens}: the last kernel of a step writes it to
        # pinned host memory, so returning a step's tokens takes one stream
        # synchronisation and no copy.
        # two records: with overlapped read-back (step_overlapped) a step writes one while
        # the host still reads the other
        # Multimodal rotary embedding: rotary position of a generated token = its cache
        # position + the delta its prompt's image tokens left behind (0 for text prompts)
        self.rope_delta = torch.zeros(B, dtype=torch.int32, device=dev)
        self._records = _rt.pin(torch.full((2, B, k + 3), -1, dtype=torch.int32))
        self._parity = 0
        self._tree_step = False  # whether the last step verified token trees (not chains)
        self._inflight = None   # (event, record parity, sequences) of a step not yet read back
        self._steps_pending = 0               # steps enqueued since the last read-back
        # speculate-or-not policy (EngineConfig.spec_policy)
        if cfg.spec_policy not in ("always", "auto"):
            raise ValueError("spec_policy must be 'always' or 'auto'")
        self._auto = (cfg.spec_policy == "auto" and self.draft is not None and not self.tree
                      and not self.host_props)
        if self._auto and tp_group is not None:
            raise ValueError("spec_policy='auto' times the steps of one rank: not with tensor parallelism")
        self._lazy_draft = self._auto      # steps that do not speculate leave the draft out
        self._lag = 0                      # such steps since the draft cache was last complete
        self._lag_max = min(8, self._slack + 1)   # (positions a catch-up pass may touch stay allocated)
        self._no_spec_now = False          # this step's decision
        self._pol: dict = {}               # batch size -> running means, see _decide_no_spec
        self._last_start = None            # (wall clock, batch size, mode) of the last step's start
        self._clean = False                # nothing but that step happened since (no admission, no capture)
        self._arange = torch.arange(16, dtype=torch.int32, device=dev)
        self.sampling = cfg.temperature > 0.0
        if self.sampling:
            f32 = dict(dtype=torch.float32, device=dev)
            self.t_probs = torch.empty(B, q_max, self.V, **f32)   # (q_max = k + 1 without trees)
            self.d_probs = torch.empty(B, max(k, 1), self.V, **f32)
            # one flat buffer of uniforms, refilled on the device at the start of every step
            kk = max(k, 1)
            ka = self.tree_nodes + 1 if self.tree else kk   # acceptance tests: one per tree row
            sizes = [kk * B, B * ka, B * (k + 1), B]
            self.u_all = torch.zeros(sum(sizes), **f32)
            parts = torch.split(self.u_all, sizes)
            self.u_draft = parts[0].view(kk, B)
            self.u_accept = parts[1].view(B, ka)
            self.u_recover = parts[2].view(B, k + 1)
            self.u_bonus = parts[3]
            # every sequence draws from a stream of its own (its request's seed, or one derived
            # from cfg.seed and the order of admission): what a sequence samples does not depend
            # on its batch mates
            # one temperature per sequence (its request's, or cfg.temperature); 0 = greedy
            self.temps = torch.full((B,), float(cfg.temperature), **f32)
            # per-sequence filters of the target distribution (off: 0, 1.0, 0.0); the draft's
            # distribution stays whole, which costs acceptance at most, never correctness
            self.top_k = torch.zeros(B, dtype=torch.int32, device=dev)
            self.top_p = torch.ones(B, **f32)
            self.min_p = torch.zeros(B, **f32)
            self.u_seeds = torch.zeros(B, dtype=torch.int64, device=dev)
            self.u_counters = torch.zeros(B, dtype=torch.int64, device=dev)
            self.host_seeds: List[int] = [0] * B
            self._admitted = 0
            self.u_regions = ops.uniform_regions([
                (0, kk, B, 1), (kk * B, ka, 1, ka), ((kk + ka) * B, k + 1, 1, k + 1),
                ((kk + ka) * B + B * (k + 1), 1, 0, 1)])
            self.bonus_ids = torch.zeros(B, dtype=torch.int64, device=dev)
            self.samp_scratch = ops.sampling_scratch(max(MAX_ROWS, B * q_max), dev)
        # Per-request logit masks (logit_bias, allowed_token_ids, banned tokens): one f32 row per
        # batch slot, allocated when the first such request arrives; the device flags tell the
        # kernel which sequences carry one (the rest are skipped)
        self.bias: Optional[torch.Tensor] = None
        self.t_raw: Optional[torch.Tensor] = None     # unmasked copy of a pass's logits, for logprobs
        self.bias_on = torch.zeros(B, dtype=torch.int32, device=dev)
        self._bias_host: List[bool] = [False] * B
        # Log-probabilities of the emitted tokens (ModelRunnerOutput.logprobs): off (None) unless
        # the caller sets logprobs_top = the number of most likely tokens to report beside each
        # emitted one (0: the emitted token's own log-probability and rank only).  While it is
        # set, every synchronous step and every admission leaves last_logprobs: per sequence (in
        # the order of the call's return value), per emitted token, (token ids [1 + top], their
        # log-probabilities under the target's raw logits, the emitted token's rank).
        self.logprobs_top: Optional[int] = None
        self.last_logprobs: Optional[List[List[tuple]]] = None
        # synthetic acceptance control (benchmarks with random weights)
        self.truth: Optional[torch.Tensor] = None
        self.accept_pattern: List[int] = []
        self.force_ctr = torch.zeros(B, dtype=torch.int32, device=dev)
        self.active = 0                       # sequences currently in the batch
        self.host_lens: List[int] = [0] * B   # upper bound of seq_lens known to the host
        self.host_gen: List[int] = [0] * B
        self._graphs: dict = {}               # (B, k, truth?, ctx bucket) -> captured step
        self._ctx = self.max_ctx              # context bound the passes are enqueued with
        # stream capture is not allowed on the legacy default stream
        self.stream = _rt.new_stream(dev)
        _rt.synchronize(dev)  # buffers above were filled on the default stream

    # ------------------------------------------------------------------ requests
    def reset(self) -> None:
        with _rt.stream_ctx(self.stream):
            self._reset()

    def _reset(self) -> None:
        self._inflight = None
        self._lag, self._no_spec_now, self._last_start, self._clean = 0, False, None, False
        for b in range(self.cfg.max_batch):
            self.t_kv.release(b)
            if self.d_kv is not None:
                self.d_kv.release(b)
            self.host_lens[b] = 0
            self.host_gen[b] = 0
            if self.ngram:
                self.history[b] = []
        self.seq_lens.zero_()
        self.gen_counts.zero_()
        self.counters.zero_()
        self.active = 0
        self._prefilling.clear()
        self.bias_on.zero_()
        self._bias_host = [False] * self.cfg.max_batch

    def set_synthetic_acceptance(self, truth: Optional[torch.Tensor], accept_prob: float = 0.0,
                                 seed: int = 0, pattern: Optional[Sequence[int]] = None,
                                 pattern_len: int = 16) -> None:
        """Benchmark control for runs with random weights (sd_debug_force_proposals; not part
        of the product path).  truth [B, L] int32: the token at every absolute position of the
        stream the target would emit.  After the draft's passes the proposals are overwritten
        so that a fixed, cyclic schedule decides how many are accepted in each step: `pattern`
        (accepted proposals per step), or the schedule acceptance_pattern(accept_prob, k,
        pattern_len) whose counts follow what independent acceptances of probability
        accept_prob would give (seed rotates its start).  None switches the control off."""
        if truth is not None and (truth.dtype != torch.int32 or truth.dim() != 2):
            raise ValueError("truth must be an int32 [B, L] tensor")
        if truth is not None and self.draft is None:
            raise ValueError("synthetic acceptance overrides a draft model's proposals")
        self.truth = truth
        if truth is not None:
            pat = list(pattern) if pattern is not None else acceptance_pattern(
                accept_prob, self.k, pattern_len, rotate=seed)
            if not 1 <= len(pat) <= ops.DEBUG_PATTERN_MAX or any(not 0 <= a <= self.k for a in pat):
                raise ValueError("pattern: 1..64 entries in 0..k")
            self.accept_pattern = pat
        self.force_ctr.zero_()
        self._drop_graphs()

    def add_request(self, slot: int, prompt: Sequence[int],
                    input_embeds: Optional[torch.Tensor] = None,
                    rope_positions: Optional[torch.Tensor] = None,
                    truth_row: Optional[Sequence[int]] = None,
                    seed: Optional[int] = None, temperature: Optional[float] = None,
                    top_k: int = 0, top_p: float = 1.0, min_p: float = 0.0,
                    bias: Optional[dict] = None, decoded_from: Optional[int] = None) -> int:
        """Prefills both models with `prompt` into batch slot `slot`, emits the
        first token from the target and returns it (this synchronises).
        Multimodal prompts (multimodal.prepare_prompt builds both): input_embeds
        [len(prompt), hidden] bf16 are the target's input rows (image embeddings in the
        placeholder tokens' places; the draft still embeds the token ids), rope_positions
        [3, len(prompt)] the rotary positions for models with mrope_section.  truth_row
        (benchmarks with synthetic acceptance, see set_synthetic_acceptance): the request's
        token stream from position 0, stored in the slot's row of the truth table.  seed
        (sampling): the request's own random stream; the tokens it samples are then the same
        whatever else is in the batch (None: a stream derived from EngineConfig.seed and the
        order of admission).  temperature (sampling engines): the request's own temperature,
        0 for a greedy request among sampling ones (None: EngineConfig.temperature); top_k /
        top_p / min_p filter the target distribution the request samples from (0 / 1.0 / 0.0: off).
        bias: the request's logit mask, {"logit_bias": {token: value}, "allowed_token_ids": [...],
        "bad_token_ids": [...]} (any subset), added to the target's logits before every argmax /
        softmax of the request (see _set_bias).  decoded_from (a sequence that is being rebuilt
        from its own tokens): the position of the first token that decode passes produced; the
        tokens from there on are recomputed by passes of the family and shape limits of a decode
        pass, so that their cache entries get the bits they had (see _prefill_range)."""
        with _rt.stream_ctx(self.stream):
            if truth_row is not None:
                if self.truth is None:
                    raise ValueError("truth_row needs set_synthetic_acceptance first")
                row = torch.tensor(list(truth_row)[:self.truth.shape[1]], dtype=torch.int32)
                self.truth[slot, :row.numel()].copy_(row)
            self._set_temperature(slot, temperature, top_k, top_p, min_p)
            self._set_bias(slot, bias)
            return self._add_request(slot, prompt, input_embeds, rope_positions, seed,
                                     decoded_from=decoded_from)

    def _bias_row(self, spec: Optional[dict]) -> Optional[torch.Tensor]:
        """The f32 row [V] of a request's logit mask, or None if it has none.  allowed_token_ids
        bans everything else (-inf), bad_token_ids bans those, logit_bias adds its values."""
        if not spec or not any(spec.get(k) for k in ("logit_bias", "allowed_token_ids", "bad_token_ids")):
            return None
        unknown = set(spec) - {"logit_bias", "allowed_token_ids", "bad_token_ids"}
        if unknown:
            raise ValueError(f"unknown logit mask entries {sorted(unknown)}")
        allowed = [int(t) for t in spec.get("allowed_token_ids") or []]
        banned = [int(t) for t in spec.get("bad_token_ids") or []]
        add = {int(t): float(v) for t, v in (spec.get("logit_bias") or {}).items()}
        for t in allowed + banned + list(add):
            if not 0 <= t < self.V:
                raise ValueError(f"token {t} of a logit mask is outside the vocabulary")
        row = torch.zeros(self.V, dtype=torch.float32)
        if allowed:
            row.fill_(float("-inf"))
            row[allowed] = 0.0
        for t, v in add.items():
            row[t] += v
        if banned:
            row[banned] = float("-inf")
        if not bool(torch.isfinite(row).any()):
            raise ValueError("the logit mask bans every token")
        return row

    def _set_bias(self, slot: int, spec: Optional[dict]) -> None:
        row = self._bias_row(spec)
        if not 0 <= slot < self.cfg.max_batch:      # validation only
            return
        if row is None:
            if self._bias_host[slot]:
                self.bias_on[slot] = 0
                self._bias_host[slot] = False
            return
        if self.bias is None:
            self.bias = torch.zeros(self.cfg.max_batch, self.V, dtype=torch.float32, device=self.device)
        self.bias[slot].copy_(row)
        self.bias_on[slot] = 1
        self._bias_host[slot] = True

    def _apply_bias(self, n_rows: int, rows_per_seq: int, slot0: int = 0) -> None:
        """Adds the sequences' logit masks to rows 0..n_rows-1 of the target's logits of the pass
        just enqueued (row r -> slot slot0 + r // rows_per_seq), whole rows or this rank's
        vocabulary slice."""
        if self.bias is None:
            return
        if self.logprobs_top is not None:
            # log-probabilities are those of the raw logits (upstream computes them before the
            # logit processors run): keep the rows as the pass left them
            if self.t_raw is None:
                self.t_raw = torch.empty_like(self.t_out)
            self.t_raw[:n_rows].copy_(self.t_out[:n_rows])
        bias = self.bias[slot0:]
        if self.vp:
            Vl = self.target.cfg.lm_vocab
            bias = bias[:, self.tp.rank * Vl:(self.tp.rank + 1) * Vl]
        ops.add_logit_bias(self.t_out[:n_rows], bias, self.bias_on[slot0:], rows_per_seq)

    def _set_temperature(self, slot: int, temperature: Optional[float], top_k: int = 0,
                         top_p: float = 1.0, min_p: float = 0.0) -> None:
        top_k = 0 if top_k is None or top_k < 0 else int(top_k)
        top_p = 1.0 if top_p is None else float(top_p)
        min_p = 0.0 if min_p is None else float(min_p)
        if not 0.0 < top_p <= 1.0 or not 0.0 <= min_p <= 1.0:
            raise ValueError(f"top_p {top_p} must be in (0, 1], min_p {min_p} in [0, 1]")
        if self.sampling:
            t = self.cfg.temperature if temperature is None else float(temperature)
            if not t >= 0.0:
                raise ValueError(f"temperature {temperature} must be >= 0")
            if 0 <= slot < self.cfg.max_batch:
                self.temps[slot] = t
                self.top_k[slot], self.top_p[slot], self.min_p[slot] = top_k, top_p, min_p
        elif temperature not in (None, 0, 0.0):
            raise ValueError("a greedy engine (EngineConfig.temperature == 0) cannot sample; build "
                             "the engine with a temperature > 0 to mix sampling and greedy requests")

    def _seed_slot(self, slot: int, seed: Optional[int]) -> float:
        """Gives batch slot `slot` its random stream; returns the uniform of its first token."""
        m = (1 << 64) - 1
        if seed is None:
            seed = (self.cfg.seed * 0x9e3779b97f4a7c15 + (self._admitted + 1) * 0xbf58476d1ce4e5b9) & m
        self._admitted += 1
        seed &= m
        self.host_seeds[slot] = seed
        self.u_seeds[slot] = seed - (1 << 64) if seed >= (1 << 63) else seed
        self.u_counters[slot] = 0
        return ops.uniform24(seed, 1 << 63, 0)    # a counter value no step reaches

    # ---- prompts prefilled in pieces (a scheduler with chunked prefill)
    def prefill_begin(self, prompt: Sequence[int], seed: Optional[int] = None,
                      temperature: Optional[float] = None, top_k: int = 0, top_p: float = 1.0,
                      min_p: float = 0.0, bias: Optional[dict] = None,
                      decoded_from: Optional[int] = None) -> int:
        """Reserves a batch slot for a text prompt that will be prefilled piece by piece and
        returns its handle.  The sequence joins the batch when prefill_advance has consumed the
        whole prompt; until then steps of the active sequences go on as usual."""
        n = len(prompt)
        if n < 1 or n + 1 > self.cfg.max_seq_len:
            raise ValueError(f"prompt length {n} outside 1..{self.cfg.max_seq_len - 1}")
        if self.active + len(self._prefilling) + 1 > self.cfg.max_batch:
            raise ValueError("no free batch slot for another prompt")
        self._set_temperature(-1, temperature, top_k, top_p, min_p)   # validates only
        self._bias_row(bias)
        used = {e["slot"] for e in self._prefilling.values()}
        slot = max(s for s in range(self.active, self.cfg.max_batch) if s not in used)
        h, self._next_handle = self._next_handle, self._next_handle + 1
        self._prefilling[h] = dict(prompt=[int(t) for t in prompt], done=0, slot=slot, seed=seed,
                                   sampling=(temperature, top_k, top_p, min_p), bias=bias,
                                   decoded_from=decoded_from)
        with _rt.stream_ctx(self.stream):
            self.t_kv.ensure_capacity(slot, n + self._slack)
            if self.d_kv is not None:
                self.d_kv.ensure_capacity(slot, n + self._slack)
        return h

    def prefill_advance(self, handle: int, n_tokens: int) -> Optional[int]:
        """Runs the next n_tokens prompt tokens of a prefill_begin prompt through both models.
        Returns None while prompt tokens remain; with the last one the sequence becomes the next
        active one (slot == the number of active sequences before the call) and its first
        generated token is returned (this synchronises).  A row's bits do not depend on how the
        prompt was cut, so the stream equals that of add_request."""
        if self._inflight is not None:
            raise RuntimeError("flush() the step in flight before changing the batch")
        e = self._prefilling.get(handle)
        if e is None:
            raise KeyError(f"no prompt is being prefilled under handle {handle}")
        prompt, done = e["prompt"], e["done"]
        self._clean = False
        n = len(prompt)
        if n_tokens < 1 or done + n_tokens > n:
            raise ValueError(f"{n_tokens} tokens asked for, {n - done} of the prompt remain")
        end = done + n_tokens
        with _rt.stream_ctx(self.stream):
            if end == n:
                # the prompt's pages move into the next active slot; a prompt still being
                # prefilled there takes this one's slot in exchange
                if e["slot"] != self.active:
                    for kv in (self.t_kv, self.d_kv):
                        if kv is not None:
                            kv.swap_slots(e["slot"], self.active)
                    for other in self._prefilling.values():
                        if other["slot"] == self.active:
                            other["slot"] = e["slot"]
                    e["slot"] = self.active
                del self._prefilling[handle]
                temperature, top_k, top_p, min_p = e["sampling"]
                self._set_temperature(self.active, temperature, top_k, top_p, min_p)
                self._set_bias(self.active, e["bias"])
                return self._add_request(self.active, prompt, seed=e["seed"], start=done,
                                         decoded_from=e["decoded_from"])
            self._prefill_range(e["slot"], prompt, done, end, final=False,
                                decoded_from=e["decoded_from"])
            e["done"] = end
            return None

    def prefill_abort(self, handle: int) -> None:
        """Drops a prompt that is being prefilled and frees its pages."""
        e = self._prefilling.pop(handle, None)
        if e is not None:
            for kv in (self.t_kv, self.d_kv):
                if kv is not None:
                    kv.release(e["slot"])

    def _vacate_active_slot(self) -> None:
        """Slot `active` is about to be filled by a whole prompt: a prompt still being prefilled
        there moves to a free slot further up (there is one: admission checked the batch size)."""
        for h, e in self._prefilling.items():
            if e["slot"] == self.active:
                used = {x["slot"] for x in self._prefilling.values()}
                free = [s for s in range(self.active + 1, self.cfg.max_batch) if s not in used]
                if not free:
                    raise ValueError("no free batch slot: finish or abort a prompt being prefilled")
                for kv in (self.t_kv, self.d_kv):
                    if kv is not None:
                        kv.swap_slots(e["slot"], free[-1])
                e["slot"] = free[-1]

    def _prefill_range(self, slot: int, prompt: Sequence[int], c_from: int, c_to: int, final: bool,
                       rope_positions=None, input_embeds=None, decoded_from: Optional[int] = None) -> None:
        """Passes over prompt tokens c_from..c_to-1 of the sequence in `slot` (both models, chunks
        of prefill_rows tokens on the wide family).  final: c_to is the prompt's end; its last
        token gets a pass of its own whose logits (target) land in t_out[0].
        decoded_from: tokens from this position on were first computed by decode passes (the
        sequence is being rebuilt from its own output): a model whose decode passes run on the
        narrow family recomputes them in narrow passes of at most a decode pass's rows.  A narrow
        row's bits depend on nothing else in its pass, so the cache entries equal those the
        decode passes wrote, whatever the grouping was then."""
        n = len(prompt)
        self.prompt_buf[c_from:c_to].copy_(torch.tensor(list(prompt[c_from:c_to]), dtype=torch.int32))
        self._ctx = self._ctx_bucket(n + self._slack)
        zl = self.zero_lens[slot:slot + 1]
        body_end = c_to - 1 if final else c_to
        for model, kv in ((self.target, self.t_kv), (self.draft, self.d_kv)):
            if model is None:
                continue
            # only the last token's logits are needed, so the prompt ends with a 1-token pass
            chunk = model.prefill_rows
            wide = chunk > MAX_ROWS or self.wide
            narrow_from = n if decoded_from is None or model.wide else max(int(decoded_from), 0)
            narrow_rows = min(MAX_ROWS, 64 // (model.cfg.n_heads // model.cfg.n_kv_heads))
            wide_end = min(body_end, narrow_from)
            passes = [(c0, min(c0 + chunk, wide_end), wide) for c0 in range(c_from, wide_end, chunk)]
            passes += [(c0, min(c0 + narrow_rows, body_end), False)
                       for c0 in range(max(c_from, wide_end), body_end, narrow_rows)]
            if final:
                passes.append((body_end, c_to, wide and narrow_from >= n))
            for c0, c1, family_wide in passes:
                if c1 <= c0:
                    continue
                want = final and c1 == n and model is self.target
                view = _KvView(kv, slot)
                mm = {}
                if rope_positions is not None and model.cfg.mrope_section:
                    mm["rope_pos"] = rope_positions[:, c0:c1].contiguous()
                if input_embeds is not None and model is self.target:
                    mm["input_embeds"] = input_embeds[c0:c1].contiguous()
                model.forward(view, self.prompt_buf, 0, c0, zl, c0, 1, c1 - c0,
                              self.t_out[:c1 - c0] if want else None, self._ctx, wide=family_wide, **mm)

    def _add_request(self, slot: int, prompt: Sequence[int], input_embeds=None,
                     rope_positions=None, seed: Optional[int] = None, start: int = 0,
                     decoded_from: Optional[int] = None) -> int:
        if self._inflight is not None:
            raise RuntimeError("flush() the step in flight before changing the batch")
        self._clean = False
        n = len(prompt)
        if n < 1 or n + 1 > self.cfg.max_seq_len:
            raise ValueError(f"prompt length {n} outside 1..{self.cfg.max_seq_len - 1}")
        if slot != self.active:
            raise ValueError("slots must be filled in order (slot == number of active sequences)")
        self._vacate_active_slot()
        self.t_kv.ensure_capacity(slot, n + self._slack)
        if self.d_kv is not None:
            self.d_kv.ensure_capacity(slot, n + self._slack)
        delta = 0
        if rope_positions is not None:
            if not self.target.cfg.mrope_section:
                raise ValueError("rope_positions need a target with mrope_section")
            if tuple(rope_positions.shape) != (3, n):
                raise ValueError(f"rope_positions must be [3, {n}]")
            delta = int(rope_positions.max().item()) + 1 - n
            rope_positions = rope_positions.to(device=self.device, dtype=torch.int32)
        if input_embeds is not None:
            if tuple(input_embeds.shape) != (n, self.target.cfg.hidden):
                raise ValueError(f"input_embeds must be [{n}, {self.target.cfg.hidden}]")
            input_embeds = input_embeds.to(device=self.device, dtype=torch.bfloat16)
        self.rope_delta[slot] = delta
        # (start > 0: the tokens before it went through prefill_advance already)
        self._prefill_range(slot, prompt, start, n, final=True, rope_positions=rope_positions,
                            input_embeds=input_embeds, decoded_from=decoded_from)
        # first generated token from the last prompt row (a 1-token pass: row 0)
        self._apply_bias(1, 1, slot0=slot)
        if self.sampling:
            self._target_full_rows(1)
            u = torch.tensor([self._seed_slot(slot, seed)], dtype=torch.float32).to(self.device)
            probs = ops.softmax_rows(self.t_logits[:1], self.temps[slot:slot + 1])
            ops.filter_probs(probs, self.top_k[slot:slot + 1], self.top_p[slot:slot + 1],
                             self.min_p[slot:slot + 1])
            first = ops.sample_rows(probs, u)
        else:
            first = self._target_argmax(1, torch.empty(1, dtype=torch.int64, device=self.device))
        first_tok = int(first.item())
        self._check_exchanges()
        self.tok_buf[slot, 0] = int(prompt[-1])
        self.tok_buf[slot, 1] = first_tok
        self.seq_lens[slot] = n + 1
        self.gen_tokens[slot, 0] = first_tok
        self.gen_counts[slot] = 1
        self.host_lens[slot] = n + 1
        if self.ngram:
            self.history[slot] = list(prompt) + [first_tok]
        self.host_gen[slot] = 1
        self.active += 1
        if self.logprobs_top is not None:
            self._logprob_rows([[0]], [[first_tok]])
        return first_tok

    def add_requests(self, prompts: Sequence[Sequence[int]],
                     truth_rows: Optional[Sequence[Sequence[int]]] = None,
                     seeds: Optional[Sequence[Optional[int]]] = None,
                     temperatures: Optional[Sequence[Optional[float]]] = None,
                     filters: Optional[Sequence[Optional[dict]]] = None,
                     biases: Optional[Sequence[Optional[dict]]] = None) -> List[int]:
        """Prefills several text prompts in the same passes and returns their first tokens (one
        synchronisation for all of them).  The prompts take the next free slots in order.
        Every pass carries a chunk of each prompt (prefill_rows // len(prompts) tokens per
        sequence, wide kernel family); rows past the end of a shorter prompt are padding whose
        cache entries are overwritten before anything reads them.  A row's bits do not depend
        on what shares its pass, so the streams equal those of add_request called once per
        prompt."""
        with _rt.stream_ctx(self.stream):
            if truth_rows is not None:
                if self.truth is None:
                    raise ValueError("truth_rows need set_synthetic_acceptance first")
                for i, tr in enumerate(truth_rows):
                    row = torch.tensor(list(tr)[:self.truth.shape[1]], dtype=torch.int32)
                    self.truth[self.active + i, :row.numel()].copy_(row)
            if seeds is not None and len(seeds) != len(prompts):
                raise ValueError("one seed (or None) per prompt")
            if temperatures is not None and len(temperatures) != len(prompts):
                raise ValueError("one temperature (or None) per prompt")
            if filters is not None and len(filters) != len(prompts):
                raise ValueError("one filter dict {top_k, top_p, min_p} (or None) per prompt")
            if biases is not None and len(biases) != len(prompts):
                raise ValueError("one logit mask (or None) per prompt")
            for i in range(len(prompts)):
                self._set_temperature(self.active + i, None if temperatures is None else temperatures[i],
                                      **((filters[i] or {}) if filters is not None else {}))
                self._set_bias(self.active + i, None if biases is None else biases[i])
            return self._add_requests([list(p) for p in prompts], seeds)

    def _add_requests(self, prompts: List[List[int]], seeds=None) -> List[int]:
        if self._inflight is not None:
            raise RuntimeError("flush() the step in flight before changing the batch")
        Bp, slot0 = len(prompts), self.active
        self._clean = False
        if Bp == 0:
            return []
        if slot0 + Bp + len(self._prefilling) > self.cfg.max_batch:
            raise ValueError(f"{Bp} prompts do not fit behind {slot0} active sequences")
        if self._prefilling:      # slots active.. must be free: one at a time moves the others up
            out, lp = [], []
            for i, p in enumerate(prompts):   # (sampling arguments were stored for slot0 + i)
                out.append(self._add_request(self.active, p, seed=None if seeds is None else seeds[i]))
                lp += self.last_logprobs or []
            if self.logprobs_top is not None:
                self.last_logprobs = lp
            return out
        lens = [len(p) for p in prompts]
        for n in lens:
            if n < 1 or n + 1 > self.cfg.max_seq_len:
                raise ValueError(f"prompt length {n} outside 1..{self.cfg.max_seq_len - 1}")
        n_max = max(lens)
        if Bp > min(m.prefill_rows for m in (self.target, self.draft) if m is not None):
            # more prompts than rows in a pass: in groups
            out: List[int] = []
            lp: List[List[tuple]] = []
            per = min(m.prefill_rows for m in (self.target, self.draft) if m is not None)
            for i in range(0, Bp, per):
                out += self._add_requests(prompts[i:i + per], None if seeds is None else seeds[i:i + per])
                lp += self.last_logprobs or []
            if self.logprobs_top is not None:
                self.last_logprobs = lp
            return out
        if self._prompt_rows is None or self._prompt_rows.shape[0] < Bp:
            self._prompt_rows = torch.zeros(self.cfg.max_batch, max(self.cfg.max_seq_len, 8),
                                            dtype=torch.int32, device=self.device)
        stride = self._prompt_rows.shape[1]
        host = torch.zeros(Bp, n_max, dtype=torch.int32)
        for i, p in enumerate(prompts):
            host[i, :lens[i]] = torch.tensor(p, dtype=torch.int32)
        self._prompt_rows[:Bp, :n_max].copy_(host)
        last = torch.tensor([p[-1] for p in prompts], dtype=torch.int32).to(self.device)
        lens_dev = torch.tensor(lens, dtype=torch.int32).to(self.device)
        for i in range(Bp):
            # padding rows write up to n_max - 2 in every sequence's cache
            self.t_kv.ensure_capacity(slot0 + i, n_max + self._slack)
            if self.d_kv is not None:
                self.d_kv.ensure_capacity(slot0 + i, n_max + self._slack)
        self.rope_delta[slot0:slot0 + Bp] = 0
        self._ctx = self._ctx_bucket(n_max + self._slack)
        zl = self.zero_lens[slot0:slot0 + Bp]
        toks = self._prompt_rows.view(-1)
        for model, kv in ((self.target, self.t_kv), (self.draft, self.d_kv)):
            if model is None:
                continue
            view = _KvView(kv, slot0)
            # all but each prompt's last token, in chunks: no logits wanted
            chunk = max(1, model.prefill_rows // Bp)
            wide = model.prefill_rows > MAX_ROWS or self.wide    # the family add_request uses
            for c0 in range(0, n_max - 1, chunk):
                model.forward(view, toks, stride, c0, zl, c0, Bp, min(chunk, n_max - 1 - c0), None,
                              self._ctx, wide=wide)
            # every prompt's last token at its own position
            model.forward(view, last, 1, 0, lens_dev, -1, Bp, 1,
                          self.t_out[:Bp] if model is self.target else None, self._ctx, wide=wide)
        self._apply_bias(Bp, 1, slot0=slot0)
        if self.sampling:
            self._target_full_rows(Bp)
            u = torch.tensor([self._seed_slot(slot0 + i, None if seeds is None else seeds[i])
                              for i in range(Bp)], dtype=torch.float32).to(self.device)
            probs = ops.softmax_rows(self.t_logits[:Bp], self.temps[slot0:slot0 + Bp])
            ops.filter_probs(probs, self.top_k[slot0:slot0 + Bp], self.top_p[slot0:slot0 + Bp],
                             self.min_p[slot0:slot0 + Bp])
            first = ops.sample_rows(probs, u)
        else:
            first = self._target_argmax(Bp, torch.empty(Bp, dtype=torch.int64, device=self.device))
        first32 = first.to(torch.int32)
        sl = slice(slot0, slot0 + Bp)
        self.tok_buf[sl, 0] = last
        self.tok_buf[sl, 1] = first32
        self.seq_lens[sl] = lens_dev + 1
        self.gen_tokens[sl, 0] = first32
        self.gen_counts[sl] = 1
        firsts = [int(t) for t in first.tolist()]
        self._check_exchanges()
        for i, n in enumerate(lens):
            self.host_lens[slot0 + i] = n + 1
            self.host_gen[slot0 + i] = 1
            if self.ngram:
                self.history[slot0 + i] = prompts[i] + [firsts[i]]
        self.active += Bp
        if self.logprobs_top is not None:
            self._logprob_rows([[i] for i in range(Bp)], [[t] for t in firsts])
        return firsts

    # ------------------------------------------------------------------ log-probabilities
    def _logprob_rows(self, rows: List[List[int]], toks: List[List[int]]) -> None:
        """last_logprobs[b][j] for token toks[b][j] chosen from row rows[b][j] of the target's
        logits of the pass just run (synchronises)."""
        n = 1 + max((r for rs in rows for r in rs), default=-1)
        if n == 0:
            self.last_logprobs = [[] for _ in rows]
            return
        raw = self.bias is not None and self.t_raw is not None    # rows as they were before the masks
        if self.vp and (raw or not self.sampling):
            # (greedy passes exchanged slice maxima only; masked passes gathered masked rows)
            self._gather_rows(self.t_raw if raw else self.t_out, self.t_logits, n)
        src = self.t_raw if raw and not self.vp else self.t_logits
        host = torch.full((n,), -1, dtype=torch.int32)
        for rs, ts in zip(rows, toks):
            for r, t in zip(rs, ts):
                host[r] = int(t)
        ids, lps, ranks = ops.token_logprobs(src[:n], host.to(self.device), int(self.logprobs_top))
        ids, lps, ranks = ids.tolist(), lps.tolist(), ranks.tolist()
        self.last_logprobs = [[(ids[r], lps[r], ranks[r]) for r in rs] for rs in rows]

    def _emitted_logprobs(self, out: List[List[int]]) -> None:
        """Log-probabilities of the tokens the step just read back emitted: token j of a sequence
        was chosen from the logits row of the position before it."""
        B, k = self.active, self.k
        if k == 0 or self._spec_disabled():
            rows = [[b] if out[b] else [] for b in range(B)]
        elif self._tree_step:                     # (the step that ran verified trees)
            q = self.tree_nodes + 1
            path = self.tree_path[:B].tolist()    # rows of the accepted nodes; the root is row 0
            rows = [[b * q + (0 if j == 0 else path[b][j - 1]) for j in range(len(out[b]))]
                    for b in range(B)]
        else:
            rows = [[b * (k + 1) + j for j in range(len(out[b]))] for b in range(B)]
        self._logprob_rows(rows, out)

    # ------------------------------------------------------------------ passes
    def _forward(self, model: DecoderModel, kv: PagedKVCache, tokens, tok_stride, tok_offset,
                 seq_lens, pos_offset, B, q_len, logits, slot0: int = 0, tree=None,
                 parts: bool = False) -> None:
        """One model pass over B sequences x q_len tokens, in sub-batches of at
        most max_rows token rows.  tree: (ancestor masks, rotary positions), each with one entry
        per token row, for a pass over token trees.  parts: the pass also leaves the slice maxima
        of its rows' logits in part_val / part_idx (greedy steps: the LM head writes them itself
        where it can, and no logits)."""
        per = max(1, model.max_rows // q_len)
        starts = list(range(0, B, per))
        lanes = len(starts) > 1 and model is self.draft and bool(self._lanes)
        main = torch.cuda.current_stream() if lanes else None
        if lanes:
            fork = torch.cuda.Event()
            fork.record(main)
        used = set()
        for i, b0 in enumerate(starts):
            nb = min(per, B - b0)
            sub_logits = None
            if logits is not None:
                sub_logits = logits[b0 * q_len:(b0 + nb) * q_len]
            view = _KvView(kv, slot0 + b0)
            delta = self.rope_delta[slot0 + b0:] if model.cfg.mrope_section else None
            lane = i % len(self._lane_ws) if lanes else 0
            stream = self._lanes[lane - 1] if lane else main
            if lane and lane not in used:
                stream.wait_event(fork)
                used.add(lane)
            extra = {}
            if tree is not None:
                extra = dict(tree_mask=tree[0].view(-1)[b0 * q_len:], rope_pos=tree[1][b0 * q_len:])
            if parts:
                extra["argmax_parts"] = (self.part_val[b0 * q_len:], self.part_idx[b0 * q_len:])
            with _rt.stream_ctx(stream if lanes else None):
                model.forward(view, tokens.view(-1)[b0 * tok_stride:] if tok_stride else tokens,
                              tok_stride, tok_offset, seq_lens[b0:], pos_offset, nb, q_len, sub_logits,
                              self._ctx, rope_delta=delta,
                              workspace=self._lane_ws[lane] if lanes else None, **extra)
        for lane in sorted(used):          # join: the step goes on when every lane is done
            done = torch.cuda.Event()
            done.record(self._lanes[lane - 1])
            main.wait_event(done)

    def _enqueue_step(self) -> None:
        """One whole step: propose (k draft passes), score (one target pass),
        accept (verification + state update).  Enqueues only."""
        fused = not self.sampling  # greedy: argmax partials feed the glue kernels directly
        if self.k == 0:
            self._enqueue_ar_step()
            return
        if self._spec_disabled():
            self._enqueue_no_spec_step()
            return
        if self.tree:
            self._enqueue_tree_step()
            return
        if self.host_props:
            self._enqueue_lookup_proposals()
        else:
            self._enqueue_propose(fused)
        self._enqueue_score(fused)
        self._enqueue_accept(fused)

    def _enqueue_tree_step(self) -> None:
        """Multi-candidate step: one target pass over every sequence's token tree (uploaded by
        _prepare_step), the greedy walk down the tree, the accepted rows' cache entries moved
        to their chain slots, the state update."""
        B, k, q = self.active, self.k, self.tree_nodes + 1
        ops.tree_prepare(self.tok_buf[:B], 1, self.tree_tok[:B], self.tree_depth[:B],
                         self.seq_lens[:B], -1, self.tree_rope)
        if self.draft_tree:
            self._enqueue_grow_tree()
        self._forward(self.target, self.t_kv, self.tree_tok, q, 0, self.seq_lens, -1, B, q,
                      self.t_out, tree=(self.tree_mask, self.tree_rope))
        self._apply_bias(B * q, q)
        if self.sampling:
            # the target's distribution after every tree row (per-sequence temperature and
            # filters), then the walk with point-mass proposals: the candidates have no draft
            # distribution (sd_tree_accept_sample)
            self._target_full_rows(B * q)
            probs = self.t_probs.view(-1, self.V)[:B * q]
            ops.softmax_rows(self.t_logits[:B * q], self.temps[:B], rows_per_seq=q, out=probs,
                             scratch=self.samp_scratch)
            ops.filter_probs(probs, self.top_k, self.top_p, self.min_p, rows_per_seq=q)
            ops.tree_accept_sample(probs, self.tree_tok[:B], self.tree_parent[:B], self.tree_len[:B], k,
                                   self.u_accept[:B], self.u_bonus[:B], self.out_ids[:B],
                                   self.tree_path[:B], self.tree_npath[:B], self.counters)
        else:
            self._target_argmax(B * q, self.tree_ids[:B * q])
            ops.tree_accept_greedy(self.tree_ids, self.tree_tok[:B], self.tree_parent[:B],
                                   self.tree_len[:B], k, self.out_ids[:B], self.tree_path[:B],
                                   self.tree_npath[:B], self.counters)
        caches = [(self.t_kv, self.target.cfg)]
        if self.draft_tree:               # the draft's cache holds the same rows at the same slots
            caches.append((self.d_kv, self.draft.cfg))
        for kv, c in caches:
            ops.kv_compact(kv.k, kv.v, kv.layer_stride_bytes, c.n_layers, kv.block_tables[:B],
                           self.seq_lens[:B], -1, self.tree_path[:B], self.tree_npath[:B],
                           c.n_kv_heads, c.head_dim)
        ops.advance_state(self.out_ids[:B], self.seq_lens[:B], self.tok_buf[:B],
                          self.gen_tokens[:B], self.gen_counts[:B], k, self._records[self._parity, :B])

    def _enqueue_grow_tree(self) -> None:
        """The draft model fills in the tree's tokens (the topology is fixed: draft_tree branches of
        depth k).  Pass 0 feeds the last two tokens as every proposal pass does; the b most likely
        next tokens are the nodes of depth 1.  The pass of level d then covers tree rows 1..d*b
        (cached at slot root + row, rotated at root + depth, each row attending its own branch)
        and the arg-max after each of its last b rows is that branch's node of depth d + 1.
        Rows of shallower levels are computed again, with the same bits: a pass over the new
        rows alone would see the other branches' cache entries as context."""
        B, k, nb, q = self.active, self.k, self.draft_tree, self.tree_nodes + 1
        self._forward(self.draft, self.d_kv, self.tok_buf, k + 2, 0, self.seq_lens, -2, B, 2, self.d_out)
        ops.topk_tokens(self.d_out[:2 * B], nb, self.tree_tok, 1, row_offset=1, row_step=2, B=B)
        rope = self.tree_rope.view(-1, q)
        for d in range(1, k):
            qd = d * nb
            d_rope = self.dtree_rope[d]
            d_rope[:B].copy_(rope[:B, 1:1 + qd])
            self._forward(self.draft, self.d_kv, self.tree_tok, q, 1, self.seq_lens, 0, B, qd,
                          self.d_out, tree=(self.dtree_mask[d], d_rope.view(-1)))
            ops.topk_tokens(self.d_out[:B * qd], 1, self.tree_tok, 1 + qd, row_offset=qd - nb, row_step=qd,
                            rows=nb, B=B)

    # -- the target's logits after a pass of n rows (whole, or one vocabulary slice per rank)
    def _target_head_fused(self) -> bool:
        """Whether a greedy target pass may leave the slice maxima itself (no logits): nothing
        downstream needs the logits (no logit masks, no log-probabilities, whole vocabulary on
        this rank)."""
        return not self.sampling and not self.vp and self.bias is None and self.logprobs_top is None

    def _target_partials(self, n: int) -> None:
        """Slice maxima of rows 0..n-1 over the whole vocabulary -> part_val / part_idx."""
        if self._target_head_fused():
            return                       # the pass left them there (_forward(parts=True))
        self._partials(self.t_out, n, self.vp)

    def _partials(self, out: torch.Tensor, n: int, split: bool) -> None:
        if not split:
            ops.argmax_partial(out[:n], self.part_val, self.part_idx)
            return
        m = n * ops.ARGMAX_PARTS
        send = self.vp_send[:2 * m]
        ops.argmax_partial(out[:n], send[:m].view(n, -1), send[m:].view(torch.int32).view(n, -1))
        recv = self.vp_recv[:self.tp.world_size * 2 * m]
        self.tp.all_gather(send, recv)
        ops.argmax_combine(recv, self.tp.world_size, n, self.target.cfg.lm_vocab,
                           self.part_val, self.part_idx)

    def _target_argmax(self, n: int, out: torch.Tensor) -> torch.Tensor:
        if not self.vp:
            return ops.argmax_rows(self.t_out[:n], out=out)
        self._target_partials(n)
        return ops.argmax_from_partials(self.part_val, self.part_idx, n, out=out)

    def _target_full_rows(self, n: int) -> None:
        """Whole logits rows 0..n-1 in t_logits (sampling needs the full distribution)."""
        if self.vp:
            self._gather_rows(self.t_out, self.t_logits, n)

    def _gather_rows(self, out: torch.Tensor, full: torch.Tensor, n: int) -> None:
        n_tp, Vl = self.tp.world_size, self.target.cfg.lm_vocab
        recv = self.vp_rows[:n_tp * n * Vl]
        self.tp.all_gather(out[:n].reshape(-1), recv)
        full[:n].view(n, n_tp, Vl).copy_(recv.view(n_tp, n, Vl).permute(1, 0, 2))

    def _enqueue_ar_step(self) -> None:
        B = self.active
        self._forward(self.target, self.t_kv, self.tok_buf, 2, 1, self.seq_lens, -1, B, 1,
                      self.t_out, parts=self._target_head_fused())
        self._apply_bias(B, 1)
        if self.sampling:
            self._target_full_rows(B)
            ops.softmax_rows(self.t_logits[:B], self.temps[:B], out=self.t_probs.view(-1, self.V),
                             scratch=self.samp_scratch)
            ops.filter_probs(self.t_probs.view(-1, self.V)[:B], self.top_k, self.top_p, self.min_p)
            ops.sample_rows(self.t_probs.view(-1, self.V)[:B], self.u_bonus, out=self.ids_tmp,
                            scratch=self.samp_scratch)
            ops.advance_state(self.ids_tmp[:B].view(-1, 1), self.seq_lens[:B], self.tok_buf[:B],
                              self.gen_tokens[:B], self.gen_counts[:B], 0, self._records[self._parity, :B])
        else:
            self._target_partials(B)
            ops.verify_from_partials(self.part_val, self.part_idx, self.tok_buf[:B], None, None,
                                     self.seq_lens[:B], self.gen_tokens[:B], self.gen_counts[:B],
                                     0, self._records[self._parity, :B])

    def _spec_disabled(self) -> bool:
        """Whether the current step (the last one _prepare_step prepared) leaves speculation out."""
        if self._auto:
            return self._no_spec_now
        t = self.cfg.disable_by_batch_size
        return self.k > 0 and t > 0 and self.active > t

    # -- speculate or not, step by step (spec_policy="auto") ---------------------------------
    def _decide_no_spec(self) -> bool:
        """True if the coming step should not speculate.  Per batch size the engine keeps running
        means of the step time in both modes and of the tokens a speculative step emits per
        sequence; the mode with the higher tokens per second runs, the other one is probed
        now and then."""
        t = self.cfg.disable_by_batch_size
        if t > 0 and self.active > t:
            return True
        st = self._pol.setdefault(self.active, {"t": {}, "n": {"spec": 0, "ar": 0}, "emit": None,
                                                "mode": "spec", "run": 0, "probe": 0})
        if st["probe"] > 0:                     # a probe of the other mode is under way
            st["probe"] -= 1
            return st["mode"] == "spec"
        if st["n"]["spec"] < 3:
            return False
        if st["n"]["ar"] < 3:
            return True
        rate_spec = (st["emit"] or 1.0) / st["t"]["spec"]
        rate_ar = 1.0 / st["t"]["ar"]
        best = "spec" if rate_spec >= rate_ar else "ar"
        if best != st["mode"] and max(rate_spec, rate_ar) > 1.03 * min(rate_spec, rate_ar):
            st["mode"], st["run"] = best, 0
        st["run"] += 1
        if st["run"] >= self.cfg.policy_probe_every:
            st["run"] = 0
            st["probe"] = max(self.cfg.policy_probe_steps - 1, 0)
            return st["mode"] == "spec"
        return st["mode"] == "ar"

    def _note_step_time(self, B: int, mode: str, seconds: float) -> None:
        st = self._pol.get(B)
        if st is None:
            return
        old = st["t"].get(mode)
        st["t"][mode] = seconds if old is None else 0.7 * old + 0.3 * seconds
        st["n"][mode] += 1

    def _note_emitted(self, B: int, out: List[List[int]], spec: bool) -> None:
        st = self._pol.get(B)
        if st is None or not spec or not out:
            return
        e = sum(len(o) for o in out) / len(out)
        st["emit"] = e if st["emit"] is None else 0.8 * st["emit"] + 0.2 * e

    def policy_state(self) -> dict:
        """The policy's running means per batch size (seconds per step in both modes, tokens a
        speculative step emits per sequence, the mode that currently runs)."""
        return {B: {"t_spec": st["t"].get("spec"), "t_ar": st["t"].get("ar"), "emitted": st["emit"],
                    "mode": st["mode"]} for B, st in self._pol.items()}

    def _draft_catch_up(self) -> None:
        """Brings the draft's cache up to date after steps that left the draft out: one pass over
        the last _lag_max positions before the two that every proposal pass re-feeds anyway
        (never before a sequence's first generated token: the prompt is always complete).
        Positions that were valid are written again with the same tokens; rows of a young
        sequence that reach past its end write slots that are rewritten before anything reads
        them.  Token ids come from the generated-token log on the device; positions from the
        device's lengths, so a step still in flight does not matter."""
        B, L = self.active, self._lag_max
        if B == 0 or self.draft is None:
            self._lag = 0
            return
        model = self.draft
        plen = self.seq_lens[:B] - self.gen_counts[:B]
        base = torch.maximum(self.seq_lens[:B] - (L + 2), plen).contiguous()
        gi = (base - plen).unsqueeze(1) + self._arange[:L].unsqueeze(0)
        toks = self.gen_tokens[:B].gather(1, gi.clamp_(0, self.gen_tokens.shape[1] - 1).long())
        toks = toks.to(torch.int32).contiguous()
        wide = model.prefill_rows > MAX_ROWS or model.wide
        rows_cap = model.prefill_rows if wide else model.max_rows
        q_cap = max(1, 64 // (model.cfg.n_heads // model.cfg.n_kv_heads))
        for b0 in range(0, B, min(B, rows_cap)):
            nb = min(min(B, rows_cap), B - b0)
            q_chunk = max(1, min(rows_cap // nb, q_cap, L))
            delta = self.rope_delta[b0:] if model.cfg.mrope_section else None
            for c in range(0, L, q_chunk):
                model.forward(_KvView(self.d_kv, b0), toks.view(-1)[b0 * L:], L, c, base[b0:], c, nb,
                              min(q_chunk, L - c), None, self._ctx, wide=wide, rope_delta=delta)
        self._lag = 0

    def _enqueue_no_spec_step(self) -> None:
        """A step of a speculative engine that does not speculate (batch above
        disable_by_batch_size): the target scores the last token only; the draft re-feeds the
        last two tokens without its LM head, so its KV cache stays complete."""
        B, k = self.active, self.k
        if self.draft is not None and not self._lazy_draft:   # (lazy: _draft_catch_up, later)
            self._forward(self.draft, self.d_kv, self.tok_buf, k + 2, 0, self.seq_lens, -2, B, 2, None)
        self._forward(self.target, self.t_kv, self.tok_buf, k + 2, 1, self.seq_lens, -1, B, 1,
                      self.t_out)
        self._apply_bias(B, 1)
        if self.sampling:
            self._target_full_rows(B)
            ops.softmax_rows(self.t_logits[:B], self.temps[:B],
                             out=self.t_probs.view(-1, self.V)[:B], scratch=self.samp_scratch)
            ops.filter_probs(self.t_probs.view(-1, self.V)[:B], self.top_k, self.top_p, self.min_p)
            ops.sample_rows(self.t_probs.view(-1, self.V)[:B], self.u_bonus, out=self.ids_tmp,
                            scratch=self.samp_scratch)
        else:
            self._target_argmax(B, self.ids_tmp[:B])
        self.out_ids[:B].fill_(-1)
        self.out_ids[:B, 0].copy_(self.ids_tmp[:B])
        ops.advance_state(self.out_ids[:B], self.seq_lens[:B], self.tok_buf[:B],
                          self.gen_tokens[:B], self.gen_counts[:B], k, self._records[self._parity, :B])

    def _enqueue_propose(self, fused: bool = False) -> None:
        """k draft passes; proposals land in tok_buf[:, 2:], their probability
        rows (sampling mode) in d_probs.  fused (greedy only): two glue kernels per
        pass instead of three, and no draft_ids gather."""
        B, k = self.active, self.k
        stride = k + 2
        head_fused = fused and not self.d_vp     # the draft's head leaves the slice maxima itself
        for i in range(k):
            if i == 0:
                self._forward(self.draft, self.d_kv, self.tok_buf, stride, 0, self.seq_lens, -2, B, 2,
                              self.d_out, parts=head_fused)
                off, step = 1, 2
            else:
                self._forward(self.draft, self.d_kv, self.tok_buf, stride, 1 + i, self.seq_lens,
                              -1 + i, B, 1, self.d_out, parts=head_fused)
                off, step = 0, 1
            if self.d_vp and (self.sampling or not fused):
                self._gather_rows(self.d_out, self.d_logits, B * step)   # whole rows on every rank
            rows = (self.d_out if self.d_map is not None else self.d_logits)[:B * step]
            if fused:
                if not head_fused:
                    self._partials(self.d_out, B * step, self.d_vp)
                ops.propose_from_partials(self.part_val, self.part_idx, off, step,
                                          self.tok_buf[:B], k, i)
                if self.d_map is not None:
                    ops.map_tokens(self.tok_buf[:B], 2 + i, self.d_map)
                continue
            if self.sampling:
                ops.softmax_rows(rows, self.temps[:B], row_offset=off, row_step=step,
                                 out_ptr=self.d_probs.data_ptr() + i * self.V * 4, out_ld=k * self.V,
                                 scratch=self.samp_scratch)
                ops.sample_rows(self.d_probs.view(-1, self.V)[:B * k], self.u_draft[i],
                                out=self.ids_tmp, row_offset=i, row_step=k, scratch=self.samp_scratch)
            else:
                ops.argmax_rows(rows, out=self.ids_tmp, row_offset=off, row_step=step)
            ops.store_proposal(self.ids_tmp[:B], self.tok_buf[:B], k, i)
            if self.d_map is not None:
                ops.map_tokens(self.tok_buf[:B], 2 + i, self.d_map)
        if self.truth is not None:    # benchmark control: impose the acceptance schedule
            ops.debug_force_proposals(self.tok_buf[:B], k, self.seq_lens[:B], self.truth,
                                      self.accept_pattern, self.force_ctr[:B], self.V)
        if not fused:
            ops.gather_proposals(self.tok_buf[:B], self.draft_ids[:B], k)

    def _enqueue_lookup_proposals(self) -> None:
        """Prompt-lookup proposals (uploaded by _prepare_step) into the token window; in
        sampling mode their one-hot 'draft distribution' into d_probs."""
        B, k = self.active, self.k
        ops.set_proposals(self.ext_props[:B], self.tok_buf[:B], k)
        ops.gather_proposals(self.tok_buf[:B], self.draft_ids[:B], k)
        if self.sampling:
            self.d_probs.zero_()
            self.d_probs.scatter_(2, self.draft_ids.unsqueeze(-1), 1.0)

    def _lookup(self, hist: List[int]) -> List[int]:
        """The k tokens that followed the most recent earlier occurrence of the sequence's
        last n tokens, longest n first (ngram_max .. ngram_min); [] if nothing matches."""
        import numpy as np
        k, arr = self.k, np.asarray(hist, dtype=np.int64)
        n_hist = arr.size
        if n_hist < 2:
            return []
        ends = np.nonzero(arr[:-1] == arr[-1])[0][::-1]   # candidate match ends, latest first
        for n in range(min(self.cfg.ngram_max, n_hist - 1), self.cfg.ngram_min - 1, -1):
            pat = arr[n_hist - n:]
            for e in ends:
                if e + 1 >= n and np.array_equal(arr[e + 1 - n:e + 1], pat):
                    return arr[e + 1:e + 1 + k].tolist()
        return []

    def _lookup_candidates(self, hist: List[int]) -> List[List[int]]:
        """Up to tree_candidates distinct continuations: the tokens after earlier occurrences of
        the sequence's last n tokens, longest n first, most recent match first (the first
        candidate is _lookup's chain)."""
        import numpy as np
        k, arr = self.k, np.asarray(hist, dtype=np.int64)
        n_hist, out = arr.size, []
        if n_hist < 2:
            return out
        ends = np.nonzero(arr[:-1] == arr[-1])[0][::-1]
        for n in range(min(self.cfg.ngram_max, n_hist - 1), self.cfg.ngram_min - 1, -1):
            pat = arr[n_hist - n:]
            for e in ends:
                if e + 1 >= n and np.array_equal(arr[e + 1 - n:e + 1], pat):
                    cont = arr[e + 1:e + 1 + k].tolist()
                    if cont and cont not in out:
                        out.append(cont)
                        if len(out) >= self.cfg.tree_candidates:
                            return out
        return out

    def _upload_trees(self, candidates: List[List[List[int]]]) -> None:
        """Merges every sequence's candidate chains into a token tree (rows in insertion order:
        a node's parent always has a smaller row) and uploads tokens, parents, depths, lengths and
        ancestor masks in one copy.  Unused rows are leaves under the root that the walk never
        looks at."""
        N, k = self.tree_nodes, self.k
        h_tok, h_par, h_dep, h_len, h_mask = self._tree_np
        for b, cands in enumerate(candidates):
            toks, parents, depths, masks = [0], [0], [0], [1]
            child = {}
            for chain in cands:
                cur = 0
                for d, tok in enumerate(chain[:k]):
                    row = child.get((cur, tok))
                    if row is None:
                        if len(toks) > N:
                            break
                        row = len(toks)
                        child[(cur, tok)] = row
                        toks.append(int(tok))
                        parents.append(cur)
                        depths.append(d + 1)
                        masks.append(masks[cur] | (1 << row))
                    cur = row
            h_len[b] = len(toks)
            for t in range(len(toks), N + 1):
                toks.append(0)
                parents.append(0)
                depths.append(1)
                masks.append(1 | (1 << t))
            h_tok[b], h_par[b], h_dep[b] = toks, parents, depths
            h_mask[b] = [m - (1 << 64) if m >= (1 << 63) else m for m in masks]
        # (column 0 of tree_tok, the root token, is written afterwards on the device by
        # sd_tree_prepare)
        self._tree_dev.copy_(self._tree_host, non_blocking=True)

    def _enqueue_score(self, fused: bool = False) -> None:
        """One target pass over {last token, k proposals} of every sequence:
        t_logits [B*(k+1), V], plus target_ids (greedy) or t_probs / bonus_ids."""
        B, k = self.active, self.k
        self._forward(self.target, self.t_kv, self.tok_buf, k + 2, 1, self.seq_lens, -1, B, k + 1,
                      self.t_out, parts=fused and self._target_head_fused())
        self._apply_bias(B * (k + 1), k + 1)
        rows = self.t_logits[:B * (k + 1)]
        if fused:
            self._target_partials(B * (k + 1))
        elif self.sampling:
            self._target_full_rows(B * (k + 1))
            ops.softmax_rows(rows, self.temps[:B], rows_per_seq=k + 1, out=self.t_probs.view(-1, self.V),
                             scratch=self.samp_scratch)
            ops.filter_probs(self.t_probs.view(-1, self.V)[:B * (k + 1)], self.top_k, self.top_p,
                             self.min_p, rows_per_seq=k + 1)
            ops.sample_rows(self.t_probs.view(-1, self.V)[:B * (k + 1)], self.u_bonus,
                            out=self.bonus_ids, row_offset=k, row_step=k + 1,
                            scratch=self.samp_scratch)
        else:
            self._target_argmax(B * (k + 1), self.target_ids.view(-1)[:B * (k + 1)])

    def _enqueue_accept(self, fused: bool = False) -> None:
        """Acceptance (rejection sampling or greedy match) and the state update."""
        B, k = self.active, self.k
        lens = self.prop_lens[:B] if self.host_props else None
        if fused:
            ops.verify_from_partials(self.part_val, self.part_idx, self.tok_buf[:B],
                                     self.out_ids[:B], self.counters, self.seq_lens[:B],
                                     self.gen_tokens[:B], self.gen_counts[:B], k,
                                     self._records[self._parity, :B], proposal_lens=lens)
            return
        if self.sampling and self.typical:
            ops.typical_acceptance(self.t_probs[:B], self.bonus_ids[:B], self.draft_ids[:B],
                                   self.cfg.posterior_threshold, self.cfg.posterior_alpha, lens,
                                   self.counters, out=self.out_ids[:B])
        elif self.sampling:
            ops.rejection_sample(self.t_probs[:B], self.bonus_ids[:B], self.d_probs[:B],
                                 self.draft_ids[:B], self.u_accept[:B], self.u_recover[:B], lens,
                                 self.counters, scratch=self.samp_scratch, out=self.out_ids[:B])
        else:
            ops.greedy_verify(self.target_ids[:B], self.draft_ids[:B], lens, self.counters,
                              out=self.out_ids)
        ops.advance_state(self.out_ids[:B], self.seq_lens[:B], self.tok_buf[:B],
                          self.gen_tokens[:B], self.gen_counts[:B], k, self._records[self._parity, :B])

    def remove_request(self, slot: int) -> None:
        """Frees batch slot `slot`; the last active sequence moves into it so
        that active sequences stay contiguous (synchronises)."""
        if self._inflight is not None:
            raise RuntimeError("flush() the step in flight before changing the batch")
        self._clean = False
        with _rt.stream_ctx(self.stream):
            last = self.active - 1
            if not 0 <= slot <= last:
                raise ValueError(f"slot {slot} is not active")
            for kv in (self.t_kv, self.d_kv):
                if kv is None:
                    continue
                kv.release(slot)
                if slot != last:
                    kv.move_slot(last, slot)
            if slot != last:
                moved = [self.seq_lens, self.tok_buf, self.gen_tokens, self.gen_counts, self.rope_delta,
                         self.force_ctr, self.bias_on]
                if self._bias_host[last]:
                    moved.append(self.bias)
                if self.sampling:
                    moved += [self.u_seeds, self.u_counters, self.temps, self.top_k, self.top_p, self.min_p]
                    self.host_seeds[slot] = self.host_seeds[last]
                if self.truth is not None:
                    moved.append(self.truth)
                for t in moved:
                    t[slot].copy_(t[last])
                self.host_lens[slot], self.host_gen[slot] = self.host_lens[last], self.host_gen[last]
                self._bias_host[slot] = self._bias_host[last]
                if self.ngram:
                    self.history[slot] = self.history[last]
            self.host_lens[last] = self.host_gen[last] = 0
            if self._bias_host[last]:
                self.bias_on[last] = 0
                self._bias_host[last] = False
            self.active = last
            _rt.synchronize(self.device)

    # ------------------------------------------------------------------ stepping
    def _refresh_uniforms(self) -> None:
        """Fresh uniforms for the coming step, drawn on the device (no host work, no sync)."""
        ops.fill_uniforms_seq(self.u_all, self.u_regions, self.u_seeds, self.u_counters, self.active)

    def set_external_proposals(self, proposals: Optional[Sequence[Sequence[int]]]) -> None:
        """Proposals of the coming step, one token list (at most k are used) per active
        sequence, in slot order; None or an empty list proposes nothing for a sequence."""
        if not self.external:
            raise RuntimeError("this engine makes its own proposals (EngineConfig.external_proposals "
                               "is for engines without a draft model)")
        if proposals is not None:
            if len(proposals) != self.active:
                raise ValueError(f"{len(proposals)} proposal lists for {self.active} sequences")
            if self.tree:   # per sequence: several candidate chains (or one flat chain)
                proposals = [[] if not p else
                             ([[int(t) for t in c] for c in p] if isinstance(p[0], (list, tuple))
                              else [[int(t) for t in p]]) for p in proposals]
                flat = [t for p in proposals for c in p for t in c]
            else:
                proposals = [[] if p is None else [int(t) for t in p] for p in proposals]
                flat = [t for p in proposals for t in p]
            if any(t < 0 or t >= self.V for t in flat):
                raise ValueError("proposed token outside the vocabulary")
        self._pending_props = proposals

    def step(self, sync: bool = True,
             proposals: Optional[Sequence[Sequence[int]]] = None) -> Optional[List[List[int]]]:
        """Runs one step for every active sequence.  With sync=True returns the
        tokens each sequence emitted in this step (the V1 worker contract:
        sampled_token_ids, one variable-length list per request).  proposals: see
        set_external_proposals."""
        if self._inflight is not None:
            raise RuntimeError("flush() the step in flight before a synchronous step")
        if proposals is not None:
            self.set_external_proposals(proposals)
        with _rt.stream_ctx(self.stream):
            return self._step(sync)

    def prepare_step(self) -> None:
        """Host work that must precede a step's kernels: cache pages for the
        tokens the step may write, fresh uniforms (sampling)."""
        with _rt.stream_ctx(self.stream):
            self._prepare_step()

    def _prepare_step(self) -> None:
        B, k = self.active, self.k
        for b in range(B):
            need = self.host_lens[b] + self._slack
            if need > self.max_ctx:
                raise RuntimeError(f"sequence {b} would exceed max_seq_len")
            # pages are taken a few at a time, so most steps upload no block-table entries
            ahead = min(need + 4 * 16, self.max_ctx)
            self.t_kv.ensure_capacity(b, need, ahead)
            if self.d_kv is not None:
                self.d_kv.ensure_capacity(b, need, ahead)
        # Attention launches one workgroup per `split_tokens` keys up to the context bound it
        # is given; a bound that follows the live lengths in powers of two keeps short
        # contexts on the single-split path (no partials, no merge kernel).  The split
        # boundaries stay at fixed absolute positions, so results do not depend on the bound.
        self._ctx = self._ctx_bucket()
        if self.tree and not self.draft_tree:   # (a draft-grown tree is filled in inside the step)
            pending, self._pending_props = self._pending_props, None
            cands = []
            for b in range(B):
                if self.ngram:
                    cands.append(self._lookup_candidates(self.history[b]))
                else:
                    cands.append(pending[b] if pending is not None else [])
            cands += [[] for _ in range(self.cfg.max_batch - B)]
            self._upload_trees(cands)
        elif self.host_props:
            pending, self._pending_props = self._pending_props, None
            for b in range(B):
                if self.ngram:
                    cont = self._lookup(self.history[b])
                else:
                    cont = list(pending[b])[:k] if pending is not None else []
                self._lens_host[b] = len(cont)
                for j in range(k):
                    self._props_host[b, j] = cont[j] if j < len(cont) else 0
            self.ext_props.copy_(self._props_host, non_blocking=True)
            self.prop_lens.copy_(self._lens_host, non_blocking=True)
        if self.sampling:
            self._refresh_uniforms()
        if self._auto:
            self._no_spec_now = self._decide_no_spec()
            if self._lag and (not self._no_spec_now or self._lag >= self._lag_max):
                self._draft_catch_up()
            if self._no_spec_now:
                self._lag += 1

    def finish_step(self) -> List[List[int]]:
        """After a step's kernels were enqueued piecewise: reads the emitted
        tokens back (synchronises)."""
        with _rt.stream_ctx(self.stream):
            for b in range(self.active):
                self.host_lens[b] += self.k + 1
            self._steps_pending += 1
            self._tree_step = False      # (enqueued piecewise: propose, score, accept of chains)
            return self._read_back()

    def _step(self, sync: bool) -> Optional[List[List[int]]]:
        B, k = self.active, self.k
        if B == 0:
            return [] if sync else None
        self._tree_step = self.tree      # (_enqueue_step verifies trees on a tree engine)
        self._prepare_step()
        if self._auto:
            import time
            now = time.perf_counter()
            if self._last_start is not None and self._clean:
                self._note_step_time(self._last_start[1], self._last_start[2], now - self._last_start[0])
            self._last_start = (now, B, "ar" if self._no_spec_now else "spec")
            self._clean = True
        stream = ops._stream()
        lib = _lib.load()
        if self.cfg.use_graph and _rt.on_gpu():   # (the CPU build has no stream capture)
            key = (B, k, self.truth is not None, self._ctx, self._spec_disabled(), self._parity,
                   self.bias is not None, self.logprobs_top is not None)
            # (a tree step has one shape too: every sequence's tree is padded to tree_nodes rows)
            g = self._graphs.get(key)
            if g is None:
                self._enqueue_step()          # eager warm-up step (also a real step)
                _rt.synchronize()
                g = C.c_void_p()
                _lib.check(lib.sd_graph_begin(stream), "sd_graph_begin")
                try:
                    self._enqueue_step()
                finally:
                    rc = lib.sd_graph_end(stream, C.byref(g))
                _lib.check(rc, "sd_graph_end")
                self._graphs[key] = g
                self._clean = False           # (a capture is no sample of the step's time)
                # the captured step has not run: the warm-up step is this call's step
            else:
                _lib.check(lib.sd_graph_launch(g, stream), "sd_graph_launch")
        else:
            self._enqueue_step()
        for b in range(B):
            self.host_lens[b] += k + 1        # upper bound until the next read-back
        self._steps_pending += 1
        if not sync:
            return None
        return self._read_back()

    def step_overlapped(self) -> Optional[List[List[int]]]:
        """Enqueues the next step and returns the tokens of the step before it (None on the
        first call): the host's read-back, bookkeeping and launch work run while the GPU is
        busy with the step just enqueued, so the card never idles between steps.  Tokens
        arrive one call late; flush() returns the last step's.  Every step depends on device
        state only; the host's length bounds (pages, context bucket) allow for the step in
        flight.  Not with prompt lookup (its proposals come from the tokens read back)."""
        if self.host_props:
            raise RuntimeError("step_overlapped: host-made proposals need each step's tokens")
        if self.logprobs_top is not None:
            raise RuntimeError("step_overlapped: log-probabilities are read from the step's logits, "
                               "which the next step overwrites")
        if not _rt.on_gpu():
            raise RuntimeError("step_overlapped needs the GPU library (the CPU build is synchronous)")
        with _rt.stream_ctx(self.stream):
            if self.active == 0:
                return []
            prev = self._inflight
            self._parity ^= 1
            self._step(False)
            ev = torch.cuda.Event()
            ev.record()
            self._inflight = (ev, self._parity, self.active, not self._no_spec_now)
            return None if prev is None else self._collect(prev, 1)

    def flush(self) -> Optional[List[List[int]]]:
        """Tokens of the step still in flight after step_overlapped (None if there is none)."""
        with _rt.stream_ctx(self.stream):
            prev, self._inflight = self._inflight, None
            return None if prev is None else self._collect(prev, 0)

    def _check_exchanges(self) -> None:
        """Called after every synchronisation of the step path.  Two in-kernel exchanges bound
        their waits instead of hanging: the one-shot collectives over peer-mapped memory (a late
        or dead peer) and the fused two-layer launches (a workgroup that never published).  A
        wait that gave up leaves unspecified values in the pass, so the tokens read back must
        not be handed out: raise, dropping them."""
        tp = self.tp
        if tp is not None and hasattr(tp, "timeouts"):
            n = tp.timeouts()
            if n != self._tp_timeouts_seen:
                new, self._tp_timeouts_seen = n - self._tp_timeouts_seen, n
                raise RuntimeError(f"tensor-parallel exchange: {new} wait(s) for a peer rank timed out; "
                                   "the step's tokens are dropped (the ranks' partial sums were not combined)")
        for model in (self.target, self.draft):
            st = getattr(model, "chain_state", None)
            if st is not None and ops.chain_errors(st):
                raise RuntimeError("fused layer launches: an exchange wait gave up; the step's tokens "
                                   "are dropped")

    def _collect(self, step, later_steps: int) -> List[List[int]]:
        ev, parity, B = step[:3]
        ev.synchronize()
        self._check_exchanges()
        rec = self._records[parity, :B].tolist()
        self._steps_pending -= 1
        out = []
        for b in range(B):
            n = rec[b][0]
            out.append(rec[b][2:2 + n])
            self.host_gen[b] += n
            # exact after that step, plus the bound of the steps enqueued since
            self.host_lens[b] = rec[b][1] + later_steps * (self.k + 1)
        if self._auto:
            self._note_emitted(B, out, step[3])
        return out

    def _ctx_bucket(self, need: Optional[int] = None) -> int:
        if need is None:
            need = max(self.host_lens[:max(self.active, 1)]) + self._slack
        bound = self.cfg.split_tokens
        while bound < need:
            bound *= 2
        return min(bound, self.max_ctx)

    def _drop_graphs(self) -> None:
        lib = _lib.load()
        for g in self._graphs.values():
            lib.sd_graph_destroy(g)
        self._graphs.clear()

    def close(self) -> None:
        """Destroys the captured step graphs (also on garbage collection)."""
        try:
            self._drop_graphs()
        except Exception:
            pass

    def __del__(self):
        if getattr(self, "_graphs", None):
            self.close()

    def read_back(self) -> List[List[int]]:
        """Synchronises and returns the tokens emitted since the last read-back."""
        with _rt.stream_ctx(self.stream):
            return self._read_back()

    def _read_back(self) -> List[List[int]]:
        B = self.active
        pending, self._steps_pending = self._steps_pending, 0
        if pending == 1:
            # the step's last kernel left its record in pinned host memory
            _rt.sync_current_stream()
            self._check_exchanges()
            rec = self._records[self._parity, :B].tolist()
            out = []
            for b in range(B):
                n = rec[b][0]
                out.append(rec[b][2:2 + n])
                self.host_gen[b] += n
                self.host_lens[b] = rec[b][1]
                if self.ngram:
                    self.history[b].extend(out[-1])
            if self.logprobs_top is not None:
                self._emitted_logprobs(out)
            if self._auto:
                self._note_emitted(B, out, not self._no_spec_now)
            return out
        if self.logprobs_top is not None and pending > 1:
            raise RuntimeError("log-probabilities need a read-back after every step (the logits of "
                               "earlier steps are gone)")
        counts = self.gen_counts[:B].tolist()
        lens = self.seq_lens[:B].tolist()
        self._check_exchanges()
        out: List[List[int]] = []
        for b in range(B):
            lo, hi = self.host_gen[b], counts[b]
            toks = self.gen_tokens[b, lo:min(hi, self.gen_tokens.shape[1])].tolist() if hi > lo else []
            out.append(toks)
            self.host_gen[b] = hi
            self.host_lens[b] = lens[b]
            if self.ngram:
                self.history[b].extend(toks)
        return out

    def metrics(self) -> dict:
        with _rt.stream_ctx(self.stream):
            acc, emit, draft = self.counters.tolist()
        return {"accepted": acc, "emitted": emit, "draft": draft}

    def metrics_begin(self) -> None:
        """Starts a copy of the acceptance counters into pinned host memory on a side stream
        (ordered behind the steps enqueued so far); does not wait for the GPU."""
        if not _rt.on_gpu():
            self._metrics_copy = (None, self.counters.clone())
            return
        if getattr(self, "_metrics_stream", None) is None:
            self._metrics_stream = torch.cuda.Stream(device=self.device)
            self._metrics_host = _rt.pin(torch.zeros(3, dtype=torch.int64))
        seen = torch.cuda.Event()
        seen.record(self.stream)
        with torch.cuda.stream(self._metrics_stream):
            self._metrics_stream.wait_event(seen)
            self._metrics_host.copy_(self.counters, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._metrics_stream)
        self._metrics_copy = (done, self._metrics_host)

    def metrics_poll(self) -> Optional[dict]:
        """The counters of the last metrics_begin once its copy has landed, else None; never
        waits."""
        pending = getattr(self, "_metrics_copy", None)
        if pending is None:
            return None
        done, host = pending
        if done is not None and not done.query():
            return None
        self._metrics_copy = None
        acc, emit, draft = host.tolist()
        return {"accepted": acc, "emitted": emit, "draft": draft}


class _KvView:
    """PagedKVCache seen from batch slot `slot0` on (block-table rows shifted)."""

    def __init__(self, kv: PagedKVCache, slot0: int):
        self.k, self.v = kv.k, kv.v
        self.layer_stride_bytes = kv.layer_stride_bytes
        self.block_tables = kv.block_tables[slot0:]
        self.max_blocks_per_seq = kv.max_blocks_per_seq
