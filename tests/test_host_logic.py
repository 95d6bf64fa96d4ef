"""Host-side logic that needs no GPU: page allocation, configuration, metrics,
the worker's request bookkeeping (with a stand-in engine), and the data-parallel
helpers under a 2-process gloo group."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vllm_omni_draft_amd import config, parallel
from vllm_omni_draft_amd.kv_cache import OutOfPages, PageAllocator
from vllm_omni_draft_amd.metrics import AsyncMetricsCollector, SpecDecodeWorkerMetrics


def test_page_allocator():
    a = PageAllocator(4)
    got = [a.alloc() for _ in range(4)]
    assert sorted(got) == [0, 1, 2, 3] and a.num_free == 0
    with pytest.raises(OutOfPages):
        a.alloc()
    a.free(got[:2])
    assert a.num_free == 2 and a.alloc() in got[:2]


def test_config_presets_and_overrides():
    c = config.preset("llama-3-8b")
    assert (c.n_layers, c.hidden, c.n_heads, c.n_kv_heads, c.head_dim, c.ffn, c.vocab) == \
        (32, 4096, 32, 8, 128, 14336, 128256)
    assert abs(c.param_bytes() / 1e9 - 15.0) < 0.2          # streamed bf16 bytes per pass
    assert c.kv_bytes_per_token() == 128 * 1024              # 128 KiB per token
    d = config.preset("llama-3.2-1b")
    assert d.tie_embeddings and d.rope_scaling["rope_type"] == "llama3"
    assert d.kv_bytes_per_token() == 32 * 1024
    with pytest.raises(ValueError):
        config.create_model_config(None, not_a_field=1)      # unknown keys are rejected
    with pytest.raises(ValueError):
        config.create_model_config(None, head_dim=80)
    hf = {"model_type": "qwen2", "num_hidden_layers": 24, "hidden_size": 896, "num_attention_heads": 14,
          "num_key_value_heads": 2, "intermediate_size": 4864, "vocab_size": 151936,
          "rms_norm_eps": 1e-6, "rope_theta": 1e6, "tie_word_embeddings": True}
    q = config.from_hf_config(hf)
    assert q.qkv_bias and q.head_dim == 64 and q.tie_embeddings


def test_metrics_definitions():
    m = SpecDecodeWorkerMetrics.from_counts(accepted=27, emitted=37, draft=50, k=5)
    assert m.draft_acceptance_rate == 27 / 50
    assert m.accept_len == 2.7 and abs(m.system_efficiency - 37 / 60) < 1e-12

    class Eng:
        k = 5
        def metrics(self):
            return {"accepted": 5, "emitted": 6, "draft": 5}
    now = [0.0]
    col = AsyncMetricsCollector(Eng(), collect_interval_s=5.0, timer=lambda: now[0])
    assert col.maybe_collect() is None
    now[0] = 6.0
    assert col.maybe_collect().accept_len == 5.0
    assert col.maybe_collect() is None and col.maybe_collect(force=True) is not None

    # an engine that copies its counters on a side stream: the interval starts a copy, a later
    # call hands the numbers out once the copy has landed; no call waits
    class AsyncEng(Eng):
        begun, landed = 0, False
        def metrics_begin(self):
            self.begun += 1
        def metrics_poll(self):
            return {"accepted": 10, "emitted": 12, "draft": 10} if self.landed else None
    eng, now[0] = AsyncEng(), 0.0
    col = AsyncMetricsCollector(eng, collect_interval_s=5.0, timer=lambda: now[0])
    assert col.maybe_collect() is None and eng.begun == 0
    now[0] = 6.0
    assert col.maybe_collect() is None and eng.begun == 1       # copy started
    assert col.maybe_collect() is None and eng.begun == 1       # still in flight
    eng.landed = True
    assert col.maybe_collect().accepted_tokens == 10
    assert col.maybe_collect() is None and eng.begun == 1       # next interval not reached
    now[0] = 12.0
    assert col.maybe_collect() is None and eng.begun == 2


class FakeEngine:
    """Emits token 100*slot_owner + step for every running sequence; records calls."""

    class _Cfg:
        max_batch = 3
        use_graph = True

    def __init__(self):
        self.cfg = self._Cfg()
        self.k = 3
        self.active = 0
        self.owners = []
        self.steps = 0

    def add_request(self, slot, prompt, decoded_from=None):
        assert slot == self.active
        self.prompts = getattr(self, "prompts", []) + [list(prompt)]
        self.decoded_from = getattr(self, "decoded_from", []) + [decoded_from]
        self.owners.append(prompt[0])
        self.active += 1
        return prompt[0] * 100

    def step(self):
        self.steps += 1
        return [[o * 100 + self.steps] * (1 + i % 2) for i, o in enumerate(self.owners)]

    def remove_request(self, slot):
        last = self.active - 1
        if slot != last:
            self.owners[slot] = self.owners[last]
        self.owners.pop()
        self.active = last

    def metrics(self):
        return {"accepted": 0, "emitted": 0, "draft": 0}


def test_worker_request_bookkeeping():
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SequenceGroupMetadata,
                                                        SpecDecodeWorker)
    eng = FakeEngine()
    w = SpecDecodeWorker(vllm_config=None, engine=eng)
    P = lambda rid, tok: SequenceGroupMetadata(rid, True, [tok, 1, 2])
    R = lambda rid: SequenceGroupMetadata(rid, False)
    out = w.execute_model(ExecuteModelRequest([P("a", 1), P("b", 2)]))
    assert out.sampled_token_ids == [[100], [200]] and out.req_ids == ["a", "b"]
    # running sequences step; a new prompt joins in the same call
    out = w.execute_model(ExecuteModelRequest([R("b"), R("a"), P("c", 3)]))
    assert out.sampled_token_ids == [[201, 201], [101], [300]]
    assert out.req_id_to_index == {"b": 0, "a": 1, "c": 2}
    # finishing the first slot moves the last sequence into it
    out = w.execute_model(ExecuteModelRequest([R("c"), R("b")], finished_requests_ids=["a"]))
    assert eng.owners == [3, 2] and out.sampled_token_ids == [[302], [202, 202]]
    with pytest.raises(KeyError):
        w.execute_model(ExecuteModelRequest([R("zzz"), R("b"), R("c")]))
    # a running request the step does not name is parked (a V1 scheduler preempts this way): its
    # slot is freed; named again, it is rebuilt from its prompt and the tokens it emitted
    out = w.execute_model(ExecuteModelRequest([R("b")]))
    assert eng.owners == [2] and "c" in w._parked and out.req_ids == ["b"]
    toks_c = list(w._tokens["c"])
    assert toks_c[:3] == [3, 1, 2] and len(toks_c) == 3 + 2       # prompt + [300] + [302]
    out = w.execute_model(ExecuteModelRequest([R("b"), R("c")]))
    assert eng.owners == [2, 3] and eng.prompts[-1] == toks_c and not w._parked
    assert out.sampled_token_ids[1] == [300]                      # the rebuilt sequence's next token
    with pytest.raises(KeyError):                                 # a new prompt under a parked id
        w.execute_model(ExecuteModelRequest([R("b")]))
        w.execute_model(ExecuteModelRequest([R("b"), P("c", 9)]))
    w.execute_model(ExecuteModelRequest([R("b")], finished_requests_ids=["c"]))   # finished while parked
    assert "c" not in w._tokens and not w._parked
    w.execute_model(ExecuteModelRequest([R("b"), P("c", 3), P("d", 4)]))
    with pytest.raises(RuntimeError):
        w.execute_model(ExecuteModelRequest([R("b"), R("c"), R("d"), P("e", 5)]))  # no free slot


def test_shard_requests():
    reqs = list(range(10))
    parts = [parallel.shard_requests(reqs, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == reqs and parts[1] == [1, 5, 9]
    assert all(parallel.owner_of(i, 4) == r for r in range(4) for i in parts[r])
    with pytest.raises(ValueError):
        parallel.shard_requests(reqs, 4, 4)
    assert parallel.aggregate_throughput(10, 2.0) == (10.0, 2.0)


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = parallel.init_from_env("gloo")
    mine = parallel.shard_requests([[i, i + 1] for i in range(5)], r, w)
    streams = [[p[0] * 10 + j for j in range(3)] for p in mine]      # stand-in generation
    tokens = sum(len(s) for s in streams)
    total, secs = parallel.aggregate_throughput(tokens, 1.0 + rank)
    gathered = parallel.gather_streams(streams, w)
    # the tensor-parallel id exchange: rank 0's 128 bytes reach every rank
    uid = parallel._broadcast_bytes(bytes(range(128)) if rank == 0 else None)
    assert uid == bytes(range(128))
    dist.barrier()
    q.put((rank, total, secs, gathered))
    dist.destroy_process_group()


def test_data_parallel_helpers_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, total, secs, gathered in res:
        assert total == 15.0 and secs == 2.0                     # sum of tokens, max of time
        assert gathered == [[[0, 1, 2], [20, 21, 22], [40, 41, 42]], [[10, 11, 12], [30, 31, 32]]]


@pytest.mark.parametrize("arch", ["llama", "opt"])
def test_tensor_parallel_shards_reproduce_the_layer_math(arch):
    """Column-parallel shards concatenate to the whole projection; row-parallel partial
    products sum to it (checked on the oracle's weights, float64)."""
    from oracle.models import OracleConfig, random_state
    from vllm_omni_draft_amd import weights
    oc = OracleConfig(arch=arch, n_layers=2, hidden=128, n_heads=4, n_kv_heads=4 if arch == "opt" else 2,
                      head_dim=64, ffn=256, vocab=64, max_pos=64)
    state = {k: v.double() for k, v in random_state(oc, seed=3, qkv_bias=(arch == "llama")).items()}
    cfg = config.create_model_config(None, arch=arch, n_layers=2, hidden=128, n_heads=4,
                                     n_kv_heads=oc.n_kv_heads, head_dim=64, ffn=256, vocab=64,
                                     max_pos=64, qkv_bias=(arch == "llama"))
    tp = 2
    shards = [weights.shard_state_dict(cfg, state, tp, r) for r in range(tp)]
    for local, _ in shards:
        assert (local.n_heads, local.n_kv_heads, local.ffn, local.hidden) == (2, oc.n_kv_heads // 2, 128, 128)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 128, generator=g, dtype=torch.float64)
    if arch == "llama":
        lp, names = "model.layers.1.", ("self_attn.q_proj", "self_attn.o_proj", "mlp.gate_proj",
                                        "mlp.up_proj", "mlp.down_proj")
    else:
        lp, names = "model.decoder.layers.1.", ("self_attn.q_proj", "self_attn.out_proj", "fc1", "fc1", "fc2")
    qn, on, gn, un, dn = (lp + n + ".weight" for n in names)
    q_full = x @ state[qn].T + state[qn.replace("weight", "bias")]
    q_parts = [x @ st[qn].T + st[qn.replace("weight", "bias")] for _, st in shards]
    torch.testing.assert_close(torch.cat(q_parts, dim=1), q_full)
    # attention output stand-in: the q columns themselves; row-parallel o_proj partial sums
    o_full = q_full @ state[on].T
    o_sum = sum(qp @ st[on].T for qp, (_, st) in zip(q_parts, shards))
    torch.testing.assert_close(o_sum, o_full)
    act = lambda st: torch.nn.functional.silu(x @ st[gn].T) * (x @ st[un].T)
    torch.testing.assert_close(sum(act(st) @ st[dn].T for _, st in shards), act(state) @ state[dn].T)
    # replicated tensors are untouched; K/V shards follow the KV heads
    kname = lp + "self_attn.k_proj.weight"
    assert torch.equal(torch.cat([st[kname] for _, st in shards]), state[kname])
    emb = "model.embed_tokens.weight" if arch == "llama" else "model.decoder.embed_tokens.weight"
    assert all(torch.equal(st[emb], state[emb]) for _, st in shards)
    with pytest.raises(ValueError):
        cfg.tp_shard(3)
    # vocabulary-parallel head: the ranks' slices are the rows of the head (the tied embedding
    # when the checkpoint has no separate head), the embedding table stays whole
    head = state.get("lm_head.weight", state[emb])
    vshards = [weights.shard_state_dict(cfg, state, tp, r, vocab_parallel=True) for r in range(tp)]
    assert all(local.lm_vocab == 32 and local.head_rows == 32 and local.vocab == 64
               for local, _ in vshards)
    assert torch.equal(torch.cat([st["lm_head.weight"] for _, st in vshards]), head)
    assert all(torch.equal(st[emb], state[emb]) for _, st in vshards)
    assert shards[0][0].lm_vocab == 0 and shards[0][0].head_rows == 64
    with pytest.raises(ValueError):
        # 48 rows do not split into 2 slices of whole 16-row tiles
        config.create_model_config(cfg, vocab=48).tp_shard(2, vocab_parallel=True)


def test_prompt_lookup_matches_longest_most_recent_ngram():
    """The host side of the n-gram proposer (no GPU needed: only the lookup is exercised)."""
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    eng = SpecEngine.__new__(SpecEngine)
    eng.k, eng.cfg = 3, EngineConfig(k=3, ngram_max=3, ngram_min=1)
    look = eng._lookup
    #        0  1  2  3  4  5  6  7  8  9
    hist = [5, 6, 7, 8, 1, 2, 6, 7, 9, 4, 6, 7]
    assert look(hist) == [9, 4, 6]                  # "6 7": the later of two earlier matches
    assert look([1, 2, 3, 9, 1, 2, 3]) == [9, 1, 2]  # 3-gram wins over shorter ones
    assert look([1, 2, 3, 4]) == []                 # nothing repeats
    assert look([7]) == []
    assert look([4, 4]) == [4]                      # continuation shorter than k
    eng.cfg = EngineConfig(k=3, ngram_max=3, ngram_min=2)
    assert look([1, 9, 2, 9]) == []                 # only a 1-gram matches: below ngram_min


def test_scheduler_continuous_batching_and_stop_conditions():
    from vllm_omni_draft_amd.scheduler import Request, SpecDecodeScheduler
    from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeWorker
    eng = FakeEngine()                      # 3 slots; request with first prompt token o emits 100*o + step
    sched = SpecDecodeScheduler(SpecDecodeWorker(vllm_config=None, engine=eng), max_num_seqs=3,
                                max_model_len=12)
    reqs = [Request(f"r{i}", [i + 1, 7, 7], max_tokens=m, eos_token_id=e)
            for i, (m, e) in enumerate([(3, None), (50, 203), (2, None), (4, None), (50, None)])]
    for r in reqs:
        sched.add_request(r)
    with pytest.raises(KeyError):
        sched.add_request(Request("r0", [1]))
    with pytest.raises(ValueError):
        sched.add_request(Request("big", list(range(12))))
    finished_order = []
    while sched.has_unfinished():
        finished_order += [r.request_id for r in sched.step()]
        assert eng.active <= 3 and len(sched.running) <= 3
    sched.drain()
    assert eng.active == 0 and not sched.running and not sched.waiting
    out = {r.request_id: r for r in reqs}
    assert out["r0"].output_token_ids[0] == 100 and len(out["r0"].output_token_ids) == 3
    assert out["r0"].finish_reason == "length"
    assert out["r1"].output_token_ids[-1] == 203 and out["r1"].finish_reason == "stop"   # eos, cut there
    assert len(out["r2"].output_token_ids) == 2
    assert len(out["r3"].output_token_ids) == 4 and out["r3"].output_token_ids[0] == 400
    # model length: 3 prompt tokens + 9 generated = 12
    assert len(out["r4"].output_token_ids) == 9 and out["r4"].finish_reason == "length"
    assert set(finished_order) == set(out) and finished_order.index("r2") < finished_order.index("r4")
    # waiting requests entered as slots became free: 5 requests over 3 slots
    assert sched.steps > 3


def test_bench_helpers_and_committed_counter_summary():
    """The bench's algorithmic-bytes model and its look-up of the committed HBM counters (a kernel
    renamed without refreshing the profile would silently report traffic = null)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from vllm_omni_draft_amd import config
    t, d = config.preset("llama-3-8b"), config.preset("llama-3.2-1b")
    total = bench.step_bytes(t, d, 5, 1024)
    # target weights once + 5 draft passes + KV of both + logits: 27-29 GB at context 1024
    assert 27e9 < total < 29.5e9
    assert bench.step_bytes(t, d, 5, 1024, 8e9, 1e9) < total
    lin = bench.pmc_traffic("linear_kernel<2, 1, false, false>")
    attn = bench.pmc_traffic("paged_attn_kernel<128, 1")
    assert lin is not None and attn is not None
    assert 1.0 <= lin / 235102208 < 1.1 and 1.0 <= attn / 537657344 < 1.1
    assert bench.pmc_traffic("no_such_kernel") is None


def test_worker_refuses_a_request_with_another_temperature():
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SamplingParams,
                                                        SequenceGroupMetadata, SpecDecodeWorker)
    eng = FakeEngine()
    eng.cfg.temperature = 0.0
    w = SpecDecodeWorker(vllm_config=None, engine=eng)
    ok = SequenceGroupMetadata("a", True, [1, 2], sampling_params=SamplingParams(temperature=0.0))
    dflt = SequenceGroupMetadata("b", True, [2, 2])
    w.execute_model(ExecuteModelRequest([ok, dflt]))
    hot = SequenceGroupMetadata("c", True, [3, 2], sampling_params=SamplingParams(temperature=0.8))
    with pytest.raises(ValueError):
        w.execute_model(ExecuteModelRequest([SequenceGroupMetadata("a", False),
                                             SequenceGroupMetadata("b", False), hot]))


def test_fast_cache_packer_equals_the_elementwise_one():
    """tests/gpu_util.pack_cache_fast (used for long sequences in the GPU attention tests)
    writes exactly the layout of the element-by-element pack_cache."""
    from tests.gpu_util import pack_cache, pack_cache_fast
    g = torch.Generator().manual_seed(0)
    for Hkv, D, n in ((2, 64, 37), (3, 128, 50)):
        pages = (n + 15) // 16 + 2
        bt = torch.randperm(pages, generator=g)[:(n + 15) // 16].to(torch.int32)
        k = torch.randn(n, Hkv, D, generator=g).to(torch.bfloat16)
        v = torch.randn(n, Hkv, D, generator=g).to(torch.bfloat16)
        kc, vc = pack_cache(k, v, bt, pages)
        kf, vf = torch.zeros_like(kc), torch.zeros_like(vc)
        pack_cache_fast(kf, vf, k, v, bt)
        assert torch.equal(kc, kf) and torch.equal(vc, vf)


# ---- the V1 step contract (fields as transcribed at /root/reference/src/doc/engine_core.md:280-313)
from dataclasses import dataclass as _dc, field as _field
from typing import Any as _Any, Optional as _Opt


@_dc
class _NewRequestData:
    req_id: str
    prompt_token_ids: list
    sampling_params: _Any = None
    mm_inputs: list = _field(default_factory=list)
    block_ids: list = _field(default_factory=list)
    num_computed_tokens: int = 0


@_dc
class _CachedRequestData:
    req_ids: list
    resumed_from_preemption: list = _field(default_factory=list)
    new_token_ids: list = _field(default_factory=list)
    new_block_ids: list = _field(default_factory=list)
    num_computed_tokens: list = _field(default_factory=list)


@_dc
class _SchedulerOutput:
    scheduled_new_reqs: list
    scheduled_cached_reqs: _Any
    num_scheduled_tokens: dict
    total_num_scheduled_tokens: int
    scheduled_spec_decode_tokens: dict = _field(default_factory=dict)
    scheduled_encoder_inputs: dict = _field(default_factory=dict)
    num_common_prefix_blocks: list = _field(default_factory=list)
    finished_req_ids: set = _field(default_factory=set)
    free_encoder_mm_hashes: list = _field(default_factory=list)
    structured_output_request_ids: dict = _field(default_factory=dict)
    grammar_bitmask: _Opt[_Any] = None
    kv_connector_metadata: _Opt[_Any] = None


def _sched(new=(), cached=(), finished=(), spec=None, old_layout=False, resumed=(), computed=None):
    """resumed: cached requests the scheduler recomputes after a preemption
    (resumed_from_preemption); computed: request -> num_computed_tokens (default 0 for a resumed
    request, 7 for any other: a running request has computed tokens)."""
    n_tok = {r.req_id: len(r.prompt_token_ids) for r in new}
    n_tok.update({r: 1 for r in cached})
    flags = [r in resumed for r in cached]
    ncomp = [(computed or {}).get(r, 0 if r in resumed else 7) for r in cached]
    c = ([type("C", (), {"req_id": r, "resumed_from_preemption": f, "num_computed_tokens": n})()
          for r, f, n in zip(cached, flags, ncomp)] if old_layout
         else _CachedRequestData(list(cached), resumed_from_preemption=flags, num_computed_tokens=ncomp))
    return _SchedulerOutput(list(new), c, n_tok, sum(n_tok.values()),
                            scheduled_spec_decode_tokens=spec or {}, finished_req_ids=set(finished))


def test_worker_accepts_the_v1_scheduler_output():
    """execute_model(scheduler_output, delay_capture) as the reference's workers are called
    (/root/reference/src/qwen_image_gen/worker.py:47-51); the result carries every field of
    ModelRunnerOutput (/root/reference/src/doc/engine_core.md:316-340)."""
    from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeWorker
    eng = FakeEngine()
    w = SpecDecodeWorker(vllm_config=None, local_rank=0, rank=0, distributed_init_method="",
                         is_driver_worker=True, engine=eng)
    N = lambda rid, tok: _NewRequestData(rid, [tok, 1, 2], sampling_params={"temperature": None})
    out = w.execute_model(_sched(new=[N("a", 1), N("b", 2)]), delay_capture=False)
    assert out.req_ids == ["a", "b"] and out.sampled_token_ids == [[100], [200]]
    assert out.req_id_to_index == {"a": 0, "b": 1}
    for name in ("logprobs", "prompt_logprobs_dict", "pooler_output", "kv_connector_output",
                 "num_nans_in_logits"):
        assert hasattr(out, name)
    assert out.logprobs is None and out.pooler_output == [None, None]
    assert out.prompt_logprobs_dict == {"a": None, "b": None}
    # running requests first (the scheduler's order), a new prompt joins, both cached layouts
    out = w.execute_model(_sched(new=[N("c", 3)], cached=["b", "a"]))
    assert out.req_ids == ["b", "a", "c"] and out.sampled_token_ids == [[201, 201], [101], [300]]
    out = w.execute_model(_sched(cached=["c", "b"], finished=["a"], old_layout=True))
    assert eng.owners == [3, 2] and out.sampled_token_ids == [[302], [202, 202]]
    # a finished id the worker never saw is ignored, as upstream's runners do
    out = w.execute_model(_sched(cached=["b", "c"], finished=["never-seen"]))
    assert out.req_ids == ["b", "c"]
    with pytest.raises(ValueError, match="chunked prefill"):
        so = _sched(new=[N("d", 4)], cached=["b", "c"])
        so.num_scheduled_tokens["d"] = 2
        w.execute_model(so)
    with pytest.raises(ValueError, match="makes its own"):        # proposals for a worker with a draft
        w.execute_model(_sched(cached=["b", "c"], spec={"b": [1, 2]}))
    with pytest.raises(TypeError):
        w.execute_model(object())
    # what the worker does not implement is refused before anything runs, not served wrongly
    steps = eng.steps
    so = _sched(cached=["b", "c"])
    so.grammar_bitmask = [[0]]
    with pytest.raises(ValueError, match="structured output"):
        w.execute_model(so)
    so = _sched(cached=["b", "c"])
    so.kv_connector_metadata = object()
    with pytest.raises(ValueError, match="connector"):
        w.execute_model(so)
    for bad in ({"prompt_logprobs": 3}, {"repetition_penalty": 1.2}, {"bad_words": ["no"]},
                {"bad_words_token_ids": [[5, 6]]}, {"presence_penalty": 0.5}):
        with pytest.raises(ValueError, match="not supported"):
            w.execute_model(_sched(new=[_NewRequestData("z", [1, 2], sampling_params=bad)], cached=["b", "c"]))
    for mask in ({"logit_bias": {5: 1.0}}, {"allowed_token_ids": [1, 2]}, {"bad_words_token_ids": [[7]]}):
        with pytest.raises(ValueError, match="applies no logit masks"):     # (this stub engine)
            w.execute_model(_sched(new=[_NewRequestData("z", [1, 2], sampling_params=mask)], cached=["b", "c"]))
    # (log-probabilities of emitted tokens need an engine that computes them; this stub does not)
    for n, msg in ((3, "computes no log-probabilities"), (99, "outside"), (-1, "outside")):
        with pytest.raises(ValueError, match=msg):
            w.execute_model(_sched(new=[_NewRequestData("z", [1, 2], sampling_params={"logprobs": n})],
                                   cached=["b", "c"]))
    assert eng.steps == steps
    # neutral values of the same fields are fine, and the filters travel to the engine's request
    ok = {"logprobs": None, "repetition_penalty": 1.0, "top_k": 40, "top_p": 0.9, "min_p": 0.0}
    from vllm_omni_draft_amd.spec_decode_worker import request_from_scheduler_output
    req = request_from_scheduler_output(_sched(new=[_NewRequestData("z", [1, 2], sampling_params=ok)]))
    sp = req.seq_group_metadata_list[0].sampling_params
    assert (sp.top_k, sp.top_p, sp.min_p) == (40, 0.9, 0.0)


class _PieceEngine(FakeEngine):
    """FakeEngine with piecewise prefill: the last piece admits the sequence."""

    def prefill_begin(self, prompt, decoded_from=None):
        self._pending = getattr(self, "_pending", {})
        h = len(self._pending) + 1
        self._pending[h] = [list(prompt), 0, decoded_from]
        return h

    def prefill_advance(self, h, n):
        p = self._pending[h]
        p[1] += n
        assert p[1] <= len(p[0])
        if p[1] < len(p[0]):
            return None
        del self._pending[h]
        return self.add_request(self.active, p[0], decoded_from=p[2])

    def prefill_abort(self, h):
        del self._pending[h]


@pytest.mark.parametrize("old_layout", [False, True])
def test_a_skipped_request_returns_whole_and_a_preempted_one_in_pieces(old_layout):
    """A V1 scheduler may leave a running request out of one step without preempting it (token
    budget): it comes back with num_scheduled_tokens = 1 + speculative tokens, resumed_from_preemption
    False and its computed tokens counted.  The worker dropped its slot, so it rebuilds the
    request WHOLE in that step and emits a token (a piecewise rebuild in pieces of 1 would emit
    nothing for many steps while the scheduler counts decode steps).  A request the scheduler
    really recomputes (resumed_from_preemption True, num_computed_tokens 0) is rebuilt in the
    pieces the scheduler names; prefix-cache hits on resume are refused before anything runs."""
    from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeWorker
    eng = _PieceEngine()
    w = SpecDecodeWorker(vllm_config=None, engine=eng)
    N = lambda rid, tok: _NewRequestData(rid, [tok, 1, 2, 3], sampling_params=None)
    S = lambda **kw: _sched(old_layout=old_layout, **kw)
    w.execute_model(S(new=[N("a", 1), N("b", 2)]))
    w.execute_model(S(cached=["a", "b"]))
    w.execute_model(S(cached=["a"]))                       # "b" left out
    assert "b" in w._parked and eng.owners == [1]
    n_b = len(w._tokens["b"])
    so = S(cached=["a", "b"])                              # back, as a decode step: 1 + 3 tokens
    so.num_scheduled_tokens["b"] = 4
    out = w.execute_model(so)
    assert not w._parked and not w._prefill_of and eng.owners == [1, 2]
    assert eng.prompts[-1] == w._tokens["b"][:n_b]         # rebuilt whole from prompt + emitted
    assert eng.decoded_from[-1] == 4 and eng.decoded_from[:2] == [None, None]   # its own tokens start at 4
    assert len(out.sampled_token_ids[1]) == 1              # and it emitted in that step
    # a real preemption: recomputed from position 0 in the scheduler's pieces
    w.execute_model(S(cached=["a"]))
    n_b = len(w._tokens["b"])
    so = S(cached=["a", "b"], resumed={"b"})
    so.num_scheduled_tokens["b"] = 3
    out = w.execute_model(so)
    assert out.sampled_token_ids[1] == [] and w._prefill_left["b"] == n_b - 3
    so = S(cached=["a", "b"])
    so.num_scheduled_tokens["b"] = n_b - 3
    out = w.execute_model(so)
    assert len(out.sampled_token_ids[1]) == 1 and not w._prefill_of and eng.owners == [1, 2]
    assert eng.decoded_from[-1] == 4                       # the pieces carry it too
    # resumed with prefix-cache hits: refused, nothing ran, the request stays parked
    w.execute_model(S(cached=["a"]))
    steps = eng.steps
    with pytest.raises(ValueError, match="prefix-cached"):
        w.execute_model(S(cached=["a", "b"], resumed={"b"}, computed={"b": 16}))
    assert eng.steps == steps and "b" in w._parked and eng.owners == [1]


def test_a_refused_step_loses_no_tokens():
    """One running request plus one bad prompt: the step raises BEFORE the running sequence is
    stepped, so a caller that drops the bad request and retries gets every token."""
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SamplingParams,
                                                        SequenceGroupMetadata, SpecDecodeWorker)
    eng = FakeEngine()
    eng.cfg.max_seq_len = 8
    eng.cfg.temperature = 0.0
    w = SpecDecodeWorker(vllm_config=None, engine=eng)
    P = lambda rid, toks, **kw: SequenceGroupMetadata(rid, True, toks, SamplingParams(**kw))
    R = lambda rid: SequenceGroupMetadata(rid, False)
    w.execute_model(ExecuteModelRequest([P("a", [1, 1])]))
    bad = [P("long", list(range(8))),                   # prompt + 1 > max_seq_len
           P("empty", []),
           P("hot", [2, 2], temperature=0.7),           # another temperature
           P("seeded", [2, 2], seed="eleven"),          # a seed that is not an integer
           P("a", [2, 2])]                              # already running
    for g in bad:
        with pytest.raises((ValueError, KeyError)):
            w.execute_model(ExecuteModelRequest([R("a"), g]))
        assert eng.steps == 0 and eng.active == 1       # nothing ran, nothing was admitted
    with pytest.raises(KeyError):                       # listed twice
        w.execute_model(ExecuteModelRequest([R("a"), R("a")]))
    with pytest.raises(RuntimeError):                   # too many prompts for the free slots
        w.execute_model(ExecuteModelRequest([R("a")] + [P(f"p{i}", [3, 3]) for i in range(3)]))
    assert eng.steps == 0 and eng.active == 1
    out = w.execute_model(ExecuteModelRequest([R("a"), P("b", [2, 2])]))
    assert out.sampled_token_ids == [[101], [200]] and eng.steps == 1
