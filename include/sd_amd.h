/*
 * sd_amd.h — C ABI of the MI355X (gfx950) speculative-decoding hot path.
 *
 * Every entry point is `extern "C"`, takes raw DEVICE pointers plus sizes and a
 * HIP stream, returns 0 on success or a negative sd_status, never throws, never
 * synchronises the device and never allocates device memory: the caller owns
 * every buffer (scratch included — ask sd_*_workspace_bytes and pass it in).
 * Work is stream-ordered; asynchronous HIP faults surface on the caller's next
 * synchronisation.  One process per GPU; calls on different streams of one GPU
 * may overlap only if they use different workspaces.
 *
 * What each group replaces.  The reference tree holds no speculative-decoding
 * code: the path lives in its un-vendored dependency `vllm`
 * (/root/reference/pyproject.toml:31, observed 0.10.2 with
 * speculative_config=None at
 * /root/reference/src/performance_enhancement/test.ipynb:148).  The seam this
 * library sits behind is the worker-injection seam the reference uses
 * (/root/reference/src/qwen_image_gen/executor.py:14-26,
 * /root/reference/src/qwen_image_gen/worker.py:34-58) and the step contract at
 * /root/reference/src/doc/engine_core.md:117-122,280-340.  Upstream symbol
 * names quoted below are the vLLM-V0 names the plugin surface keeps.
 */
#ifndef SD_AMD_H
#define SD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sd_stream_t; /* a hipStream_t, passed as an opaque pointer */

typedef enum {
  SD_OK = 0,
  SD_ERR_INVALID = -1,   /* bad argument (shape, alignment, null pointer) */
  SD_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are built for */
  SD_ERR_HIP = -3,       /* a HIP runtime call failed at launch time */
  SD_ERR_WORKSPACE = -4  /* workspace too small */
} sd_status;

/* Library version (major*10000 + minor*100 + patch). */
int sd_version(void);
/* Copies the calling thread's last error text (NUL-terminated) into buf. */
int sd_last_error(char* buf, size_t n);
/* Number of CUs / name of the current device; also proves HIP is usable. */
int sd_device_info(int* cu_count, char* name, size_t n);

/* ------------------------------------------------------------------------
 * Acceptance (replaces vllm.model_executor.layers.rejection_sampler
 * RejectionSampler.forward, which the task statement calls
 * `_custom_ops.rejection_sample`; contract rows at
 * /root/reference/src/doc/engine_core.md:323-324 — one variable-length token
 * list per request).
 * ---------------------------------------------------------------------- */

/* Modified rejection sampling (Leviathan et al. 2023, Alg. 1), batched.
 *   target_probs   [B, k+1, V] f32   p_i, row k is the bonus position
 *   bonus_ids      [B]        i64    token already sampled from p_k
 *   draft_probs    [B, k, V]  f32    q_i
 *   draft_ids      [B, k]     i64    x_i ~ q_i
 *   u_accept       [B, k]     f32    uniforms in [0,1)
 *   u_recover      [B, k+1]   f32    uniforms in [0,1) for the resampled token
 *   proposal_lens  [B] i32 or NULL   (NULL = every sequence proposes k)
 *   out_ids        [B, k+1]   i64    accepted prefix, then one more token,
 *                                    then -1
 *   counters       [3] i64 or NULL   += {accepted, emitted, draft} tokens
 * Accept x_i iff u_i * q_i(x_i) < p_i(x_i).  At the first rejected position j
 * the token is drawn from norm(max(p_j - q_j, 0)) by inverse CDF with
 * u_recover[b, j] (f64 running sum, first index whose sum exceeds
 * u * total).  With all L = proposal_lens[b] accepted: L == k emits
 * bonus_ids[b]; L < k draws from p_L with u_recover[b, L].
 * Requires 1 <= k <= 63, V % 4 == 0, 16-byte aligned rows.                */
int sd_rejection_sample(const float* target_probs, const int64_t* bonus_ids,
                        const float* draft_probs, const int64_t* draft_ids,
                        const float* u_accept, const float* u_recover,
                        const int32_t* proposal_lens, int64_t* out_ids,
                        int64_t* counters, int B, int k, int V,
                        sd_stream_t stream);

/* Greedy acceptance: accept x_i while x_i == target_ids[b, i]; then emit
 * target_ids[b, j] at the first mismatch j (or at j == L).  Same outputs and
 * counters as above.  target_ids [B, k+1] i64, draft_ids [B, k] i64.       */
int sd_greedy_verify(const int64_t* target_ids, const int64_t* draft_ids,
                     const int32_t* proposal_lens, int64_t* out_ids,
                     int64_t* counters, int B, int k, sd_stream_t stream);

/* Row-wise helpers over logits [R, V] f32 (row stride ld, in elements).
 * argmax: smallest index among equal maxima.  softmax: exp((x-max)/T)/sum.
 * sample: inverse CDF of probs with one uniform per row (f64 running sum).  */
/* Typical acceptance (Cai et al. 2024, "Medusa"; replaces upstream
 * `TypicalAcceptanceSampler.forward`, the alternative acceptor behind the same
 * sampler surface as sd_rejection_sample).  Proposal x_i stands iff
 *   p_i(x_i) > min(posterior_threshold, posterior_alpha * exp(-H(p_i))),
 * H(p) = -sum p log(p + 1e-5) (f32 terms, f64 sum).  The first rejected position
 * emits argmax p_j; k accepted proposals are followed by bonus_ids[b]; a short
 * proposal (L < k) fully accepted emits argmax p_L.  Arguments as in
 * sd_rejection_sample; no uniforms (the rule is deterministic).              */
int sd_typical_acceptance(const float* target_probs, const int64_t* bonus_ids,
                          const int64_t* draft_ids, const int32_t* proposal_lens,
                          float posterior_threshold, float posterior_alpha,
                          int64_t* out_ids, int64_t* counters, int B, int k, int V,
                          sd_stream_t stream);

int sd_argmax_rows(const float* logits, int64_t ld, int64_t* ids, int R, int V,
                   sd_stream_t stream);
int sd_softmax_rows(const float* logits, int64_t ld, float temperature,
                    float* probs, int64_t ld_probs, int R, int V,
                    sd_stream_t stream);
int sd_sample_rows(const float* probs, int64_t ld, const float* uniforms,
                   int64_t* ids, int R, int V, sd_stream_t stream);

/* top-k / top-p / min-p filtering of probability rows, in place, with per-sequence parameters
 * (SamplingParams.top_k / top_p / min_p; upstream: vllm/v1/sample/ops/topk_topp_sampler.py
 * apply_top_k_top_p and the sampler's min-p step; not in the reference tree).  Row r belongs to
 * sequence r / rows_per_seq.  Order: top-k (keep entries >= the k-th largest value), min-p (keep
 * entries >= min_p x the row maximum), then top-p on what is left (keep entries >= the largest
 * value t for which the entries >= t hold at least top_p of the remaining mass); the kept entries
 * are renormalised.  Entries equal to a cut-off are kept.  top_k[s] <= 0, top_p[s] >= 1,
 * min_p[s] <= 0 switch a filter off; any of the arrays may be NULL.  One workgroup per row. */
int sd_filter_probs_seq(float* probs, int64_t ld, const int32_t* top_k, const float* top_p,
                        const float* min_p, int rows_per_seq, int R, int V, sd_stream_t stream);

/* Per-request additive logit masks (SamplingParams.logit_bias, allowed_token_ids, bad_words of one
 * token; upstream: vllm/v1/sample/logits_processor, not in the reference tree): row r of logits
 * [R, ld] belongs to sequence r / rows_per_seq; where enabled[seq] != 0, bias[seq, 0..V-1] (f32, row
 * stride ld_bias; -inf bans a token) is added to the row in place.  Applied to the target's
 * logits before argmax / softmax, so acceptance and every emitted token see the masked
 * distribution; the draft needs none (a banned proposal is simply rejected).                   */
int sd_add_logit_bias(float* logits, int64_t ld, const float* bias, int64_t ld_bias,
                      const int32_t* enabled, int rows_per_seq, int R, int V, sd_stream_t stream);

/* Log-probabilities of chosen tokens, for ModelRunnerOutput.logprobs
 * (/root/reference/src/doc/engine_core.md:326-329: token ids, log-probabilities and ranks per
 * sampled token, max_num_logprobs + 1 columns).  Row r of logits [R, ld] (f32, raw: no
 * temperature, no filters) and tokens[r] give out_ids[r, 0] = tokens[r], out_logprobs[r, 0] =
 * logits[r, tokens[r]] - logsumexp(row) and out_ranks[r] = the number of entries >= the chosen
 * one (1 for the most likely token; upstream counts ties the same way); columns 1..n_top hold
 * the n_top largest entries of the row, largest first, ties by lower index (id -1 and -inf past
 * the row's finite entries).  A token id outside 0..V-1 skips row r (its outputs are left alone).
 * out_ids / out_logprobs are [R, n_top + 1]; 0 <= n_top <= SD_LOGPROBS_TOP_MAX.  One workgroup
 * per row; n_top + 3 passes over the row.                                                   */
#define SD_LOGPROBS_TOP_MAX 32
int sd_token_logprobs(const float* logits, int64_t ld, const int32_t* tokens, int R, int V,
                      int n_top, int32_t* out_ids, float* out_logprobs, int32_t* out_ranks,
                      sd_stream_t stream);

/* sd_softmax_rows(_split) with one temperature per sequence (per-request SamplingParams.
 * temperature): row r uses temperatures[r / rows_per_seq] (device, f32).  A temperature of 0 asks
 * for the greedy distribution: probability 1 on the first maximum of the row, so that the
 * acceptance kernels treat a greedy request exactly like token matching.  scratch == NULL: one
 * workgroup per row; else the split form (sd_sampling_scratch_bytes(R) bytes).          */
int sd_softmax_rows_seq(const float* logits, int64_t ld, const float* temperatures,
                        int rows_per_seq, float* probs, int64_t ld_probs, int R, int V,
                        void* scratch, sd_stream_t stream);

/* Split forms of the three vocabulary-wide steps of the sampling path: each row
 * is cut into 64 slices worked on by one workgroup each (a handful of rows
 * otherwise occupies a handful of CUs), a per-row step finishes.  Same results
 * as the one-workgroup forms up to the rounding of the f32 softmax sum and, for
 * draws, of an f64 sum taken in a different order.  `scratch` holds
 * sd_sampling_scratch_bytes(R) bytes (R = rows, or sequences for the rejection
 * form); contents need not survive the call.                                */
size_t sd_sampling_scratch_bytes(int R);
int sd_softmax_rows_split(const float* logits, int64_t ld, float temperature,
                          float* probs, int64_t ld_probs, int R, int V,
                          void* scratch, sd_stream_t stream);
int sd_sample_rows_split(const float* probs, int64_t ld, const float* uniforms,
                         int64_t* ids, int R, int V, void* scratch,
                         sd_stream_t stream);
int sd_rejection_sample_split(const float* target_probs, const int64_t* bonus_ids,
                              const float* draft_probs, const int64_t* draft_ids,
                              const float* u_accept, const float* u_recover,
                              const int32_t* proposal_lens, int64_t* out_ids,
                              int64_t* counters, int B, int k, int V,
                              void* scratch, sd_stream_t stream);

/* out[i] = uniform in [0, 1) (24 bits) from a counter-based hash of (seed,
 * *counter, i); then *counter += 1.  Lets a captured step draw fresh uniforms
 * on every replay without host work.  (The acceptance entry points take their
 * uniforms as inputs; this is one way to fill them.)                         */
int sd_fill_uniforms(float* out, int n, uint64_t seed, uint64_t* counter,
                     sd_stream_t stream);

/* Per-sequence streams of uniforms (a request's SamplingParams.seed; upstream hands
 * RejectionSampler.forward a `seeded_seqs` map of per-sequence generators): sequence b draws
 * its j-th number of a step from hash(seeds[b], counters[b], j), whatever else is in the batch,
 * so a seeded request samples the same tokens alone and among others.  The numbers land where
 * the step's kernels read them: region r holds n numbers per sequence at
 * out[offset + i * stride_j + b * stride_b], i < n; j runs through the regions in order.
 * Afterwards counters[b] += 1 for b < B.  At most SD_UNIFORM_REGIONS_MAX regions.          */
#define SD_UNIFORM_REGIONS_MAX 8
typedef struct {
  int64_t offset;
  int32_t n;
  int64_t stride_j, stride_b;
} sd_uniform_region;
int sd_fill_uniforms_seq(float* out, const sd_uniform_region* regions, int n_regions,
                         const uint64_t* seeds, uint64_t* counters, int B, sd_stream_t stream);

/* ------------------------------------------------------------------------
 * Paged KV cache and attention (replaces vLLM's paged_attention_v1/v2 as used
 * by spec_decode.batch_expansion's scorer, with k+1 query tokens per sequence
 * scored in ONE pass instead of k+1 expanded single-query sequences).
 *
 * Page = 16 tokens of one KV head.  Layouts (bf16), chosen so that a wave's
 * MFMA operand fragments are contiguous 16-byte pieces of a page:
 *   K page  [D/8][16 tokens][8]      (8-element chunks of the head dim)
 *   V page  [D][16 slots]            (transposed; token t sits in slot
 *                                     (t&3) | ((t>>3)&1)<<2 | ((t>>2)&1)<<3)
 *   cache   [num_pages][Hkv][page]
 * The cache must be zero-filled when allocated (unwritten slots are read and
 * masked, so they must hold finite values).
 * ---------------------------------------------------------------------- */

/* Rotary embedding + append.  For token t of sequence b (t < q_len) at
 * position pos = seq_lens[b] + pos_offset + t: rotates q in place, rotates k,
 * and writes k, v into the page block_tables[b][pos/16].
 *   qkv        [B*q_len, (Hq + 2*Hkv) * D] bf16, q heads then k then v
 *   rope_cs    [max_pos, D] f32: cos (D/2) then sin (D/2); NULL = no rotation */
int sd_rope_append(void* qkv, const float* rope_cs, int max_pos, void* k_cache,
                   void* v_cache, const int32_t* block_tables,
                   int max_blocks_per_seq, const int32_t* seq_lens,
                   int pos_offset, int B, int q_len, int Hq, int Hkv, int D,
                   sd_stream_t stream);

/* Bytes of scratch sd_paged_attention needs for these bounds. */
size_t sd_paged_attention_workspace_bytes(int B, int q_len, int Hq, int D,
                                          int max_ctx, int split_tokens);

/* Causal attention of q_len new query tokens per sequence over its paged KV.
 * Query token t of sequence b sits at position seq_lens[b] + pos_offset + t
 * and attends keys 0..that position (its own K/V must already be appended).
 *   q    [B*q_len, Hq, D] bf16 with row stride ldq (elements)
 *   out  [B*q_len, Hq*D]  bf16 with row stride ldo
 *   max_ctx       host upper bound of seq_lens[b] + pos_offset + q_len
 *   split_tokens  keys per workgroup (multiple of 128); results for a given
 *                 query depend on split_tokens but not on B, q_len or max_ctx
 * Supported: D in {64, 128}, q_len * (Hq/Hkv) <= 64.
 * With more than one split (max_ctx > split_tokens) the workspace receives the
 * per-split partials: [B, Hkv, n_splits, m_pad, D] f32 un-normalised outputs,
 * then [B, Hkv, n_splits, m_pad, 2] f32 (row max in log2 units, row sum), with
 * m_pad = 32 if q_len*(Hq/Hkv) <= 32 else 64 and row t*(Hq/Hkv)+g for query
 * token t, head g of the KV group; a merge launch then writes `out`.  out ==
 * NULL skips that launch and leaves the partials to sd_linear_attn_merge.    */
int sd_paged_attention(const void* q, int64_t ldq, const void* k_cache,
                       const void* v_cache, const int32_t* block_tables,
                       int max_blocks_per_seq, const int32_t* seq_lens,
                       int pos_offset, void* out, int64_t ldo, int B, int q_len,
                       int Hq, int Hkv, int D, float scale, int max_ctx,
                       int split_tokens, void* workspace,
                       size_t workspace_bytes, sd_stream_t stream);

/* The same for prompt chunks: every sequence has q_total new tokens, attended
 * in blocks of q_block tokens (q_block * (Hq/Hkv) <= 64; the last block may be
 * shorter) by ONE launch: the blocks of a sequence are independent workgroup
 * columns over the same block table, each with its own causal bound, exactly
 * as ceil(q_total / q_block) calls of sd_paged_attention with pos_offset
 * advanced by q_block would compute them.  q and out hold B*q_total rows; the
 * workspace is sized for B * ceil(q_total / q_block) sequences
 * (sd_paged_attention_workspace_bytes); out must not be NULL unless q_total ==
 * q_block.                                                                  */
int sd_paged_attention_blocked(const void* q, int64_t ldq, const void* k_cache,
                               const void* v_cache, const int32_t* block_tables,
                               int max_blocks_per_seq, const int32_t* seq_lens,
                               int pos_offset, void* out, int64_t ldo, int B,
                               int q_total, int q_block, int Hq, int Hkv, int D,
                               float scale, int max_ctx, int split_tokens,
                               void* workspace, size_t workspace_bytes,
                               sd_stream_t stream);

/* sd_paged_attention for callers that cannot allocate or hold a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in
 * the reference: it exists for those callers only.  Same arguments and bit-identical
 * results, minus the workspace: the split partials go to a SD_STATIC_SCRATCH_BYTES array
 * inside the library's code object (one copy per GPU, held until the process ends).  The
 * call allocates nothing (no hipMalloc / hipFree); `out` must not be NULL; shapes whose
 * partials ([B, Hkv, n_splits, m_pad, D + 2] f32) exceed the array return
 * SD_ERR_WORKSPACE.  Every call on a GPU shares that GPU's array: calls that overlap on
 * one GPU (two streams, two host threads) race; calls on one stream are safe.  The first
 * call on a device looks the array's address up (hipGetSymbolAddress): make it outside
 * stream capture.                                                              */
#define SD_STATIC_SCRATCH_BYTES (16u << 20)
int sd_paged_attention_static(const void* q, int64_t ldq, const void* k_cache,
                              const void* v_cache, const int32_t* block_tables,
                              int max_blocks_per_seq, const int32_t* seq_lens,
                              int pos_offset, void* out, int64_t ldo, int B, int q_len,
                              int Hq, int Hkv, int D, float scale, int max_ctx,
                              int split_tokens, sd_stream_t stream);

/* tokens[b*stride + col] = map[tokens[b*stride + col]] for b < B (ids outside 0..n_map-1 are
 * clamped).  For a draft whose LM head covers a subset of the vocabulary (sd_model_desc.lm_vocab
 * rows in the order of `map`): turns the head's row index into the token id after
 * sd_propose_from_partials / sd_store_proposal.  (Upstream counterpart: none in the reference;
 * the technique is a frequency-ranked draft vocabulary.)                               */
int sd_map_tokens(int32_t* tokens, int64_t stride, int col, const int32_t* map, int n_map,
                  int B, sd_stream_t stream);

/* ------------------------------------------------------------------------
 * Token trees (multi-candidate verification; upstream: vllm/spec_decode's
 * tree-style scoring is not in the reference tree, the semantics follow
 * SpecInfer / Medusa tree attention: every node attends the context and its
 * own ancestors).  One verify pass scores q_len = 1 + N rows per sequence: row
 * 0 is the sequence's last token (the root), rows 1..N are candidate tokens in
 * topological order (a node's parent has a smaller row index).  Row t is
 * cached at slot seq_lens[b] + pos_offset + t but rotated at position
 * seq_lens[b] + pos_offset + depth(t); after acceptance the accepted path's
 * cache entries are moved to their chain slots (sd_kv_compact).
 *
 * sd_paged_attention_tree: sd_paged_attention with an ancestor mask.
 * tree_mask [B*q_len] u64 (device): bit j of entry b*q_len + t set iff new
 * token t attends new token j (its ancestors and itself, j <= t < 64).  The
 * arithmetic of a row is that of sd_paged_attention over the keys it sees;
 * masked keys sit at other slots than in a chain, so sums run in another
 * order: results match a chain pass to rounding, not bit for bit.          */
int sd_paged_attention_tree(const void* q, int64_t ldq, const void* k_cache,
                            const void* v_cache, const int32_t* block_tables,
                            int max_blocks_per_seq, const int32_t* seq_lens,
                            int pos_offset, const uint64_t* tree_mask, void* out,
                            int64_t ldo, int B, int q_len, int Hq, int Hkv, int D,
                            float scale, int max_ctx, int split_tokens,
                            void* workspace, size_t workspace_bytes, sd_stream_t stream);
/* Before the verify pass: tree_tok[b][0] = tok_buf[b*tok_stride + tok_col] (the
 * root = last emitted token) and rope_pos[b*q_len + t] = seq_lens[b] +
 * pos_offset + tree_depth[b*q_len + t] (depth of the root = 0).             */
int sd_tree_prepare(const int32_t* tok_buf, int64_t tok_stride, int tok_col,
                    int32_t* tree_tok, const int32_t* tree_depth,
                    const int32_t* seq_lens, int pos_offset, int32_t* rope_pos, int B,
                    int q_len, sd_stream_t stream);
/* Levels of a tree grown by the draft model: for every sequence s and r < rows the b largest
 * entries of logits row row_offset + s*row_step + r as tokens, largest first (ties: lowest
 * index), into out_tok[s*out_stride + out_col + r*b .. + b).  The first level takes rows = 1,
 * b = the branches; a deeper level rows = the branches, b = 1 (every branch's arg-max).
 * (Upstream's top-k proposers: the Medusa / EAGLE-style top-k candidate selection of
 * vllm/spec_decode; no draft distribution is kept: the candidates are verified as point
 * masses.)                                                                          */
#define SD_TOPK_TOKENS_MAX 8
int sd_topk_tokens(const float* logits, int64_t ld, int V, int row_offset, int row_step,
                   int rows, int b, int32_t* out_tok, int64_t out_stride, int out_col, int B,
                   sd_stream_t stream);
/* Greedy acceptance over a tree.  target_ids [B*q_len] i64: the target's argmax
 * after every row.  tree_tok / tree_parent [B, q_len] i32 (parent row of rows
 * 1..; entry 0 unused), tree_len [B] rows in use (root included, 1..q_len).
 * Walks from the root: while the target's token after the current row equals
 * the token of one of its children (lowest row wins) and fewer than k are
 * accepted, descend.  out_ids [B, k+1] i64: accepted tokens, then the target's
 * token after the last accepted row, then -1.  path [B, k] i32: accepted rows;
 * n_path [B].  counters (or NULL) += {accepted, emitted, depth of the tree}.  */
int sd_tree_accept_greedy(const int64_t* target_ids, const int32_t* tree_tok,
                          const int32_t* tree_parent, const int32_t* tree_len, int q_len,
                          int k, int64_t* out_ids, int32_t* path, int32_t* n_path,
                          int64_t* counters, int B, sd_stream_t stream);
/* Acceptance over a tree under sampling, for candidates without a draft
 * distribution (prompt lookup, proposals made outside: point-mass proposals; the
 * multi-step speculative sampling of SpecInfer, Miao et al. 2024, section 4.3,
 * with q = one-hot; not in the reference tree).  target_probs [B*q_len, ld]: the
 * target's distribution after every tree row (temperature and filters applied).
 * At the current node, with r = its row: the children are tried in row order;
 * child x stands iff u_accept[b, row of x] * (1 - removed) < r(x) in f64, where
 * removed is the mass of the children rejected so far at this node; a rejected
 * child's token leaves r (a later sibling that carries the same token is skipped:
 * nothing of it is left).  A row's parent must be a smaller row (as
 * sd_tree_prepare's callers build them); tokens outside 0..V-1 are skipped.
 * With no child standing, or k accepted, the emitted
 * token is the inverse-CDF draw (u_final[b], f64 running sum as in
 * sd_rejection_sample) from r without the rejected tokens.  Outputs and counters
 * as sd_tree_accept_greedy.  u_accept [B, q_len], u_final [B] f32 in [0, 1).    */
int sd_tree_accept_sample(const float* target_probs, int64_t ld, const int32_t* tree_tok,
                          const int32_t* tree_parent, const int32_t* tree_len, int q_len,
                          int k, const float* u_accept, const float* u_final,
                          int64_t* out_ids, int32_t* path, int32_t* n_path,
                          int64_t* counters, int B, int V, sd_stream_t stream);
/* Moves the accepted rows' K/V (every layer, every KV head) from slot
 * seq_lens[b] + pos_offset + path[b][d] to slot seq_lens[b] + pos_offset + d + 1,
 * d = 0..n_path[b]-1, in that order.  Call before seq_lens advances.        */
int sd_kv_compact(void* k_cache, void* v_cache, int64_t layer_stride_bytes, int n_layers,
                  const int32_t* block_tables, int max_blocks_per_seq,
                  const int32_t* seq_lens, int pos_offset, const int32_t* path,
                  const int32_t* n_path, int k, int B, int Hkv, int D, sd_stream_t stream);

/* ------------------------------------------------------------------------
 * Dense ops of a decoder layer at <= 16 token rows (weight streaming; replaces
 * the per-step model execution inside spec_decode.multi_step_worker and the
 * scorer's target forward).
 *
 * Weight matrices are consumed PRE-TILED for the 16x16x32 bf16 MFMA: tile
 * (nt, ks) = 16 rows x 32 columns stored as [64 lanes][8 elements], lane l
 * holding row (l & 15), columns 32*ks + 8*(l >> 4) .. +8; tiles ordered
 * [nt][ks].  One wave load is then 1 KiB of contiguous memory.  Which source
 * rows form tile nt depends on the mode:
 *   SD_TILE_PLAIN   rows 16*nt .. 16*nt+15
 *   SD_TILE_GATED   src = [2F, K] (gate rows, then up rows): tile nt holds gate
 *                   rows 8*nt..8*nt+7 and up rows 8*nt..8*nt+7 (SILU_MUL epilogue)
 *   SD_TILE_ROTARY  src = [heads*D, K]: tile nt holds dims i0..i0+7 of a head and
 *                   their rotation partners i0+D/2..i0+D/2+7 (sd_norm_qkv_rope)
 * sd_tile_weights converts a row-major bf16 matrix once, at load time.
 * ---------------------------------------------------------------------- */

typedef enum { SD_TILE_PLAIN = 0, SD_TILE_GATED = 1, SD_TILE_ROTARY = 2 } sd_tile_mode;

/* Elements (bf16) of the tiled copy of an N x K matrix (N = F outputs for
 * SD_TILE_GATED).  K must be a multiple of 32.                              */
size_t sd_tiled_weight_elems(int N, int K, int mode);
/* src row-major -> dst tiled; param = head_dim for SD_TILE_ROTARY, else 0.  */
int sd_tile_weights(const void* src, void* dst, int N, int K, int mode,
                    int param, sd_stream_t stream);
/* The same for fp8 (OCP e4m3) weights, one byte per element: tile (nt, ks2) holds
 * two 32-column k-steps per lane (16 bytes), so a wave load is still 1 KiB.
 * dst holds sd_tiled_weight_elems(N, K, mode) bytes.  K must be a multiple of 64.
 * The linear layers below take such a buffer as W when w_scale is not NULL:
 * w_scale[r] (f32, indexed by source row like bias) multiplies the finished sum
 * of output row r, i.e. the matrix is W[r, :] * w_scale[r]; weights are
 * converted to bf16 in registers and activations stay bf16 (weight-only fp8:
 * these layers are bound by the weight stream, which halves).                  */
int sd_tile_weights_fp8(const void* src, void* dst, int N, int K, int mode,
                        int param, sd_stream_t stream);

typedef enum {
  SD_EPI_STORE = 0,     /* out = bf16(acc + bias)                           */
  SD_EPI_RESIDUAL = 1,  /* out = bf16(bf16(acc + bias) + residual)          */
  SD_EPI_SILU_MUL = 2,  /* GATED tiling: out = silu(gate row n) * up row n   */
  SD_EPI_RELU = 3,      /* out = bf16(max(acc + bias, 0))                   */
  SD_EPI_F32 = 4,       /* out (f32) = acc + bias                           */
  SD_EPI_GELU_TANH = 5  /* out = bf16(gelu_tanh(acc + bias))                */
} sd_epilogue;

/* out[m, n] = epilogue(sum_k x[m, k] * W[n, k]),  m < M <= 16.
 *   W tiled (PLAIN; GATED for SD_EPI_SILU_MUL, then N = F outputs);
 *   x [M, K] bf16 (row stride ldx); bias [N] (GATED: [2F]) bf16 or NULL;
 *   residual [M, N] bf16 (row stride ldo) or NULL.
 * Each output element is accumulated in an order that depends only on K, so
 * a row's result does not depend on M or on the other rows.
 * Requires K % 32 == 0 and 16-byte aligned rows.  w_scale: NULL for bf16
 * tiles, else W holds fp8 tiles (sd_tile_weights_fp8) and K % 64 == 0.       */
int sd_linear(const void* W, const float* w_scale, const void* x, int64_t ldx,
              const void* bias, const void* residual, void* out, int64_t ldo,
              int M, int N, int K, int epilogue, sd_stream_t stream);

/* The same linear layers at up to 256 token rows (batched draft passes, batched
 * verify; replaces the batch-expanded target pass of
 * spec_decode.batch_expansion at batch > 2), over the same tiled weights.
 * Activations pass through LDS shared by the waves of a workgroup; matrices
 * with few rows split K over workgroups, partial sums are added in a fixed
 * order by a second launch that applies the epilogue.  A token row's result
 * depends on (N, K) only, not on M; it differs in rounding from sd_linear's
 * (another summation order), so a caller uses one family for all its passes.
 * No fused norm: normalise first (sd_rmsnorm / sd_layernorm).  w_scale as in
 * sd_linear (NULL: bf16 tiles; else fp8 tiles and K % 64 == 0).
 * workspace: sd_linear_wide_workspace_bytes(M, N, K, epilogue) bytes; for
 * sd_qkv_rope_wide pass N = (Hq + 2*Hkv) * D and SD_EPI_STORE.              */
size_t sd_linear_wide_workspace_bytes(int M, int N, int K, int epilogue);

/* Rotary positions that differ from the cache position (the multimodal rotary
 * embedding of Qwen2-VL: transformers/models/qwen2_vl/modeling_qwen2_vl.py,
 * apply_multimodal_rotary_pos_emb and get_rope_index).  Frequency index i of a
 * head (0 <= i < D/2) takes the temporal position when i < section_t, the
 * height position when i < section_t + section_h, else the width position.
 * pos != NULL: the three positions of every token row, [3][B*q_len] (prompt
 * chunks with image tokens); else delta != NULL: all three are the cache
 * position + delta[b] (text after an image; [B], one entry per sequence).
 * A NULL sd_rope_positions*, or both pointers NULL, is the ordinary rotary
 * embedding.  Positions are clamped to the rope table.  The cache slot a
 * token's K/V goes to is never affected.                                     */
typedef struct {
  const int32_t* pos;
  const int32_t* delta;
  int section_t, section_h;
} sd_rope_positions;
int sd_linear_wide(const void* W, const float* w_scale, const void* x, int64_t ldx,
                   const void* bias,
                   const void* residual, void* out, int64_t ldo, int M, int N,
                   int K, int epilogue, void* workspace, size_t workspace_bytes,
                   sd_stream_t stream);
int sd_qkv_rope_wide(const void* W, const float* w_scale, const void* x, int64_t ldx,
                     const void* bias,
                     void* q_out, int64_t ldq, const float* rope_cs, int max_pos,
                     void* k_cache, void* v_cache, const int32_t* block_tables,
                     int max_blocks_per_seq, const int32_t* seq_lens,
                     int pos_offset, int B, int q_len, int Hq, int Hkv, int D,
                     int K, const sd_rope_positions* rope_pos, void* workspace,
                     size_t workspace_bytes, sd_stream_t stream);

/* fp8 activations for the wide family (opt-in; fp8 weight tiles only, K % 128 == 0):
 * sd_row_scales_fp8 leaves, for each of M token rows of x, scales[m] = max |x[m]| / 448 and
 * scales[M + m] = 448 / max |x[m]| (both 1 for a row of zeros; scales: 2*M floats);
 * sd_linear_wide_a8 / sd_qkv_rope_wide_a8 are sd_linear_wide / sd_qkv_rope_wide with the
 * rows quantised on their way into LDS, q = e4m3(clamp(x * scales[M + m], +-448)), round to
 * nearest even, the products on the fp8 matrix instruction of gfx950
 * (v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales) and a sum multiplied by
 * scales[m] and the weight row's scale before the epilogue.  Epilogues: all but RELU and
 * GELU_TANH.  A row's result depends on that row, N and K only.  Workspace as for the
 * bf16-activation forms.  No counterpart in the reference (it names "fp8 weights (CDNA4 fp8
 * MFMA)" for its Qwen2-VL configuration without code for it).                          */
int sd_row_scales_fp8(const void* x, int64_t ldx, int M, int K, float* scales,
                      sd_stream_t stream);
int sd_linear_wide_a8(const void* W, const float* w_scale, const float* x_scales,
                      const void* x, int64_t ldx, const void* bias, const void* residual,
                      void* out, int64_t ldo, int M, int N, int K, int epilogue,
                      void* workspace, size_t workspace_bytes, sd_stream_t stream);
int sd_qkv_rope_wide_a8(const void* W, const float* w_scale, const float* x_scales,
                        const void* x, int64_t ldx, const void* bias, void* q_out,
                        int64_t ldq, const float* rope_cs, int max_pos, void* k_cache,
                        void* v_cache, const int32_t* block_tables, int max_blocks_per_seq,
                        const int32_t* seq_lens, int pos_offset, int B, int q_len, int Hq,
                        int Hkv, int D, int K, const sd_rope_positions* rope_pos,
                        void* workspace, size_t workspace_bytes, sd_stream_t stream);

/* Output projection straight from attention partials: sd_paged_attention with
 * out == NULL (and more than one split) leaves its per-split partials in the
 * workspace; this linear merges them while it stages its activations (same
 * arithmetic as the stand-alone merge) and applies W [N, Hq*D] (PLAIN tiles).
 * Same B, q_len, heads, max_ctx and split_tokens as the attention call; at most
 * 8 splits, B*q_len <= 16; epilogue STORE or RESIDUAL.  Every workgroup redoes
 * the merge, so it pays only for few rows and splits (decode, short verify).  */
int sd_linear_attn_merge(const void* W, const float* w_scale,
                         const void* workspace, size_t workspace_bytes,
                         const int32_t* seq_lens,
                         int pos_offset, int B, int q_len, int Hq, int Hkv, int D,
                         int max_ctx, int split_tokens, const void* bias,
                         const void* residual, void* out, int64_t ldo, int N,
                         int epilogue, sd_stream_t stream);

/* The same with a norm fused in front: x is first normalised per row (RMSNorm,
 * or LayerNorm when layer_norm != 0; norm_b may be NULL) exactly as sd_rmsnorm /
* sd_layernorm would, then multiplied.  K is the hidden size; needs
 * (M + 1) * K * 2 <= 128 KiB (the normalised rows live in LDS).  Epilogues: STORE,
 * SILU_MUL, RELU, F32.                                                        */
int sd_norm_linear(const void* W, const float* w_scale, const void* x,
                   int64_t ldx, const void* norm_w,
                   const void* norm_b, float eps, int layer_norm,
                   const void* bias, void* out, int64_t ldo, int M, int N,
                   int K, int epilogue, sd_stream_t stream);

/* Fused  norm -> QKV projection -> rotary embedding -> KV-cache append.
 * W is [(Hq + 2*Hkv) * D, K] (q heads, k heads, v heads), tiled SD_TILE_ROTARY;
 * bias is indexed by source row.  Rotated q goes to
 * q_out [B*q_len, Hq*D] (row stride ldq); rotated k and v go straight into the
 * cache pages (arguments as sd_rope_append).  norm_w == NULL skips the norm.
 * rope_pos: NULL, or multimodal rotary positions (sd_rope_positions above). */
int sd_norm_qkv_rope(const void* W, const float* w_scale, const void* x,
                     int64_t ldx,
                     const void* norm_w, const void* norm_b, float eps,
                     int layer_norm, const void* bias, void* q_out, int64_t ldq,
                     const float* rope_cs, int max_pos, void* k_cache,
                     void* v_cache, const int32_t* block_tables,
                     int max_blocks_per_seq, const int32_t* seq_lens,
                     int pos_offset, int B, int q_len, int Hq, int Hkv, int D,
                     int K, const sd_rope_positions* rope_pos, sd_stream_t stream);

/* ------------------------------------------------------------------------
 * A decoder layer of the DRAFT model as three launches instead of five, for
 * passes of 1..SD_DRAFT_LAYER_MAX_ROWS token rows (one or two per sequence):
 *   1. sd_norm_qkv_rope_adds   RMSNorm + QKV + rotary + cache append
 *   2. sd_draft_attn_oproj     attention over the whole context + output projection
 *   3. sd_draft_mlp            RMSNorm + gate/up + SiLU*mul + down projection
 * A linear layer whose input is produced distributed over workgroups is computed
 * by the producers: each multiplies its slice of the input with the matching
 * weight columns and contributes a partial output row (launch 2: one f32 slab per
 * KV head, summed by launch 3 while it stages its input; launch 3: float atomics
 * into one f32 accumulator row).  The residual stream travels as (h, pending
 * sums): a normalising launch computes bf16(h + sums) itself and one workgroup
 * stores it for the launches behind.  Rows differ from the five-launch layer's in
 * the last bits (other summation order; the atomics' order is not fixed): for the
 * draft model, whose rounding shows only in what is proposed.  No counterpart in
 * the reference tree (replaces launches of spec_decode.multi_step_worker's model
 * passes).                                                                     */
#define SD_DRAFT_LAYER_MAX_ROWS 4
#define SD_DRAFT_LAYER_MAX_CTX 2048   /* keys one workgroup attends per KV head */

/* Pending f32 partial sums of the rows a fused norm reads: the staging normalises
 * x + add (rounded to bf16 first) and, when x_out is given, one workgroup stores
 * that sum as the new residual rows.  add == NULL: none.                       */
typedef struct {
  const float* add;   /* [M][ld_add] f32 */
  int64_t ld_add;
  void* x_out;        /* [M][ldx] bf16, or NULL */
} sd_row_adds;

/* sd_norm_linear / sd_norm_qkv_rope with an additive input (RMSNorm, bf16 weights;
 * epilogue F32 for sd_norm_linear_adds).  adds == NULL: the plain forms.        */
int sd_norm_linear_adds(const void* W, const float* w_scale, const void* x, int64_t ldx,
                        const void* norm_w, const void* norm_b, float eps, int layer_norm,
                        const void* bias, void* out, int64_t ldo, int M, int N, int K,
                        int epilogue, const sd_row_adds* adds, sd_stream_t stream);
int sd_norm_qkv_rope_adds(const void* W, const float* w_scale, const void* x, int64_t ldx,
                          const void* norm_w, const void* norm_b, float eps, int layer_norm,
                          const void* bias, void* q_out, int64_t ldq, const float* rope_cs,
                          int max_pos, void* k_cache, void* v_cache,
                          const int32_t* block_tables, int max_blocks_per_seq,
                          const int32_t* seq_lens, int pos_offset, int B, int q_len, int Hq,
                          int Hkv, int D, int K, const sd_rope_positions* rope_pos,
                          const sd_row_adds* adds, sd_stream_t stream);
/* 1 if launches 2 and 3 take the shape: rows = B*q_len <= SD_DRAFT_LAYER_MAX_ROWS,
 * q_len <= 2, head_dim 64 or 128, (Hq/Hkv)*q_len <= 32, (Hq/Hkv)*D <= 512,
 * hidden a multiple of 128 up to 4096, ffn a multiple of 32, max_ctx <=
 * SD_DRAFT_LAYER_MAX_CTX.                                                      */
int sd_draft_layer_supported(int rows, int q_len, int hidden, int n_heads, int n_kv_heads,
                             int head_dim, int ffn, int max_ctx);
/* Launch 2.  q [B*q_len, Hq*D] (rotated, from launch 1), caches and block table as
 * sd_paged_attention; o_w [H, Hq*D] tiled SD_TILE_PLAIN.  One workgroup per (KV
 * head, 128 projection rows, sequence) attends the head's WHOLE context (no key
 * splits across workgroups: max_ctx <= SD_DRAFT_LAYER_MAX_CTX) and multiplies the
 * head group's normalised output with its columns of o_w.  slabs [Hkv][B*q_len][H]
 * f32: every element is written.  zero / zero_n: f32 words this launch clears
 * (the accumulator launch 3 adds into), or NULL / 0.                           */
int sd_draft_attn_oproj(const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                        const int32_t* block_tables, int max_blocks_per_seq,
                        const int32_t* seq_lens, int pos_offset, const void* o_w, float* slabs,
                        float* zero, int zero_n, int B, int q_len, int Hq, int Hkv, int D, int H,
                        float scale, int max_ctx, sd_stream_t stream);
/* Launch 3.  x = bf16(h + sum of the n_add slabs add[s][m][:], s in order); h_out
 * (or NULL) receives x; acc[m][:] += down(silu(gate(norm(x))) * up(norm(x))) by
 * float atomics (acc must hold the sums pending so far, usually zeros).  up_w:
 * [2F, H] tiled SD_TILE_GATED; down_w_km: the SD_TILE_PLAIN tiles of [H, F] ordered
 * k-step first ([F/32][H/16][64][8]), so that the 32 input columns a workgroup
 * owns are contiguous.  M <= SD_DRAFT_LAYER_MAX_ROWS, n_add <= 8.              */
int sd_draft_mlp(const void* h, int64_t ldh, const float* add, int n_add, void* h_out,
                 const void* norm_w, float eps, const void* up_w, const void* down_w_km,
                 float* acc, int M, int H, int F, sd_stream_t stream);

/* out = bf16(bf16(x * rsqrt(mean(x^2) + eps)) * w) per row; rows x H bf16. */
int sd_rmsnorm(const void* x, int64_t ldx, const void* w, void* out,
               int64_t ldo, int rows, int H, float eps, sd_stream_t stream);
/* out = bf16((x - mean) * rsqrt(var + eps) * w + b) per row.               */
int sd_layernorm(const void* x, int64_t ldx, const void* w, const void* b,
                 void* out, int64_t ldo, int rows, int H, float eps,
                 sd_stream_t stream);
/* out[r] = table[ids[r*id_stride]] (+ pos_table[pos(r) + pos_bias]) bf16.
 * pos(r) = seq_lens[r / q_len] + pos_offset + r % q_len.  table_tiled != 0:
 * the table is stored in the SD_TILE_PLAIN layout (a tied LM head).  H % 32 == 0. */
int sd_embed(const int32_t* ids, int64_t id_stride_seq, int64_t id_offset,
             const void* table, int table_tiled, const void* pos_table, int pos_bias,
             const int32_t* seq_lens, int pos_offset, void* out, int B,
             int q_len, int H, int vocab, sd_stream_t stream);

/* ------------------------------------------------------------------------
 * Two dependent linear layers of a decoder layer in one launch, for passes of at
 * most SD_CHAIN_MAX_ROWS token rows (a draft model's decode passes, where every
 * launch boundary costs as much as the bytes a small matrix streams).
 * (No counterpart in the reference tree; upstream's draft worker runs one
 * kernel per linear layer.)
 *
 *   sd_chain_o_up      h += o_w . attention;  act = silu(gate) * up of RMSNorm(h)
 *                      == sd_linear_attn_merge (or sd_linear) RESIDUAL, then
 *                         sd_norm_linear SILU_MUL
 *   sd_chain_down_qkv  h += down_w . act;  q, K/V-cache append of the NEXT layer's
 *                      RMSNorm(h) . qkv_w with rotary embedding
 *                      == sd_linear RESIDUAL, then sd_norm_qkv_rope
 *
 * Rows have exactly the bits of those calls (same summation order and
 * roundings).  Inside a launch the first layer's outputs travel between
 * workgroups as tagged 8-byte words in `state` while the second matrix is
 * already streaming.  `state`: sd_chain_state_bytes(hidden) bytes of device
 * memory, 256-byte aligned, prepared once by sd_chain_state_init and then owned by
 * the calls (it carries the launch counter the tags come from); passes that may
 * overlap in time (other streams) need a state each.  Shapes: Hq*D, hidden and
 * ffn multiples of 2048, hidden <= 8192, RMSNorm, gated SiLU MLP, ordinary rotary
 * positions; w_scale pointers as in sd_linear (both matrices of a call bf16 or
 * both fp8); biases may be NULL.  Every wait inside a launch is bounded (0.2 s):
 * a launch whose workgroups cannot all run at once (the grid is one workgroup
 * per CU at most) raises the failure word read by sd_chain_errors and returns
 * unspecified rows instead of hanging.  The launch counter is 32 bits wide:
 * prepare the state again before 2^32 launches (days of decoding).  On
 * MI355X these launches measured slower than the ones they replace (an
 * in-launch exchange costs what a launch boundary costs): a building block
 * for a resident draft pass, not the default path.                           */
#define SD_CHAIN_MAX_ROWS 4
size_t sd_chain_state_bytes(int hidden);
/* Zeroes the state and sets the launch counter; synchronises `stream`.       */
int sd_chain_state_init(void* state, size_t bytes, int hidden, sd_stream_t stream);
/* *failed != 0: some launch on this state gave up waiting (synchronous read). */
int sd_chain_errors(const void* state, int* failed);
/* 1 if both fused calls take these shapes (rows = B*q_len, q_dim = Hq*D).    */
int sd_chain_supported(int rows, int hidden, int q_dim, int ffn);
/* attn != NULL: attention output rows [B*q_len, Hq*D] (row stride ld_attn);
 * attn == NULL: the per-split partials sd_paged_attention (out == NULL) left in
 * attn_workspace are merged while staging (arguments as sd_linear_attn_merge).
 * h [B*q_len, hidden] is read (residual) and overwritten; act [B*q_len, ffn].  */
int sd_chain_o_up(void* state, const void* o_w, const float* o_s, const void* o_b,
                  const void* attn, int64_t ld_attn, const void* attn_workspace,
                  size_t attn_workspace_bytes, const int32_t* seq_lens,
                  int pos_offset, int B, int q_len, int Hq, int Hkv, int D,
                  int max_ctx, int split_tokens, void* h, int hidden,
                  const void* norm_w, float eps, const void* up_w,
                  const float* up_s, const void* up_b, void* act, int64_t ld_act,
                  int ffn, sd_stream_t stream);
/* norm_w / qkv_w / qkv_b / the cache pointers belong to the layer AFTER the one
 * whose down projection this is; other arguments as sd_norm_qkv_rope.         */
int sd_chain_down_qkv(void* state, const void* down_w, const float* down_s,
                      const void* down_b, const void* act, int64_t ld_act, int ffn,
                      void* h, int hidden, const void* norm_w, float eps,
                      const void* qkv_w, const float* qkv_s, const void* qkv_b,
                      void* q_out, int64_t ldq, const float* rope_cs, int max_pos,
                      void* k_cache, void* v_cache, const int32_t* block_tables,
                      int max_blocks_per_seq, const int32_t* seq_lens,
                      int pos_offset, int B, int q_len, int Hq, int Hkv, int D,
                      sd_stream_t stream);

/* ------------------------------------------------------------------------
 * Decoder model: a host-side descriptor over caller-owned device weights.
 * ---------------------------------------------------------------------- */

typedef enum { SD_ARCH_LLAMA = 0, SD_ARCH_OPT = 1 } sd_arch;

/* Sum of a bf16 buffer over the ranks of a tensor-parallel group, in place,
 * enqueued on `stream`.  sd_tp_allreduce (RCCL, below) has this signature; a
 * host may supply its own.                                                  */
typedef int (*sd_allreduce_fn)(void* buf, size_t n_bf16, void* ctx,
                               sd_stream_t stream);

typedef struct {
  const void* attn_norm_w;  /* [H] */
  const void* attn_norm_b;  /* [H] or NULL (OPT) */
  const void* qkv_w;        /* [(Hq+2Hkv)*D, H], tiled SD_TILE_ROTARY */
  const void* qkv_b;        /* or NULL */
  const void* o_w;          /* [H, Hq*D], tiled SD_TILE_PLAIN */
  const void* o_b;          /* or NULL */
  const void* mlp_norm_w;   /* [H] */
  const void* mlp_norm_b;   /* or NULL */
  const void* up_w;         /* LLAMA: [2F, H] tiled SD_TILE_GATED; OPT: [F, H] tiled PLAIN */
  const void* up_b;         /* or NULL */
  const void* down_w;       /* [H, F], tiled SD_TILE_PLAIN */
  const void* down_b;       /* or NULL */
  /* fp8 weights (sd_model_desc.weight_fp8): f32 scale per source row of the
   * matrix (qkv: (Hq+2Hkv)*D, o: H, up: 2F or F, down: H); else NULL.        */
  const float* qkv_s;
  const float* o_s;
  const float* up_s;
  const float* down_s;
  /* down_w's tiles ordered k-step first (sd_draft_mlp), or NULL: needed only by a
   * model whose passes may run with sd_forward_args.draft_fused.              */
  const void* down_w_km;
} sd_layer_weights;

typedef struct {
  int arch;        /* sd_arch */
  int n_layers, hidden, n_heads, n_kv_heads, head_dim, ffn, vocab;
  int max_pos;     /* rows of rope_cs / pos_table */
  int pos_bias;    /* OPT learned positions start at row 2 */
  float norm_eps;
  float attn_scale;        /* usually head_dim^-0.5 */
  const void* embed;       /* [vocab, H] bf16, row-major (tiled if embed_tiled) */
  int embed_tiled;
  const void* pos_table;   /* [max_pos + pos_bias, H] bf16 or NULL */
  const float* rope_cs;    /* [max_pos, D] f32 or NULL */
  const void* final_norm_w;
  const void* final_norm_b; /* or NULL */
  const void* lm_head;     /* [vocab, H] bf16 tiled SD_TILE_PLAIN (== embed when tied) */
  const sd_layer_weights* layers; /* n_layers entries (copied) */
  /* Tensor parallelism (allreduce == NULL: none).  n_heads, n_kv_heads and ffn
   * are this rank's shares: qkv_w / up_w hold the rows of the rank's heads /
   * ffn columns, o_w / down_w the matching input columns (hidden stays whole,
   * embedding, norms and LM head are replicated).  The two row-parallel
   * projections of a layer are summed over the ranks with `allreduce`; rank 0
   * alone adds the residual and the biases.                                  */
  int tp_rank;
  sd_allreduce_fn allreduce;
  void* allreduce_ctx;
  /* weight_fp8 != 0: every layer matrix and the LM head are fp8 tiles
   * (sd_tile_weights_fp8) with per-row scales (layers[i].*_s, lm_head_s); the
   * embedding table, norms and biases stay bf16.                             */
  int weight_fp8;
  const float* lm_head_s;
  /* Vocabulary-parallel LM head under tensor parallelism: lm_head holds lm_vocab
   * rows (this rank's slice of the vocabulary) and logits are [rows, lm_vocab];
   * 0 = the whole vocabulary.  The ranks' slice maxima are exchanged with
   * sd_tp_allgather and merged by sd_argmax_combine (greedy), instead of
   * gathering logits.                                                       */
  int lm_vocab;
  /* Multimodal rotary sections (Qwen2-VL: 16, 24 of D/2 = 64); 0, 0 = ordinary
   * rotary embedding.  With sections set, sd_forward_args.rope_pos / rope_delta
   * give the rotary positions.                                               */
  int mrope_section_t, mrope_section_h;
} sd_model_desc;

typedef struct sd_model sd_model;

sd_model* sd_model_create(const sd_model_desc* desc);
void sd_model_destroy(sd_model* m);

typedef struct {
  void* k_cache; /* [n_layers][num_pages][Hkv][page] */
  void* v_cache;
  int64_t layer_stride_bytes; /* bytes between consecutive layers' caches */
  const int32_t* block_tables; /* [B, max_blocks_per_seq] i32 (device) */
  int max_blocks_per_seq;
} sd_kv_cache;

typedef struct {
  int B, q_len;              /* q_len tokens per sequence, B*q_len <= 16 (256 if wide) */
  const int32_t* token_ids;  /* token t of seq b at [b*tok_stride + tok_offset + t] */
  int64_t tok_stride, tok_offset;
  const int32_t* seq_lens;   /* [B] device; first new position = seq_lens[b] + pos_offset */
  int pos_offset;
  int max_ctx;               /* host bound on seq_lens[b] + pos_offset + q_len */
  int split_tokens;          /* attention keys per workgroup */
  float* logits;             /* [B*q_len, vocab] f32 out, or NULL to skip the head */
  int wide;                  /* 0: B*q_len <= 16 (sd_linear family); 1: up to 256 rows
                                through sd_linear_wide / sd_qkv_rope_wide.  The two families
                                round differently: a caller keeps one for all its passes.
                                sd_model_workspace_bytes plans for it when max_rows > 16. */
  /* Multimodal inputs (all may be NULL).  input_embeds: [B*q_len, hidden] bf16
   * rows that replace the token-embedding lookup (image embeddings already
   * merged in by the caller).  rope_pos / rope_delta: as sd_rope_positions;
   * used only by a model with mrope sections.                               */
  const void* input_embeds;
  const int32_t* rope_pos;
  const int32_t* rope_delta;
  /* Token tree pass (NULL: a chain).  tree_mask [B*q_len] u64 as for
   * sd_paged_attention_tree; needs a rotary model without mrope sections, q_len *
   * (Hq/Hkv) <= 64, and rope_pos = the rows' rotary positions ([B*q_len] i32; a model
   * without sections reads one component only).                              */
  const uint64_t* tree_mask;
  /* NULL, or a prepared sd_chain state (sd_chain_state_init, this model's hidden
   * size): passes whose shapes sd_chain_supported accepts (narrow family, RMSNorm
   * model, no tensor parallelism, no tree, no multimodal positions) then run
   * every layer as attention + sd_chain_o_up + sd_chain_down_qkv, three launches
   * instead of five, with the same bits.                                      */
  void* chain_state;
  /* != 0: a pass whose shape sd_draft_layer_supported accepts (narrow family, RMSNorm
   * model with bf16 weights and down_w_km in every layer, no tensor parallelism, no
   * tree, no multimodal positions) runs every layer as three launches
   * (sd_norm_qkv_rope_adds, sd_draft_attn_oproj, sd_draft_mlp).  Logits differ from
   * the five-launch pass in the last bits: for draft models only.              */
  int draft_fused;
  /* > 0: on the five-launch path of the narrow family, the idle compute units of every
   * layer's attention launch read ahead up to this many MiB of the weights that the next
   * launches stream (output projection, then gate/up), into the L2 of the XCD that will
   * ask for them.  Results are unchanged (nothing is written); needs a KV-head count that
   * is a multiple of 8, else ignored.                                          */
  int read_ahead_mb;
  /* != 0: passes of the wide family over a model whose matrices are all fp8 (and whose
   * hidden, query and ffn widths are multiples of 128) quantise their activations to fp8 per
   * token row (sd_row_scales_fp8 + the _a8 linear layers, LM head included).  Changes the
   * rounding of those passes: choose it for all wide passes of a model or none.          */
  int wide_a8;
  /* Both non-NULL: the pass leaves the slice maxima of every row's logits here
   * ([B*q_len, SD_ARGMAX_PARTS] each, for sd_propose_from_partials / sd_verify_from_partials
   * / sd_argmax_from_partials) — straight from the LM head where its shape allows
   * (sd_norm_linear_argmax; `logits` is then not written and may be NULL), else through
   * `logits` (which must then be given) and sd_argmax_partial.                    */
  float* argmax_val;
  int32_t* argmax_idx;
} sd_forward_args;

size_t sd_model_workspace_bytes(const sd_model* m, int max_rows, int max_ctx,
                                int split_tokens);
/* One forward pass of B*q_len token rows: embeds, runs every layer (appending
 * K/V to the cache), final norm and LM head.  Enqueues kernels only.        */
int sd_model_forward(sd_model* m, const sd_kv_cache* kv,
                     const sd_forward_args* a, void* workspace,
                     size_t workspace_bytes, sd_stream_t stream);

/* ------------------------------------------------------------------------
 * One-shot collectives over peer-mapped device memory (xGMI inside a node) for the small
 * messages of a tensor-parallel verify pass (rows x hidden bf16: 96 KB at 70B / 6 rows), where
 * a ring's 2(N-1) dependent steps cost more than every rank reading every peer once.
 * (Upstream: vLLM's custom_all_reduce for small messages; not in the reference tree.)
 *
 * Every rank creates one exchange buffer (sd_ipc_buffer_create: fine-grained device memory +
 * its 64-byte IPC handle), the host code passes the handles around (any channel; the Python
 * host uses torch.distributed), every rank maps its peers' buffers (sd_ipc_buffer_open) and
 * builds a communicator over the world_size pointers (own buffer at index rank).  A call
 * publishes the message in the rank's buffer, raises per-slice flags, waits for the peers'
 * flags and combines the ranks' data IN RANK ORDER, so every rank ends with the same bits.
 * Calls are stream-ordered kernels only (capturable); call counters live in device memory.
 * Every rank must make the same sequence of calls.  A peer that does not arrive within ~20 s
 * makes the waiting workgroup give up (garbage result, sd_ipc_comm_errors > 0), never a hang.
 * Limits: <= 8 ranks, messages <= max_message_bytes <= 16 MiB, multiples of 16 bytes.
 * sd_ipc_allreduce is an sd_allreduce_fn (ctx = the sd_ipc_comm*); sd_ipc_allgather has
 * sd_tp_allgather's meaning.                                                            */
#define SD_IPC_HANDLE_BYTES 64
typedef struct sd_ipc_comm sd_ipc_comm;
size_t sd_ipc_buffer_bytes(size_t max_message_bytes);
int sd_ipc_buffer_create(size_t bytes, void** ptr, void* handle_out);
int sd_ipc_buffer_open(const void* handle, void** ptr);
int sd_ipc_buffer_close(void* ptr);   /* a peer's mapping */
int sd_ipc_buffer_destroy(void* ptr); /* the rank's own buffer */
sd_ipc_comm* sd_ipc_comm_create(int rank, int world_size, void* const* bufs,
                                size_t max_message_bytes);
void sd_ipc_comm_destroy(sd_ipc_comm* comm);
int sd_ipc_comm_errors(sd_ipc_comm* comm, int* n_timeouts); /* synchronises */
/* How long a wait for a peer may last in the calls enqueued from now on (0: back to the
 * default, ~20 s; calls already captured into a graph keep theirs).  Processes that have just
 * started may lag each other by more than the default on the host side (cold page cache,
 * first launches): give them a generous bound for a warm-up pass, then restore it.         */
int sd_ipc_comm_set_timeout_ms(sd_ipc_comm* comm, int milliseconds);
int sd_ipc_allreduce(void* buf, size_t n_bf16, void* ctx, sd_stream_t stream);
int sd_ipc_allgather(const void* send, void* recv, size_t bytes_per_rank, void* ctx,
                     sd_stream_t stream);

/* ------------------------------------------------------------------------
 * Tensor-parallel group over RCCL (one process per GPU; xGMI inside a node).
 * Replaces the NCCL tensor_model_parallel_all_reduce of the target model's
 * row-parallel layers.  Rank 0 makes an id, the host spreads its
 * SD_TP_ID_BYTES to the other ranks by any means, every rank then creates its
 * communicator (collective call).
 * ---------------------------------------------------------------------- */
#define SD_TP_ID_BYTES 128
typedef struct sd_tp_comm sd_tp_comm;
int sd_tp_unique_id(void* id_out);
sd_tp_comm* sd_tp_comm_create(const void* id, int rank, int world_size);
void sd_tp_comm_destroy(sd_tp_comm* comm);
/* An sd_allreduce_fn: ctx is the sd_tp_comm*.                                */
int sd_tp_allreduce(void* buf, size_t n_bf16, void* ctx, sd_stream_t stream);
/* recv[r * bytes_per_rank ...] = rank r's send buffer, on every rank (the
 * exchange of the vocabulary-parallel head: slice maxima, or logits when a
 * full row is needed for sampling).                                          */
int sd_tp_allgather(const void* send, void* recv, size_t bytes_per_rank, void* ctx,
                    sd_stream_t stream);

/* ------------------------------------------------------------------------
 * Speculative step glue that must stay on the device (replaces the tensor
 * bookkeeping of spec_decode.spec_decode_worker._verify_tokens /
 * _create_output_sampler_list and top1_proposer's proposal assembly).
 * Per-sequence device state:
 *   seq_lens [B] i32   tokens in the sequence so far (n)
 *   tok_buf  [B, k+2] i32 = {t[n-2], t[n-1], d_1 .. d_k}
 * ---------------------------------------------------------------------- */

/* Writes proposal ids (i64 -> i32) into tok_buf[:, 2 + step].                */
int sd_store_proposal(const int64_t* ids, int32_t* tok_buf, int k, int step, int B,
                      sd_stream_t stream);

/* Writes externally made proposals [B, k] i32 (an n-gram / prompt-lookup
 * proposer, which needs no draft model) into tok_buf[:, 2:2+k]; entries past a
 * sequence's proposal length are ignored by the verification (proposal_lens). */
int sd_set_proposals(const int32_t* proposals, int32_t* tok_buf, int B, int k,
                     sd_stream_t stream);

/* Gathers tok_buf[:, 2:2+k] as i64 draft_ids [B, k].                        */
int sd_gather_proposals(const int32_t* tok_buf, int64_t* draft_ids, int B,
                        int k, sd_stream_t stream);

/* Applies out_ids [B, k+1] (i64, -1 padded) to the state: appends the emitted
 * tokens to gen_tokens[b, gen_counts[b]..] (bounded by gen_capacity),
 * advances seq_lens and gen_counts, and rebuilds tok_buf[:, 0:2].
 * step_out (optional) [B, k+3] i32 receives the step record
 * {n_emitted, seq_len after the step, token_0 .. token_k (-1 padded)}; it may
 * be pinned host memory, so that a host needs one stream synchronisation and
 * no copy to return a step's tokens (the V1 worker's sampled_token_ids).    */
int sd_advance_state(const int64_t* out_ids, int32_t* seq_lens, int32_t* tok_buf,
                     int32_t* gen_tokens, int64_t gen_stride, int gen_capacity,
                     int32_t* gen_counts, int32_t* step_out, int B, int k,
                     sd_stream_t stream);

/* Greedy steps without a logits round trip per helper.  The row argmax is
 * split in two: sd_argmax_partial scans each row in SD_ARGMAX_PARTS slices
 * (part_val / part_idx [R, SD_ARGMAX_PARTS]); the consumers below finish it
 * (smallest index among equal maxima, as sd_argmax_rows) and go straight on:
 *  - sd_propose_from_partials: rows row_offset + b*row_step are the draft's
 *    last-token logits; stores proposal `step` like sd_store_proposal
 *    (replaces top1_proposer's sampler + token append of one draft step);
 *  - sd_verify_from_partials: rows b*(k+1)+i are the target's k+1 rows;
 *    greedy acceptance against tok_buf[:, 2:2+k] (as sd_greedy_verify;
 *    out_ids and counters optional), then the state update and step record
 *    of sd_advance_state.  k == 0 is plain autoregressive decoding; k <= 15.  */
#define SD_ARGMAX_PARTS 512
/* The LM head of a greedy pass without the logits round trip: sd_norm_linear with the F32
 * epilogue replaced by the first stage of the row argmax.  Every workgroup of the launch (at
 * most SD_ARGMAX_PARTS) leaves the largest of the f32 logits it computed for token row m, and
 * its column, in part_val / part_idx[m][workgroup]; slots no workgroup owns get -inf.  The
 * logits are the F32 epilogue's bit for bit and ties go to the lowest column, so the consumers
 * below finish the same argmax as after sd_argmax_partial (upstream samples from the logits
 * tensor: vllm/model_executor/layers/sampler.py; its greedy branch is an argmax over the row).
 * Shapes as sd_norm_linear (M <= 16 rows that fit LDS: sd_model_forward falls back to logits +
 * sd_argmax_partial otherwise).                                                */
int sd_norm_linear_argmax(const void* W, const float* w_scale, const void* x, int64_t ldx,
                          const void* norm_w, const void* norm_b, float eps, int layer_norm,
                          float* part_val, int32_t* part_idx, int M, int N, int K,
                          sd_stream_t stream);
int sd_argmax_partial(const float* logits, int64_t ld, float* part_val,
                      int32_t* part_idx, int R, int V, sd_stream_t stream);
/* Distributed argmax of a vocabulary-parallel head.  part_val / part_idx hold
 * n_ranks blocks of [R, SD_ARGMAX_PARTS] slice maxima (sd_argmax_partial run on
 * each rank's [R, shard_vocab] logits, indices local to the shard, blocks in rank
 * order as sd_tp_allgather leaves them, rank_stride elements apart: values and
 * indices may be packed into one gathered buffer); out_* receive [R, SD_ARGMAX_PARTS]
 * maxima over the whole vocabulary with global indices (ties: lowest index),
 * which sd_verify_from_partials / sd_propose_from_partials /
 * sd_argmax_from_partials finish as usual.                                   */
int sd_argmax_combine(const float* part_val, const int32_t* part_idx, int n_ranks,
                      int R, int shard_vocab, int64_t rank_stride, float* out_val,
                      int32_t* out_idx, sd_stream_t stream);
/* ids[r] = first argmax of row r from its slice maxima.                      */
int sd_argmax_from_partials(const float* part_val, const int32_t* part_idx,
                            int64_t* ids, int R, sd_stream_t stream);

int sd_propose_from_partials(const float* part_val, const int32_t* part_idx,
                             int row_offset, int row_step, int32_t* tok_buf,
                             int k, int step, int B, sd_stream_t stream);
int sd_verify_from_partials(const float* part_val, const int32_t* part_idx,
                            int32_t* tok_buf, const int32_t* proposal_lens,
                            int64_t* out_ids, int64_t* counters,
                            int32_t* seq_lens, int32_t* gen_tokens,
                            int64_t gen_stride, int gen_capacity,
                            int32_t* gen_counts, int32_t* step_out, int B,
                            int k, sd_stream_t stream);

/* ------------------------------------------------------------------------
 * Benchmark instrumentation (not part of the product path; no counterpart upstream).
 * With randomly initialised weights a draft model never agrees with its target, so a
 * benchmark imposes the acceptance: after the draft's k passes this overwrites the proposals
 * in tok_buf[:, 2:2+k] so that exactly pattern[step_counter[b] % pattern_len] of them are
 * accepted (the first a become truth[b, seq_lens[b] + i], proposal a a token that is not the
 * true one), then step_counter[b] += 1.  truth [B, truth_stride] i32 holds the target's own
 * greedy stream by absolute position; pattern is a HOST array of pattern_len <=
 * SD_DEBUG_PATTERN_MAX entries; step_counter [B] i32 (device) makes replays of a captured
 * step walk through the pattern.  Deterministic: a run's accept-len is the pattern's mean. */
#define SD_DEBUG_PATTERN_MAX 64
int sd_debug_force_proposals(int32_t* tok_buf, int k, const int32_t* seq_lens,
                             const int32_t* truth, int64_t truth_stride,
                             const uint8_t* pattern, int pattern_len,
                             int32_t* step_counter, int vocab, int B, sd_stream_t stream);

/* ------------------------------------------------------------------------
 * Stream capture helpers (one spec step = one hipGraph launch).
 * ---------------------------------------------------------------------- */
typedef struct sd_graph sd_graph;
int sd_graph_begin(sd_stream_t stream);
int sd_graph_end(sd_stream_t stream, sd_graph** out);
int sd_graph_launch(sd_graph* g, sd_stream_t stream);
void sd_graph_destroy(sd_graph* g);

#ifdef __cplusplus
}
#endif
#endif /* SD_AMD_H */
