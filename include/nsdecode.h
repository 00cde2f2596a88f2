/*
 * nsdecode.h - C ABI of libnsdecode.so: the NeuroSymbolic (NS) beam scorer of
 * Silent-Speech as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference implements this path in Python only; there is no native
 * interface to bind.  Each entry point therefore cites the reference code
 * whose work it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every pointer marked "device" is caller-owned HIP device memory
 *     (e.g. torch.Tensor.data_ptr()); the library allocates device memory
 *     only inside nsd_tables_create and frees it in nsd_tables_destroy.
 *     One exception, host side only: nsd_decode_synth called with step_ms or
 *     check_every > 0 keeps a pool of HIP events per device and one pinned
 *     host word between calls; nsd_shutdown() releases them;
 *   - kernels are enqueued on the caller's stream and never synchronise;
 *   - return value: 0 = NSD_OK, < 0 = argument error (NSD_E_*), > 0 = a
 *     hipError_t value; nsd_strerror() describes all three;
 *   - vocabulary layout is the reference TextTransform
 *     (scripts/data_utils.py:44-50): ids 0-25 a-z, 26-35 0-9, 36 space,
 *     37 BOS, 38 EOS, 39 PAD; V = 40.
 */
#ifndef NSDECODE_H
#define NSDECODE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSD_VERSION        103
#define NSD_V              40
#define NSD_NBASE          37
#define NSD_SPACE          36
#define NSD_BOS            37
#define NSD_EOS            38
#define NSD_PAD            39
#define NSD_MAX_BEAM       64
#define NSD_MAX_STEPS      128
#define NSD_MAX_ORDER      10
#define NSD_MAX_LDS_TRIE_NODES 1024
#define NSD_MAX_HYPS       (1 << 26)  /* N * B of one nsd_step launch (the kernel indexes hypotheses and logits rows in 32 bits) */
#define NSD_NODE_DEAD      0xFFFFFFFFu
#define NSD_KEY_EMPTY      0xFFFFFFFFFFFFFFFFull
#define NSD_TOKEN_CARRY    0xFFu     /* "no token this step": a finished hypothesis carried over */

/* error codes (negative) */
#define NSD_OK              0
#define NSD_E_NULL         -1
#define NSD_E_BEAM         -2   /* beam outside [1, NSD_MAX_BEAM] */
#define NSD_E_VOCAB        -3   /* V != NSD_V */
#define NSD_E_ORDER        -4   /* n-gram order outside [0, NSD_MAX_ORDER] */
#define NSD_E_TABLE        -5   /* malformed table (capacity not a power of two, empty trie, a log-prob that is not finite, ...) */
#define NSD_E_VARIANT      -6
#define NSD_E_SIZE         -7   /* negative / oversized count */
#define NSD_E_NOALLOC_SIZE -8   /* batch larger than the built-in scratch of nsd_decode_synth_noalloc */
#define NSD_E_ALIGN        -9   /* a row pointer of the AR-step kernels is not a multiple of 16 bytes */

/* beam flags (low byte of NsdBeamState.meta) */
#define NSD_F_FINISHED      1u  /* last token is EOS */
#define NSD_F_DEAD          2u  /* left the lexicon: neural top-k fall-back from now on */
#define NSD_F_PENDING_SPACE 4u  /* surface ends with a space not yet part of the LM sum */

/* per-utterance word NsdBeamState.utt: nbeams | DONE | ERR | nsteps << 16 */
#define NSD_U_NBEAMS_MASK   0xFFu
#define NSD_U_DONE          0x100u  /* every hypothesis finished (ns_utils.py:274-275) */
#define NSD_U_ERR_TOPK      0x200u  /* fall-back asked for k > V in variant I/E (torch.topk raises) */
#define NSD_U_ERR_INTERNAL  0x400u  /* a step's candidate buffer would have overflowed: a defect of the library, never
                                       expected; the utterance's result is not to be trusted */
#define NSD_U_STEPS_SHIFT   16

/* ---- replicated symbolic tables ------------------------------------ */

/* One trie node.  Replaces the dict-of-dicts Trie (scripts/ns_utils.py:31-56).
 * Nodes are numbered breadth-first, root = 0; the children of a node are
 * stored contiguously in ascending token order starting at child_base, so
 * child(tok) = child_base + popcount(mask & ((1 << tok) - 1)).
 *   mask bits 0..35 : child exists for that base character
 *   mask bit  36    : space may follow (is_word, or a literal space edge)
 *   mask bit  37    : the prefix is a lexicon word (is_word)            */
typedef struct NsdTrieNode {
    uint64_t mask;
    uint32_t child_base;
    uint32_t reserved;
} NsdTrieNode;

typedef struct NsdTrieCSR {            /* host memory */
    const NsdTrieNode *nodes;
    uint32_t n_nodes;
} NsdTrieCSR;

/* Open-addressed tables of host-precomputed fp64 log-probs.
 * Replaces CharNgramLM.counts/.context and the math.log per n-gram of
 * CharNgramLM.logp (scripts/ns_utils.py:138-150).
 * Keys pack symbol codes 6 bits each, oldest symbol highest:
 *   code = token id + 1 for the 37 base characters, 38 for '^', 39 for '$'.
 * Empty slots hold NSD_KEY_EMPTY.
 *
 * n-gram table: linear probing from slot nsd_hash64(key) & (cap - 1) (whole-string scoring only). */
typedef struct NsdGramSlot {           /* key: `order` symbols */
    uint64_t key;
    double   logp;                     /* log((c+1)/(C+|V|+1)) */
} NsdGramSlot;

/* Context table: two-choice (cuckoo) placement.  With h = nsd_hash64(key), a key sits either in slot
 * h & (cap - 1) or in slot (h >> 32) & (cap - 1), never anywhere else: a look-up is two independent 32-byte
 * reads issued together, for a key that is present as for one that is not (the decoder asks mostly for
 * contexts the corpus never showed, and a wave waits for its slowest lane). */
typedef struct NsdCtxSlot {            /* key: `order-1` symbols */
    uint64_t key;
    double   term;                     /* log((c(ctx+'$')+1)/(C+|V|+1)): the terminal n-gram */
    uint64_t ctx_id;                   /* index of this context in ctx_recs */
    double   backoff;                  /* log(1/(C+|V|+1)): value of an unseen n-gram in a seen context */
} NsdCtxSlot;

/* The same counts as a context automaton, which is what nsd_step walks: one
 * record per seen context, its seen n-grams stored contiguously in ascending
 * symbol-code order starting at child_base, so
 *   rec(ctx, sym) = gram_recs[child_base + popcount(symmask & ((1 << sym) - 1))]
 * when bit `sym` of symmask is set.  A step then costs one direct load per
 * candidate instead of dependent hash probes; the hash tables above remain
 * for unseen transitions and for whole-string scoring (nsd_lm_logp). */
typedef struct NsdCtxRec {
    double   backoff;
    double   term;
    uint64_t symmask;                  /* bit c set: n-gram (ctx, symbol code c) was seen */
    uint32_t child_base;
    uint32_t reserved;
} NsdCtxRec;

typedef struct NsdGramRec {            /* 24 bytes */
    double   logp;                     /* log((c+1)/(C+|V|+1)) of this n-gram */
    double   term_next;                /* term of the context reached by appending the symbol */
    uint32_t next_ctx;                 /* id of that context, NSD_CTX_NONE if it was never seen */
    uint32_t reserved;
} NsdGramRec;

#define NSD_CTX_NONE 0xFFFFFFFFu

typedef struct NsdNgramHash {          /* host memory */
    int32_t  order;                    /* 0 = no language model */
    uint32_t init_ctx;                 /* id of the context of the empty surface ('^' for order 2, the empty
                                          context for order 1), NSD_CTX_NONE for longer orders */
    uint64_t gram_cap;                 /* power of two */
    uint64_t ctx_cap;                  /* power of two, at most 2^32 */
    const NsdGramSlot *grams;
    const NsdCtxSlot  *ctxs;
    double   unseen;                   /* log(1/(|V|+1)): any n-gram in an unseen context */
    uint32_t n_ctx_recs;
    uint32_t n_gram_recs;
    const NsdCtxRec  *ctx_recs;
    const NsdGramRec *gram_recs;
} NsdNgramHash;

typedef struct NsdTables NsdTables;    /* opaque, device resident, immutable after create */

/* ---- decode parameters --------------------------------------------- */
#define NSD_VARIANT_U 0   /* scripts/ns_utils.py:194-278   EOS offered at every step        */
#define NSD_VARIANT_I 1   /* inference_emg_llm.py:222-300  no EOS option, top-k unclamped   */
#define NSD_VARIANT_E 2   /* eval_ar.py:214-317            word bonus, boundary EOS, length norm */

typedef struct NsdParams {
    int32_t variant;
    int32_t use_lm;                    /* lm is not None */
    int32_t allow_eos_anytime;         /* ns_utils.py:203 (variant U only) */
    int32_t min_len_for_eos;           /* eval_ar.py:226 (variant E only) */
    int32_t lanes_per_utt;             /* 0 = choose from N and B; else 16, 32 or 64 (>= B): lanes of a wave
                                          that work on one utterance (a tuning knob, results do not depend on it) */
    int32_t lds_trie_nodes;            /* 0 = trie records are read from memory (default).  n > 0: every workgroup first
                                          copies the first n node records (breadth-first: the upper trie levels) to LDS
                                          and hypotheses at those nodes read them there (a tuning knob, measured not to
                                          pay; results do not depend on it); at most NSD_MAX_LDS_TRIE_NODES */
    double  beta;                      /* lm_weight   */
    double  gamma;                     /* oov_penalty */
    double  kappa;                     /* word_bonus  (variant E) */
    double  eos_len_nudge;             /* 0.002, eval_ar.py:293 */
    /* variant E pruning key = score / len_denom[L], L = generated tokens,
     * len_denom[L] = ((5+L)**alpha)/(6**alpha) computed by the host
     * (eval_ar.py:237-241); entry 0 is unused. */
    double  len_denom[NSD_MAX_STEPS + 4];
} NsdParams;

/* ---- beam state (structure of device arrays) ------------------------ */
typedef struct NsdBeamState {
    double   *score;     /* device [N*B] accumulated raw score (fp64, as the reference's Python floats) */
    double   *lmsum;     /* device [N*B] running sum of interior n-gram log-probs of the stripped surface */
    double   *lmsum_e;   /* device [N*B] the same including a pending trailing space */
    uint64_t *ctx;       /* device [N*B] last order-1 symbols of '^' + stripped surface */
    uint64_t *hist;      /* device [N*B] rolling hash of the token history (consumed by batched step functions) */
    uint32_t *ctx_id;    /* device [N*B] automaton id of ctx, NSD_CTX_NONE if unseen or still shorter than order-1 */
    uint32_t *ctxe_id;   /* device [N*B] automaton id of ctx followed by the pending space (= ctx_id without one) */
    uint32_t *node;      /* device [N*B] trie node of the current word prefix, or NSD_NODE_DEAD */
    uint32_t *meta;      /* device [N*B] flags | generated length << 8 */
    uint32_t *utt;       /* device [N]   see NSD_U_* */
    uint32_t *done_count;/* device [1]   number of utterances that reached NSD_U_DONE; a step adds the
                                         utterances that finish in it to the word of its `out` state, so two
                                         states used as a ping-pong pair must share this one word */
} NsdBeamState;

/* ---- entry points ---------------------------------------------------- */

int nsd_version(void);
const char *nsd_strerror(int code);

/* Release what nsd_decode_synth keeps between calls (HIP events, one pinned host word).  Optional; nothing
 * else in the library holds state outside NsdTables.  Not to be called while a decode is running. */
int nsd_shutdown(void);

/* hash used for both tables (exported so that table builders can be checked) */
uint64_t nsd_hash64(uint64_t key);

/* Upload the tables.  `lm` may be NULL or have order 0 (lm=None).
 * Replaces build_trie_from_words / CharNgramLM.fit as far as the decoder
 * reads them (scripts/ns_utils.py:59-65, 99-109). */
int nsd_tables_create(const NsdTrieCSR *trie, const NsdNgramHash *lm, NsdTables **out);
int nsd_tables_destroy(NsdTables *t);
int nsd_tables_order(const NsdTables *t);

/* beams = [(0.0, [BOS], "")] for N utterances (scripts/ns_utils.py:214). */
int nsd_state_init(const NsdTables *t, const NsdBeamState *st, int32_t N, int32_t B, void *stream);

/* One decode step for N utterances x B beams: the body of the loop at
 * scripts/ns_utils.py:218-275 (and its copies, see NSD_VARIANT_*).
 *   logits   device [N*B, V] fp32; row u*B+r is read only if hypothesis r of
 *            utterance u is live (r < nbeams and not finished)
 *   in/out   beam state before/after; may be the same object
 *   backptr  device [N*B] u8: rank of the parent hypothesis   (this step's slice)
 *   token    device [N*B] u8: appended token or NSD_TOKEN_CARRY (this step's slice)
 *   mask_dbg device [N*B] u64 or NULL: lexicon mask of each live parent
 * Utterances already NSD_U_DONE are left untouched.  N * B above NSD_MAX_HYPS
 * is refused with NSD_E_SIZE before anything is enqueued (split the batch).
 * The launcher picks among several forms of the kernel (lanes per utterance, compile-time beam / n-gram order for
 * beams 8, 16, 32 with a 5-gram model, 4 lanes per logits row for the copy without an EOS option, a form for dense
 * lexicons chosen from the trie's fan-out); all give the same bits.  Two environment variables restrict the choice
 * for debugging, read once per process: NSD_STEP_MODE (0 = run-time form only, 1 = no 4-lanes-per-row form) and
 * NSD_STEP_DENSE (0 / 1 = never / always the dense-lexicon form). */
int nsd_step(const NsdTables *t, const NsdParams *p, const float *logits,
             const NsdBeamState *in, const NsdBeamState *out,
             uint8_t *backptr, uint8_t *token, uint64_t *mask_dbg,
             int32_t N, int32_t B, int32_t V, void *stream);

/* Follow the back-pointers of the best hypothesis (rank 0: the first maximal
 * element, scripts/ns_utils.py:277-278).
 *   backptr/token device [T, N*B] as written by nsd_step, step-major
 *   out_ids  device [N, T+1] u8, starts with BOS;  out_len device [N] i32
 *   out_score device [N] f64 or NULL */
int nsd_backtrace(const NsdBeamState *st, const uint8_t *backptr, const uint8_t *token,
                  int32_t T, int32_t N, int32_t B,
                  uint8_t *out_ids, int32_t *out_len, double *out_score, void *stream);

/* Token histories of all current hypotheses, for batched step functions that
 * need ids (inference_emg_llm.py:427-429 builds them per beam on the host).
 *   out_ids device [N*B, T+1] u8 (BOS first), out_len device [N*B] i32 (0 = no such hypothesis) */
int nsd_gather_ids(const NsdBeamState *st, const uint8_t *backptr, const uint8_t *token,
                   int32_t T, int32_t N, int32_t B, uint8_t *out_ids, int32_t *out_len, void *stream);

/* CTC greedy proposal: frame argmax, collapse repeats, drop PAD/BOS/EOS
 * (inference_emg_llm.py:445-455, eval_ar.py:558-568).
 *   logits device [N, Tp, V] fp32; out_ids device [N, Tp] u8; out_len device [N] i32 */
int nsd_ctc_greedy(const float *logits, int32_t N, int32_t Tp, int32_t V,
                   uint8_t *out_ids, int32_t *out_len, void *stream);

/* CharNgramLM.logp of N strings (scripts/ns_utils.py:138-150), used by the
 * LM rerank (inference_emg_llm.py:461-464).
 *   text device [N, stride] u8 base-character ids (0..36), len device [N] i32
 *   out  device [N] f64 */
int nsd_lm_logp(const NsdTables *t, const uint8_t *text, const int32_t *len,
                int32_t N, int32_t stride, double *out, void *stream);

/* LM rerank of the AR hypothesis against the CTC proposal (inference_emg_llm.py:457-464, before
 * postprocess): both id sequences are turned into text as ids_to_text does (skip BOS, stop at EOS, keep
 * base characters, strip), scored with lm.logp(s) - 0.002 * len(s), and the maximum is taken with the AR
 * hypothesis winning ties.
 *   ar_ids  device [N, ar_stride] u8 starting with BOS, ar_len device [N] i32
 *   ctc_ids device [N, ctc_stride] u8 collapsed CTC ids (no BOS/EOS), ctc_len device [N] i32
 *   choice  device [N] u8: 0 = AR, 1 = CTC;  scores device [N, 2] f64 or NULL */
int nsd_rerank_lm(const NsdTables *t, const uint8_t *ar_ids, const int32_t *ar_len, int32_t ar_stride,
                  const uint8_t *ctc_ids, const int32_t *ctc_len, int32_t ctc_stride,
                  int32_t N, uint8_t *choice, double *scores, void *stream);

/* Teacher-forced sum of token log-probs (eval_ar.py:574-589):
 *   out[u] = sum_i log_softmax(logits[u, i, :])[target[u, i]], i < len[u]
 *   logits device [N, L, V] fp32; target device [N, L] u8; len device [N] i32 */
int nsd_rerank_gather(const float *logits, const uint8_t *target, const int32_t *len,
                      int32_t N, int32_t L, int32_t V, double *out, void *stream);

/* Which copy of CharNgramLM.cond_dist forms the context (they differ only at order 1):
 *   U  scripts/ns_utils.py:111-136  last order-1 characters, left-padded with '^'
 *   T  train_emg_llm.py:88-107      one '^' in front when fewer than order-1 characters are there;
 *                                   order 1 keeps the whole context, so only the empty one matches */
#define NSD_PRIOR_U 0
#define NSD_PRIOR_T 1

/* Next-character prior of the n-gram tables: add-one smoothed P(ch | context) over the 37 base
 * characters as fp32, renormalised (CharNgramLM.cond_dist, see above).
 *   ctx_ids device [N, stride] u8 base-character ids of the context (0..36; any other value is a
 *           character outside the alphabet, which no count matches), ctx_len device [N] i32
 *   probs   device [N, 37] fp32 */
int nsd_cond_dist(const NsdTables *t, const uint8_t *ctx_ids, const int32_t *ctx_len, int32_t N, int32_t stride,
                  int32_t form, float *probs, void *stream);

/* The LM regulariser of training (train_emg_llm.py:254-316): mean over the positions whose target is
 * not -100 of CE(prior || softmax(logits)), the prior being nsd_cond_dist of the row's earlier
 * base-character targets, spread over the model vocabulary with 1e-8 added per entry and renormalised.
 *   logits  device [B, L, V] fp32;  out_ids device [B, L] i64 targets, -100 = padding
 *   pos_loss device [B*L] fp32: per-position cross entropy (0 at skipped positions)
 *   stats    device [3] f64: sum of pos_loss, number of valid positions, mean (0 when there are none);
 *            reduced in a fixed order, so repeated calls give the same bits */
int nsd_lm_prior_ce(const NsdTables *t, const float *logits, const int64_t *out_ids, int32_t B, int32_t L, int32_t V,
                    int32_t form, float *pos_loss, double *stats, void *stream);

/* Gradient of that mean with respect to the logits (what autograd derives from
 * train_emg_llm.py:276,300-303): grad_out * (softmax * sum(prior) - prior) / count at valid positions,
 * 0 elsewhere.  stats is the array nsd_lm_prior_ce filled for the same inputs; grad_out device [1] fp32.
 *   grad_logits device [B, L, V] fp32 */
int nsd_lm_prior_ce_grad(const NsdTables *t, const float *logits, const int64_t *out_ids, int32_t B, int32_t L,
                         int32_t V, int32_t form, const double *stats, const float *grad_out, float *grad_logits,
                         void *stream);

/* Beam re-ordering of a key/value cache after a scorer step: hypothesis m of the new beam continues
 * hypothesis parent(m) of the old one, so the cached rows of the generated positions follow it.  The
 * reference keeps no cache (EMGTextModel.step re-runs the whole sequence, inference_emg_llm.py:83-88); this
 * is the copy a KV-cached batched step needs between two nsd_step launches.
 *   src, dst  device [L, M, H, T, row_bytes] (layers, hypotheses M = N*B, kv heads, positions, bytes of one
 *             head row); distinct buffers; row_bytes a multiple of 16
 *   backptr   device [M] u8: the scorer's back-pointers of the step (index inside the utterance's beam),
 *             parent(m) = (m / B) * B + backptr[m]
 *   t         positions 0..t-1 are copied: dst[l, m, h, i] = src[l, parent(m), h, i] */
int nsd_kv_reorder(const void *src, void *dst, const uint8_t *backptr, int32_t L, int32_t M, int32_t B, int32_t H,
                   int32_t T, int32_t t, int32_t row_bytes, void *stream);

/* ---- the non-GEMM part of a KV-cached batched step of the AR head ----------------------------------------
 * EMGTextModel.step (inference_emg_llm.py:83-88) re-runs the frozen LLM over the whole sequence for one
 * hypothesis at a time.  With a cache, a step feeds ONE token per hypothesis; the dense contractions stay
 * hipBLASLt, these kernels are what lies between them.  The generated keys / values are never moved: position s
 * of the generated part is a slab [M, kv_heads, hd] written once (M = N*B hypotheses), and hypothesis m reads,
 * at position s, the slab row of its ancestor at that step, found through the scorer's back-pointers.
 * Element type of activations, weights and caches: dtype 0 = fp32, 1 = bf16; arithmetic in fp32.
 * nsd_add_rmsnorm, nsd_swiglu, nsd_attn_step and nsd_attn_ctx move rows in 16-byte pieces: every array pointer
 * must be a multiple of 16 bytes (a tensor view with a storage offset may not be), else NSD_E_ALIGN. */

/* anc[s, m] (u8 [>= t, M]) = in-beam rank of the hypothesis that hypothesis m (of the beam after scorer step
 * t-1) descends from at generated position s < t.   backptr device [>= t, M] as written by nsd_step. */
int nsd_ancestors(const uint8_t *backptr, int32_t t, int32_t N, int32_t B, uint8_t *anc, void *stream);

/* x_out = x + delta (skipped when delta is NULL; x_out may be x); h = w * (x_out * rsqrt(mean(x_out^2) + eps)),
 * the normalised row rounded to the element type before the weight is applied (LlamaRMSNorm).  rows x hidden,
 * hidden a multiple of 4 (fp32) / 8 (bf16). */
int nsd_add_rmsnorm(const void *x, const void *delta, const void *w, float eps, int32_t rows, int32_t hidden,
                    int32_t dtype, void *x_out, void *h, void *stream);

/* Rotary embedding (rotate-half form) of the q and k heads of qkv [M, (heads + 2 kv_heads) * hd] with the
 * step's cos / sin (fp32 [hd], both halves hold the same angles); q to q_out [M, heads, hd], k and v to the
 * step's slabs k_out / v_out [M, kv_heads, hd] of the generated cache. */
int nsd_rope_kv_append(const void *qkv, const float *cos_t, const float *sin_t, int32_t M, int32_t heads,
                       int32_t kvh, int32_t hd, int32_t dtype, void *q_out, void *k_out, void *v_out, void *stream);

/* Attention of the new token of every hypothesis over [context ; generated positions 0..t]:
 *   q [M, heads, hd] (rotated); kctx / vctx [N, kv_heads, S, hd]: the utterance's context, shared by its B
 *   hypotheses; kgen / vgen [>= t+1, M, kv_heads, hd]: slabs of the generated positions (position t is the
 *   hypothesis' own slot, earlier ones go through anc); out [M, heads, hd].
 *   part_o / part_m / part_l: NULL, or the softmax state of the context part as nsd_attn_ctx left it
 *   (fp32 [M*heads, hd], [M*heads], [M*heads]); the context is then not read again (kctx / vctx may be NULL).
 * hd: a power of two, 4 (fp32) / 8 (bf16) to 256; heads / kv_heads in {1, 2, 3, 4, 8}. */
int nsd_attn_step(const void *q, const void *kctx, const void *vctx, const void *kgen, const void *vgen,
                  const uint8_t *anc, void *out, int32_t N, int32_t B, int32_t heads, int32_t kvh, int32_t hd,
                  int32_t S, int32_t t, float scale, int32_t dtype, const float *part_o, const float *part_m,
                  const float *part_l, void *stream);

/* The context part of that attention on the matrix cores (bf16 only): the B hypotheses of an utterance and the
 * query heads of a kv head share the context keys, so scores and weighted sums are small dense products
 * (v_mfma_f32_16x16x32_bf16).  kctx_pad [N, kv_heads, S32, hd]: context keys, rows S..S32-1 zero; vctx_t
 * [N, kv_heads, hd, S32]: context values transposed (keys contiguous), columns S..S32-1 zero; S32 a multiple of
 * 32, at most 256; hd 64 or 128.  The zero padding is a precondition the caller must establish: padded keys are
 * masked out of the softmax, but padded value columns are multiplied by a weight of 0, so an Inf or NaN left
 * there turns whole output rows into NaN.  Leaves per query row the unnormalised weighted value sum, the maximum and the
 * sum of the weights (part_o fp32 [M*heads, hd], part_m / part_l fp32 [M*heads]) for nsd_attn_step to continue. */
int nsd_attn_ctx(const void *q, const void *kctx_pad, const void *vctx_t, int32_t N, int32_t B, int32_t heads,
                 int32_t kvh, int32_t hd, int32_t S, int32_t S32, float scale, float *part_o, float *part_m,
                 float *part_l, void *stream);

/* out[r, i] = silu(gate_up[r, i]) * gate_up[r, inter + i]  (the SwiGLU between the two MLP contractions);
 * inter a multiple of 4 (fp32) / 8 (bf16). */
int nsd_swiglu(const void *gate_up, int64_t rows, int32_t inter, int32_t dtype, void *out, void *stream);

/* Row-wise deterministic fp32 log-softmax, the one nsd_step applies
 * (F.log_softmax at scripts/ns_utils.py:227).  x, out device [rows, V]. */
int nsd_log_softmax(const float *x, int32_t rows, int32_t V, float *out, void *stream);

/* Synthetic stand-in for the AR head: integer-hash logits with a planted
 * target, a function of (utterance seed, step, token-history hash) only.
 *   targets device [N, stride] i8 planted token ids, target_len device [N] i32
 *   logits  device [N*B, V] fp32 (rows of live hypotheses are written) */
int nsd_synth_logits(const NsdBeamState *st, uint64_t base_seed, int64_t utt0,
                     const int8_t *targets, const int32_t *target_len, int32_t stride,
                     float margin, float *logits, int32_t N, int32_t B, void *stream);

/* The whole batched decode loop of inference_emg_llm.py:416-435 for the
 * synthetic AR head, enqueued from native code: state init, then up to T
 * rounds of { nsd_synth_logits, nsd_step }, then nsd_backtrace.
 *   check_every > 0: every that many steps the done counter is copied to the
 *                    host and the loop stops once all N utterances are done
 *                    (one stream synchronisation per check); 0 = run all T
 *                    steps (finished utterances cost an early-exit wave each)
 *   step_ms   host [T] or NULL: when given, every nsd_step launch carries a pair
 *             of HIP events that take the kernel's own start and end timestamps
 *             and its duration in milliseconds is
 *             stored (entries of steps that did not run are set to -1); the
 *             call then ends with one stream synchronisation
 *   steps_run host [1] or NULL: number of steps enqueued
 * Buffers as in the single calls: logits device [N*B, V], backptr/token
 * device [T, N*B], out_ids device [N, T+1], out_len device [N], out_score
 * device [N] or NULL. */
int nsd_decode_synth(const NsdTables *t, const NsdParams *p, const NsdBeamState *st,
                     uint64_t base_seed, int64_t utt0,
                     const int8_t *targets, const int32_t *target_len, int32_t stride, float margin,
                     float *logits, uint8_t *backptr, uint8_t *token,
                     int32_t T, int32_t N, int32_t B,
                     uint8_t *out_ids, int32_t *out_len, double *out_score,
                     int32_t check_every, float *step_ms, int32_t *steps_run, void *stream);

/* nsd_decode_synth for callers that cannot allocate or hand over workspaces (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same decode, same results as nsd_decode_synth with check_every = 0 and no step_ms, but the beam state, the
 * logits rows and the back-pointer / token arrays live in a scratch array built into the library
 * (NSD_NOALLOC_SCRATCH_BYTES of device memory on every GPU the library runs on, held until the process ends);
 * the caller passes only the inputs (tables, parameters, targets) and the outputs.  The call allocates
 * nothing and does not synchronise.  The array's device address is looked up once per GPU by nsd_tables_create
 * (which a caller has run on that GPU anyway to get `t`, outside any capture); a call on a GPU where no tables
 * were ever created looks it up itself.  What the HIP runtime does the first time a process launches a kernel
 * (loading the library's code objects) is outside the library's hands: make one warm-up call, or any other decode
 * through the library, before the first call that must not allocate (e.g. before a capture begins).
 *   limits   N * B <= NSD_NOALLOC_MAX_HYPS and N * B * T <= NSD_NOALLOC_MAX_CELLS, else NSD_E_NOALLOC_SIZE
 *   sharing  all calls on one GPU use that GPU's one array: calls that overlap on a GPU (two streams, two
 *            threads) race; calls on one stream, or ordered by events, are fine */
#define NSD_NOALLOC_MAX_HYPS      65536u                  /* e.g. 4096 utterances x beam 16 */
#define NSD_NOALLOC_MAX_CELLS     (65536u * 64u)          /* ... x 64 steps */
#define NSD_NOALLOC_SCRATCH_BYTES (22u * 1024u * 1024u)
int nsd_decode_synth_noalloc(const NsdTables *t, const NsdParams *p, uint64_t base_seed, int64_t utt0,
                             const int8_t *targets, const int32_t *target_len, int32_t stride, float margin,
                             int32_t T, int32_t N, int32_t B,
                             uint8_t *out_ids, int32_t *out_len, double *out_score, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NSDECODE_H */
