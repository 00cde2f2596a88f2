// Tables, state initialisation, back-trace, CTC greedy, LM string scoring,
// rerank gather, log-softmax and the synthetic logits generator (gfx950).
#include <hip/hip_runtime.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "nsd_device.h"

// ---------------------------------------------------------------------
// version / errors
// ---------------------------------------------------------------------
extern "C" int nsd_version(void) { return NSD_VERSION; }

extern "C" const char *nsd_strerror(int code)
{
    switch (code) {
    case NSD_OK:        return "ok";
    case NSD_E_NULL:    return "null pointer argument";
    case NSD_E_BEAM:    return "beam width outside [1, 64]";
    case NSD_E_VOCAB:   return "vocabulary size must be 40";
    case NSD_E_ORDER:   return "n-gram order outside [0, 10]";
    case NSD_E_TABLE:   return "malformed table";
    case NSD_E_VARIANT: return "unknown decoder variant";
    case NSD_E_SIZE:    return "size argument out of range";
    case NSD_E_NOALLOC_SIZE: return "batch exceeds the built-in scratch of nsd_decode_synth_noalloc "
                                    "(N*B <= 65536 and N*B*T <= 4194304)";
    case NSD_E_ALIGN:   return "array pointer not a multiple of 16 bytes";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown nsdecode error";
}

extern "C" uint64_t nsd_hash64(uint64_t key) { return nsd_mix64(key); }

// ---------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------
static bool is_pow2(uint64_t x) { return x && !(x & (x - 1)); }

static int nsd_scratch_address(unsigned char **out);

extern "C" int nsd_tables_create(const NsdTrieCSR *trie, const NsdNgramHash *lm, NsdTables **out)
{
    if (!trie || !out || !trie->nodes) return NSD_E_NULL;
    if (trie->n_nodes < 1) return NSD_E_TABLE;
    const int order = lm ? lm->order : 0;
    if (order < 0 || order > NSD_MAX_ORDER) return NSD_E_ORDER;
    {   // The library's set-up call also resolves this device's address of the built-in scratch array (the first
        // look-up of a device symbol loads the code object and may allocate): nsd_decode_synth_noalloc then finds
        // it in the per-device cache and calls nothing that allocates, from its first call on.
        unsigned char *sp = nullptr;
        const int rc = nsd_scratch_address(&sp);
        if (rc != NSD_OK) return rc;
    }
    double lo = 0.0, hi = 0.0;                     // range of the values a table look-up can return
    if (order > 0) {
        if (!lm->grams || !lm->ctxs) return NSD_E_NULL;
        if (!is_pow2(lm->gram_cap) || !is_pow2(lm->ctx_cap) || lm->ctx_cap > (1ull << 32)) return NSD_E_TABLE;
        if ((lm->n_ctx_recs && !lm->ctx_recs) || (lm->n_gram_recs && !lm->gram_recs)) return NSD_E_NULL;
        // every automaton edge must stay inside the arrays
        for (uint32_t i = 0; i < lm->n_ctx_recs; ++i) {
            const NsdCtxRec &r = lm->ctx_recs[i];
            if (r.symmask && (uint64_t)r.child_base + (uint64_t)__builtin_popcountll(r.symmask) > lm->n_gram_recs)
                return NSD_E_TABLE;
        }
        for (uint32_t i = 0; i < lm->n_gram_recs; ++i)
            if (lm->gram_recs[i].next_ctx != NSD_CTX_NONE && lm->gram_recs[i].next_ctx >= lm->n_ctx_recs)
                return NSD_E_TABLE;
        for (uint64_t i = 0; i < lm->ctx_cap; ++i) {
            // a context key sits in one of its two slots and names a record that exists
            if (lm->ctxs[i].key == NSD_KEY_EMPTY) continue;
            const uint64_t m = nsd_mix64(lm->ctxs[i].key);
            if ((m & (lm->ctx_cap - 1)) != i && ((m >> 32) & (lm->ctx_cap - 1)) != i) return NSD_E_TABLE;
            if (lm->ctxs[i].ctx_id >= lm->n_ctx_recs) return NSD_E_TABLE;
        }
        if (lm->init_ctx != NSD_CTX_NONE && lm->init_ctx >= lm->n_ctx_recs) return NSD_E_TABLE;
        {   // Range of the values a look-up can return: the bounds of a candidate's score before its look-up are
            // built from it, so every log-prob must be a finite number (a NaN would lie outside any range and a
            // bound that ignores it could drop a candidate that belongs to the top-B).
            bool finite = __builtin_isfinite(lm->unseen);
            lo = hi = lm->unseen;
            auto see = [&](double v) { finite = finite && __builtin_isfinite(v); if (v < lo) lo = v; if (v > hi) hi = v; };
            for (uint64_t i = 0; i < lm->gram_cap; ++i)
                if (lm->grams[i].key != NSD_KEY_EMPTY) see(lm->grams[i].logp);
            for (uint64_t i = 0; i < lm->ctx_cap; ++i)
                if (lm->ctxs[i].key != NSD_KEY_EMPTY) { see(lm->ctxs[i].backoff); see(lm->ctxs[i].term); }
            for (uint32_t i = 0; i < lm->n_ctx_recs; ++i) { see(lm->ctx_recs[i].backoff); see(lm->ctx_recs[i].term); }
            for (uint32_t i = 0; i < lm->n_gram_recs; ++i) { see(lm->gram_recs[i].logp); see(lm->gram_recs[i].term_next); }
            if (!finite) return NSD_E_TABLE;
        }
    }
    // children of every node must stay inside the array
    for (uint32_t i = 0; i < trie->n_nodes; ++i) {
        const uint64_t kids = trie->nodes[i].mask & ((1ull << 36) - 1);
        if (kids && (uint64_t)trie->nodes[i].child_base + (uint64_t)__builtin_popcountll(kids) > trie->n_nodes)
            return NSD_E_TABLE;
    }
    NsdTables *t = new NsdTables();
    memset(t, 0, sizeof(*t));
    hipError_t e;
    const size_t nb = sizeof(NsdTrieNode) * (size_t)trie->n_nodes;
    if ((e = hipMalloc((void **)&t->nodes, nb)) != hipSuccess) { delete t; return (int)e; }
    if ((e = hipMemcpy(t->nodes, trie->nodes, nb, hipMemcpyHostToDevice)) != hipSuccess) goto fail;
    t->n_nodes = trie->n_nodes;
    t->order = order;
    {   // fan-out of the upper levels (nodes are numbered breadth-first): a tuning hint for nsd_step
        const uint32_t top = trie->n_nodes < 1369u ? trie->n_nodes : 1369u;        // 1 + 36 + 36 * 37
        uint64_t kids = 0, inner = 0;
        for (uint32_t i = 0; i < top; ++i) {
            const int c = __builtin_popcountll(trie->nodes[i].mask & ((1ull << NSD_NBASE) - 1));
            if (c) { kids += (uint64_t)c; ++inner; }
        }
        t->fanout_x16 = inner ? (uint32_t)(16 * kids / inner) : 0u;
    }
    if (order > 0) {
        const size_t gb = sizeof(NsdGramSlot) * (size_t)lm->gram_cap;
        const size_t cb = sizeof(NsdCtxSlot) * (size_t)lm->ctx_cap;
        if ((e = hipMalloc((void **)&t->grams, gb)) != hipSuccess) goto fail;
        if ((e = hipMemcpy(t->grams, lm->grams, gb, hipMemcpyHostToDevice)) != hipSuccess) goto fail;
        if ((e = hipMalloc((void **)&t->ctxs, cb)) != hipSuccess) goto fail;
        if ((e = hipMemcpy(t->ctxs, lm->ctxs, cb, hipMemcpyHostToDevice)) != hipSuccess) goto fail;
        {   // automaton records (at least one element is allocated so that the pointers are never null)
            const size_t rb = sizeof(NsdCtxRec) * (size_t)(lm->n_ctx_recs ? lm->n_ctx_recs : 1);
            const size_t qb = sizeof(NsdGramRec) * (size_t)(lm->n_gram_recs ? lm->n_gram_recs : 1);
            if ((e = hipMalloc((void **)&t->ctx_recs, rb)) != hipSuccess) goto fail;
            if ((e = hipMalloc((void **)&t->gram_recs, qb)) != hipSuccess) goto fail;
            if (lm->n_ctx_recs && (e = hipMemcpy(t->ctx_recs, lm->ctx_recs, sizeof(NsdCtxRec) * (size_t)lm->n_ctx_recs,
                                                 hipMemcpyHostToDevice)) != hipSuccess) goto fail;
            if (lm->n_gram_recs && (e = hipMemcpy(t->gram_recs, lm->gram_recs, sizeof(NsdGramRec) * (size_t)lm->n_gram_recs,
                                                  hipMemcpyHostToDevice)) != hipSuccess) goto fail;
            t->n_ctx_recs = lm->n_ctx_recs;
            t->n_gram_recs = lm->n_gram_recs;
            t->init_ctx = lm->init_ctx;
        }
        if (order == 1) {
            // a unigram model scores the '^' marker itself: log-prob of the n-gram "^", else the empty
            // context's back-off, else the unseen value
            t->lm_start = lm->unseen;
            {
                const uint64_t m = nsd_mix64(0);
                const uint64_t h1 = m & (lm->ctx_cap - 1), h2 = (m >> 32) & (lm->ctx_cap - 1);
                if (lm->ctxs[h1].key == 0) t->lm_start = lm->ctxs[h1].backoff;
                else if (lm->ctxs[h2].key == 0) t->lm_start = lm->ctxs[h2].backoff;
            }
            for (uint64_t h = nsd_mix64(NSD_SYM_HAT) & (lm->gram_cap - 1);; h = (h + 1) & (lm->gram_cap - 1)) {
                if (lm->grams[h].key == NSD_SYM_HAT) { t->lm_start = lm->grams[h].logp; break; }
                if (lm->grams[h].key == NSD_KEY_EMPTY) break;
            }
        }
        t->min_logp = lo;
        t->max_logp = hi;
        t->gram_mask = lm->gram_cap - 1;
        t->ctx_mask = lm->ctx_cap - 1;
        t->unseen = lm->unseen;
        t->ctx_sym_mask = order >= 2 ? ((1ull << (6 * (order - 1))) - 1ull) : 0ull;
    }
    if (order == 0) t->init_ctx = NSD_CTX_NONE;
    *out = t;
    return NSD_OK;
fail:
    if (t->nodes) (void)hipFree(t->nodes);
    if (t->grams) (void)hipFree(t->grams);
    if (t->ctxs) (void)hipFree(t->ctxs);
    if (t->ctx_recs) (void)hipFree(t->ctx_recs);
    if (t->gram_recs) (void)hipFree(t->gram_recs);
    delete t;
    return (int)e;
}

extern "C" int nsd_tables_destroy(NsdTables *t)
{
    if (!t) return NSD_OK;
    if (t->nodes) (void)hipFree(t->nodes);
    if (t->grams) (void)hipFree(t->grams);
    if (t->ctxs) (void)hipFree(t->ctxs);
    if (t->ctx_recs) (void)hipFree(t->ctx_recs);
    if (t->gram_recs) (void)hipFree(t->gram_recs);
    delete t;
    return NSD_OK;
}

extern "C" int nsd_tables_order(const NsdTables *t) { return t ? t->order : NSD_E_NULL; }

// ---------------------------------------------------------------------
// state init
// ---------------------------------------------------------------------
__global__ void nsd_state_init_kernel(NsdBeamState st, int N, int B, uint64_t ctx0, uint32_t id0, double lm0)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u == 0 && st.done_count) *st.done_count = 0;
    if (u >= N) return;
    const size_t i = (size_t)u * B;
    st.score[i] = 0.0;
    st.lmsum[i] = lm0;
    st.lmsum_e[i] = lm0;
    st.ctx[i] = ctx0;
    st.hist[i] = NSD_HIST0;
    st.ctx_id[i] = id0;
    st.ctxe_id[i] = id0;
    st.node[i] = 0;
    st.meta[i] = 0;
    st.utt[u] = 1u;            // one hypothesis, not done, zero steps
}

extern "C" int nsd_state_init(const NsdTables *t, const NsdBeamState *st, int32_t N, int32_t B, void *stream)
{
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (N < 0) return NSD_E_SIZE;
    if (N == 0) return NSD_OK;
    if (!t || !st || !st->score || !st->lmsum || !st->lmsum_e || !st->ctx || !st->hist || !st->ctx_id ||
        !st->ctxe_id || !st->node || !st->meta || !st->utt) return NSD_E_NULL;
    // context of the empty surface: '^' cut to the last order-1 symbols
    const uint64_t ctx0 = t->order >= 2 ? (uint64_t)NSD_SYM_HAT : 0ull;
    hipLaunchKernelGGL(nsd_state_init_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, *st, N, B,
                       ctx0, t->init_ctx, t->lm_start);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------
// back-trace
// ---------------------------------------------------------------------
// One wave per utterance.  The utterance's back-pointers and tokens of all executed steps (nsteps * B bytes
// each) are first copied into LDS with independent loads, so that following a chain costs LDS latency per
// step instead of a dependent global load; lane r then walks hypothesis r from its last step to step 0 and
// writes BOS + tokens in forward order.  `all` = every hypothesis of the beam (token histories for the next
// model call), otherwise only rank 0 (the result).
__global__ void __launch_bounds__(64)
nsd_walk_kernel(NsdBeamState st, const uint8_t *backptr, const uint8_t *token, int T, int N, int B, int all,
                uint8_t *out_ids, int32_t *out_len, double *out_score)
{
    extern __shared__ unsigned char walk_lds[];
    const int u = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t um = st.utt[u];
    const int nbeams = (int)(um & NSD_U_NBEAMS_MASK);
    int nsteps = (int)(um >> NSD_U_STEPS_SHIFT);
    if (nsteps > T) nsteps = T;
    uint8_t *bp = walk_lds, *tk = walk_lds + (size_t)T * B;
    const size_t NB = (size_t)N * B, ubase = (size_t)u * B;
    for (int i = lane; i < nsteps * B; i += 64) {
        const size_t o = (size_t)(i / B) * NB + ubase + (size_t)(i % B);
        bp[i] = backptr[o];
        tk[i] = token[o];
    }
    __syncthreads();
    const int rows = all ? B : 1;
    if (lane < rows) {
        const size_t g = all ? ubase + lane : (size_t)u;
        int len = 0;
        if (lane < nbeams) {
            int r = lane, n = 0;
            for (int t = nsteps - 1; t >= 0; --t) {
                if (tk[t * B + r] != NSD_TOKEN_CARRY) ++n;
                r = bp[t * B + r];
            }
            uint8_t *dst = out_ids + g * (size_t)(T + 1);
            dst[0] = NSD_BOS;
            r = lane;
            int k = n;
            for (int t = nsteps - 1; t >= 0; --t) {
                const uint8_t c = tk[t * B + r];
                if (c != NSD_TOKEN_CARRY) dst[k--] = c;
                r = bp[t * B + r];
            }
            len = n + 1;
        }
        out_len[g] = len;
        if (!all && out_score) out_score[u] = st.score[ubase];
    }
}

static int nsd_walk_launch(const NsdBeamState *st, const uint8_t *backptr, const uint8_t *token, int32_t T, int32_t N,
                           int32_t B, int all, uint8_t *out_ids, int32_t *out_len, double *out_score, void *stream)
{
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (N < 0 || T < 0 || T > NSD_MAX_STEPS) return NSD_E_SIZE;
    if (N == 0) return NSD_OK;
    if (!st || !backptr || !token || !out_ids || !out_len) return NSD_E_NULL;
    hipLaunchKernelGGL(nsd_walk_kernel, dim3((unsigned)N), dim3(64), (size_t)2 * (T > 0 ? T : 1) * B, (hipStream_t)stream,
                       *st, backptr, token, T, N, B, all, out_ids, out_len, out_score);
    return (int)hipGetLastError();
}

extern "C" int nsd_backtrace(const NsdBeamState *st, const uint8_t *backptr, const uint8_t *token,
                             int32_t T, int32_t N, int32_t B,
                             uint8_t *out_ids, int32_t *out_len, double *out_score, void *stream)
{
    return nsd_walk_launch(st, backptr, token, T, N, B, 0, out_ids, out_len, out_score, stream);
}

extern "C" int nsd_gather_ids(const NsdBeamState *st, const uint8_t *backptr, const uint8_t *token,
                              int32_t T, int32_t N, int32_t B, uint8_t *out_ids, int32_t *out_len, void *stream)
{
    return nsd_walk_launch(st, backptr, token, T, N, B, 1, out_ids, out_len, nullptr, stream);
}

// ---------------------------------------------------------------------
// CTC greedy: one wave per utterance, one lane per frame, 64 frames a round
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
nsd_ctc_greedy_kernel(const float *logits, int N, int Tp, uint8_t *out_ids, int32_t *out_len)
{
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= N) return;
    const float *base = logits + (size_t)u * Tp * NSD_V;
    uint8_t *dst = out_ids + (size_t)u * Tp;
    int prev_tail = -1;          // argmax of the last frame of the previous round
    int n = 0;
    for (int t0 = 0; t0 < Tp; t0 += 64) {
        const int t = t0 + lane;
        int best = -2;
        if (t < Tp) {
            const float4 *row = reinterpret_cast<const float4 *>(base + (size_t)t * NSD_V);
            float bv = -__builtin_inff();
            best = 0;
            bool first = true;
#pragma unroll
            for (int q = 0; q < NSD_V / 4; ++q) {
                const float4 f = row[q];
                const float v4[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (first || v4[c] > bv) { bv = v4[c]; best = q * 4 + c; first = false; }
                }
            }
        }
        int prev = __shfl_up(best, 1);
        if (lane == 0) prev = prev_tail;
        const bool keep = (t < Tp) && best != prev && best != NSD_PAD && best != NSD_BOS && best != NSD_EOS;
        const uint64_t bal = __ballot(keep);
        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (keep) dst[n + (int)pre] = (uint8_t)best;
        n += __popcll(bal);
        const int last = (Tp - t0 >= 64) ? 63 : (Tp - t0 - 1);
        prev_tail = __shfl(best, last);
    }
    if (lane == 0) out_len[u] = n;
}

extern "C" int nsd_ctc_greedy(const float *logits, int32_t N, int32_t Tp, int32_t V,
                              uint8_t *out_ids, int32_t *out_len, void *stream)
{
    if (V != NSD_V) return NSD_E_VOCAB;
    if (N < 0 || Tp < 0) return NSD_E_SIZE;
    if (N == 0) return NSD_OK;
    if (!logits || !out_ids || !out_len) return NSD_E_NULL;
    hipLaunchKernelGGL(nsd_ctc_greedy_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       logits, N, Tp, out_ids, out_len);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------
// whole-string LM score: sum over the windows of '^' + strip(text) + '$',
// left to right (the summation order of CharNgramLM.logp)
// ---------------------------------------------------------------------
// One wave scores one string.  text[0, len) are the base-character ids of the stripped text, in LDS.  Every lane
// takes one window of a round of 64 (its context key is rebuilt from the order-1 symbols before it), so the table
// probes of a round are in flight together; lane 0 then adds the round's values in text order.  The result is
// valid on lane 0.
NSD_D double nsd_wave_logp(const NsdTables &tab, const uint8_t *text, int len, double *vals, int lane)
{
    const int order = tab.order;
    const int m = len + 2;                         // symbols in '^' + text + '$'
    auto sym_at = [&](int p) -> uint32_t {
        return p == 0 ? NSD_SYM_HAT : (p == m - 1 ? NSD_SYM_DOLLAR : (uint32_t)text[p - 1] + 1u);
    };
    double lp = 0.0;
    for (int i0 = 0; i0 < m; i0 += 64) {
        const int i = i0 + lane;
        double v = 0.0;
        if (i < m && i + 1 >= order) {             // window ending at symbol i exists
            uint64_t ctx = 0;
            for (int k = order - 1; k >= 1; --k)
                if (i - k >= 0) ctx = (ctx << 6) | sym_at(i - k);
            if (!nsd_gram_find(tab, (ctx << 6) | sym_at(i), v)) v = nsd_ctx_backoff(tab, ctx);
        }
        vals[lane] = v;
        nsd_wave_sync();
        if (lane == 0) {
            const int cnt = m - i0 < 64 ? m - i0 : 64;
            for (int q = 0; q < cnt; ++q)
                if (i0 + q + 1 >= order) lp += vals[q];
        }
        nsd_wave_sync();
    }
    return lp;
}

// first and last non-space position of buf[0, k) (LDS), -1 when there is none; uniform over the wave
NSD_D void nsd_wave_strip(const uint8_t *buf, int k, int lane, int &first, int &last)
{
    first = last = -1;
    for (int j0 = 0; j0 < k; j0 += 64) {
        const int j = j0 + lane;
        const uint64_t nsb = __builtin_amdgcn_ballot_w64(j < k && buf[j] != NSD_SPACE);
        if (nsb) {
            if (first < 0) first = j0 + __ffsll((long long)nsb) - 1;
            last = j0 + 63 - __clzll((long long)nsb);
        }
    }
}

constexpr int LM_WAVE_MAX_STRIDE = 16384;           // longest string the wave-per-string kernels stage in LDS

__global__ void __launch_bounds__(256)
nsd_lm_logp_kernel(NsdTables tab, const uint8_t *text, const int32_t *len, int N, int stride, int buf_bytes, double *out)
{
    extern __shared__ __align__(16) unsigned char lm_smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int u = blockIdx.x * (blockDim.x >> 6) + wid;
    if (u >= N) return;
    double *vals = reinterpret_cast<double *>(lm_smem + (size_t)wid * (512 + buf_bytes));
    uint8_t *buf = reinterpret_cast<uint8_t *>(vals + 64);
    const uint8_t *s = text + (size_t)u * stride;
    int b = len[u];
    b = b < 0 ? 0 : (b > stride ? stride : b);     // a length beyond the row would run past the LDS buffer
    for (int i = lane; i < b; i += 64) buf[i] = s[i];
    nsd_wave_sync();
    if (b == 0) { if (lane == 0) out[u] = -1e9; return; }
    int first, last;
    nsd_wave_strip(buf, b, lane, first, last);
    const int n = first < 0 ? 0 : last - first + 1;          // a text of spaces only scores "^$"
    const double lp = nsd_wave_logp(tab, buf + (first < 0 ? 0 : first), n, vals, lane);
    if (lane == 0) out[u] = lp;
}

// the same with one lane per string: strings too long to stage in LDS
__global__ void nsd_lm_logp_serial_kernel(NsdTables tab, const uint8_t *text, const int32_t *len, int N, int stride, double *out)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= N) return;
    const uint8_t *s = text + (size_t)u * stride;
    int a = 0, b = len[u];
    if (b == 0) { out[u] = -1e9; return; }
    while (a < b && s[a] == NSD_SPACE) ++a;
    while (b > a && s[b - 1] == NSD_SPACE) --b;
    const int order = tab.order;
    const int n = (b - a) + 2;                 // symbols in '^' + text + '$'
    double lp = 0.0;
    uint64_t ctx = 0;                          // last order-1 symbols before the current one
    for (int i = 0; i < n; ++i) {
        const uint32_t sym = (i == 0) ? NSD_SYM_HAT : (i == n - 1 ? NSD_SYM_DOLLAR : (uint32_t)s[a + i - 1] + 1u);
        if (i + 1 >= order) {                  // window ending at symbol i exists
            double v;
            if (!nsd_gram_find(tab, (ctx << 6) | sym, v)) v = nsd_ctx_backoff(tab, ctx);
            lp += v;
        }
        ctx = nsd_ctx_shift(tab, ctx, sym);
    }
    out[u] = lp;
}

extern "C" int nsd_lm_logp(const NsdTables *t, const uint8_t *text, const int32_t *len,
                           int32_t N, int32_t stride, double *out, void *stream)
{
    if (!t) return NSD_E_NULL;
    if (t->order < 1) return NSD_E_ORDER;
    if (N < 0 || stride < 0) return NSD_E_SIZE;
    if (N == 0) return NSD_OK;
    if (!text || !len || !out) return NSD_E_NULL;
    if (stride > LM_WAVE_MAX_STRIDE) {
        hipLaunchKernelGGL(nsd_lm_logp_serial_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                           *t, text, len, N, stride, out);
    } else {
        const int buf_bytes = (stride + 15) & ~15;
        const int waves = buf_bytes > 8192 ? 1 : 4;
        hipLaunchKernelGGL(nsd_lm_logp_kernel, dim3((N + waves - 1) / waves), dim3(64 * waves),
                           (size_t)waves * (512 + buf_bytes), (hipStream_t)stream, *t, text, len, N, stride, buf_bytes, out);
    }
    return (int)hipGetLastError();
}

// lm.logp(text) - 0.002 * len(text) of the text that ids_to_text would produce from `ids`
// (scripts/ns_utils.py:284-296): optional leading BOS skipped, stop at EOS, ids >= 37 dropped, stripped.
// One wave per string: the kept characters are compacted into `buf` (LDS, room for n bytes) chunk by chunk,
// stripped, and scored by nsd_wave_logp.  Valid on lane 0.
NSD_D double nsd_wave_hyp_lm_score(const NsdTables &tab, const uint8_t *ids, int n, bool skip_first, uint8_t *buf,
                                   double *vals, int lane)
{
    int k = 0;
    for (int i0 = skip_first ? 1 : 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const uint32_t c = i < n ? ids[i] : 0xffu;
        const uint64_t eos = __builtin_amdgcn_ballot_w64(i < n && c == NSD_EOS);
        const int limit = eos ? __ffsll((long long)eos) - 1 : 64;          // lanes from the first EOS on are out
        const bool keep = i < n && lane < limit && c < NSD_NBASE;
        const uint64_t kb = __builtin_amdgcn_ballot_w64(keep);
        if (keep) buf[k + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(kb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kb, 0u))] = (uint8_t)c;
        k += __popcll(kb);
        if (eos) break;
    }
    nsd_wave_sync();
    if (k == 0) return -1e9;                       // empty string: CharNgramLM returns -1e9
    int first, last;
    nsd_wave_strip(buf, k, lane, first, last);
    // a text of spaces only is not empty for logp (it scores "^$"), but ids_to_text has already stripped it,
    // so logp sees "" -> -1e9
    if (first < 0) return -1e9;
    const int len = last - first + 1;
    return nsd_wave_logp(tab, buf + first, len, vals, lane) - 0.002 * (double)len;
}

// one workgroup of two waves per utterance: wave 0 scores the AR transcript, wave 1 the CTC proposal
__global__ void __launch_bounds__(128)
nsd_rerank_lm_kernel(NsdTables tab, const uint8_t *ar_ids, const int32_t *ar_len, int ar_stride,
                     const uint8_t *ctc_ids, const int32_t *ctc_len, int ctc_stride,
                     int N, int buf_bytes, uint8_t *choice, double *scores)
{
    extern __shared__ __align__(16) unsigned char lm_smem[];
    __shared__ double both[2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int u = blockIdx.x;
    double *vals = reinterpret_cast<double *>(lm_smem + (size_t)wid * (512 + buf_bytes));
    uint8_t *buf = reinterpret_cast<uint8_t *>(vals + 64);
    // (lengths are held to their rows: the compacted characters of a row must fit the LDS buffer)
    const int na = ar_len[u] < ar_stride ? ar_len[u] : ar_stride, nc = ctc_len[u] < ctc_stride ? ctc_len[u] : ctc_stride;
    const double sc = wid == 0
        ? nsd_wave_hyp_lm_score(tab, ar_ids + (size_t)u * ar_stride, na, true, buf, vals, lane)
        : nsd_wave_hyp_lm_score(tab, ctc_ids + (size_t)u * ctc_stride, nc, false, buf, vals, lane);
    if (lane == 0) both[wid] = sc;
    __syncthreads();
    if (threadIdx.x == 0) {
        choice[u] = both[1] > both[0] ? 1 : 0;     // max() keeps the first (AR) on ties
        if (scores) { scores[2 * (size_t)u] = both[0]; scores[2 * (size_t)u + 1] = both[1]; }
    }
}

// the same with one lane per utterance: strings too long to stage in LDS
__device__ double nsd_hyp_lm_score(const NsdTables &tab, const uint8_t *ids, int n, bool skip_first)
{
    // pass 1: bounds of the stripped text in the filtered character stream
    int first = -1, last = -1, k = 0;
    for (int i = skip_first ? 1 : 0; i < n; ++i) {
        const uint8_t c = ids[i];
        if (c == NSD_EOS) break;
        if (c >= NSD_NBASE) continue;
        if (c != NSD_SPACE) { if (first < 0) first = k; last = k; }
        ++k;
    }
    if (k == 0) return -1e9;
    const int len = first < 0 ? 0 : last - first + 1;
    if (len == 0) return -1e9;
    const int order = tab.order;
    double lp = 0.0;
    uint64_t ctx = 0;
    int pos = 0;                                   // symbols consumed of '^' + text + '$'
    auto feed = [&](uint32_t sym) {
        if (pos + 1 >= order) {
            double v;
            if (!nsd_gram_find(tab, (ctx << 6) | sym, v)) v = nsd_ctx_backoff(tab, ctx);
            lp += v;
        }
        ctx = nsd_ctx_shift(tab, ctx, sym);
        ++pos;
    };
    feed(NSD_SYM_HAT);
    k = 0;
    for (int i = skip_first ? 1 : 0; i < n; ++i) {
        const uint8_t c = ids[i];
        if (c == NSD_EOS) break;
        if (c >= NSD_NBASE) continue;
        if (k >= first && k <= last) feed((uint32_t)c + 1u);
        ++k;
    }
    feed(NSD_SYM_DOLLAR);
    return lp - 0.002 * (double)len;
}

__global__ void nsd_rerank_lm_serial_kernel(NsdTables tab, const uint8_t *ar_ids, const int32_t *ar_len, int ar_stride,
                                            const uint8_t *ctc_ids, const int32_t *ctc_len, int ctc_stride,
                                            int N, uint8_t *choice, double *scores)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= N) return;
    const double sa = nsd_hyp_lm_score(tab, ar_ids + (size_t)u * ar_stride, ar_len[u], true);
    const double sc = nsd_hyp_lm_score(tab, ctc_ids + (size_t)u * ctc_stride, ctc_len[u], false);
    choice[u] = sc > sa ? 1 : 0;                   // max() keeps the first (AR) on ties
    if (scores) { scores[2 * (size_t)u] = sa; scores[2 * (size_t)u + 1] = sc; }
}

extern "C" int nsd_rerank_lm(const NsdTables *t, const uint8_t *ar_ids, const int32_t *ar_len, int32_t ar_stride,
                             const uint8_t *ctc_ids, const int32_t *ctc_len, int32_t ctc_stride,
                             int32_t N, uint8_t *choice, double *scores, void *stream)
{
    if (!t) return NSD_E_NULL;
    if (t->order < 1) return NSD_E_ORDER;
    if (N < 0 || ar_stride < 0 || ctc_stride < 0) return NSD_E_SIZE;
    if (N == 0) return NSD_OK;
    if (!ar_ids || !ar_len || !ctc_ids || !ctc_len || !choice) return NSD_E_NULL;
    const int longest = ar_stride > ctc_stride ? ar_stride : ctc_stride;
    if (longest > LM_WAVE_MAX_STRIDE) {
        hipLaunchKernelGGL(nsd_rerank_lm_serial_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                           *t, ar_ids, ar_len, ar_stride, ctc_ids, ctc_len, ctc_stride, N, choice, scores);
    } else {
        const int buf_bytes = (longest + 15) & ~15;
        hipLaunchKernelGGL(nsd_rerank_lm_kernel, dim3((unsigned)N), dim3(128), (size_t)2 * (512 + buf_bytes),
                           (hipStream_t)stream, *t, ar_ids, ar_len, ar_stride, ctc_ids, ctc_len, ctc_stride, N,
                           buf_bytes, choice, scores);
    }
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------
// log-softmax rows / rerank gather: 8 lanes per row
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
nsd_log_softmax_kernel(const float *x, long long rows, float *out)
{
    const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int j = threadIdx.x & 7;
    float v[5];
    const bool ok = row < rows;
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = ok ? x[row * NSD_V + j * 5 + k] : 0.0f;
    nsd_row_log_softmax5(v);
    if (ok) {
#pragma unroll
        for (int k = 0; k < 5; ++k) out[row * NSD_V + j * 5 + k] = v[k];
    }
}

extern "C" int nsd_log_softmax(const float *x, int32_t rows, int32_t V, float *out, void *stream)
{
    if (V != NSD_V) return NSD_E_VOCAB;
    if (rows < 0) return NSD_E_SIZE;
    if (rows == 0) return NSD_OK;
    if (!x || !out) return NSD_E_NULL;
    const long long threads = (long long)rows * 8;
    hipLaunchKernelGGL(nsd_log_softmax_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, (long long)rows, out);
    return (int)hipGetLastError();
}

// 8 lanes per utterance; positions are visited in order so that the fp64 sum
// is accumulated left to right like the reference's Python loop
__global__ void __launch_bounds__(256)
nsd_rerank_gather_kernel(const float *logits, const uint8_t *target, const int32_t *len, int N, int Lmax, double *out)
{
    const int u = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int j = threadIdx.x & 7;
    const bool ok = u < N;
    const int n = ok ? len[u] : 0;
    int nmax = n;
    nmax = max(nmax, __shfl_xor(nmax, 8));
    nmax = max(nmax, __shfl_xor(nmax, 16));
    nmax = max(nmax, __shfl_xor(nmax, 32));
    double tot = 0.0;
    for (int i = 0; i < nmax; ++i) {
        const bool act = i < n;
        float v[5];
        const float *row = logits + ((size_t)(ok ? u : 0) * Lmax + (act ? i : 0)) * NSD_V + j * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = act ? row[k] : 0.0f;
        nsd_row_log_softmax5(v);
        const int tg = act ? (int)target[(size_t)u * Lmax + i] : 0;
        float mine = 0.0f;
#pragma unroll
        for (int k = 0; k < 5; ++k) mine = (tg == j * 5 + k) ? v[k] : mine;
        const float val = __shfl(mine, (threadIdx.x & 56) + tg / 5);
        if (act) tot += (double)val;
    }
    if (ok && j == 0) out[u] = tot;
}

extern "C" int nsd_rerank_gather(const float *logits, const uint8_t *target, const int32_t *len,
                                 int32_t N, int32_t L, int32_t V, double *out, void *stream)
{
    if (V != NSD_V) return NSD_E_VOCAB;
    if (N < 0 || L < 0) return NSD_E_SIZE;
    if (N == 0) return NSD_OK;
    if (!logits || !target || !len || !out) return NSD_E_NULL;
    const long long threads = (long long)N * 8;
    hipLaunchKernelGGL(nsd_rerank_gather_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, logits, target, len, N, L, out);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------
// beam re-ordering of a KV cache: a batched copy of L*M*H chunks of t*row_bytes contiguous bytes, 16 bytes per
// lane; chunks are dealt to waves so that consecutive waves read consecutive heads of one source hypothesis
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
nsd_kv_reorder_kernel(const uint4 *src, uint4 *dst, const uint8_t *backptr, uint32_t M, uint32_t B, uint32_t H,
                      uint32_t chunk_stride /* uint4 per (l, m, h) */, uint32_t chunk_len /* uint4 to copy */,
                      unsigned long long n_chunks)
{
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= n_chunks) return;
    const uint32_t h = (uint32_t)(wave % H);
    const unsigned long long lm = wave / H;
    const uint32_t m = (uint32_t)(lm % M);
    const unsigned long long l = lm / M;
    uint32_t bp = backptr[m];
    if (bp >= B) bp = 0;                                   // slots the beam does not use: any in-range source
    const uint32_t pm = (m / B) * B + bp;
    const uint4 *s = src + ((l * M + pm) * H + h) * (unsigned long long)chunk_stride;
    uint4 *d = dst + ((l * M + m) * H + h) * (unsigned long long)chunk_stride;
    for (uint32_t i = lane; i < chunk_len; i += 64) d[i] = s[i];
}

extern "C" int nsd_kv_reorder(const void *src, void *dst, const uint8_t *backptr, int32_t L, int32_t M, int32_t B,
                              int32_t H, int32_t T, int32_t t, int32_t row_bytes, void *stream)
{
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (L < 0 || M < 0 || H < 1 || T < 0 || t < 0 || t > T || row_bytes < 16 || (row_bytes & 15) || (M % B) != 0)
        return NSD_E_SIZE;
    if (L == 0 || M == 0 || t == 0) return NSD_OK;
    if (!src || !dst || !backptr || src == dst) return NSD_E_NULL;
    if (((uintptr_t)src | (uintptr_t)dst) & 15u) return NSD_E_SIZE;
    const unsigned long long n_chunks = (unsigned long long)L * M * H;
    const unsigned long long stride = (unsigned long long)T * row_bytes / 16, len = (unsigned long long)t * row_bytes / 16;
    if (stride > 0xFFFFFFFFull || (n_chunks + 3) / 4 > 0x7FFFFFFFull) return NSD_E_SIZE;
    hipLaunchKernelGGL(nsd_kv_reorder_kernel, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4 *)src, (uint4 *)dst, backptr, (uint32_t)M, (uint32_t)B, (uint32_t)H,
                       (uint32_t)stride, (uint32_t)len, n_chunks);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------
// synthetic logits: a wave takes ROWS consecutive rows.  Lane = row for the row's hash (state loads and
// three 64-bit mixes, once per row), then the wave writes its ROWS * 160 bytes in rounds of one dwordx4 store
// per lane (1 KB of consecutive bytes per round), fetching the row's hash from the lane that owns it.
// ROWS = 64 does the least work per row; 16 has the shortest wave and suits launches that do not fill the chip.
// ---------------------------------------------------------------------
template <uint32_t ROWS>
__global__ void __launch_bounds__(256)
nsd_synth_logits_kernel(NsdBeamState st, uint64_t base_seed, long long utt0, const int8_t *targets,
                        const int32_t *target_len, int stride, float margin, float *logits, uint32_t nrows, uint32_t B)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= nrows) return;
    const uint32_t row = row0 + lane;
    uint32_t rk = 0;
    int tgt = -1;                                            // -1: nothing to write for this row
    if (lane < ROWS && row < nrows) {
        const uint32_t u = row / B, r = row - u * B;
        // independent loads first (slots past the beam count hold stale values that are dropped below)
        const uint32_t um = st.utt[u];
        const uint32_t meta = st.meta[row];
        const uint64_t hist = st.hist[row];
        const int tlen = target_len[u];
        if (!(um & NSD_U_DONE) && r < (um & NSD_U_NBEAMS_MASK) && !(meta & NSD_F_FINISHED)) {
            const int step = (int)(meta >> 8);
            tgt = step < tlen ? (int)targets[(size_t)u * stride + step] : NSD_EOS;
            const uint64_t useed = nsd_mix64(base_seed + (uint64_t)(utt0 + u) * 0xD1B54A32D192ED03ull);
            const uint64_t k = nsd_mix64(hist ^ nsd_mix64(useed + (uint64_t)step * NSD_GOLD));
            rk = (uint32_t)(k ^ (k >> 32));
        }
    }
    float4 *out = reinterpret_cast<float4 *>(logits) + (size_t)row0 * 10u;
    constexpr uint32_t NF4 = ROWS * 10u;                     // float4s of the wave's block
#pragma unroll 2
    for (uint32_t i = 0; i < (NF4 + 63u) / 64u; ++i) {
        const uint32_t f = i * 64u + lane;
        const uint32_t rl = (f / 10u) & 63u, c = f - (f / 10u) * 10u;
        const uint32_t rk_r = (uint32_t)__shfl((int)rk, (int)rl);
        const int tgt_r = __shfl(tgt, (int)rl);
        if (f >= NF4 || tgt_r < 0) continue;
        float x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t v = c * 4u + q;
            const uint32_t h = nsd_fmix32(rk_r + v * 0x9E3779B9u);
            x[q] = ((float)(h >> 8) * 5.9604644775390625e-08f - 0.5f) * 4.0f;
            if ((int)v == tgt_r) x[q] = x[q] + margin;
        }
        out[f] = make_float4(x[0], x[1], x[2], x[3]);
    }
}

extern "C" int nsd_synth_logits(const NsdBeamState *st, uint64_t base_seed, int64_t utt0,
                                const int8_t *targets, const int32_t *target_len, int32_t stride,
                                float margin, float *logits, int32_t N, int32_t B, void *stream)
{
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (N < 0 || stride < 0) return NSD_E_SIZE;
    if (N == 0) return NSD_OK;
    if (!st || !targets || !target_len || !logits) return NSD_E_NULL;
    const long long rows = (long long)N * B;
    if (rows * 10 > 0xFFFFFFFFll) return NSD_E_SIZE;
    if (rows >= 262144)
        hipLaunchKernelGGL(nsd_synth_logits_kernel<64>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, *st, base_seed, (long long)utt0, targets, target_len, stride,
                           margin, logits, (uint32_t)rows, (uint32_t)B);
    else
        hipLaunchKernelGGL(nsd_synth_logits_kernel<16>, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0,
                           (hipStream_t)stream, *st, base_seed, (long long)utt0, targets, target_len, stride,
                           margin, logits, (uint32_t)rows, (uint32_t)B);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------
// native decode driver for the synthetic AR head
// ---------------------------------------------------------------------
int nsd_step_with_events(const NsdTables *t, const NsdParams *p, const float *logits,
                         const NsdBeamState *in, const NsdBeamState *out,
                         uint8_t *backptr, uint8_t *token, uint64_t *mask_dbg,
                         int32_t N, int32_t B, int32_t V, void *stream, hipEvent_t ev0, hipEvent_t ev1);   // nsd_step.hip

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_ev_mutex;
std::vector<hipEvent_t> g_events[MAX_DEVICES];   // per device: grown on demand, reused by every timed decode
uint32_t *g_host_done = nullptr;                 // pinned word for the done-counter poll

int ensure_events(int dev, size_t n)
{
    std::vector<hipEvent_t> &ev = g_events[dev];
    while (ev.size() < n) {
        hipEvent_t e;
        hipError_t err = hipEventCreate(&e);
        if (err != hipSuccess) return (int)err;
        ev.push_back(e);
    }
    return 0;
}
} // namespace

extern "C" int nsd_shutdown(void)
{
    std::lock_guard<std::mutex> lock(g_ev_mutex);
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < MAX_DEVICES; ++d) {
        if (g_events[d].empty()) continue;
        if (hipSetDevice(d) == hipSuccess)
            for (hipEvent_t e : g_events[d]) (void)hipEventDestroy(e);
        g_events[d].clear();
    }
    if (have_cur) (void)hipSetDevice(cur);
    if (g_host_done) { (void)hipHostFree(g_host_done); g_host_done = nullptr; }
    return NSD_OK;
}

extern "C" int nsd_decode_synth(const NsdTables *t, const NsdParams *p, const NsdBeamState *st,
                                uint64_t base_seed, int64_t utt0,
                                const int8_t *targets, const int32_t *target_len, int32_t stride, float margin,
                                float *logits, uint8_t *backptr, uint8_t *token,
                                int32_t T, int32_t N, int32_t B,
                                uint8_t *out_ids, int32_t *out_len, double *out_score,
                                int32_t check_every, float *step_ms, int32_t *steps_run, void *stream)
{
    if (T < 0 || T > NSD_MAX_STEPS) return NSD_E_SIZE;
    if (steps_run) *steps_run = 0;
    int rc = nsd_state_init(t, st, N, B, stream);
    if (rc != 0 || N == 0) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t NB = (size_t)N * B;
    std::unique_lock<std::mutex> lock(g_ev_mutex, std::defer_lock);
    hipEvent_t *ev = nullptr;
    if (step_ms || check_every > 0) lock.lock();
    if (step_ms) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev < 0 || dev >= MAX_DEVICES) return NSD_E_SIZE;
        if ((rc = ensure_events(dev, 2 * (size_t)T)) != 0) return rc;
        ev = g_events[dev].data();
        for (int i = 0; i < T; ++i) step_ms[i] = -1.0f;
    }
    if (check_every > 0 && !g_host_done) {
        hipError_t e = hipHostMalloc((void **)&g_host_done, sizeof(uint32_t), hipHostMallocPortable);
        if (e != hipSuccess) return (int)e;
    }
    int ran = 0;
    for (int step = 0; step < T; ++step) {
        rc = nsd_synth_logits(st, base_seed, utt0, targets, target_len, stride, margin, logits, N, B, stream);
        if (rc != 0) return rc;
        rc = nsd_step_with_events(t, p, logits, st, st, backptr + (size_t)step * NB, token + (size_t)step * NB,
                                  nullptr, N, B, NSD_V, stream, ev ? ev[2 * step] : nullptr,
                                  ev ? ev[2 * step + 1] : nullptr);
        if (rc != 0) return rc;
        ran = step + 1;
        if (check_every > 0 && ran % check_every == 0 && ran < T) {
            hipError_t e = hipMemcpyAsync(g_host_done, st->done_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return (int)e;
            if (*g_host_done >= (uint32_t)N) break;
        }
    }
    if (steps_run) *steps_run = ran;
    rc = nsd_backtrace(st, backptr, token, T, N, B, out_ids, out_len, out_score, stream);
    if (rc != 0) return rc;
    if (step_ms) {
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return (int)e;
        for (int i = 0; i < ran; ++i) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess) step_ms[i] = ms;
        }
    }
    return NSD_OK;
}

// ---------------------------------------------------------------------
// nsd_decode_synth without any allocation: the beam state, the logits rows and the back-pointer / token
// arrays live in one file-scope device array instead of caller-owned workspaces.  Every GPU the library runs
// on holds its own copy of the array until the process ends; its address on a device is looked up once, with
// that device current, and from then on only that address is handed to kernels.
// ---------------------------------------------------------------------
namespace {
constexpr size_t NA_HYPS = NSD_NOALLOC_MAX_HYPS;                   // N * B
constexpr size_t NA_CELLS = NSD_NOALLOC_MAX_CELLS;                 // N * B * T
constexpr size_t na_align(size_t x) { return (x + 255) & ~(size_t)255; }
// offsets of the parts (all multiples of 256 bytes)
constexpr size_t NA_SCORE = 0;
constexpr size_t NA_LMSUM = NA_SCORE + na_align(8 * NA_HYPS);
constexpr size_t NA_LMSUM_E = NA_LMSUM + na_align(8 * NA_HYPS);
constexpr size_t NA_CTX = NA_LMSUM_E + na_align(8 * NA_HYPS);
constexpr size_t NA_HIST = NA_CTX + na_align(8 * NA_HYPS);
constexpr size_t NA_CTX_ID = NA_HIST + na_align(8 * NA_HYPS);
constexpr size_t NA_CTXE_ID = NA_CTX_ID + na_align(4 * NA_HYPS);
constexpr size_t NA_NODE = NA_CTXE_ID + na_align(4 * NA_HYPS);
constexpr size_t NA_META = NA_NODE + na_align(4 * NA_HYPS);
constexpr size_t NA_UTT = NA_META + na_align(4 * NA_HYPS);         // N <= N * B
constexpr size_t NA_DONE = NA_UTT + na_align(4 * NA_HYPS);
constexpr size_t NA_LOGITS = NA_DONE + 256;
constexpr size_t NA_BACKPTR = NA_LOGITS + na_align(4 * NSD_V * NA_HYPS);
constexpr size_t NA_TOKEN = NA_BACKPTR + na_align(NA_CELLS);
constexpr size_t NA_BYTES = NA_TOKEN + na_align(NA_CELLS);
static_assert(NA_BYTES <= (size_t)NSD_NOALLOC_SCRATCH_BYTES, "nsdecode.h states a smaller scratch size than the layout needs");

std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[MAX_DEVICES];                         // device address of `scratch` on each device
}

static __device__ __align__(256) unsigned char scratch[NSD_NOALLOC_SCRATCH_BYTES];

// device address of the scratch array on the current device (looked up on the first call per device)
static int nsd_scratch_address(unsigned char **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return NSD_E_SIZE;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!ptr) return NSD_E_NULL;
        g_scratch_ptr[dev] = static_cast<unsigned char *>(ptr);
    }
    *out = g_scratch_ptr[dev];
    return NSD_OK;
}

extern "C" int nsd_decode_synth_noalloc(const NsdTables *t, const NsdParams *p, uint64_t base_seed, int64_t utt0,
                                        const int8_t *targets, const int32_t *target_len, int32_t stride,
                                        float margin, int32_t T, int32_t N, int32_t B,
                                        uint8_t *out_ids, int32_t *out_len, double *out_score, void *stream)
{
    if (!t || !p) return NSD_E_NULL;
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (T < 0 || T > NSD_MAX_STEPS || N < 0 || stride < 0) return NSD_E_SIZE;
    if ((size_t)N * (size_t)B > NA_HYPS || (size_t)N * (size_t)B * (size_t)T > NA_CELLS) return NSD_E_NOALLOC_SIZE;
    if (N == 0) return NSD_OK;
    if (!targets || !target_len || !out_ids || !out_len) return NSD_E_NULL;
    unsigned char *base = nullptr;
    int rc = nsd_scratch_address(&base);
    if (rc != 0) return rc;
    NsdBeamState st;
    st.score = reinterpret_cast<double *>(base + NA_SCORE);
    st.lmsum = reinterpret_cast<double *>(base + NA_LMSUM);
    st.lmsum_e = reinterpret_cast<double *>(base + NA_LMSUM_E);
    st.ctx = reinterpret_cast<uint64_t *>(base + NA_CTX);
    st.hist = reinterpret_cast<uint64_t *>(base + NA_HIST);
    st.ctx_id = reinterpret_cast<uint32_t *>(base + NA_CTX_ID);
    st.ctxe_id = reinterpret_cast<uint32_t *>(base + NA_CTXE_ID);
    st.node = reinterpret_cast<uint32_t *>(base + NA_NODE);
    st.meta = reinterpret_cast<uint32_t *>(base + NA_META);
    st.utt = reinterpret_cast<uint32_t *>(base + NA_UTT);
    st.done_count = reinterpret_cast<uint32_t *>(base + NA_DONE);
    // The array keeps what the last call left.  nsd_state_init rewrites the first hypothesis of every
    // utterance, the utterance words and the done counter; every other hypothesis slot, logits row and
    // back-pointer cell is written by the step that first makes it valid before anything reads it.
    return nsd_decode_synth(t, p, &st, base_seed, utt0, targets, target_len, stride, margin,
                            reinterpret_cast<float *>(base + NA_LOGITS), base + NA_BACKPTR, base + NA_TOKEN,
                            T, N, B, out_ids, out_len, out_score, 0, nullptr, nullptr, stream);
}
