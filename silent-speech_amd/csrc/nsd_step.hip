// nsd_step: one NS decode step for N utterances x B beams (gfx950).
//
// Work split: one 64-lane wave per utterance.
//   phase 1  lane b loads hypothesis b's state, its trie node record and the
//            per-hypothesis n-gram context values into LDS
//   phase 2  8 lanes per logits row (5 consecutive logits per lane), 8 rows
//            per pass: deterministic log-softmax, lexicon mask applied, the
//            allowed (hypothesis, token) pairs are compacted into an LDS list
//   phase 3  one lane per listed candidate: n-gram lookups, beta/gamma/kappa
//            fusion, fp64 score and pruning key
//   phase 4  top-B of the list by (key descending, emission order ascending)
//   phase 5  lane r writes surviving hypothesis r: new score, n-gram state,
//            trie child, flags, history hash, back-pointer and token
//
// Restates the loop body of scripts/ns_utils.py:218-275 (variant U),
// inference_emg_llm.py:239-297 (I) and eval_ar.py:243-313 (E) of the
// reference; all score arithmetic is fp64 in the reference's operation order.
#include "nsd_device.h"

namespace {

enum : uint32_t { K_CHAR = 0, K_EOS = 1, K_CARRY = 2, K_FALLBACK = 3 };
constexpr uint32_t CONSUMED = 0x80000000u;

// candidate info word: tok[0:6) kind[6:8) order[8:20) (= parent*64 + w)
NSD_D uint32_t pack_info(uint32_t tok, uint32_t kind, uint32_t w, uint32_t parent)
{
    return tok | (kind << 6) | (w << 8) | (parent << 14);
}

struct StepArgs {
    NsdTables    tab;
    const float *logits;
    NsdBeamState in, out;
    uint8_t     *backptr;
    uint8_t     *token;
    uint64_t    *mask_dbg;
    int32_t      N, B;
    uint32_t     lds_per_wave;
    uint32_t     maxc;
};

struct WaveLds {
    double   *bScore, *bG, *bGe, *bBo;
    uint64_t *bCtx, *bCtxe, *bHist, *bMask;
    uint32_t *bChild, *bMeta, *bNode;
    double   *cKey, *cScore;
    uint32_t *cInfo;
    float    *rowbuf;
};

NSD_D WaveLds carve(unsigned char *base, int B, uint32_t maxc, bool with_score)
{
    WaveLds w;
    unsigned char *p = base;
    w.bScore = (double *)p;   p += 8 * B;
    w.bG     = (double *)p;   p += 8 * B;
    w.bGe    = (double *)p;   p += 8 * B;
    w.bBo    = (double *)p;   p += 8 * B;
    w.bCtx   = (uint64_t *)p; p += 8 * B;
    w.bCtxe  = (uint64_t *)p; p += 8 * B;
    w.bHist  = (uint64_t *)p; p += 8 * B;
    w.bMask  = (uint64_t *)p; p += 8 * B;
    w.cKey   = (double *)p;   p += 8 * (size_t)maxc;
    w.cScore = (double *)p;   if (with_score) p += 8 * (size_t)maxc;
    w.cInfo  = (uint32_t *)p; p += 4 * (size_t)maxc;
    w.bChild = (uint32_t *)p; p += 4 * B;
    w.bMeta  = (uint32_t *)p; p += 4 * B;
    w.bNode  = (uint32_t *)p; p += 4 * B;
    w.rowbuf = (float *)p;
    return w;
}

} // namespace

static size_t nsd_step_lds_per_wave(int B, int variant, uint32_t *maxc_out)
{
    uint32_t maxc = (uint32_t)B * 41u;
    size_t bytes = (size_t)B * (8 * 8 + 4 * 3) + (size_t)maxc * (8 + 4 + (variant == NSD_VARIANT_E ? 8 : 0)) + 8 * 40 * 4;
    bytes = (bytes + 15) & ~(size_t)15;
    if (maxc_out) *maxc_out = maxc;
    return bytes;
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
nsd_step_kernel(StepArgs a, NsdParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int u = blockIdx.x * WAVES + wid;
    if (u >= a.N) return;                       // waves are independent: no block barrier anywhere

    const int B = a.B;
    const uint32_t um = a.in.utt[u];
    const int nbeams = (int)(um & NSD_U_NBEAMS_MASK);
    const bool aliased = a.in.score == a.out.score;

    if (um & NSD_U_DONE) {
        if (!aliased) {
            if (lane < nbeams) {
                const size_t i = (size_t)u * B + lane;
                a.out.score[i] = a.in.score[i]; a.out.lmsum[i] = a.in.lmsum[i];
                a.out.ctx[i] = a.in.ctx[i];     a.out.hist[i] = a.in.hist[i];
                a.out.node[i] = a.in.node[i];   a.out.meta[i] = a.in.meta[i];
            }
            if (lane == 0) a.out.utt[u] = um;
        }
        return;
    }

    const int variant = P.variant;
    const bool use_lm = P.use_lm != 0 && a.tab.order > 0;
    const int order = a.tab.order;
    WaveLds L = carve(smem + (size_t)wid * a.lds_per_wave, B, a.maxc, variant == NSD_VARIANT_E);

    // ---------------- phase 1: hypothesis state -> LDS ----------------
    if (lane < nbeams) {
        const size_t i = (size_t)u * B + lane;
        const double score = a.in.score[i];
        const double G = a.in.lmsum[i];
        const uint64_t ctx = a.in.ctx[i];
        const uint64_t hist = a.in.hist[i];
        const uint32_t node = a.in.node[i];
        uint32_t meta = a.in.meta[i];
        const uint32_t flags = meta & 0xffu;
        const int glen = (int)(meta >> 8);
        uint64_t mask = 0;
        uint32_t child = 0;
        const bool live = !(flags & NSD_F_FINISHED);
        if (live && !(flags & NSD_F_DEAD) && node != NSD_NODE_DEAD) {
            const ulonglong2 rec = *reinterpret_cast<const ulonglong2 *>(&a.tab.nodes[node]);
            mask = rec.x;
            child = (uint32_t)rec.y;
        }
        if ((mask & ((1ull << NSD_NBASE) - 1)) == 0) mask = 0;     // nothing lexicon-consistent: fall-back
        double Ge = G, bo = 0.0;
        uint64_t ctxe = ctx;
        if (use_lm && live && mask != 0) {
            const int pend = (flags & NSD_F_PENDING_SPACE) ? 1 : 0;
            const int ms = 1 + glen - pend;            // symbols in '^' + stripped surface
            int me = ms;
            if (pend) {
                if (ms + 1 >= order) Ge = G + nsd_gram_logp(a.tab, ctx, NSD_SPACE + 1u, nsd_ctx_backoff(a.tab, ctx));
                ctxe = nsd_ctx_shift(a.tab, ctx, NSD_SPACE + 1u);
                me = ms + 1;
            }
            if (me + 1 >= order) bo = nsd_ctx_backoff(a.tab, ctxe);
        }
        L.bScore[lane] = score; L.bG[lane] = G; L.bGe[lane] = Ge; L.bBo[lane] = bo;
        L.bCtx[lane] = ctx; L.bCtxe[lane] = ctxe; L.bHist[lane] = hist; L.bMask[lane] = mask;
        L.bChild[lane] = child; L.bMeta[lane] = meta; L.bNode[lane] = node;
        if (a.mask_dbg) a.mask_dbg[i] = mask & ((1ull << NSD_NBASE) - 1);
    }
    nsd_wave_sync();

    // ---------------- phase 2: log-softmax + candidate list ----------------
    uint32_t C = 0;                               // wave-uniform candidate count
    bool err_topk = false;
    auto emit = [&](bool pred, float lp, uint32_t info) {
        const uint64_t bal = __ballot(pred);
        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (pred) {
            const uint32_t s = C + pre;
            L.cInfo[s] = info;
            reinterpret_cast<float *>(L.cKey)[2 * s] = lp;
        }
        C += (uint32_t)__popcll(bal);
    };

    const int j = lane & 7;
    for (int pass = 0; pass * 8 < nbeams; ++pass) {
        const int r = pass * 8 + (lane >> 3);
        const bool rowvalid = r < nbeams;
        const uint32_t meta = rowvalid ? L.bMeta[r] : NSD_F_FINISHED;
        const bool live = rowvalid && !(meta & NSD_F_FINISHED);
        const uint64_t mask = rowvalid ? L.bMask[r] : 0;
        const int glen = (int)(meta >> 8);
        float x[5];
        if (live) {
            const float *row = a.logits + ((size_t)u * B + r) * NSD_V + j * 5;
#pragma unroll
            for (int k = 0; k < 5; ++k) x[k] = row[k];
        } else {
#pragma unroll
            for (int k = 0; k < 5; ++k) x[k] = 0.0f;
        }
        nsd_row_log_softmax5(x);

        const bool whole = (mask >> 37) & 1;
        const bool pend = (meta & NSD_F_PENDING_SPACE) != 0;
        const bool eos_u = live && variant == NSD_VARIANT_U && P.allow_eos_anytime;
        const bool eos_e = live && variant == NSD_VARIANT_E && mask != 0 &&
                           glen >= P.min_len_for_eos && (pend || whole);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint32_t v = (uint32_t)(j * 5 + k);
            const bool is_char = live && v < NSD_NBASE && ((mask >> v) & 1);
            const bool is_eos = (v == NSD_EOS) && (eos_u || eos_e);
            uint32_t w = v, kind = K_CHAR;
            if (variant == NSD_VARIANT_U) w = v + 1;
            if (is_eos) { kind = K_EOS; w = (variant == NSD_VARIANT_U) ? 0u : 40u; }
            emit(is_char || is_eos, x[k], pack_info(v, kind, w, (uint32_t)r));
        }

        // fall-back: no lexicon-consistent option -> neural top-k over all 40 ids
        const bool fb = live && mask == 0 && !eos_e;
        if (__ballot(fb) != 0) {
            float *rb = L.rowbuf + (lane >> 3) * NSD_V;
#pragma unroll
            for (int k = 0; k < 5; ++k) rb[j * 5 + k] = x[k];
            nsd_wave_sync();
            int kfb = B;
            if (variant == NSD_VARIANT_U) { if (kfb > NSD_V) kfb = NSD_V; }
            else if (kfb > NSD_V) { kfb = NSD_V; if (fb) err_topk = true; }
            uint32_t rank[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int v = j * 5 + k;
                uint32_t c = 0;
                for (int i = 0; i < NSD_V; ++i) {
                    const float o = rb[i];
                    c += (o > x[k] || (o == x[k] && i < v)) ? 1u : 0u;
                }
                rank[k] = c;
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint32_t v = (uint32_t)(j * 5 + k);
                const uint32_t w = rank[k] + (variant == NSD_VARIANT_U ? 1u : 0u);
                emit(fb && rank[k] < (uint32_t)kfb, x[k], pack_info(v, K_FALLBACK, w, (uint32_t)r));
            }
            nsd_wave_sync();
        }
    }
    {   // finished hypotheses compete unchanged (ns_utils.py:222-224)
        const bool carry = lane < nbeams && (L.bMeta[lane] & NSD_F_FINISHED);
        emit(carry, 0.0f, pack_info(0x3f, K_CARRY, 0, (uint32_t)lane));
    }
    nsd_wave_sync();

    // ---------------- phase 3: score every candidate ----------------
    for (uint32_t i = lane; i < C; i += 64) {
        const uint32_t info = L.cInfo[i];
        const uint32_t tok = info & 0x3f, kind = (info >> 6) & 3, p = info >> 14;
        const double lp = (double)reinterpret_cast<const float *>(L.cKey)[2 * i];
        const double sb = L.bScore[p];
        const uint32_t meta = L.bMeta[p];
        const int glen = (int)(meta >> 8);
        double score;
        int Lnew = glen + 1;
        if (kind == K_CARRY) {
            score = sb;
            Lnew = glen > 1 ? glen : 1;
        } else if (kind == K_FALLBACK) {
            score = sb + (lp - P.gamma);
        } else if (kind == K_EOS && variant == NSD_VARIANT_U) {
            score = sb + lp;
        } else {
            const int pend = (meta & NSD_F_PENDING_SPACE) ? 1 : 0;
            const int ms = 1 + glen - pend;
            const bool whole = (L.bMask[p] >> 37) & 1;
            double add = lp;
            if (kind == K_EOS) {                       // variant E boundary EOS (eval_ar.py:290-296)
                add -= P.eos_len_nudge * (double)glen;
                if (use_lm && glen > 0) {
                    double lg = L.bG[p];
                    if (ms + 1 >= order) lg = lg + nsd_ctx_term(a.tab, L.bCtx[p]);
                    add += P.beta * lg;
                }
            } else if (tok == NSD_SPACE) {
                if (use_lm) {
                    double lg = L.bG[p];               // logp(surf + " ") == logp(surf)
                    if (ms + 1 >= order) lg = lg + nsd_ctx_term(a.tab, L.bCtx[p]);
                    add += P.beta * lg;
                }
                if (!whole) add -= P.gamma;
                if (variant == NSD_VARIANT_E && whole) add += P.kappa;
            } else {
                if (use_lm) {
                    const int me = ms + pend;
                    const uint64_t ctxe = L.bCtxe[p];
                    double lg = L.bGe[p];
                    if (me + 1 >= order) lg = lg + nsd_gram_logp(a.tab, ctxe, tok + 1u, L.bBo[p]);
                    if (me + 2 >= order) lg = lg + nsd_ctx_term(a.tab, nsd_ctx_shift(a.tab, ctxe, tok + 1u));
                    add += P.beta * lg;
                }
            }
            score = sb + add;
        }
        if (variant == NSD_VARIANT_E) {
            L.cScore[i] = score;
            L.cKey[i] = score / P.len_denom[Lnew];
        } else {
            L.cKey[i] = score;
        }
    }
    nsd_wave_sync();

    // ---------------- phase 4: top-B by (key desc, emission order asc) ----------------
    const int nsel = (int)C < B ? (int)C : B;
    uint32_t mysel = 0;
    {
        double myk = 0.0;
        uint32_t myt = 0xFFFFFFFFu;
        bool stale = true;
        for (int r = 0; r < nsel; ++r) {
            if (stale) {
                myt = 0xFFFFFFFFu; myk = -__builtin_inf();
                for (uint32_t i = lane; i < C; i += 64) {
                    const uint32_t info = L.cInfo[i];
                    if (info & CONSUMED) continue;
                    const double k = L.cKey[i];
                    const uint32_t t = (((info >> 8) & 0xFFFu) << 12) | i;
                    if (myt == 0xFFFFFFFFu || k > myk || (k == myk && t < myt)) { myk = k; myt = t; }
                }
                stale = false;
            }
            double bk = myk;
            uint32_t bt = myt;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ok = __shfl_xor(bk, off);
                const uint32_t ot = __shfl_xor(bt, off);
                const bool take = (ot != 0xFFFFFFFFu) && (bt == 0xFFFFFFFFu || ok > bk || (ok == bk && ot < bt));
                if (take) { bk = ok; bt = ot; }
            }
            const uint32_t widx = bt & 0xFFFu;
            if (lane == r) mysel = widx;
            if ((int)(widx & 63u) == lane) { L.cInfo[widx] |= CONSUMED; stale = true; }
        }
    }

    nsd_wave_sync();

    // ---------------- phase 5: write the surviving hypotheses ----------------
    bool fin = true;
    if (lane < nsel) {
        const uint32_t info = L.cInfo[mysel];
        const uint32_t tok = info & 0x3f, kind = (info >> 6) & 3, p = (info >> 14) & 0x3f;
        const double score = (variant == NSD_VARIANT_E) ? L.cScore[mysel] : L.cKey[mysel];
        uint32_t meta = L.bMeta[p];
        uint32_t flags = meta & 0xffu;
        int glen = (int)(meta >> 8);
        double G = L.bG[p];
        uint64_t ctx = L.bCtx[p];
        uint64_t hist = L.bHist[p];
        uint32_t node = L.bNode[p];
        uint8_t out_tok = NSD_TOKEN_CARRY;
        if (kind != K_CARRY) {
            out_tok = (uint8_t)tok;
            hist = nsd_hist_next(hist, tok);
            const int pend = (flags & NSD_F_PENDING_SPACE) ? 1 : 0;
            if (kind == K_EOS) {
                flags |= NSD_F_FINISHED;
            } else if (kind == K_FALLBACK) {
                flags = (flags | NSD_F_DEAD) & ~NSD_F_PENDING_SPACE;
                if (tok == NSD_EOS) flags |= NSD_F_FINISHED;
                node = NSD_NODE_DEAD;
            } else if (tok == NSD_SPACE) {
                flags |= NSD_F_PENDING_SPACE;
                node = 0;
            } else {
                if (use_lm) {
                    const int ms = 1 + glen - pend;
                    const int me = ms + pend;
                    const uint64_t ctxe = L.bCtxe[p];
                    G = L.bGe[p];
                    if (me + 1 >= order) G = G + nsd_gram_logp(a.tab, ctxe, tok + 1u, L.bBo[p]);
                    ctx = nsd_ctx_shift(a.tab, ctxe, tok + 1u);
                }
                flags &= ~NSD_F_PENDING_SPACE;
                const uint64_t m = L.bMask[p];
                node = L.bChild[p] + (uint32_t)__popcll(m & ((1ull << tok) - 1ull) & ((1ull << 36) - 1ull));
            }
            glen += 1;
        }
        meta = flags | ((uint32_t)glen << 8);
        fin = (flags & NSD_F_FINISHED) != 0;
        const size_t o = (size_t)u * B + lane;
        a.out.score[o] = score; a.out.lmsum[o] = G; a.out.ctx[o] = ctx; a.out.hist[o] = hist;
        a.out.node[o] = node;   a.out.meta[o] = meta;
        a.backptr[o] = (uint8_t)p;
        a.token[o] = out_tok;
    }
    const bool all_fin = __ballot(!fin) == 0;
    const bool any_err = __ballot(err_topk) != 0;
    if (lane == 0) {
        uint32_t nu = (uint32_t)nsel | (((um >> NSD_U_STEPS_SHIFT) + 1u) << NSD_U_STEPS_SHIFT) | (um & NSD_U_ERR_TOPK);
        if (all_fin) nu |= NSD_U_DONE;
        if (any_err) nu |= NSD_U_ERR_TOPK;
        a.out.utt[u] = nu;
        if (all_fin && a.out.done_count) atomicAdd(a.out.done_count, 1u);
    }
}

extern "C" int nsd_step(const NsdTables *t, const NsdParams *p, const float *logits,
                        const NsdBeamState *in, const NsdBeamState *out,
                        uint8_t *backptr, uint8_t *token, uint64_t *mask_dbg,
                        int32_t N, int32_t B, int32_t V, void *stream)
{
    if (!t || !p) return NSD_E_NULL;
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (V != NSD_V) return NSD_E_VOCAB;
    if (N < 0) return NSD_E_SIZE;
    if (p->variant < NSD_VARIANT_U || p->variant > NSD_VARIANT_E) return NSD_E_VARIANT;
    if (N == 0) return NSD_OK;
    if (!logits || !in || !out || !backptr || !token) return NSD_E_NULL;
    StepArgs a;
    a.tab = *t; a.logits = logits; a.in = *in; a.out = *out;
    a.backptr = backptr; a.token = token; a.mask_dbg = mask_dbg;
    a.N = N; a.B = B;
    a.lds_per_wave = (uint32_t)nsd_step_lds_per_wave(B, p->variant, &a.maxc);
    hipStream_t s = (hipStream_t)stream;
    // waves per workgroup chosen so that a workgroup's LDS stays under 64 KiB
    if (a.lds_per_wave * 4u <= 65536u) {
        hipLaunchKernelGGL(nsd_step_kernel<4>, dim3((N + 3) / 4), dim3(256), a.lds_per_wave * 4u, s, a, *p);
    } else if (a.lds_per_wave * 2u <= 65536u) {
        hipLaunchKernelGGL(nsd_step_kernel<2>, dim3((N + 1) / 2), dim3(128), a.lds_per_wave * 2u, s, a, *p);
    } else {
        hipLaunchKernelGGL(nsd_step_kernel<1>, dim3(N), dim3(64), a.lds_per_wave, s, a, *p);
    }
    return (int)hipGetLastError();
}
