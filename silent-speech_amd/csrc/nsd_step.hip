// nsd_step: one NS decode step for N utterances x B beams (gfx950).
//
// Work split: one 64-lane wave per utterance, no block-level barrier.
//   phase 1  lane b loads hypothesis b's state, then (second dependent hop)
//            its trie node record and its n-gram context record, into LDS
//   phase 2  8 lanes per logits row (5 consecutive logits per lane), live rows
//            only, 8 rows per pass: deterministic log-softmax, lexicon mask
//            applied, the allowed (hypothesis, token) pairs are compacted into
//            an LDS candidate list
//   phase 3  one lane per listed candidate: one context-automaton record load
//            (third hop), beta/gamma/kappa fusion, fp64 score and pruning key
//   phase 4  top-B of the list: every lane counts, over broadcast LDS reads of
//            all keys, how many beat its own candidate; that count is the rank
//            (exact score ties take a slower pass that also compares the
//            emission order)
//   phase 5  lane r writes surviving hypothesis r: new score, n-gram state,
//            trie child, flags, history hash, back-pointer and token
//
// Restates the loop body of scripts/ns_utils.py:218-275 (variant U),
// inference_emg_llm.py:239-297 (I) and eval_ar.py:243-313 (E) of the
// reference; all score arithmetic is fp64 in the reference's operation order.
#include "nsd_device.h"

namespace {

enum : uint32_t { K_CHAR = 0, K_EOS = 1, K_CARRY = 2, K_FALLBACK = 3 };
constexpr uint32_t SEL_NONE = 0xFFFFFFFFu;

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
    double   *bScore, *bG, *bGe, *bBo, *bTermS;
    uint64_t *bCtx, *bHist, *bMask, *bSym;
    uint32_t *bChild, *bLmBase, *bMeta, *bNode, *bCtxId, *bCtxeId, *live, *sel;
    double   *cKey, *cScore;
    uint32_t *cInfo;
    float    *rowbuf;
};

constexpr int BEAM_BYTES = 9 * 8 + 8 * 4;

NSD_D WaveLds carve(unsigned char *base, int B, uint32_t maxc, bool with_score)
{
    WaveLds w;
    unsigned char *p = base;
    w.bScore = (double *)p;   p += 8 * B;
    w.bG     = (double *)p;   p += 8 * B;
    w.bGe    = (double *)p;   p += 8 * B;
    w.bBo    = (double *)p;   p += 8 * B;
    w.bTermS = (double *)p;   p += 8 * B;
    w.bCtx   = (uint64_t *)p; p += 8 * B;
    w.bHist  = (uint64_t *)p; p += 8 * B;
    w.bMask  = (uint64_t *)p; p += 8 * B;
    w.bSym   = (uint64_t *)p; p += 8 * B;
    w.cKey   = (double *)p;   p += 8 * (size_t)maxc;
    w.cScore = (double *)p;   if (with_score) p += 8 * (size_t)maxc;
    w.cInfo  = (uint32_t *)p; p += 4 * (size_t)maxc;
    w.bChild = (uint32_t *)p; p += 4 * B;
    w.bLmBase = (uint32_t *)p; p += 4 * B;
    w.bMeta  = (uint32_t *)p; p += 4 * B;
    w.bNode  = (uint32_t *)p; p += 4 * B;
    w.bCtxId = (uint32_t *)p; p += 4 * B;
    w.bCtxeId = (uint32_t *)p; p += 4 * B;
    w.live   = (uint32_t *)p; p += 4 * B;
    w.sel    = (uint32_t *)p; p += 4 * B;
    w.rowbuf = (float *)p;
    return w;
}

// most candidates one live hypothesis can emit (a carried, finished one emits exactly one)
NSD_HD uint32_t max_cands_per_beam(int B, int variant)
{
    const uint32_t kfb = (uint32_t)(B < NSD_V ? B : NSD_V);
    const uint32_t lex = NSD_NBASE + (variant == NSD_VARIANT_I ? 0u : 1u);   // 37 characters (+ EOS)
    const uint32_t fb = kfb + (variant == NSD_VARIANT_U ? 1u : 0u);
    return lex > fb ? lex : fb;
}

} // namespace

static size_t nsd_step_lds_per_wave(int B, int variant, uint32_t *maxc_out)
{
    uint32_t maxc = (uint32_t)B * max_cands_per_beam(B, variant);
    maxc = (maxc + 1u) & ~1u;
    size_t bytes = (size_t)B * BEAM_BYTES + (size_t)maxc * (8 + 4 + (variant == NSD_VARIANT_E ? 8 : 0)) + 8 * 40 * 4;
    bytes = (bytes + 15) & ~(size_t)15;
    if (maxc_out) *maxc_out = maxc;
    return bytes;
}

template <int WAVES, int VARIANT>
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
                a.out.score[i] = a.in.score[i]; a.out.lmsum[i] = a.in.lmsum[i]; a.out.lmsum_e[i] = a.in.lmsum_e[i];
                a.out.ctx[i] = a.in.ctx[i];     a.out.hist[i] = a.in.hist[i];
                a.out.ctx_id[i] = a.in.ctx_id[i]; a.out.ctxe_id[i] = a.in.ctxe_id[i];
                a.out.node[i] = a.in.node[i];   a.out.meta[i] = a.in.meta[i];
            }
            if (lane == 0) a.out.utt[u] = um;
        }
        return;
    }

    const bool use_lm = P.use_lm != 0 && a.tab.order > 0;
    const int order = a.tab.order;
    WaveLds L = carve(smem + (size_t)wid * a.lds_per_wave, B, a.maxc, VARIANT == NSD_VARIANT_E);

    // ---------------- phase 1: hypothesis state -> LDS ----------------
    bool my_live = false;
    if (lane < B) L.sel[lane] = SEL_NONE;
    if (lane < nbeams) {
        const size_t i = (size_t)u * B + lane;
        const uint32_t meta = a.in.meta[i];
        const uint32_t node = a.in.node[i];
        const uint32_t ctx_id = a.in.ctx_id[i];
        const uint32_t ctxe_id = a.in.ctxe_id[i];
        const double score = a.in.score[i];
        const double G = a.in.lmsum[i];
        const double Ge = a.in.lmsum_e[i];
        const uint64_t ctx = a.in.ctx[i];
        const uint64_t hist = a.in.hist[i];
        const uint32_t flags = meta & 0xffu;
        const int glen = (int)(meta >> 8);
        my_live = !(flags & NSD_F_FINISHED);
        // second hop: trie node and n-gram context records
        uint64_t sym = 0;
        uint32_t lmbase = 0;
        double bo = 0.0, term_s = 0.0;
        const bool in_lex = my_live && !(flags & NSD_F_DEAD) && node != NSD_NODE_DEAD;
        ulonglong2 nrec = {0ull, 0ull};
        if (in_lex) nrec = *reinterpret_cast<const ulonglong2 *>(&a.tab.nodes[node]);
        if (use_lm && in_lex) {
            const int pend = (flags & NSD_F_PENDING_SPACE) ? 1 : 0;
            const int ms = 1 + glen - pend;                        // symbols in '^' + stripped surface
            if (ctxe_id != NSD_CTX_NONE) {
                const NsdCtxRec *r = a.tab.ctx_recs + ctxe_id;
                const ulonglong2 q0 = reinterpret_cast<const ulonglong2 *>(r)[0];   // backoff, term
                const ulonglong2 q1 = reinterpret_cast<const ulonglong2 *>(r)[1];   // symmask, child_base
                bo = __longlong_as_double((long long)q0.x);
                sym = q1.x;
                lmbase = (uint32_t)q1.y;
                if (!pend) term_s = __longlong_as_double((long long)q0.y);
            }
            if (ms + 1 >= order) {                                 // terminal window of the stripped surface exists
                if (pend) term_s = ctx_id != NSD_CTX_NONE ? a.tab.ctx_recs[ctx_id].term : a.tab.unseen;
                else if (ctxe_id == NSD_CTX_NONE) term_s = a.tab.unseen;
            }
        }
        uint64_t mask = nrec.x;
        const uint32_t child = (uint32_t)nrec.y;
        if ((mask & ((1ull << NSD_NBASE) - 1)) == 0) mask = 0;     // nothing lexicon-consistent: fall-back
        L.bScore[lane] = score; L.bG[lane] = G; L.bGe[lane] = Ge; L.bBo[lane] = bo; L.bTermS[lane] = term_s;
        L.bCtx[lane] = ctx; L.bHist[lane] = hist; L.bMask[lane] = mask; L.bSym[lane] = sym;
        L.bChild[lane] = child; L.bLmBase[lane] = lmbase; L.bMeta[lane] = meta; L.bNode[lane] = node;
        L.bCtxId[lane] = ctx_id; L.bCtxeId[lane] = ctxe_id;
        if (a.mask_dbg) a.mask_dbg[i] = mask & ((1ull << NSD_NBASE) - 1);
    }
    const uint64_t live_mask = __ballot(my_live);
    const int nlive = __popcll(live_mask);
    if (my_live) {
        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(live_mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)live_mask, 0u));
        L.live[pre] = (uint32_t)lane;
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
    for (int pass = 0; pass * 8 < nlive; ++pass) {
        const int slot = pass * 8 + (lane >> 3);
        const bool live = slot < nlive;
        const int r = live ? (int)L.live[slot] : 0;
        const uint32_t meta = L.bMeta[r];
        const uint64_t mask = live ? L.bMask[r] : 0;
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
        const bool eos_u = live && VARIANT == NSD_VARIANT_U && P.allow_eos_anytime;
        const bool eos_e = live && VARIANT == NSD_VARIANT_E && mask != 0 &&
                           glen >= P.min_len_for_eos && (pend || whole);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint32_t v = (uint32_t)(j * 5 + k);
            const bool is_char = live && v < NSD_NBASE && ((mask >> v) & 1);
            const bool is_eos = (v == NSD_EOS) && (eos_u || eos_e);
            uint32_t w = v, kind = K_CHAR;
            if (VARIANT == NSD_VARIANT_U) w = v + 1;
            if (is_eos) { kind = K_EOS; w = (VARIANT == NSD_VARIANT_U) ? 0u : 40u; }
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
            if (VARIANT == NSD_VARIANT_U) { if (kfb > NSD_V) kfb = NSD_V; }
            else if (kfb > NSD_V) { kfb = NSD_V; if (fb) err_topk = true; }
            uint32_t rank[5] = {0, 0, 0, 0, 0};
#pragma unroll 1
            for (int i = 0; i < NSD_V; ++i) {
                const float o = rb[i];
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    rank[k] += (o > x[k] || (o == x[k] && i < j * 5 + k)) ? 1u : 0u;
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint32_t v = (uint32_t)(j * 5 + k);
                const uint32_t w = rank[k] + (VARIANT == NSD_VARIANT_U ? 1u : 0u);
                emit(fb && rank[k] < (uint32_t)kfb, x[k], pack_info(v, K_FALLBACK, w, (uint32_t)r));
            }
            nsd_wave_sync();
        }
    }
    {   // finished hypotheses compete unchanged (ns_utils.py:222-224)
        const bool carry = lane < nbeams && !my_live;
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
        } else if (kind == K_EOS && VARIANT == NSD_VARIANT_U) {
            score = sb + lp;
        } else {
            const int pend = (meta & NSD_F_PENDING_SPACE) ? 1 : 0;
            const int ms = 1 + glen - pend;
            const bool whole = (L.bMask[p] >> 37) & 1;
            double add = lp;
            if (kind == K_EOS || tok == NSD_SPACE) {
                // both score the stripped surface: logp(surf + " ") == logp(surf)
                double lg = L.bG[p];
                if (ms + 1 >= order) lg = lg + L.bTermS[p];
                if (kind == K_EOS) {                   // variant E boundary EOS (eval_ar.py:290-296)
                    add -= P.eos_len_nudge * (double)glen;
                    if (use_lm && glen > 0) add += P.beta * lg;
                } else {
                    if (use_lm) add += P.beta * lg;
                    if (!whole) add -= P.gamma;
                    if (VARIANT == NSD_VARIANT_E && whole) add += P.kappa;
                }
            } else if (use_lm) {
                const uint64_t ctx = L.bCtx[p];
                const uint64_t ctxe = pend ? nsd_ctx_shift(a.tab, ctx, NSD_SPACE + 1u) : ctx;
                const NsdLmStep s = nsd_lm_advance(a.tab, 1 + glen, ctxe, L.bCtxeId[p], L.bSym[p], L.bLmBase[p],
                                                   L.bBo[p], tok + 1u);
                double lg = L.bGe[p];
                if (s.has_int) lg = lg + s.lint;
                if (s.full_new) lg = lg + s.term;
                add += P.beta * lg;
            }
            score = sb + add;
        }
        if (VARIANT == NSD_VARIANT_E) {
            L.cScore[i] = score;
            L.cKey[i] = score / P.len_denom[Lnew];
        } else {
            L.cKey[i] = score;
        }
    }
    nsd_wave_sync();

    // ---------------- phase 4: top-B by (key desc, emission order asc) ----------------
    // Rank = number of keys that beat mine, counted over broadcast LDS reads.  With more than 64
    // candidates a threshold tau with nsel <= #{key >= tau} <= 64 is found first by a ballot quickselect
    // (pivot = a key inside the open interval still in play), the survivors are compacted and ranked.
    const int nsel = (int)C < B ? (int)C : B;
    {
        double *sKey = reinterpret_cast<double *>(L.rowbuf);          // [64] compacted survivors
        uint32_t *sIdx = reinterpret_cast<uint32_t *>(L.rowbuf) + 128; // [64]
        const double *keys = L.cKey;
        uint32_t F = C;
        bool compacted = false, full_scan = false;
        if (C > 64) {
            double lo = -__builtin_inf(), hi = __builtin_inf(), tau = 0.0;
            bool ok = false;
            for (int it = 0; it < 40 && !ok; ++it) {
                // pivot: first key in (lo, hi), starting at a lane that rotates with the iteration
                bool have = false;
                double pivot = 0.0;
                for (uint32_t base = 0; base < C && !have; base += 64) {
                    const uint32_t i = base + lane;
                    const double k = i < C ? L.cKey[i] : 0.0;
                    const uint64_t act = __ballot(i < C && k > lo && k < hi);
                    if (act) {
                        const int rot = (it * 23 + 11) & 63;
                        const uint64_t hiPart = act & ~((1ull << rot) - 1ull);
                        const int pl = __ffsll((long long)(hiPart ? hiPart : act)) - 1;
                        const uint64_t kb = (uint64_t)__double_as_longlong(k);
                        const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kb, pl);
                        const uint32_t phi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(kb >> 32), pl);
                        pivot = __longlong_as_double((long long)(((uint64_t)phi << 32) | plo));
                        have = true;
                    }
                }
                if (!have) break;
                uint32_t c = 0;
                for (uint32_t base = 0; base < C; base += 64) {
                    const uint32_t i = base + lane;
                    c += (uint32_t)__popcll(__ballot(i < C && L.cKey[i] >= pivot));
                }
                if (c < (uint32_t)nsel) hi = pivot;
                else if (c > 64u) lo = pivot;
                else { ok = true; tau = pivot; F = c; }
            }
            if (ok) {
                uint32_t n = 0;
                for (uint32_t base = 0; base < C; base += 64) {
                    const uint32_t i = base + lane;
                    const double k = i < C ? L.cKey[i] : 0.0;
                    const bool pred = i < C && k >= tau;
                    const uint64_t bal = __ballot(pred);
                    const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (pred) { sKey[n + pre] = k; sIdx[n + pre] = i; }
                    n += (uint32_t)__popcll(bal);
                }
                nsd_wave_sync();
                keys = sKey;
                compacted = true;
            } else {
                full_scan = true;            // clustered / tied keys: exact pass over everything below
            }
        }
        uint32_t claimed = (uint32_t)nsel;
        if (!full_scan) {
            const bool valid = (uint32_t)lane < F;
            const double mk = valid ? keys[lane] : 0.0;
            uint32_t rank = 0;
#pragma unroll 4
            for (uint32_t q = 0; q < F; ++q) rank += (keys[q] > mk) ? 1u : 0u;
            const bool win = valid && rank < (uint32_t)nsel;
            if (win) L.sel[rank] = compacted ? sIdx[lane] : (uint32_t)lane;
            claimed = (uint32_t)__popcll(__ballot(win));
        }
        nsd_wave_sync();
        // equal keys show up as a rank claimed twice: inside the top-B that leaves a hole, at its lower
        // edge it makes more claimants than places
        const bool hole = lane < nsel && L.sel[lane] == SEL_NONE;
        if (full_scan || claimed != (uint32_t)nsel || __ballot(hole) != 0) {
            // exact score ties (or no usable threshold): rank everything with the emission order as tie-break
            if (lane < B) L.sel[lane] = SEL_NONE;
            nsd_wave_sync();
            for (uint32_t base = 0; base < C; base += 64) {
                const uint32_t i = base + lane;
                const bool valid = i < C;
                const double mk = valid ? L.cKey[i] : 0.0;
                const uint32_t mo = valid ? ((L.cInfo[i] >> 8) & 0xFFFu) : 0u;
                uint32_t rank = 0;
                for (uint32_t q = 0; q < C; ++q) {
                    const double k = L.cKey[q];
                    const uint32_t o = (L.cInfo[q] >> 8) & 0xFFFu;
                    rank += (k > mk || (k == mk && (o < mo || (o == mo && q < i)))) ? 1u : 0u;
                }
                if (valid && rank < (uint32_t)nsel) L.sel[rank] = i;
            }
            nsd_wave_sync();
        }
    }

    // ---------------- phase 5: write the surviving hypotheses ----------------
    bool fin = true;
    if (lane < nsel) {
        uint32_t mysel = L.sel[lane];
        if (mysel >= C) mysel = 0;                 // only reachable with NaN keys
        const uint32_t info = L.cInfo[mysel];
        const uint32_t tok = info & 0x3f, kind = (info >> 6) & 3, p = (info >> 14) & 0x3f;
        const double score = (VARIANT == NSD_VARIANT_E) ? L.cScore[mysel] : L.cKey[mysel];
        uint32_t meta = L.bMeta[p];
        uint32_t flags = meta & 0xffu;
        int glen = (int)(meta >> 8);
        double G = L.bG[p], Ge = L.bGe[p];
        uint64_t ctx = L.bCtx[p];
        uint64_t hist = L.bHist[p];
        uint32_t node = L.bNode[p];
        uint32_t ctx_id = L.bCtxId[p], ctxe_id = L.bCtxeId[p];
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
                // The space joins the LM sum only once a character follows it: keep the stripped state
                // and prepare the effective one (context + space).  A hypothesis with a pending space
                // sits at the trie root, which never allows a space, so pend == 0 here.
                if (use_lm) {
                    const NsdLmStep s = nsd_lm_advance(a.tab, 1 + glen, ctx, ctx_id, L.bSym[p], L.bLmBase[p],
                                                       L.bBo[p], NSD_SPACE + 1u);
                    Ge = s.has_int ? G + s.lint : G;
                    ctxe_id = s.full_new ? s.next_id : NSD_CTX_NONE;
                }
                flags |= NSD_F_PENDING_SPACE;
                node = 0;
            } else {
                if (use_lm) {
                    const uint64_t ctxe = pend ? nsd_ctx_shift(a.tab, ctx, NSD_SPACE + 1u) : ctx;
                    const NsdLmStep s = nsd_lm_advance(a.tab, 1 + glen, ctxe, ctxe_id, L.bSym[p], L.bLmBase[p],
                                                       L.bBo[p], tok + 1u);
                    G = s.has_int ? Ge + s.lint : Ge;
                    Ge = G;
                    ctx = s.ctxn;
                    ctx_id = ctxe_id = s.full_new ? s.next_id : NSD_CTX_NONE;
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
        a.out.score[o] = score; a.out.lmsum[o] = G; a.out.lmsum_e[o] = Ge; a.out.ctx[o] = ctx; a.out.hist[o] = hist;
        a.out.ctx_id[o] = ctx_id; a.out.ctxe_id[o] = ctxe_id;
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

template <int VARIANT>
static void launch_variant(const StepArgs &a, const NsdParams &p, int N, hipStream_t s)
{
    // waves per workgroup: as many as keep a workgroup's LDS at or under 40 KiB (4 workgroups per CU)
    if (a.lds_per_wave * 4u <= 40960u)
        hipLaunchKernelGGL((nsd_step_kernel<4, VARIANT>), dim3((N + 3) / 4), dim3(256), a.lds_per_wave * 4u, s, a, p);
    else if (a.lds_per_wave * 2u <= 40960u)
        hipLaunchKernelGGL((nsd_step_kernel<2, VARIANT>), dim3((N + 1) / 2), dim3(128), a.lds_per_wave * 2u, s, a, p);
    else
        hipLaunchKernelGGL((nsd_step_kernel<1, VARIANT>), dim3(N), dim3(64), a.lds_per_wave, s, a, p);
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
    switch (p->variant) {
    case NSD_VARIANT_U: launch_variant<NSD_VARIANT_U>(a, *p, N, s); break;
    case NSD_VARIANT_I: launch_variant<NSD_VARIANT_I>(a, *p, N, s); break;
    default:            launch_variant<NSD_VARIANT_E>(a, *p, N, s); break;
    }
    return (int)hipGetLastError();
}
