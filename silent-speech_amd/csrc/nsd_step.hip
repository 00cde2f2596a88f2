// nsd_step: one NS decode step for N utterances x B beams (gfx950).
//
// Work split: G lanes per utterance (G = 16, 32 or 64, at least B; chosen by the
// launcher from the batch size), so a 64-lane wave carries 64/G utterances; every
// ballot, prefix sum and LDS region is local to such a lane group and no
// block-level barrier is used.
//   phase 1  lane b of a group loads hypothesis b's state (requested together
//            with the utterance word), then (second dependent hop) its trie node
//            record and its n-gram context record, into the group's LDS region
//   passes   G/8 live logits rows per pass and utterance, 8 lanes per row
//            (5 consecutive logits per lane): deterministic log-softmax,
//            lexicon mask applied, the allowed (hypothesis, token) pairs are
//            appended to the group's LDS candidate buffer.
//            The buffer is bounded: when a pass might overflow it, what it holds
//            is scored and pruned to its top-B first (which cannot change the
//            final top-B)
//   score    one lane per candidate: one context-automaton record load (third
//            hop; a hash probe for an unseen transition), beta/gamma/kappa
//            fusion, fp64 score and pruning key; the transition itself is left in
//            the unused top of the candidate buffer for the write phase
//   select   top-B of the buffer: up to G candidates are ranked directly, one per
//            lane, each lane counting over broadcast LDS reads how many keys beat
//            its own; beyond G a bound from cell maxima first drops what cannot
//            be in the top-B, the rest is compacted and ranked one or two per
//            lane; a ballot quickselect only runs if more than 2G remain (exact
//            score ties take a slower pass that also compares the emission order)
//   write    lane r writes surviving hypothesis r: new score, n-gram state (from
//            the kept transition), trie child, flags, history hash, back-pointer
//            and token
//
// Restates the loop body of scripts/ns_utils.py:218-275 (variant U),
// inference_emg_llm.py:239-297 (I) and eval_ar.py:243-313 (E) of the
// reference; all score arithmetic is fp64 in the reference's operation order.
#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>
#include "nsd_device.h"

namespace {

enum : uint32_t { K_CHAR = 0, K_EOS = 1, K_CARRY = 2, K_FALLBACK = 3 };
constexpr uint32_t SEL_NONE = 0xFFFFFFFFu;

#ifdef NSD_STAMPS
// Diagnostic build only (never the shipped library): lane 0 of every wave stores the shader clock at the
// phase boundaries into a buffer of its own that no other code reads.
__device__ unsigned long long *g_nsd_stamps = nullptr;
#ifndef NSD_REPEAT
#define NSD_REPEAT 1
#endif
#define NSD_STAMP(k) do { if (g_nsd_stamps && lane == 0) g_nsd_stamps[((size_t)wave_id * NSD_REPEAT + rep) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define NSD_STAMP_VALUE(k, v) do { if (g_nsd_stamps && lane == 0) g_nsd_stamps[((size_t)wave_id * NSD_REPEAT + rep) * 16 + (k)] = (unsigned long long)(v); } while (0)
#elif defined(NSD_STOP_AT)
// Diagnostic build only: the kernel ends at one phase boundary, so that instruction counters of builds that end
// at successive boundaries, launched on the same input, give the instructions of each phase.
#define NSD_STAMP(k) do { if ((k) == NSD_STOP_AT || ((k) == 4 && (NSD_STOP_AT == 9 || NSD_STOP_AT == 12))) return; } while (0)
#define NSD_STAMP_VALUE(k, v) do { } while (0)
#else
#define NSD_STAMP(k) do { } while (0)
#define NSD_STAMP_VALUE(k, v) do { } while (0)
#endif

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
    uint32_t     lds_per_utt;
    uint32_t     cap;            // candidate buffer entries per utterance
    uint32_t     lds_nodes;      // trie node records every workgroup stages at the front of its LDS (0 = none)
};

// The kernel's argument block read anew: loads through the returned pointer are not merged with earlier loads of the
// same arguments, so pointers a phase needs are fetched (one scalar load) where that phase starts instead of being
// held in scalar registers, or spilled to vector lanes, from the top of the kernel.  StepArgs is the kernel's first
// parameter, i.e. it sits at offset 0 of the kernel-argument segment.
using StepArgsK = const __attribute__((address_space(4))) StepArgs;
NSD_D StepArgsK *fresh_args()
{
    StepArgsK *p = (StepArgsK *)__builtin_amdgcn_kernarg_segment_ptr();
#ifndef NSD_NO_FRESH
    asm volatile("" : "+s"(p));
#endif
    return p;
}

NSD_D NsdTables fresh_tables_raw(StepArgsK *k)
{
    NsdTables t;
    __builtin_memcpy(&t, &k->tab, sizeof(t));
    return t;
}

// ORD >= 0: the launcher guarantees that the tables have this n-gram order, so everything derived from it is a
// constant of the instantiation
template <int ORD>
NSD_D NsdTables fresh_tables(StepArgsK *k)
{
    NsdTables t = fresh_tables_raw(k);
    if constexpr (ORD >= 0) {
        t.order = ORD;
        t.ctx_sym_mask = ORD >= 2 ? ((1ull << (6 * (ORD >= 2 ? ORD - 1 : 0))) - 1ull) : 0ull;
    }
    return t;
}

// One utterance's LDS region.  Hypothesis fields are interleaved in 104-byte records (one address per
// hypothesis, fields at constant offsets); the candidate arrays follow at offsets that depend on the beam
// and the buffer size only, so no per-array pointer has to be kept in registers.
struct GroupLds {
    unsigned char *base;
    uint32_t c_off, s_off, i_off, t_off;       // candidate keys, compaction arrays, candidate info words, total
    uint32_t cap;
    static constexpr uint32_t REC = 104;
    template <class T> NSD_D T &rec(uint32_t p, uint32_t off) const { return *reinterpret_cast<T *>(base + __umul24(p, REC) + off); }   // p < 64: the 24-bit multiply-add is one full-rate instruction
    NSD_D double &bScore(uint32_t p) const { return rec<double>(p, 0); }
    NSD_D double &bG(uint32_t p) const { return rec<double>(p, 8); }
    NSD_D double &bGe(uint32_t p) const { return rec<double>(p, 16); }
    NSD_D double &bBo(uint32_t p) const { return rec<double>(p, 24); }
    NSD_D double &bTermS(uint32_t p) const { return rec<double>(p, 32); }
    NSD_D uint64_t &bCtx(uint32_t p) const { return rec<uint64_t>(p, 40); }
    NSD_D uint64_t &bHist(uint32_t p) const { return rec<uint64_t>(p, 48); }
    NSD_D uint64_t &bMask(uint32_t p) const { return rec<uint64_t>(p, 56); }
    NSD_D uint64_t &bSym(uint32_t p) const { return rec<uint64_t>(p, 64); }
    NSD_D uint32_t &bChild(uint32_t p) const { return rec<uint32_t>(p, 72); }
    NSD_D uint32_t &bLmBase(uint32_t p) const { return rec<uint32_t>(p, 76); }
    NSD_D uint32_t &bMeta(uint32_t p) const { return rec<uint32_t>(p, 80); }
    NSD_D uint32_t &bNode(uint32_t p) const { return rec<uint32_t>(p, 84); }
    NSD_D uint32_t &bCtxId(uint32_t p) const { return rec<uint32_t>(p, 88); }
    NSD_D uint32_t &bCtxeId(uint32_t p) const { return rec<uint32_t>(p, 92); }
    NSD_D uint32_t &live(uint32_t p) const { return rec<uint32_t>(p, 96); }
    NSD_D uint32_t &sel(uint32_t p) const { return rec<uint32_t>(p, 100); }
    NSD_D double *cKey() const { return reinterpret_cast<double *>(base + c_off); }
    NSD_D double *cScore() const { return reinterpret_cast<double *>(base + c_off + 8u * cap); }
    NSD_D double *sKey() const { return reinterpret_cast<double *>(base + s_off); }
    NSD_D float *rowbuf() const { return reinterpret_cast<float *>(base + s_off); }
    NSD_D uint32_t *cInfo() const { return reinterpret_cast<uint32_t *>(base + i_off); }
    NSD_D uint32_t *total() const { return reinterpret_cast<uint32_t *>(base + t_off); }
    NSD_D double &tauSeen() const { return *reinterpret_cast<double *>(base + t_off + 8u); }   // t_off is a multiple of 8
    uint32_t sidx_off;
    NSD_D uint32_t *sIdx() const { return reinterpret_cast<uint32_t *>(base + sidx_off); }
};

constexpr int BEAM_BYTES = 9 * 8 + 8 * 4;

NSD_HD constexpr uint32_t max_cands_per_beam(int B, int variant)
{
    const uint32_t kfb = (uint32_t)(B < NSD_V ? B : NSD_V);
    const uint32_t lex = NSD_NBASE + (variant == NSD_VARIANT_I ? 0u : 1u);   // 37 characters (+ EOS)
    const uint32_t fb = kfb + (variant == NSD_VARIANT_U ? 1u : 0u);
    return lex > fb ? lex : fb;
}

// Candidate buffer entries per utterance: room for a pruned buffer (B) plus the most one pass can add, and enough
// that ordinary steps (about 10 candidates per hypothesis) never prune before the final selection.
NSD_HD constexpr uint32_t step_cap(int B, int G, int variant)
{
    uint32_t cap = (uint32_t)B + (uint32_t)(G / 8) * max_cands_per_beam(B, variant);
    if (cap < 10u * (uint32_t)B) cap = 10u * (uint32_t)B;
    if (cap < 6u * (uint32_t)G) cap = 6u * (uint32_t)G;
    return (cap + 1u) & ~1u;
}

// LDS bytes of one utterance's region
NSD_HD constexpr uint32_t step_lds_per_utt(int B, int G, int variant)
{
    const uint32_t bytes = (uint32_t)((B + 1) & ~1) * (uint32_t)BEAM_BYTES +
                           step_cap(B, G, variant) * (8u + 4u + (variant == NSD_VARIANT_E ? 8u : 0u)) + 2u * (uint32_t)G * 12u + 16u;
    return (bytes + 15u) & ~15u;
}

// fp64 score of a character candidate: hypothesis score + (log-prob + beta * LM value of the extended text), the LM
// value being the running sum plus the new n-gram (when its window exists) plus the terminal n-gram (when it
// exists), added left to right.  Scoring calls it with the table's values, the bounds with the table's extremes;
// every operation is monotone in lint / term, so the result lies between the two bounds bit for bit.
NSD_D double char_score(double sb, double lp, double ge, bool has_int, double lint, bool full_new, double term,
                        bool use_lm, double beta)
{
    double add = lp;
    if (use_lm) {
        double lg = ge;
        if (has_int) lg = lg + lint;
        if (full_new) lg = lg + term;
        add += beta * lg;
    }
    return sb + add;
}

// the smallest float that is not below x (NaN stays NaN)
NSD_D float float_not_below(double x)
{
    float f = (float)x;
    if ((double)f < x) {
        const uint32_t b = __float_as_uint(f);
        f = __uint_as_float((b >> 31) ? b - 1u : b + 1u);        // one step towards +inf
    }
    return f;
}

template <int G>
NSD_D GroupLds carve(unsigned char *base, int B, uint32_t cap, bool with_score)
{
    GroupLds w;
    B = (B + 1) & ~1;                              // keeps cKey / sKey 16-byte aligned (paired key reads)
    w.base = base;
    w.cap = cap;
    w.c_off = GroupLds::REC * (uint32_t)B;
    w.s_off = w.c_off + 8u * cap * (with_score ? 2u : 1u);
    // the fall-back's row buffer (G/8 rows of 40 floats = 20 G bytes) shares the 24 G bytes of the
    // selection's compaction arrays: the two are never live at the same time
    w.sidx_off = w.s_off + 8u * 2u * G;
    w.i_off = w.s_off + 12u * 2u * G;
    w.t_off = w.i_off + 4u * cap;
    return w;
}

} // namespace

// Lanes per utterance.  A wave carries 64/G utterances: narrow groups cut the instructions per utterance
// when few hypotheses are live, but serialise an utterance's rows over more passes, which only pays once
// the launch has far more waves than the chip holds at a time.
static int nsd_step_group(int B, int N, int hint, int variant)
{
    const int gmin = B <= 16 ? 16 : (B <= 32 ? 32 : 64);
    if (hint == 16 || hint == 32 || hint == 64) return hint < gmin ? gmin : hint;
    // measured at beam 16: two utterances per wave win from 16 384 utterances on for the copy without an EOS option
    // (all hypotheses live: 57.1 against 59.8 us at 24 576, 37.9 against 35.0 us at 12 288), from 8 192 on for the
    // copy that offers EOS at every step (few live hypotheses: 14.8 against 15.3 us at 8 192, 21.2 against 24.5 us at
    // 16 384), and only beyond 65 536 for the length-normalised copy (115.6 against 112.2 us at 65 536, 239.0 against
    // 244.6 us at 131 072)
    const int many = variant == NSD_VARIANT_U ? 8192 : (variant == NSD_VARIANT_E ? 98304 : 16384);
    return N >= many ? (gmin < 32 ? 32 : gmin) : 64;
}


// BT: the beam, known at compile time (0 = taken from the arguments).  ORD: the language model is on and has this
// order (-1 = both taken from the arguments).  The launcher picks the instantiation that matches the call; the
// run-time form serves every other shape.
// QUAD: a logits row is spread over 4 lanes (10 elements each) instead of 8, rows belong to hypotheses by position
// (row slot h of a pass = hypothesis pass * RPP + h, no compaction of the live ones), and the first pass's rows are
// requested together with the hypothesis state: for copies of the decoder whose hypotheses are (nearly) all live.
// DENSE: the lexicon's fan-out makes steps whose candidates exceed the buffer the rule (a launcher's guess from the trie,
// a tuning knob like G: both forms give the same bits).  Rows that emit after a sweep are then filtered by the sweep's
// threshold before they write anything, and a 4-lanes-per-row pass is split into halves only when it does not fit.
template <int G, int VARIANT, int BT, int ORD, bool QUAD, bool DENSE>
NSD_D void nsd_step_body(const StepArgs &a, const NsdParams &P, const int rep)
{
    constexpr int UPW = 64 / G;                   // utterances per wave
    constexpr int RL = QUAD ? 4 : 8;              // lanes per logits row
    constexpr int EPL = NSD_V / RL;               // logits per lane
    constexpr int RPP = G / RL;                   // logits rows per pass and utterance
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int g = lane / G;
    const int gl = lane % G;
    const int gshift = g * G;
    const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull);
    const long long wave_id = (long long)blockIdx.x * (blockDim.x >> 6) + wid;
    const long long ull = wave_id * UPW + g;
    const int B = BT ? BT : a.B;
    bool act = ull < (long long)a.N;
    const int u = act ? (int)ull : 0;
    // The hypothesis records are requested together with the utterance word, for every slot of the beam
    // (slots past the current beam count hold stale values, which are never used): one memory round trip
    // instead of two.  Both the early-out below and the step proper consume them.
    const bool slot = act && gl < B;
    const uint32_t si = (uint32_t)u * (uint32_t)B + (uint32_t)(slot ? gl : 0);   // N * B <= NSD_MAX_HYPS: 32-bit indices
    uint32_t um = NSD_U_DONE, ld_meta = 0, ld_node = 0, ld_ctx_id = 0, ld_ctxe_id = 0;
    double ld_score = 0.0, ld_G0 = 0.0, ld_Ge = 0.0;
    uint64_t ld_ctx = 0, ld_hist = 0;
    if (act) um = a.in.utt[u];
    if (slot) {
        ld_meta = a.in.meta[si];   ld_node = a.in.node[si];
        ld_ctx_id = a.in.ctx_id[si]; ld_ctxe_id = a.in.ctxe_id[si];
        ld_score = a.in.score[si]; ld_G0 = a.in.lmsum[si]; ld_Ge = a.in.lmsum_e[si];
        ld_ctx = a.in.ctx[si];     ld_hist = a.in.hist[si];
    }
    const int nbeams = (int)(um & NSD_U_NBEAMS_MASK);
    // logits of the next pass are requested before the current pass is processed; with rows assigned by position
    // the first pass's are requested here, next to the state (rows of all B slots exist, whatever the beam count)
    const int j = lane & (RL - 1);
    const int rslot = gl / RL;                    // row slot of this lane inside a pass
    float xn[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) xn[k] = 0.0f;
    if (QUAD && act && rslot < B) {
        const float2 *row = reinterpret_cast<const float2 *>(
            a.logits + (((uint32_t)u * (uint32_t)B + (uint32_t)rslot) * (uint32_t)NSD_V + (uint32_t)(j * EPL)));
#pragma unroll
        for (int k = 0; k < EPL / 2; ++k) { const float2 v = row[k]; xn[2 * k] = v.x; xn[2 * k + 1] = v.y; }
    }

    // optional: the upper trie levels (the first node records in breadth-first order) copied to LDS by the whole
    // workgroup, before any wave leaves
    const uint32_t KN = a.lds_nodes;
    if (KN) {
        ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(smem);
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(a.tab.nodes);
        for (uint32_t i = threadIdx.x; i < KN; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }

    if (act && (um & NSD_U_DONE)) {
        StepArgsK *E = fresh_args();
        if (E->in.score != E->out.score) {
            if (gl < nbeams) {
                E->out.score[si] = ld_score; E->out.lmsum[si] = ld_G0; E->out.lmsum_e[si] = ld_Ge;
                E->out.ctx[si] = ld_ctx;     E->out.hist[si] = ld_hist;
                E->out.ctx_id[si] = ld_ctx_id; E->out.ctxe_id[si] = ld_ctxe_id;
                E->out.node[si] = ld_node;   E->out.meta[si] = ld_meta;
            }
            if (gl == 0) E->out.utt[u] = um;
        }
        act = false;
    }
    if (__builtin_amdgcn_ballot_w64(act) == 0) return;               // every utterance of this wave is finished or out of range
    NSD_STAMP(0);

    auto gballot = [&](bool pred) -> uint64_t { return (__builtin_amdgcn_ballot_w64(pred) >> gshift) & gmask; };
    // maximum over the wave's groups of a value that is uniform inside each group (scalar result)
    auto wave_max = [&](uint32_t v) -> uint32_t {
        uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
#pragma unroll
        for (int q = 1; q < UPW; ++q) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)v, q * G);
            m = o > m ? o : m;
        }
        return m;
    };
    // set bits of a group ballot below this lane (one utterance per wave: the wave-wide count instruction)
    auto gprefix = [&](uint64_t gb) -> uint32_t {
        if (UPW == 1)
            return __builtin_amdgcn_mbcnt_hi((uint32_t)(gb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)gb, 0u));
        return (uint32_t)__popcll(gb & ((1ull << gl) - 1ull));
    };

    const bool use_lm = ORD >= 0 ? ORD > 0 : (P.use_lm != 0 && a.tab.order > 0);
    const int order = ORD >= 0 ? ORD : a.tab.order;
    const uint32_t CAP = BT ? step_cap(BT, G, VARIANT) : a.cap;
    const uint32_t lds_per_utt = BT ? step_lds_per_utt(BT, G, VARIANT) : a.lds_per_utt;
    GroupLds L = carve<G>(smem + (size_t)KN * 16 + ((size_t)wid * UPW + g) * lds_per_utt, B, CAP, VARIANT == NSD_VARIANT_E);

    // ---------------- phase 1: hypothesis state -> LDS ----------------
    bool my_live = false;
    if (gl == 0) { L.total()[0] = 0; if (DENSE) L.tauSeen() = -__builtin_inf(); }
    if (act && gl < nbeams) {
        const uint32_t i = (uint32_t)u * (uint32_t)B + (uint32_t)gl;
        const uint32_t meta = ld_meta;
        const uint32_t node = ld_node;
        const uint32_t ctx_id = ld_ctx_id;
        const uint32_t ctxe_id = ld_ctxe_id;
        const double score = ld_score;
        const double G0 = ld_G0;
        const double Ge = ld_Ge;
        const uint64_t ctx = ld_ctx;
        const uint64_t hist = ld_hist;
        const uint32_t flags = meta & 0xffu;
        const int glen = (int)(meta >> 8);
        my_live = !(flags & NSD_F_FINISHED);
        // second hop: trie node and n-gram context records.  The three reads (node record, context record, terminal
        // log-prob of the stripped context under a pending space) are requested together, outside any divergent
        // branch: a lane without such a record reads its own logits row instead (160 readable bytes, discarded).
        uint64_t sym = 0;
        uint32_t lmbase = 0;
        double bo = 0.0, term_s = 0.0;
        const bool in_lex = my_live && !(flags & NSD_F_DEAD) && node != NSD_NODE_DEAD;
        const int pend = (flags & NSD_F_PENDING_SPACE) ? 1 : 0;
        const int ms = 1 + glen - pend;                            // symbols in '^' + stripped surface
        const bool lm_here = use_lm && in_lex;
        const bool has_ce = lm_here && ctxe_id != NSD_CTX_NONE;
        const bool tail_s = lm_here && ms + 1 >= order;            // terminal window of the stripped surface exists
        const bool want_t = tail_s && pend && ctx_id != NSD_CTX_NONE;
        const char *own_row = reinterpret_cast<const char *>(a.logits + (size_t)i * NSD_V);
        const char *pn = in_lex ? reinterpret_cast<const char *>(a.tab.nodes + node) : own_row;
        const char *pc = has_ce ? reinterpret_cast<const char *>(a.tab.ctx_recs + ctxe_id) : own_row;
        const char *pt = want_t ? reinterpret_cast<const char *>(&a.tab.ctx_recs[ctx_id].term) : own_row;
        ulonglong2 nrec = *reinterpret_cast<const ulonglong2 *>(pn);
        const ulonglong2 q0 = reinterpret_cast<const ulonglong2 *>(pc)[0];   // backoff, term
        const ulonglong2 q1 = reinterpret_cast<const ulonglong2 *>(pc)[1];   // symmask, child_base
        const double tterm = *reinterpret_cast<const double *>(pt);
        if (KN && in_lex && node < KN) nrec = reinterpret_cast<const ulonglong2 *>(smem)[node];
        if (!in_lex) nrec = ulonglong2{0ull, 0ull};
        if (has_ce) {
            bo = __longlong_as_double((long long)q0.x);
            sym = q1.x;
            lmbase = (uint32_t)q1.y;
            if (!pend) term_s = __longlong_as_double((long long)q0.y);
        }
        if (tail_s) {
            if (pend) term_s = ctx_id != NSD_CTX_NONE ? tterm : a.tab.unseen;
            else if (ctxe_id == NSD_CTX_NONE) term_s = a.tab.unseen;
        }
        uint64_t mask = nrec.x;
        const uint32_t child = (uint32_t)nrec.y;
        if ((mask & ((1ull << NSD_NBASE) - 1)) == 0) mask = 0;     // nothing lexicon-consistent: fall-back
        L.bScore(gl) = score; L.bG(gl) = G0; L.bGe(gl) = Ge; L.bBo(gl) = bo; L.bTermS(gl) = term_s;
        L.bCtx(gl) = ctx; L.bHist(gl) = hist; L.bMask(gl) = mask; L.bSym(gl) = sym;
        L.bChild(gl) = child; L.bLmBase(gl) = lmbase; L.bMeta(gl) = meta; L.bNode(gl) = node;
        L.bCtxId(gl) = ctx_id; L.bCtxeId(gl) = ctxe_id;
        if (a.mask_dbg) a.mask_dbg[i] = mask & ((1ull << NSD_NBASE) - 1);
        // upper bound of what this hypothesis adds to the candidate buffer
        uint32_t cnt = 1;                                          // carried unchanged
        if (my_live) cnt = mask ? (uint32_t)__popcll(mask & ((1ull << NSD_NBASE) - 1)) + 1u
                                : (uint32_t)(B < NSD_V ? B : NSD_V) + 1u;
        atomicAdd(L.total(), cnt);
    }
    const uint64_t live_gb = gballot(my_live);
    const int nlive = __popcll(live_gb);
    if (!QUAD && my_live) L.live(gprefix(live_gb)) = (uint32_t)gl;
    nsd_wave_sync();
    NSD_STAMP(1);

    uint32_t C = 0;                               // candidates in this utterance's buffer (uniform in the group)
#ifdef NSD_STAMPS
    uint32_t dbg_redo = 0, dbg_iters = 0, dbg_prunes = 0, dbg_slowthr = 0;
    unsigned long long dbg_t_prune = 0, dbg_t_score = 0, dbg_t0 = 0;
#define NSD_TIC() do { dbg_t0 = __builtin_amdgcn_s_memtime(); } while (0)
#define NSD_TOC(acc) do { (acc) += __builtin_amdgcn_s_memtime() - dbg_t0; } while (0)
#define NSD_COUNT(x) do { ++(x); } while (0)
#else
#define NSD_COUNT(x) do { } while (0)
#define NSD_TIC() do { } while (0)
#define NSD_TOC(acc) do { } while (0)
#endif
    bool err_topk = false;
    uint32_t scored = 0;                          // candidates [0, scored) already have their key (group-uniform)

    // (The overflow handling of the passes keeps every append inside the buffer; the s < CAP tests only make sure
    // that a slip in it shows as NSD_U_ERR_INTERNAL instead of as writes into a neighbour's LDS.)
    bool err_cap = false;
    auto emit = [&](bool pred, float lp, uint32_t info) {
        const uint64_t gb = gballot(pred);
        if (pred) {
            const uint32_t s = C + gprefix(gb);
            if (s < CAP) {
                L.cInfo()[s] = info;
                reinterpret_cast<float *>(L.cKey())[2 * s] = lp;
            } else {
                err_cap = true;
            }
        }
        C += (uint32_t)__popcll(gb);
        if (C > CAP) C = CAP;
    };

    // score candidates [from, C) of this group, one lane each.  With `keep` (group-uniform; needs
    // 2 C <= CAP) the n-gram transition of a character or space candidate is left in the unused top of
    // the buffer (entry CAP-1-i for candidate i), where the write phase finds it for the survivors.
    auto score_new = [&](uint32_t from, bool keep) {
        const NsdTables T = fresh_tables<ORD>(fresh_args());
        const uint32_t rounds = wave_max(act && C > from ? (C - from + G - 1) / G : 0u);
        for (uint32_t rd = 0; rd < rounds; ++rd) {
            const uint32_t i = from + rd * G + gl;
            if (!(act && i < C)) continue;
            const uint32_t info = L.cInfo()[i];
            const uint32_t tok = info & 0x3f, kind = (info >> 6) & 3, p = info >> 14;
            const double lp = (double)reinterpret_cast<const float *>(L.cKey())[2 * i];
            const double sb = L.bScore(p);
            const uint32_t meta = L.bMeta(p);
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
                const bool whole = (L.bMask(p) >> 37) & 1;
                const bool is_eos = kind == K_EOS;
                const bool is_sp = !is_eos && tok == NSD_SPACE;
                double add = lp;
                // One transition serves both kinds: a character extends the effective context; a space
                // (only offered without a pending one, when effective and stripped state coincide) is
                // scored on the stripped surface and its transition is only kept for the write phase.
                NsdLmStep s;
                s.full_new = false; s.has_int = false; s.term = 0.0; s.lint = 0.0;
                double lgn = 0.0;
                if (use_lm && !is_eos && (keep || !is_sp)) {
                    const uint64_t ctx = L.bCtx(p);
                    const uint64_t ctxe = pend ? nsd_ctx_shift(T, ctx, NSD_SPACE + 1u) : ctx;
                    s = nsd_lm_advance(T, 1 + glen, ctxe, L.bCtxeId(p), L.bSym(p), L.bLmBase(p), L.bBo(p), tok + 1u);
                    lgn = L.bGe(p);
                    if (s.has_int) lgn = lgn + s.lint;
                    if (keep) {
                        L.cKey()[CAP - 1u - i] = lgn;
                        L.cInfo()[CAP - 1u - i] = s.full_new ? s.next_id : NSD_CTX_NONE;
                    }
                }
                if (is_eos || is_sp) {
                    // both score the stripped surface: logp(surf + " ") == logp(surf)
                    double lg = L.bG(p);
                    if (ms + 1 >= order) lg = lg + L.bTermS(p);
                    if (is_eos) {                          // variant E boundary EOS (eval_ar.py:290-296)
                        add -= P.eos_len_nudge * (double)glen;
                        if (use_lm && glen > 0) add += P.beta * lg;
                    } else {
                        if (use_lm) add += P.beta * lg;
                        if (!whole) add -= P.gamma;
                        if (VARIANT == NSD_VARIANT_E && whole) add += P.kappa;
                    }
                    score = sb + add;
                } else {
                    score = char_score(sb, lp, L.bGe(p), s.has_int, s.lint, s.full_new, s.term, use_lm, P.beta);
                }
            }
            if (VARIANT == NSD_VARIANT_E) {
                L.cScore()[i] = score;
                L.cKey()[i] = score / P.len_denom[Lnew];
            } else {
                L.cKey()[i] = score;
            }
        }
    };

    // Pruning key of the not yet scored candidate i with the two table values of its n-gram transition replaced
    // by `sub_hi` / `sub_lo`: candidates whose key needs no look-up (carried, fall-back, EOS, space) come out
    // exact, hi == lo.  Same operations in the same order as score_new.
    auto bounds = [&](uint32_t i, double sub_hi, double sub_lo, double &hi, double &lo) {
        const uint32_t info = L.cInfo()[i];
        const uint32_t tok = info & 0x3f, kind = (info >> 6) & 3, p = info >> 14;
        const double lp = (double)reinterpret_cast<const float *>(L.cKey())[2 * i];
        const double sb = L.bScore(p);
        const uint32_t meta = L.bMeta(p);
        const int glen = (int)(meta >> 8);
        int Lnew = glen + 1;
        if (kind == K_CARRY) {
            hi = lo = sb;
            Lnew = glen > 1 ? glen : 1;
        } else if (kind == K_FALLBACK) {
            hi = lo = sb + (lp - P.gamma);
        } else if (kind == K_EOS && VARIANT == NSD_VARIANT_U) {
            hi = lo = sb + lp;
        } else {
            const int pend = (meta & NSD_F_PENDING_SPACE) ? 1 : 0;
            const int ms = 1 + glen - pend;
            const bool is_eos = kind == K_EOS;
            const bool is_sp = !is_eos && tok == NSD_SPACE;
            if (is_eos || is_sp) {
                const bool whole = (L.bMask(p) >> 37) & 1;
                double add = lp;
                double lg = L.bG(p);
                if (ms + 1 >= order) lg = lg + L.bTermS(p);
                if (is_eos) {
                    add -= P.eos_len_nudge * (double)glen;
                    if (use_lm && glen > 0) add += P.beta * lg;
                } else {
                    if (use_lm) add += P.beta * lg;
                    if (!whole) add -= P.gamma;
                    if (VARIANT == NSD_VARIANT_E && whole) add += P.kappa;
                }
                hi = lo = sb + add;
            } else {
                const bool has_int = 1 + glen + 1 >= order, full_new = 1 + glen + 2 >= order;
                const double ge = L.bGe(p);
                hi = char_score(sb, lp, ge, has_int, sub_hi, full_new, sub_hi, use_lm, P.beta);
                lo = char_score(sb, lp, ge, has_int, sub_lo, full_new, sub_lo, use_lm, P.beta);
            }
        }
        if (VARIANT == NSD_VARIANT_E) {
            const double den = P.len_denom[Lnew];
            hi = hi / den;
            lo = lo / den;
        }
    };

    // Drop candidates that cannot be among the top-B before a table look-up is spent on them (groups with `want`,
    // which need C > G; every lane of the wave must call it).  Every candidate's key lies between a lower and an
    // upper bound that need no look-up (exact keys are their own bounds).  The lanes form >= B cells of neighbouring
    // lanes, and the lanes of a cell take candidates that lie `cells` apart in the buffer, so that neighbouring
    // candidates, which tend to share a parent, fall into different cells;
    // every cell holds a candidate, hence at least B candidates have a key of at least tau = the smallest of the
    // cells' largest lower bounds, and a candidate whose upper bound is below tau is beaten by B others whatever
    // the look-ups return.  Survivors are moved to the front in their order (emission order and the scored
    // prefix are preserved).
    auto bound_prune = [&](bool want) {
        NSD_COUNT(dbg_slowthr);
        const bool up = !(P.beta < 0.0);
        const double sub_hi = up ? a.tab.max_logp : a.tab.min_logp;
        const double sub_lo = up ? a.tab.min_logp : a.tab.max_logp;
        const uint32_t chunks = wave_max(want ? (C + G - 1) / G : 0u);
        int cells = 1;
        while (cells < B) cells <<= 1;
        if (cells > G) cells = G;
        const int cell = G / cells;               // lanes per cell (contiguous); lane gl takes candidate perm of a chunk
        const uint32_t perm = (uint32_t)((gl % cell) * cells + gl / cell);
        double m = -__builtin_inf();
        auto bound_one = [&](uint32_t i) {
            if (!(want && i < C)) return;
            double hi, lo;
            if (i < scored) {
                hi = lo = L.cKey()[i];
            } else {
                bounds(i, sub_hi, sub_lo, hi, lo);
                // kept for the second sweep, rounded up (so that it never drops below the fp64 bound)
                reinterpret_cast<float *>(L.cKey())[2 * i + 1] = float_not_below(hi);
            }
            m = lo > m ? lo : m;                   // a NaN bound never raises the threshold
        };
        for (uint32_t ch = 0; ch + 1 < chunks; ch += 2) {       // two candidates per lane and trip: their LDS reads overlap
            bound_one(ch * G + perm);
            bound_one((ch + 1) * G + perm);
        }
        if (chunks & 1u) bound_one((chunks - 1) * G + perm);
        // largest per cell first (distances below the cell width), then the smallest over the cells
        auto red = [&](auto dist) {
            constexpr int D = decltype(dist)::value;
            if (D < G) {
                const double o = nsd_xchg_f64<D>(m);
                m = D < cell ? fmax(m, o) : fmin(m, o);
            }
        };
        red(std::integral_constant<int, 1>{});  red(std::integral_constant<int, 2>{});
        red(std::integral_constant<int, 4>{});  red(std::integral_constant<int, 8>{});
        red(std::integral_constant<int, 16>{}); red(std::integral_constant<int, 32>{});
        const double tau = m;
        // B candidates reach tau whatever follows: rows that emit later drop what cannot reach it (see the passes)
        if (DENSE && want && gl == 0 && tau > L.tauSeen()) L.tauSeen() = tau;
        uint32_t n = 0, nsc = 0;
        for (uint32_t ch = 0; ch < chunks; ++ch) {
            const uint32_t i = ch * G + gl;
            const bool valid = want && i < C;
            uint32_t info = 0;
            unsigned long long k8 = 0;
            double sc = 0.0;
            bool pred = false;
            if (valid) {
                info = L.cInfo()[i];
                k8 = reinterpret_cast<const unsigned long long *>(L.cKey())[i];
                if (VARIANT == NSD_VARIANT_E) sc = L.cScore()[i];
                const double hi = i < scored ? __longlong_as_double((long long)k8)
                                             : (double)__uint_as_float((uint32_t)(k8 >> 32));
                pred = !(hi < tau);
            }
            const uint64_t gb = gballot(pred);
            const uint32_t pos = n + gprefix(gb);
            nsc += (uint32_t)__popcll(gballot(pred && i < scored));
            nsd_wave_sync();                       // the whole chunk is in registers before any of it is overwritten
            if (pred) {
                L.cInfo()[pos] = info;
                reinterpret_cast<unsigned long long *>(L.cKey())[pos] = k8;
                if (VARIANT == NSD_VARIANT_E) L.cScore()[pos] = sc;
            }
            n += (uint32_t)__popcll(gb);
        }
        nsd_wave_sync();
        if (want) { C = n; scored = nsc; }
    };

    // top-`nsel` of candidates [0, C): afterwards L.sel(r) = buffer index of rank r (r < nsel), for the
    // groups with `want`; every lane of the wave must call it
    auto select_top = [&](bool want, int nsel) {
        if (want && gl < B) L.sel(gl) = SEL_NONE;
        const uint32_t chunks = wave_max(want ? (C + G - 1) / G : 0u);   // buffer chunks of G candidates
        uint32_t F = want ? C : 0u;
        const double *keys = L.cKey();
        bool compacted = false, full_scan = false;
        // ---- threshold: at most fcap survivors, two per lane ----
        const uint32_t fcap = 2u * (uint32_t)G;
        constexpr int KMAX = 4;                    // keys a lane keeps in registers during the search
        if (__builtin_amdgcn_ballot_w64(want && C > (uint32_t)G) != 0 && chunks <= (uint32_t)KMAX) {
            // Runs on registers: lane gl holds candidates gl, gl + G, ...
            const bool in = want && C > (uint32_t)G;
            double kk[KMAX];
            bool kv[KMAX];
#pragma unroll
            for (int c = 0; c < KMAX; ++c) {
                const uint32_t i = (uint32_t)c * G + gl;
                kv[c] = in && i < C;
                kk[c] = kv[c] ? L.cKey()[i] : 0.0;
            }
            // A bound from cell maxima first.  The lanes form `cells` >= nsel cells (every lane holds a
            // candidate, as C > G); the nsel-th best key is at least the smallest of the cells' best keys,
            // so everything below that value is out.  Usually that already leaves fewer than G candidates.
            double lo = -__builtin_inf(), hi = __builtin_inf(), tau = 0.0;
            bool ok = false, searching = false;
            {
                int cells = 1;
                while (cells < nsel) cells <<= 1;
                const int cell = G / cells;        // lanes per cell (nsel <= B <= G)
                double m = kk[0];
#pragma unroll
                for (int c = 1; c < KMAX; ++c) m = kv[c] ? fmax(m, kk[c]) : m;
                // (every aligned block of d lanes agrees before the step at distance d: register moves up to 8)
                auto red = [&](auto dist) {
                    constexpr int D = decltype(dist)::value;
                    if (D < G) {
                        const double o = nsd_xchg_f64<D>(m);
                        m = D < cell ? fmax(m, o) : fmin(m, o);
                    }
                };
                red(std::integral_constant<int, 1>{});  red(std::integral_constant<int, 2>{});
                red(std::integral_constant<int, 4>{});  red(std::integral_constant<int, 8>{});
                red(std::integral_constant<int, 16>{}); red(std::integral_constant<int, 32>{});
                uint32_t n = 0;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) {
                    if ((uint32_t)c >= chunks) continue;          // uniform: chunks the buffer does not have
                    n += (uint32_t)__popcll(gballot(kv[c] && kk[c] >= m));
                }
                if (in) {
                    if (n < (uint32_t)nsel) full_scan = true;                 // only with NaN keys
                    else if (n <= fcap) { ok = true; tau = m; F = n; }
                    else { searching = true; lo = m; }
                }
            }
            // Beyond fcap survivors: a pivot is broadcast from the lane that owns it and counted with one
            // compare + ballot per held key.
            for (int it = 0; it < 48 && __builtin_amdgcn_ballot_w64(searching) != 0; ++it) {
                NSD_COUNT(dbg_iters);
                bool have = false;
                double pivot = 0.0;
                const int rot = (it * 7 + 3) % G;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) {
                    if ((uint32_t)c >= chunks) continue;          // uniform: chunks the buffer does not have
                    const uint64_t gb = gballot(searching && !have && kv[c] && kk[c] > lo && kk[c] < hi);
                    const bool pick = gb != 0 && searching && !have;
                    int src = lane;
                    if (pick) {
                        const uint64_t hiPart = gb & ~((1ull << rot) - 1ull);
                        src = gshift + __ffsll((long long)(hiPart ? hiPart : gb)) - 1;
                    }
                    double cand;
                    if (UPW == 1) {
                        // one utterance per wave: the owner lane is the same for all lanes
                        const int sl = __builtin_amdgcn_readfirstlane(src);
                        const long long bits = __double_as_longlong(kk[c]);
                        const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bits, sl);
                        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)bits >> 32), sl);
                        cand = __longlong_as_double((long long)(((uint64_t)bhi << 32) | blo));
                    } else {
                        cand = __shfl(kk[c], src);
                    }
                    if (pick) { pivot = cand; have = true; }
                }
                uint32_t cnt = 0;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) {
                    if ((uint32_t)c >= chunks) continue;
                    cnt += (uint32_t)__popcll(gballot(searching && have && kv[c] && kk[c] >= pivot));
                }
                if (searching) {
                    if (!have) { full_scan = true; searching = false; }       // clustered / tied keys
                    else if (cnt < (uint32_t)nsel) hi = pivot;
                    else if (cnt > fcap) lo = pivot;
                    else { ok = true; tau = pivot; F = cnt; searching = false; }
                }
            }
            if (searching) { full_scan = true; searching = false; }
            // compact the survivors
            uint32_t n = 0;
#pragma unroll
            for (int c = 0; c < KMAX; ++c) {
                if ((uint32_t)c >= chunks) continue;
                const bool pred = ok && kv[c] && kk[c] >= tau;
                const uint64_t gb = gballot(pred);
                if (pred) { const uint32_t sidx = n + gprefix(gb); L.sKey()[sidx] = kk[c]; L.sIdx()[sidx] = (uint32_t)c * G + gl; }
                n += (uint32_t)__popcll(gb);
            }
            nsd_wave_sync();
            if (ok) { keys = L.sKey(); compacted = true; }
        } else if (__builtin_amdgcn_ballot_w64(want && C > fcap) != 0) {
            bool searching = want && C > fcap;
            double lo = -__builtin_inf(), hi = __builtin_inf(), tau = 0.0;
            bool ok = false;
            NSD_COUNT(dbg_slowthr);
            for (int it = 0; it < 48 && __builtin_amdgcn_ballot_w64(searching) != 0; ++it) {
                NSD_COUNT(dbg_iters);
                // pivot: first key inside (lo, hi), starting at a lane that rotates with the iteration
                bool have = false;
                double pivot = 0.0;
                for (uint32_t ch = 0; ch < chunks && __builtin_amdgcn_ballot_w64(searching && !have) != 0; ++ch) {
                    const uint32_t i = ch * G + gl;
                    const bool inb = searching && !have && i < C;
                    const double k = inb ? L.cKey()[i] : 0.0;
                    const uint64_t gb = gballot(inb && k > lo && k < hi);
                    const bool pick = gb != 0 && searching && !have;
                    int src = lane;
                    if (pick) {
                        const int rot = (it * 7 + 3) % G;
                        const uint64_t hiPart = gb & ~((1ull << rot) - 1ull);
                        src = gshift + __ffsll((long long)(hiPart ? hiPart : gb)) - 1;
                    }
                    const double cand = __shfl(k, src);
                    if (pick) { pivot = cand; have = true; }
                }
                uint32_t c = 0;
                for (uint32_t ch = 0; ch < chunks; ++ch) {
                    const uint32_t i = ch * G + gl;
                    c += (uint32_t)__popcll(gballot(searching && have && i < C && L.cKey()[i] >= pivot));
                }
                if (searching) {
                    if (!have) { full_scan = true; searching = false; }       // clustered / tied keys
                    else if (c < (uint32_t)nsel) hi = pivot;
                    else if (c > fcap) lo = pivot;
                    else { ok = true; tau = pivot; F = c; searching = false; }
                }
            }
            if (searching) { full_scan = true; searching = false; }
            // compact the survivors
            uint32_t n = 0;
            for (uint32_t ch = 0; ch < chunks; ++ch) {
                const uint32_t i = ch * G + gl;
                const double k = (ok && i < C) ? L.cKey()[i] : 0.0;
                const bool pred = ok && i < C && k >= tau;
                const uint64_t gb = gballot(pred);
                if (pred) { const uint32_t s = n + gprefix(gb); L.sKey()[s] = k; L.sIdx()[s] = i; }
                n += (uint32_t)__popcll(gb);
            }
            nsd_wave_sync();
            if (ok) { keys = L.sKey(); compacted = true; }
        }
        NSD_STAMP(9);
        // ---- rank among the F <= 2G survivors: two per lane ----
        uint32_t claimed = 0;
        {
            const uint32_t f0 = (uint32_t)gl, f1 = (uint32_t)gl + G;
            const bool run = want && !full_scan;
            const bool v0 = run && f0 < F, v1 = run && f1 < F;
            const double k0 = v0 ? keys[f0] : 0.0, k1 = v1 ? keys[f1] : 0.0;
            uint32_t r0 = 0, r1 = 0;
            const uint32_t Fmax = wave_max(run ? F : 0u);
            if (Fmax <= (uint32_t)G) {
                // one candidate per lane; reads past a group's F compare against that group's own stale
                // buffer entries and are discarded by the q < F test
                if (UPW == 1) {
                    // two keys per LDS read, eight reads in flight
                    const double2 *k2 = reinterpret_cast<const double2 *>(keys);
                    const uint32_t pairs = Fmax >> 1;
#pragma unroll 8
                    for (uint32_t q = 0; q < pairs; ++q) {
                        const double2 k = k2[q];
                        r0 += (k.x > k0) ? 1u : 0u;
                        r0 += (k.y > k0) ? 1u : 0u;
                    }
                    if (Fmax & 1u) r0 += (keys[Fmax - 1] > k0) ? 1u : 0u;
                } else {
                    // several utterances per wave: each group reads its own keys, pairs past its F are skipped
                    const double2 *k2 = reinterpret_cast<const double2 *>(keys);
                    const uint32_t pairs = Fmax >> 1, mine = run ? F >> 1 : 0u;
#pragma unroll 4
                    for (uint32_t q = 0; q < pairs; ++q) {
                        if (q < mine) {
                            const double2 k = k2[q];
                            r0 += (k.x > k0) ? 1u : 0u;
                            r0 += (k.y > k0) ? 1u : 0u;
                        }
                    }
                    if (run && (F & 1u)) r0 += (keys[F - 1] > k0) ? 1u : 0u;
                }
            } else if (UPW == 1) {
                const double2 *k2 = reinterpret_cast<const double2 *>(keys);
                const uint32_t pairs = Fmax >> 1;
#pragma unroll 4
                for (uint32_t q = 0; q < pairs; ++q) {
                    const double2 k = k2[q];
                    r0 += (k.x > k0) ? 1u : 0u;
                    r1 += (k.x > k1) ? 1u : 0u;
                    r0 += (k.y > k0) ? 1u : 0u;
                    r1 += (k.y > k1) ? 1u : 0u;
                }
                if (Fmax & 1u) {
                    const double k = keys[Fmax - 1];
                    r0 += (k > k0) ? 1u : 0u;
                    r1 += (k > k1) ? 1u : 0u;
                }
            } else {
                const double2 *k2 = reinterpret_cast<const double2 *>(keys);
                const uint32_t pairs = Fmax >> 1, mine = run ? F >> 1 : 0u;
#pragma unroll 2
                for (uint32_t q = 0; q < pairs; ++q) {
                    if (q < mine) {
                        const double2 k = k2[q];
                        r0 += (k.x > k0) ? 1u : 0u;
                        r1 += (k.x > k1) ? 1u : 0u;
                        r0 += (k.y > k0) ? 1u : 0u;
                        r1 += (k.y > k1) ? 1u : 0u;
                    }
                }
                if (run && (F & 1u)) {
                    const double k = keys[F - 1];
                    r0 += (k > k0) ? 1u : 0u;
                    r1 += (k > k1) ? 1u : 0u;
                }
            }
            const bool w0 = v0 && r0 < (uint32_t)nsel, w1 = v1 && r1 < (uint32_t)nsel;
            if (w0) L.sel(r0) = compacted ? L.sIdx()[f0] : f0;
            if (w1) L.sel(r1) = compacted ? L.sIdx()[f1] : f1;
            claimed = (uint32_t)__popcll(gballot(w0)) + (uint32_t)__popcll(gballot(w1));
            NSD_STAMP(12);
        }
        nsd_wave_sync();
        // equal keys show up as a rank claimed twice: inside the top-B that leaves a hole, at its lower
        // edge it makes more claimants than places
        const bool hole = want && gl < nsel && L.sel(gl) == SEL_NONE;
        const bool redo = want && (full_scan || claimed != (uint32_t)nsel || gballot(hole) != 0);
        if (__builtin_amdgcn_ballot_w64(redo) != 0) {
            NSD_COUNT(dbg_redo);
            if (redo && gl < B) L.sel(gl) = SEL_NONE;
            nsd_wave_sync();
            const uint32_t Cmax = wave_max(redo ? C : 0u);
            for (uint32_t ch = 0; ch < chunks; ++ch) {
                const uint32_t i = ch * G + gl;
                const bool valid = redo && i < C;
                const double mk = valid ? L.cKey()[i] : 0.0;
                const uint32_t mo = valid ? ((L.cInfo()[i] >> 8) & 0xFFFu) : 0u;
                uint32_t rank = 0;
                for (uint32_t q = 0; q < Cmax; ++q) {
                    if (redo && q < C) {
                        const double k = L.cKey()[q];
                        const uint32_t o = (L.cInfo()[q] >> 8) & 0xFFFu;
                        rank += (k > mk || (k == mk && (o < mo || (o == mo && q < i)))) ? 1u : 0u;
                    }
                }
                if (valid && rank < (uint32_t)nsel) L.sel(rank) = i;
            }
            nsd_wave_sync();
        }
    };

    // prune the buffer of the groups with `over` to its top-B, moved to the front
    auto prune = [&](bool over) {
        const int keep = (int)C < B ? (int)C : B;
        NSD_COUNT(dbg_prunes);
        select_top(over, keep);
        uint32_t info = 0;
        double key = 0.0, sc = 0.0;
        if (over && gl < keep) {
            const uint32_t s = L.sel(gl);
            info = L.cInfo()[s]; key = L.cKey()[s];
            if (VARIANT == NSD_VARIANT_E) sc = L.cScore()[s];
        }
        nsd_wave_sync();
        if (over && gl < keep) {
            L.cInfo()[gl] = info; L.cKey()[gl] = key;
            if (VARIANT == NSD_VARIANT_E) L.cScore()[gl] = sc;
        }
        if (over) C = (uint32_t)keep;
        nsd_wave_sync();
        if (DENSE) {                               // the B-th best key so far is a threshold for the rows that follow
            if (over && keep == B && gl == 0) {
                const double kth = L.cKey()[B - 1];
                if (kth > L.tauSeen()) L.tauSeen() = kth;
            }
            nsd_wave_sync();
        }
    };

    // finished hypotheses compete unchanged (ns_utils.py:222-224); they are scored with the first pass
    emit(act && gl < nbeams && !my_live, 0.0f, pack_info(0x3f, K_CARRY, 0, (uint32_t)gl));
    // when everything the step can emit fits the buffer (the usual case) no pass needs an overflow check
    const bool may_overflow = __builtin_amdgcn_ballot_w64(act && L.total()[0] > CAP) != 0;

    // ---------------- passes: log-softmax, candidates, scores ----------------
    const int npass = act ? ((QUAD ? nbeams : nlive) + RPP - 1) / RPP : 0;
    // with rows by position and a beam that one pass covers, the loop below runs at most once: the instantiation
    // knows it (no next-pass request, no copy of the requested row)
    constexpr bool ONE_PASS = QUAD && BT > 0 && BT <= RPP;
    const int npass_max = ONE_PASS ? (wave_max((uint32_t)npass) != 0u ? 1 : 0) : (int)wave_max((uint32_t)npass);
    auto fetch_row = [&](int pass) {
        const int slot = pass * RPP + rslot;
        if (act && slot < (QUAD ? nbeams : nlive)) {
            const uint32_t hyp = QUAD ? (uint32_t)slot : L.live(slot);
            const float *row = fresh_args()->logits + (((uint32_t)u * (uint32_t)B + hyp) * (uint32_t)NSD_V + (uint32_t)(j * EPL));
            if (QUAD) {                            // 40-byte pieces: 8-byte aligned
#pragma unroll
                for (int k = 0; k < EPL / 2; ++k) {
                    const float2 v = reinterpret_cast<const float2 *>(row)[k];
                    xn[2 * k] = v.x; xn[2 * k + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < EPL; ++k) xn[k] = row[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < EPL; ++k) xn[k] = 0.0f;
        }
    };
    if (!QUAD && npass_max > 0) fetch_row(0);
    for (int pass = 0; pass < npass_max; ++pass) {
        const int slot = pass * RPP + rslot;
        bool live;
        int r;
        uint32_t meta;
        if (QUAD) {
            r = (act && slot < nbeams) ? slot : 0;
            meta = L.bMeta(r);
            live = act && slot < nbeams && !(meta & NSD_F_FINISHED);
        } else {
            live = act && slot < nlive;
            r = live ? (int)L.live(slot) : 0;
            meta = L.bMeta(r);
        }
        const uint64_t mask = live ? L.bMask(r) : 0;
        const int glen = (int)(meta >> 8);
        const bool whole = (mask >> 37) & 1;
        const bool pend = (meta & NSD_F_PENDING_SPACE) != 0;
        const bool eos_u = live && VARIANT == NSD_VARIANT_U && P.allow_eos_anytime;
        const bool eos_e = live && VARIANT == NSD_VARIANT_E && mask != 0 &&
                           glen >= P.min_len_for_eos && (pend || whole);
        const bool fb = live && mask == 0 && !eos_e;
        int kfb = B;
        if (kfb > NSD_V) { kfb = NSD_V; if (VARIANT != NSD_VARIANT_U && fb) err_topk = true; }

        float x[EPL];
#pragma unroll
        for (int k = 0; k < EPL; ++k) x[k] = xn[k];
        if (!ONE_PASS && pass + 1 < npass_max) fetch_row(pass + 1);
        if (pass == 0) NSD_STAMP(6);
        if constexpr (QUAD) nsd_row_log_softmax10(x);
        else nsd_row_log_softmax5(x);

        // With 4 lanes per row a pass covers twice the rows the buffer is sized for, and the fall-back's row
        // buffer holds half of them: when the pass has a fall-back row or might not fit the buffer (both rare
        // unless the lexicon is dense), its rows emit their candidates in two halves; with a dense lexicon only
        // once a pass turns out not to fit.
        bool split = QUAD && ((!DENSE && may_overflow) || __builtin_amdgcn_ballot_w64(fb) != 0);
        int nhalf = split ? 2 : 1;
#pragma unroll 1
        for (int eh = 0; eh < nhalf; ++eh) {
            bool part = !split || (rslot / (RPP / 2)) == eh;

            // this lane's lexicon-allowed tokens of its row
            uint32_t own = part ? (uint32_t)((mask & ((1ull << NSD_NBASE) - 1ull)) >> (j * EPL)) & ((1u << EPL) - 1u) : 0u;
            uint32_t n = 0, incl = 0, total = 0;
            // One prefix sum over the lanes' candidate counts places all of a lane's tokens (a ballot per token costs
            // more, most of all with several utterances per wave).  The order of candidates in the buffer carries no
            // meaning: ties are broken by the order field of the info word.
            auto count_own = [&]() {
                n = (uint32_t)__popc(own);
                incl = n;
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);         // row_shr:1
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);         // row_shr:2
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);         // row_shr:4
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, true);         // row_shr:8
                if (G >= 32) incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, true);   // row_bcast:15 into rows 1, 3
                if (G == 64) incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, true);   // row_bcast:31 into rows 2, 3
                if (G == 64) {
                    total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                } else if (G == 32) {
                    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31), t1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    total = g ? t1 : t0;
                } else {
                    total = (uint32_t)__shfl((int)incl, gshift + G - 1);
                }
            };
            if constexpr (!DENSE) {
                // the buffer must hold what these rows can add; otherwise prune it to its top-B first
                if (may_overflow) {
                    uint32_t add = 0;
                    if (live && part && j == 0)
                        add = fb ? (uint32_t)kfb + (eos_u ? 1u : 0u)
                                 : (uint32_t)__popcll(mask & ((1ull << NSD_NBASE) - 1)) + ((eos_u || eos_e) ? 1u : 0u);
#pragma unroll
                    for (int off = RL; off < G; off <<= 1) add += __shfl_xor(add, off);
                    add = __shfl(add, gshift);
                    bool over = act && C + add > CAP;
                    if (__builtin_amdgcn_ballot_w64(over && C > (uint32_t)G) != 0) {
                        nsd_wave_sync();
                        bound_prune(over && C > (uint32_t)G);          // cheap: no look-ups, no ranking
                        over = act && C + add > CAP;
                    }
                    if (__builtin_amdgcn_ballot_w64(over) != 0) {
                        NSD_TIC();
                        // score_new covers every group of the wave; when it does not run, the groups that are not
                        // over keep their unscored tail
                        if (__builtin_amdgcn_ballot_w64(over && scored < C) != 0) { nsd_wave_sync(); score_new(scored, false); nsd_wave_sync(); scored = C; }
                        NSD_TOC(dbg_t_score);
                        NSD_TIC();
                        prune(over);
                        NSD_TOC(dbg_t_prune);
                        if (over) scored = C;
                    }
                }
                count_own();
            } else {
                bool swept = false;
#pragma unroll 1
                for (;;) {
                    if (may_overflow) {
                        // Rows that emit after a sweep drop, before anything is written, the characters that cannot
                        // reach the threshold the sweep established: B candidates with a key of at least that value
                        // exist whatever follows, and the upper bound is the one the sweep itself would compute for
                        // the candidate.
                        const double tau_seen = L.tauSeen();
                        const bool known = act && tau_seen > -__builtin_inf();
                        if (__builtin_amdgcn_ballot_w64(known && own != 0) != 0) {
                            if (known && own != 0) {
                                const double sb = L.bScore(r), ge = L.bGe(r);
                                const bool has_int = 1 + glen + 1 >= order, full_new = 1 + glen + 2 >= order;
                                const double sub_hi = !(P.beta < 0.0) ? a.tab.max_logp : a.tab.min_logp;
                                const double den = VARIANT == NSD_VARIANT_E ? P.len_denom[glen + 1] : 1.0;
#pragma unroll
                                for (int k = 0; k < EPL; ++k) {
                                    if (j * EPL + k == (int)NSD_SPACE) continue;      // a space is its own (exact) bound
                                    double hi = char_score(sb, (double)x[k], ge, has_int, sub_hi, full_new, sub_hi, use_lm, P.beta);
                                    if (VARIANT == NSD_VARIANT_E) hi = hi / den;
                                    if (hi < tau_seen) own &= ~(1u << k);
                                }
                            }
                        }
                    }
                    count_own();
                    if (!may_overflow) break;

                    // the buffer must hold what these rows add (their characters, an EOS option each, the picks of a
                    // fall-back row); otherwise it is swept, and pruned to its top-B if need be
                    const uint32_t n_fb = (uint32_t)__popcll(gballot(fb && part && j == 0));
                    const uint32_t n_eos = (uint32_t)__popcll(gballot(live && part && j == 0 && (eos_u || eos_e)));
                    const uint32_t add = total + n_fb * (uint32_t)kfb + n_eos;
                    const bool over = act && C + add > CAP;
                    if (QUAD && !split && __builtin_amdgcn_ballot_w64(over) != 0) {
                        split = true;                                  // the whole pass does not fit: first half now
                        nhalf = 2;
                        part = (rslot / (RPP / 2)) == eh;
                        if (!part) own = 0;
                        continue;
                    }
                    if (!swept && __builtin_amdgcn_ballot_w64(over && C > (uint32_t)G) != 0) {
                        nsd_wave_sync();
                        bound_prune(over && C > (uint32_t)G);          // cheap: no look-ups, no ranking
                        swept = true;
                        continue;                                      // these rows are filtered again with its threshold
                    }
                    if (__builtin_amdgcn_ballot_w64(over) != 0) {
                        NSD_TIC();
                        // score_new covers every group of the wave; when it does not run, the groups that are not
                        // over keep their unscored tail
                        if (__builtin_amdgcn_ballot_w64(over && scored < C) != 0) { nsd_wave_sync(); score_new(scored, false); nsd_wave_sync(); scored = C; }
                        NSD_TOC(dbg_t_score);
                        NSD_TIC();
                        prune(over);
                        NSD_TOC(dbg_t_prune);
                        if (over) scored = C;
                    }
                    break;
                }
            }

            {
                uint32_t s = C + incl - n;
                const bool room = C + incl <= CAP;               // all of this lane's tokens fit
                if (!room && n) err_cap = true;
                const uint32_t v0 = (uint32_t)(j * EPL);
                const uint32_t info0 = pack_info(v0, K_CHAR, (VARIANT == NSD_VARIANT_U) ? v0 + 1u : v0, (uint32_t)r);
#pragma unroll
                for (int k = 0; k < EPL; ++k) {
                    const bool pred = room && (own & (1u << k)) != 0;
                    if (pred) {
                        L.cInfo()[s] = info0 + (uint32_t)k * 0x101u;     // token and order both advance by one
                        reinterpret_cast<float *>(L.cKey())[2 * s] = x[k];
                    }
                    s += pred ? 1u : 0u;
                }
                C += total;
                if (C > CAP) C = CAP;
            }
            // EOS option (variants U and E): token 38 is element 3 of the eighth lane of a row, element 8 of the fourth
            if (VARIANT != NSD_VARIANT_I)
                emit((eos_u || eos_e) && part && j == RL - 1, x[NSD_EOS - (RL - 1) * EPL],
                     pack_info(NSD_EOS, K_EOS, (VARIANT == NSD_VARIANT_U) ? 0u : 40u, (uint32_t)r));

            // fall-back: no lexicon-consistent option -> neural top-k over all 40 ids
            if (__builtin_amdgcn_ballot_w64(fb && part) != 0) {
                float *rb = L.rowbuf() + (QUAD ? rslot % (RPP / 2) : rslot) * NSD_V;
                if (part) {
#pragma unroll
                    for (int k = 0; k < EPL; ++k) rb[j * EPL + k] = x[k];
                }
                nsd_wave_sync();
                uint32_t rank[EPL];
#pragma unroll
                for (int k = 0; k < EPL; ++k) rank[k] = 0;
#pragma unroll 1
                for (int i = 0; i < NSD_V; ++i) {
                    const float o = rb[i];
#pragma unroll
                    for (int k = 0; k < EPL; ++k)
                        rank[k] += (o > x[k] || (o == x[k] && i < j * EPL + k)) ? 1u : 0u;
                }
#pragma unroll
                for (int k = 0; k < EPL; ++k) {
                    const uint32_t v = (uint32_t)(j * EPL + k);
                    const uint32_t w = rank[k] + (VARIANT == NSD_VARIANT_U ? 1u : 0u);
                    emit(fb && part && rank[k] < (uint32_t)kfb, x[k], pack_info(v, K_FALLBACK, w, (uint32_t)r));
                }
                nsd_wave_sync();                   // rowbuf is reused by the next half / pass
            }
        }
    }
    NSD_STAMP(2);
    nsd_wave_sync();
    // many candidates: bound them away before a table look-up is spent on them (repeated only while the buffer still
    // exceeds the threshold and the last sweep removed something)
    {
        // measured: with one utterance per wave the sweep pays from a quarter round beyond the first (mean launch
        // 17.1 -> 16.9 us at beam 16, the busiest step 33 -> 19 us; from 64 candidates on it costs more than it
        // saves, 17.7 us; at beam 32, where a step has some 160 candidates, the lower threshold only repeats the
        // sweep: 47.9 against 46.8 us); with two or four utterances per wave it pays from two rounds (281 us against
        // 308 from one)
        const uint32_t NSD_SWEEP_AT = (G == 64 && B <= 16) ? 80u : 2u * (uint32_t)G;
        bool again = act && C > NSD_SWEEP_AT;
        for (int it = 0; it < 3 && __builtin_amdgcn_ballot_w64(again) != 0; ++it) {
            const uint32_t before = C;
            bound_prune(again);
            again = again && C > NSD_SWEEP_AT && C < before;
        }
    }
    // nothing scored or pruned so far and the upper half of the buffer free: transitions can be kept
    const bool kept = act && scored == 0 && 2u * C <= CAP;
    score_new(scored, kept);
    nsd_wave_sync();
    NSD_STAMP(3);

    // ---------------- final selection ----------------
    const int nsel = (int)C < B ? (int)C : B;
    select_top(act, nsel);
    NSD_STAMP(4);
    NSD_STAMP_VALUE(10, C);
    NSD_STAMP_VALUE(11, nlive);
#ifdef NSD_STAMPS
    NSD_STAMP_VALUE(12, dbg_redo);
    NSD_STAMP_VALUE(13, dbg_iters);
    NSD_STAMP_VALUE(14, dbg_prunes);
    NSD_STAMP_VALUE(15, dbg_slowthr);
    NSD_STAMP_VALUE(8, dbg_t_prune);
    NSD_STAMP_VALUE(7, dbg_t_score);
#endif

    // ---------------- write the surviving hypotheses ----------------
    bool fin = true;
    StepArgsK *W = fresh_args();                   // output pointers are fetched here, not held from the top
    const NsdTables WT = fresh_tables<ORD>(W);
    if (act && gl < nsel) {
        uint32_t mysel = L.sel(gl);
        if (mysel >= C) mysel = 0;                 // only reachable with NaN keys
        const uint32_t info = L.cInfo()[mysel];
        const uint32_t tok = info & 0x3f, kind = (info >> 6) & 3, p = (info >> 14) & 0x3f;
        const double score = (VARIANT == NSD_VARIANT_E) ? L.cScore()[mysel] : L.cKey()[mysel];
        uint32_t meta = L.bMeta(p);
        uint32_t flags = meta & 0xffu;
        int glen = (int)(meta >> 8);
        double G0 = L.bG(p), Ge = L.bGe(p);
        uint64_t ctx = L.bCtx(p);
        uint64_t hist = L.bHist(p);
        uint32_t node = L.bNode(p);
        uint32_t ctx_id = L.bCtxId(p), ctxe_id = L.bCtxeId(p);
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
                if (use_lm && kept) {
                    Ge = L.cKey()[CAP - 1u - mysel];
                    ctxe_id = L.cInfo()[CAP - 1u - mysel];
                } else if (use_lm) {
                    const NsdLmStep s = nsd_lm_advance(WT, 1 + glen, ctx, ctx_id, L.bSym(p), L.bLmBase(p),
                                                       L.bBo(p), NSD_SPACE + 1u);
                    Ge = s.has_int ? G0 + s.lint : G0;
                    ctxe_id = s.full_new ? s.next_id : NSD_CTX_NONE;
                }
                flags |= NSD_F_PENDING_SPACE;
                node = 0;
            } else {
                if (use_lm && kept) {
                    const uint64_t ctxe = pend ? nsd_ctx_shift(WT, ctx, NSD_SPACE + 1u) : ctx;
                    G0 = Ge = L.cKey()[CAP - 1u - mysel];
                    ctx = nsd_ctx_shift(WT, ctxe, tok + 1u);
                    ctx_id = ctxe_id = L.cInfo()[CAP - 1u - mysel];
                } else if (use_lm) {
                    const uint64_t ctxe = pend ? nsd_ctx_shift(WT, ctx, NSD_SPACE + 1u) : ctx;
                    const NsdLmStep s = nsd_lm_advance(WT, 1 + glen, ctxe, ctxe_id, L.bSym(p), L.bLmBase(p),
                                                       L.bBo(p), tok + 1u);
                    G0 = s.has_int ? Ge + s.lint : Ge;
                    Ge = G0;
                    ctx = s.ctxn;
                    ctx_id = ctxe_id = s.full_new ? s.next_id : NSD_CTX_NONE;
                }
                flags &= ~NSD_F_PENDING_SPACE;
                const uint64_t m = L.bMask(p);
                node = L.bChild(p) + (uint32_t)__popcll(m & ((1ull << tok) - 1ull) & ((1ull << 36) - 1ull));
            }
            glen += 1;
        }
        meta = flags | ((uint32_t)glen << 8);
        fin = (flags & NSD_F_FINISHED) != 0;
        const uint32_t o = (uint32_t)u * (uint32_t)B + (uint32_t)gl;
        W->out.score[o] = score; W->out.lmsum[o] = G0; W->out.lmsum_e[o] = Ge; W->out.ctx[o] = ctx; W->out.hist[o] = hist;
        W->out.ctx_id[o] = ctx_id; W->out.ctxe_id[o] = ctxe_id;
        W->out.node[o] = node;   W->out.meta[o] = meta;
        W->backptr[o] = (uint8_t)p;
        W->token[o] = out_tok;
    }
    const bool all_fin = gballot(!fin) == 0;
    const bool any_err = gballot(err_topk) != 0;
    const bool any_cap = gballot(err_cap) != 0;
    if (act && gl == 0) {
        uint32_t nu = (uint32_t)nsel | (((um >> NSD_U_STEPS_SHIFT) + 1u) << NSD_U_STEPS_SHIFT) |
                      (um & (NSD_U_ERR_TOPK | NSD_U_ERR_INTERNAL));
        if (all_fin) nu |= NSD_U_DONE;
        if (any_err) nu |= NSD_U_ERR_TOPK;
        if (any_cap) nu |= NSD_U_ERR_INTERNAL;
        W->out.utt[u] = nu;
        if (all_fin && W->out.done_count) atomicAdd(W->out.done_count, 1u);
    }
    NSD_STAMP(5);
}

template <int G, int VARIANT, int BT, int ORD, bool QUAD, bool DENSE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
nsd_step_kernel(StepArgs a, NsdParams P)
{
#ifdef NSD_REPEAT
    // Diagnostic build only: the whole step runs NSD_REPEAT times on the same input (the caller passes distinct
    // in / out states), so that stamps of the later runs show a wave's time with caches and translations warm.
    for (int rep = 0; rep < NSD_REPEAT; ++rep) {
        nsd_step_body<G, VARIANT, BT, ORD, QUAD, DENSE>(a, P, rep);
        __builtin_amdgcn_s_waitcnt(0);
        nsd_wave_sync();
    }
#else
    // The workgroups that are resident when a launch starts (four per CU) would run their phases in step, the four
    // waves of a SIMD asking for the vector unit at the same moments and waiting for memory at the same moments.
    // The second half of them starts 1 024 cycles late (measured: 16.2 -> 16.0 us per launch at 4 096 utterances,
    // 10.8 -> 10.65 us for the copy with an EOS option, nothing at other launch sizes; delays of 320-1 536 cycles
    // spread over four groups do the same, 2 560 cycles and groupings that do not follow the dispatch order lose).
    if (blockIdx.x >= 512u && blockIdx.x < 1024u) __builtin_amdgcn_s_sleep(16);
    nsd_step_body<G, VARIANT, BT, ORD, QUAD, DENSE>(a, P, 0);
#endif
}

#ifdef NSD_STAMPS
extern "C" int nsd_debug_set_stamps(unsigned long long *p)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nsd_stamps), &p, sizeof(p));
}
#endif

template <int G, int VARIANT, int BT, int ORD, bool QUAD, bool DENSE>
static void launch_gvd(const StepArgs &a, const NsdParams &p, int N, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    constexpr int UPW = 64 / G;
    const uint32_t lds_per_wave = a.lds_per_utt * UPW;
    // waves per workgroup: the count (4, 2 or 1) that lets most waves share a CU's 160 KiB of LDS, among
    // those whose workgroup stays within the 64 KiB of dynamic LDS a launch may ask for without opting in
    const uint32_t lds_fixed = a.lds_nodes * 16u;
    int waves = 1, best = 0;
    for (int w = 4; w >= 1; w >>= 1) {
        const uint32_t tot = lds_fixed + lds_per_wave * (uint32_t)w;
        if (tot > 65536u) continue;
        const int per_cu = (int)(163840u / tot) * w;
        if (per_cu > best) { best = per_cu; waves = w; }
    }
    const long long n_waves = ((long long)N + UPW - 1) / UPW;
    const unsigned grid = (unsigned)((n_waves + waves - 1) / waves);
    if (ev0 || ev1)      // the events take the kernel's own start / end timestamps (no marker packets in between)
        hipExtLaunchKernelGGL((nsd_step_kernel<G, VARIANT, BT, ORD, QUAD, DENSE>), dim3(grid), dim3(64 * waves),
                              lds_fixed + lds_per_wave * (uint32_t)waves, s, ev0, ev1, 0, a, p);
    else
        hipLaunchKernelGGL((nsd_step_kernel<G, VARIANT, BT, ORD, QUAD, DENSE>), dim3(grid), dim3(64 * waves),
                           lds_fixed + lds_per_wave * (uint32_t)waves, s, a, p);
}

// Dense lexicon?  The trie's mean fan-out over its upper levels (NsdTables::fanout_x16, sixteenths) times the beam is
// what a step emits when every hypothesis sits at such a node; beyond the candidate buffer the form that filters at
// emission pays.  NSD_STEP_DENSE=0/1 in the environment overrides the guess (both forms give the same bits).
template <int G, int VARIANT, int BT, int ORD, bool QUAD>
static void launch_gv(const StepArgs &a, const NsdParams &p, int N, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    static const int forced = [] { const char *e = getenv("NSD_STEP_DENSE"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }();
    const bool dense = forced >= 0 ? forced == 1 : (uint64_t)a.tab.fanout_x16 * (uint64_t)a.B > 16ull * a.cap;
    if (dense) launch_gvd<G, VARIANT, BT, ORD, QUAD, true>(a, p, N, s, ev0, ev1);
    else       launch_gvd<G, VARIANT, BT, ORD, QUAD, false>(a, p, N, s, ev0, ev1);
}

// The instantiation for this call: beams 8, 16 and 32 with a 5-gram model (the shapes of the benchmark
// configurations) have kernels of their own at 32 and 64 lanes per utterance; everything else runs the form that
// reads beam and order from its arguments.  The copy of the decoder without an EOS option (all hypotheses stay live)
// takes the 4-lanes-per-row form there.  NSD_STEP_MODE in the environment restricts the choice (a debugging aid:
// all forms must give the same bits): 0 = run-time form only, 1 = no 4-lanes-per-row form.
template <int G, int VARIANT, int BT>
static void launch_beam(const StepArgs &a, const NsdParams &p, int N, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, bool quad)
{
    if constexpr (VARIANT == NSD_VARIANT_I) {
        if (quad) return launch_gv<G, VARIANT, BT, 5, true>(a, p, N, s, ev0, ev1);
    }
    launch_gv<G, VARIANT, BT, 5, false>(a, p, N, s, ev0, ev1);
}

template <int G, int VARIANT>
static void launch_shape(const StepArgs &a, const NsdParams &p, int N, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    static const int mode = [] { const char *e = getenv("NSD_STEP_MODE"); return e && e[0] >= '0' && e[0] <= '9' ? e[0] - '0' : 9; }();
    const bool lm5 = p.use_lm != 0 && a.tab.order == 5;
    if constexpr (G >= 32) {
        if (lm5 && mode >= 1) {
            const bool quad = mode >= 2;
            if (a.B == 16) return launch_beam<G, VARIANT, 16>(a, p, N, s, ev0, ev1, quad);
            if (a.B == 8)  return launch_beam<G, VARIANT, 8>(a, p, N, s, ev0, ev1, quad);
            if (a.B == 32) return launch_beam<G, VARIANT, 32>(a, p, N, s, ev0, ev1, quad);
        }
    }
    launch_gv<G, VARIANT, 0, -1, false>(a, p, N, s, ev0, ev1);
}

template <int VARIANT>
static void launch_variant(StepArgs &a, const NsdParams &p, int N, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    const int G = nsd_step_group(a.B, N, p.lanes_per_utt, p.variant);
    a.cap = step_cap(a.B, G, p.variant);
    a.lds_per_utt = step_lds_per_utt(a.B, G, p.variant);
    {   // staged trie records: what was asked for, within the trie, the documented maximum and a workgroup's LDS
        uint32_t kn = p.lds_trie_nodes > 0 ? (uint32_t)p.lds_trie_nodes : 0u;
        if (kn > a.tab.n_nodes) kn = a.tab.n_nodes;
        if (kn > NSD_MAX_LDS_TRIE_NODES) kn = NSD_MAX_LDS_TRIE_NODES;
        while (kn && kn * 16u + a.lds_per_utt * (uint32_t)(64 / G) > 65536u) kn >>= 1;
        a.lds_nodes = kn;
    }
    switch (G) {
    case 16: launch_shape<16, VARIANT>(a, p, N, s, ev0, ev1); break;
    case 32: launch_shape<32, VARIANT>(a, p, N, s, ev0, ev1); break;
    default: launch_shape<64, VARIANT>(a, p, N, s, ev0, ev1); break;
    }
}

// nsd_step with optional events that receive the kernel's own start and end timestamps
int nsd_step_with_events(const NsdTables *t, const NsdParams *p, const float *logits,
                         const NsdBeamState *in, const NsdBeamState *out,
                         uint8_t *backptr, uint8_t *token, uint64_t *mask_dbg,
                         int32_t N, int32_t B, int32_t V, void *stream, hipEvent_t ev0, hipEvent_t ev1)
{
    if (!t || !p) return NSD_E_NULL;
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (V != NSD_V) return NSD_E_VOCAB;
    if (N < 0 || (long long)N * B > (long long)NSD_MAX_HYPS) return NSD_E_SIZE;
    if (p->variant < NSD_VARIANT_U || p->variant > NSD_VARIANT_E) return NSD_E_VARIANT;
    if (N == 0) return NSD_OK;
    if (!logits || !in || !out || !backptr || !token) return NSD_E_NULL;
    StepArgs a;
    a.tab = *t; a.logits = logits; a.in = *in; a.out = *out;
    a.backptr = backptr; a.token = token; a.mask_dbg = mask_dbg;
    a.N = N; a.B = B;
    hipStream_t s = (hipStream_t)stream;
    switch (p->variant) {
    case NSD_VARIANT_U: launch_variant<NSD_VARIANT_U>(a, *p, N, s, ev0, ev1); break;
    case NSD_VARIANT_I: launch_variant<NSD_VARIANT_I>(a, *p, N, s, ev0, ev1); break;
    default:            launch_variant<NSD_VARIANT_E>(a, *p, N, s, ev0, ev1); break;
    }
    return (int)hipGetLastError();
}

extern "C" int nsd_step(const NsdTables *t, const NsdParams *p, const float *logits,
                        const NsdBeamState *in, const NsdBeamState *out,
                        uint8_t *backptr, uint8_t *token, uint64_t *mask_dbg,
                        int32_t N, int32_t B, int32_t V, void *stream)
{
    return nsd_step_with_events(t, p, logits, in, out, backptr, token, mask_dbg, N, B, V, stream, nullptr, nullptr);
}
