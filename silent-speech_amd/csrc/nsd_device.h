// Device-side helpers shared by the NS decoder kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nsdecode.h"

struct NsdTables {
    NsdTrieNode *nodes;
    uint32_t     n_nodes;
    int32_t      order;
    NsdGramSlot *grams;          // hash: n-gram key -> logp (whole-string scoring)
    uint64_t     gram_mask;      // cap - 1
    NsdCtxSlot  *ctxs;           // hash: context key -> backoff, term, automaton id
    uint64_t     ctx_mask;       // cap - 1
    NsdCtxRec   *ctx_recs;       // automaton: one record per seen context
    NsdGramRec  *gram_recs;      // automaton: seen n-grams grouped by context
    uint32_t     n_ctx_recs, n_gram_recs;
    uint32_t     init_ctx;       // automaton id of the empty surface's context, or NSD_CTX_NONE
    uint32_t     fanout_x16;     // mean children per node with children over the trie's upper levels, in sixteenths
    double       unseen;
    uint64_t     ctx_sym_mask;   // keeps the last order-1 symbols of a packed context
    double       lm_start;       // LM sum of the empty surface: the '^' window, which exists for order 1 only
    double       min_logp, max_logp;   // smallest / largest log-prob any table look-up can return
};

#define NSD_SYM_HAT    38u
#define NSD_SYM_DOLLAR 39u
#define NSD_GOLD  0x9E3779B97F4A7C15ull
#define NSD_HIST0 0x6A09E667F3BCC908ull

#define NSD_HD __host__ __device__ __forceinline__
#define NSD_D  __device__ __forceinline__

NSD_HD uint64_t nsd_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

NSD_HD uint32_t nsd_fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

NSD_HD uint64_t nsd_hist_next(uint64_t h, uint32_t tok)
{
    return nsd_mix64(h ^ ((uint64_t)(tok + 1u) * NSD_GOLD));
}

// ---------------------------------------------------------------------
// Deterministic fp32 exp / log: individually rounded fp32 operations and
// fused multiply-adds only, so that a host implementation of the same
// sequence gives the same bits.  Build with -ffp-contract=off.
// ---------------------------------------------------------------------
NSD_D float nsd_det_expf(float d)
{
    if (!(d >= -80.0f)) return 0.0f;
    float k = __builtin_rintf(d * 1.44269504088896341f);
    float r = __builtin_fmaf(k, -0.693359375f, d);
    r = __builtin_fmaf(k, 2.12194440e-4f, r);
    float z = r * r;
    float y = 1.9875691500e-4f;
    y = __builtin_fmaf(y, r, 1.3981999507e-3f);
    y = __builtin_fmaf(y, r, 8.3334519073e-3f);
    y = __builtin_fmaf(y, r, 4.1665795894e-2f);
    y = __builtin_fmaf(y, r, 1.6666665459e-1f);
    y = __builtin_fmaf(y, r, 5.0000001201e-1f);
    y = __builtin_fmaf(y, z, r);
    y = y + 1.0f;
    return __builtin_ldexpf(y, (int)k);
}

NSD_D float nsd_det_logf(float t)
{
    uint32_t bits = __float_as_uint(t);
    int e = (int)((bits >> 23) & 0xffu) - 126;
    float f = __uint_as_float((bits & 0x007fffffu) | 0x3f000000u);
    if (f < 0.707106781186547524f) {
        e -= 1;
        f = (f + f) - 1.0f;
    } else {
        f = f - 1.0f;
    }
    float z = f * f;
    float y = 7.0376836292e-2f;
    y = __builtin_fmaf(y, f, -1.1514610310e-1f);
    y = __builtin_fmaf(y, f, 1.1676998740e-1f);
    y = __builtin_fmaf(y, f, -1.2420140846e-1f);
    y = __builtin_fmaf(y, f, 1.4249322787e-1f);
    y = __builtin_fmaf(y, f, -1.6668057665e-1f);
    y = __builtin_fmaf(y, f, 2.0000714765e-1f);
    y = __builtin_fmaf(y, f, -2.4999993993e-1f);
    y = __builtin_fmaf(y, f, 3.3333331174e-1f);
    y = y * f;
    y = y * z;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = f + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

// Log-softmax of one 40-wide row spread over 8 consecutive lanes, 5
// consecutive elements per lane.  Summation tree: each lane sums its 5
// exponentials left to right, then the 8 partial sums are combined pairwise
// at lane distance 1, 2, 4 (every lane ends with the same total).
// x[] is replaced by the log-probabilities.
// value of the lane at distance 1, 2 (inside the quad) and of the mirrored lane of the 8-lane row, by DPP
NSD_D float nsd_dpp_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
NSD_D float nsd_dpp_xor2(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
NSD_D float nsd_dpp_half_mirror(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
}

// fp64 value of the lane at xor-distance d (a power of two below 64) for reductions in which the lanes of every
// aligned block of d lanes already agree: the mirrored lane of the next larger block then holds what the lane at
// distance d holds, so distances up to 8 are register moves (DPP) instead of LDS-crossbar permutes.
template <int D>
NSD_D double nsd_xchg_f64(double v)
{
    const long long b = __double_as_longlong(v);
    int lo = (int)(uint32_t)b, hi = (int)(uint32_t)((uint64_t)b >> 32);
    if (D == 1) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    } else if (D == 2) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
    } else if (D == 4) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true);       // row_half_mirror
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true);
    } else if (D == 8) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true);       // row_mirror
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true);
    } else {
        lo = __shfl_xor(lo, D);
        hi = __shfl_xor(hi, D);
    }
    return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}

// Two exponentials at once with the packed fp32 instructions (v_pk_mul/fma/add_f32 are IEEE per element,
// so each half has the bits of nsd_det_expf).
typedef float nsd_f2 __attribute__((ext_vector_type(2)));

NSD_D nsd_f2 nsd_det_expf2(nsd_f2 d)
{
    const nsd_f2 t = d * 1.44269504088896341f;
    const nsd_f2 k = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
    nsd_f2 r = __builtin_elementwise_fma(k, (nsd_f2)(-0.693359375f), d);
    r = __builtin_elementwise_fma(k, (nsd_f2)(2.12194440e-4f), r);
    const nsd_f2 z = r * r;
    nsd_f2 y = (nsd_f2)(1.9875691500e-4f);
    y = __builtin_elementwise_fma(y, r, (nsd_f2)(1.3981999507e-3f));
    y = __builtin_elementwise_fma(y, r, (nsd_f2)(8.3334519073e-3f));
    y = __builtin_elementwise_fma(y, r, (nsd_f2)(4.1665795894e-2f));
    y = __builtin_elementwise_fma(y, r, (nsd_f2)(1.6666665459e-1f));
    y = __builtin_elementwise_fma(y, r, (nsd_f2)(5.0000001201e-1f));
    y = __builtin_elementwise_fma(y, z, r);
    y = y + 1.0f;
    nsd_f2 o;
    o.x = d.x >= -80.0f ? __builtin_ldexpf(y.x, (int)k.x) : 0.0f;
    o.y = d.y >= -80.0f ? __builtin_ldexpf(y.y, (int)k.y) : 0.0f;
    return o;
}

NSD_D void nsd_row_log_softmax5(float (&x)[5])
{
    // After the distance-1 and distance-2 steps the four lanes of a quad hold the same value, so the
    // mirrored lane (7 - i) of the other quad carries exactly what the lane at distance 4 carries.
    // Written out: from fmaxf the compiler first re-quiets every operand (v_max x, x, x) and moves the permuted
    // value through a register of its own, 16 instructions for these five.  NaN operands are ignored as by fmaxf.
    float m;
    asm("v_max3_f32 %0, %1, %2, %3\n\t"
        "v_max3_f32 %0, %0, %4, %5\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(m) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]));
    nsd_f2 a = {x[0], x[1]}, b = {x[2], x[3]};
    a = a - m;
    b = b - m;
    const float c = x[4] - m;
    const nsd_f2 ea = nsd_det_expf2(a), eb = nsd_det_expf2(b);
    const float ec = nsd_det_expf(c);
    float s = ea.x;                               // left to right, as a scalar loop would
    s = s + ea.y;
    s = s + eb.x;
    s = s + eb.y;
    s = s + ec;
    s = s + nsd_dpp_xor1(s);
    s = s + nsd_dpp_xor2(s);
    s = s + nsd_dpp_half_mirror(s);
    const float lse = nsd_det_logf(s);
    a = a - lse;
    b = b - lse;
    x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y; x[4] = c - lse;
}

// The same log-softmax with the row spread over 4 consecutive lanes, 10 consecutive elements per lane.  Same
// summation tree, hence the same bits: a lane sums its two groups of five left to right and adds the two sums
// (the distance-1 step of the 8-lane form), then lane distances 1 and 2 here are distances 2 and 4 there.
NSD_D void nsd_row_log_softmax10(float (&x)[10])
{
    float m;
    asm("v_max3_f32 %0, %1, %2, %3\n\t"
        "v_max3_f32 %0, %0, %4, %5\n\t"
        "v_max3_f32 %0, %0, %6, %7\n\t"
        "v_max3_f32 %0, %0, %8, %9\n\t"
        "v_max_f32 %0, %0, %10\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(m) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
          "v"(x[8]), "v"(x[9]));
    nsd_f2 d[5], e[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        d[k] = nsd_f2{x[2 * k], x[2 * k + 1]} - m;
        e[k] = nsd_det_expf2(d[k]);
    }
    float sa = e[0].x;                            // elements 0..4, left to right
    sa = sa + e[0].y;
    sa = sa + e[1].x;
    sa = sa + e[1].y;
    sa = sa + e[2].x;
    float sb = e[2].y;                            // elements 5..9
    sb = sb + e[3].x;
    sb = sb + e[3].y;
    sb = sb + e[4].x;
    sb = sb + e[4].y;
    float s = sa + sb;
    s = s + nsd_dpp_xor1(s);
    s = s + nsd_dpp_xor2(s);
    const float lse = nsd_det_logf(s);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        d[k] = d[k] - lse;
        x[2 * k] = d[k].x;
        x[2 * k + 1] = d[k].y;
    }
}

// ---------------------------------------------------------------------
// n-gram tables
// ---------------------------------------------------------------------
NSD_D bool nsd_gram_find(const NsdTables &t, uint64_t key, double &val)
{
    uint64_t h = nsd_mix64(key) & t.gram_mask;
    for (;;) {
        const ulonglong2 s = *reinterpret_cast<const ulonglong2 *>(&t.grams[h]);
        if (s.x == key) { val = __longlong_as_double((long long)s.y); return true; }
        if (s.x == NSD_KEY_EMPTY) return false;
        h = (h + 1) & t.gram_mask;
    }
}

// The two slots a context key may occupy (two-choice placement): both keys are requested before either is
// looked at, so a look-up costs one memory round trip whether the key is there or not.
NSD_D const NsdCtxSlot *nsd_ctx_find(const NsdTables &t, uint64_t key)
{
    const uint64_t m = nsd_mix64(key);
    const NsdCtxSlot *a = t.ctxs + (m & t.ctx_mask), *b = t.ctxs + ((m >> 32) & t.ctx_mask);
    const uint64_t ka = a->key, kb = b->key;
    return ka == key ? a : (kb == key ? b : nullptr);
}

// id and terminal log-prob of a context (NSD_CTX_NONE / the unseen value when the corpus never showed it); the
// 20 bytes read of each of the two slots are requested together
NSD_D void nsd_ctx_lookup(const NsdTables &t, uint64_t key, uint32_t &id, double &term)
{
    const uint64_t m = nsd_mix64(key);
    const NsdCtxSlot *a = t.ctxs + (m & t.ctx_mask), *b = t.ctxs + ((m >> 32) & t.ctx_mask);
    const ulonglong2 qa = *reinterpret_cast<const ulonglong2 *>(a);    // key, term
    const ulonglong2 qb = *reinterpret_cast<const ulonglong2 *>(b);
    const uint32_t ia = (uint32_t)a->ctx_id, ib = (uint32_t)b->ctx_id;
    const bool ha = qa.x == key, hb = qb.x == key;
    id = ha ? ia : (hb ? ib : NSD_CTX_NONE);
    term = ha ? __longlong_as_double((long long)qa.y) : (hb ? __longlong_as_double((long long)qb.y) : t.unseen);
}

NSD_D double nsd_ctx_backoff(const NsdTables &t, uint64_t ctx)
{
    const NsdCtxSlot *s = nsd_ctx_find(t, ctx);
    return s ? s->backoff : t.unseen;
}

NSD_D double nsd_ctx_term(const NsdTables &t, uint64_t ctx)
{
    const NsdCtxSlot *s = nsd_ctx_find(t, ctx);
    return s ? s->term : t.unseen;
}

// log-prob of the n-gram (ctx, sym); `backoff` is the value for a seen
// context whose n-gram is unseen (already looked up by the caller)
NSD_D double nsd_gram_logp(const NsdTables &t, uint64_t ctx, uint32_t sym, double backoff)
{
    double v;
    return nsd_gram_find(t, (ctx << 6) | sym, v) ? v : backoff;
}

NSD_D uint64_t nsd_ctx_shift(const NsdTables &t, uint64_t ctx, uint32_t sym)
{
    return ((ctx << 6) | sym) & t.ctx_sym_mask;
}

// One transition of the context automaton.  The hypothesis' effective context
// has `me` symbols ('^' + surface), packed key `ctxe`, automaton id `ide`
// (NSD_CTX_NONE when unseen or shorter than order-1) and, when the id is
// valid, the record fields symmask / lmbase / backoff.  Appending symbol code
// `sym`:
//   has_int  the n-gram window ending at the new symbol exists (me+1 >= order);
//            its log-prob is `lint`
//   full_new the new context has order-1 symbols, i.e. the terminal window of
//            the extended text exists (me+2 >= order); its log-prob is `term`
//   ctxn / next_id  key and automaton id of the new context
struct NsdLmStep {
    double   lint, term;
    uint64_t ctxn;
    uint32_t next_id;
    bool     has_int, full_new;
};

NSD_D NsdLmStep nsd_lm_advance(const NsdTables &t, int me, uint64_t ctxe, uint32_t ide,
                               uint64_t symmask, uint32_t lmbase, double backoff, uint32_t sym)
{
    NsdLmStep r;
    const int order = t.order;
    r.has_int = me + 1 >= order;
    r.full_new = me + 2 >= order;
    r.ctxn = nsd_ctx_shift(t, ctxe, sym);
    // Either the n-gram is in the automaton (one record: log-prob, terminal log-prob and id of the next context)
    // or, when the next context has order-1 symbols, that context is looked up in the two-choice table.  Both
    // kinds of read go through the same two 16-byte + 4-byte loads with lane-specific addresses, outside any
    // divergent branch, so a scoring round is one memory round trip whatever mix of lanes it holds.
    const bool seen = r.has_int && ide != NSD_CTX_NONE && ((symmask >> sym) & 1ull);
    const bool ask = r.full_new && !seen;
    const char *pa = reinterpret_cast<const char *>(t.gram_recs), *pb = pa;
    if (seen) {
        pa = pb = reinterpret_cast<const char *>(t.gram_recs + lmbase + __popcll(symmask & ((1ull << sym) - 1ull)));
    }
    if (__builtin_amdgcn_ballot_w64(ask) != 0) {
        if (ask) {
            const uint64_t m = nsd_mix64(r.ctxn);
            pa = reinterpret_cast<const char *>(t.ctxs + (m & t.ctx_mask));
            pb = reinterpret_cast<const char *>(t.ctxs + ((m >> 32) & t.ctx_mask));
        }
    }
    const ulonglong2 qa = *reinterpret_cast<const ulonglong2 *>(pa);
    const ulonglong2 qb = *reinterpret_cast<const ulonglong2 *>(pb);
    const uint32_t ia = *reinterpret_cast<const uint32_t *>(pa + 16), ib = *reinterpret_cast<const uint32_t *>(pb + 16);
    r.next_id = NSD_CTX_NONE;
    r.lint = 0.0;
    r.term = 0.0;
    if (r.has_int) r.lint = ide != NSD_CTX_NONE ? backoff : t.unseen;
    if (seen) {                                   // gram record: logp, term_next | next_ctx
        r.lint = __longlong_as_double((long long)qa.x);
        r.term = __longlong_as_double((long long)qa.y);
        r.next_id = ia;
    } else if (ask) {                             // context slots: key, term | ctx_id
        const bool ha = qa.x == r.ctxn, hb = qb.x == r.ctxn;
        r.next_id = ha ? ia : (hb ? ib : NSD_CTX_NONE);
        r.term = ha ? __longlong_as_double((long long)qa.y) : (hb ? __longlong_as_double((long long)qb.y) : t.unseen);
    }
    return r;
}

// compiler + LDS ordering point between lanes of one wave
NSD_D void nsd_wave_sync()
{
    // LDS only: lanes of a wave exchange data through LDS, never through global memory
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
