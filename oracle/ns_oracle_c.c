/*
 * TEST INFRASTRUCTURE ONLY - C restatement of the Silent-Speech NS decoder.
 *
 * Not part of the shipped library.  It is the checker that the HIP kernels
 * are compared with at sizes the Python restatement (oracle/ns_oracle.py) is
 * too slow for, and it defines on the CPU the two pieces of arithmetic that
 * must be bit-reproducible between host and device:
 *
 *   nso_log_softmax   deterministic fp32 log-softmax (fixed summation tree,
 *                     polynomial exp/log built from individually rounded
 *                     fp32 operations and fused multiply-adds only)
 *   nso_synth_row     integer-hash synthetic logits (see oracle/synth.py)
 *
 * The decoder follows the reference structure on purpose (string surfaces,
 * whole-string n-gram scoring per candidate, stable sort by score), not the
 * flat tables of the HIP path, so that it is an independent check of them.
 *
 * Reference locations restated (paths relative to /root/reference):
 *   beam search  U: scripts/ns_utils.py:194-278   I: inference_emg_llm.py:222-300
 *                E: eval_ar.py:214-317
 *   trie            scripts/ns_utils.py:31-65
 *   n-gram LM       scripts/ns_utils.py:99-109, 138-150
 *   ids_to_text     scripts/ns_utils.py:284-296
 *
 * Limits of this C form: corpus and lexicon bytes are treated as characters,
 * so n-gram statistics equal the reference's only for ASCII corpora (the
 * Python restatement covers the rest); V = 40, beam <= 64, order <= 12.
 *
 * Build: see oracle/Makefile (-ffp-contract=off is required).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NSO_V 40
#define NSO_NBASE 37
#define NSO_SPACE 36
#define NSO_BOS 37
#define NSO_EOS 38
#define NSO_PAD 39
#define NSO_MAXB 64
#define NSO_MAXORDER 12

static const char BASE_CHARS[NSO_NBASE + 1] = "abcdefghijklmnopqrstuvwxyz0123456789 ";

/* ------------------------------------------------------------------ */
/* deterministic fp32 log-softmax                                      */
/* ------------------------------------------------------------------ */
static float det_expf(float d)
{
    /* d <= 0.  Below -80 the term cannot change a sum that contains exp(0). */
    if (!(d >= -80.0f)) return 0.0f;
    float k = rintf(d * 1.44269504088896341f);
    float r = fmaf(k, -0.693359375f, d);
    r = fmaf(k, 2.12194440e-4f, r);
    float z = r * r;
    float y = 1.9875691500e-4f;
    y = fmaf(y, r, 1.3981999507e-3f);
    y = fmaf(y, r, 8.3334519073e-3f);
    y = fmaf(y, r, 4.1665795894e-2f);
    y = fmaf(y, r, 1.6666665459e-1f);
    y = fmaf(y, r, 5.0000001201e-1f);
    y = fmaf(y, z, r);
    y = y + 1.0f;
    return ldexpf(y, (int)k);
}

static float det_logf(float t)
{
    /* t is a normal positive float (here 1 <= t <= 64). */
    uint32_t bits;
    memcpy(&bits, &t, 4);
    int e = (int)((bits >> 23) & 0xffu) - 126;
    bits = (bits & 0x007fffffu) | 0x3f000000u;       /* f in [0.5, 1) */
    float f;
    memcpy(&f, &bits, 4);
    if (f < 0.707106781186547524f) {
        e -= 1;
        f = (f + f) - 1.0f;
    } else {
        f = f - 1.0f;
    }
    float z = f * f;
    float y = 7.0376836292e-2f;
    y = fmaf(y, f, -1.1514610310e-1f);
    y = fmaf(y, f, 1.1676998740e-1f);
    y = fmaf(y, f, -1.2420140846e-1f);
    y = fmaf(y, f, 1.4249322787e-1f);
    y = fmaf(y, f, -1.6668057665e-1f);
    y = fmaf(y, f, 2.0000714765e-1f);
    y = fmaf(y, f, -2.4999993993e-1f);
    y = fmaf(y, f, 3.3333331174e-1f);
    y = y * f;
    y = y * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* out[i] = (x[i] - max) - log(sum exp(x - max)), V <= 64.
 * Summation tree: 8 groups of ceil(V/8) consecutive elements, each summed
 * left to right, then the 8 group sums combined pairwise (1, 2, 4 apart). */
void nso_log_softmax(const float *x, int V, float *out)
{
    int per = (V + 7) / 8;
    float m = x[0];
    for (int i = 1; i < V; ++i) m = x[i] > m ? x[i] : m;
    float s[8];
    for (int j = 0; j < 8; ++j) {
        float acc = 0.0f;
        for (int k = 0; k < per; ++k) {
            int i = j * per + k;
            float e = (i < V) ? det_expf(x[i] - m) : 0.0f;
            acc = (k == 0) ? e : acc + e;
        }
        s[j] = acc;
    }
    float t0 = s[0] + s[1], t2 = s[2] + s[3], t4 = s[4] + s[5], t6 = s[6] + s[7];
    float u0 = t0 + t2, u4 = t4 + t6;
    float total = u0 + u4;
    float lse = det_logf(total);
    for (int i = 0; i < V; ++i) out[i] = (x[i] - m) - lse;
}

/* ------------------------------------------------------------------ */
/* synthetic logits (oracle/synth.py)                                  */
/* ------------------------------------------------------------------ */
static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
#define GOLD 0x9E3779B97F4A7C15ull
#define HIST0 0x6A09E667F3BCC908ull

uint64_t nso_hist_next(uint64_t h, int tok) { return mix64(h ^ ((uint64_t)(tok + 1) * GOLD)); }
uint64_t nso_utt_seed(uint64_t base, uint64_t utt) { return mix64(base + utt * 0xD1B54A32D192ED03ull); }

void nso_synth_row(uint64_t useed, uint64_t hist, int step, int tgt_tok, float margin, float *out)
{
    uint64_t k = mix64(hist ^ mix64(useed + (uint64_t)step * GOLD));
    uint32_t rk = (uint32_t)(k ^ (k >> 32));
    for (int v = 0; v < NSO_V; ++v) {
        uint32_t h = fmix32(rk + (uint32_t)v * 0x9E3779B9u);
        float r = (float)(h >> 8);
        float x = (r * 5.9604644775390625e-08f - 0.5f) * 4.0f;
        if (v == tgt_tok) x = x + margin;
        out[v] = x;
    }
}

/* ------------------------------------------------------------------ */
/* model: trie + n-gram counts                                         */
/* ------------------------------------------------------------------ */
typedef struct {
    int32_t child[NSO_NBASE];   /* -1 = none; index NSO_SPACE = literal space edge */
    uint8_t end;
} TNode;

typedef struct {
    char key[NSO_MAXORDER];
    int32_t count;
    uint8_t used;
} GSlot;

typedef struct {
    GSlot *slots;
    uint64_t cap;               /* power of two */
    uint64_t n;
    int klen;
} GMap;

typedef struct NsoModel {
    TNode *nodes;
    int n_nodes, cap_nodes;
    int order;                  /* 0 = no LM */
    int vsz;                    /* value used in the smoothing denominator */
    GMap grams, ctxs;
    uint8_t seen[256];
} NsoModel;

static int base_index(unsigned char c)
{
    if (c >= 'a' && c <= 'z') return c - 'a';
    if (c >= '0' && c <= '9') return 26 + (c - '0');
    if (c == ' ') return NSO_SPACE;
    return -1;
}

static int new_node(NsoModel *m)
{
    if (m->n_nodes == m->cap_nodes) {
        m->cap_nodes = m->cap_nodes ? m->cap_nodes * 2 : 1024;
        m->nodes = (TNode *)realloc(m->nodes, sizeof(TNode) * (size_t)m->cap_nodes);
    }
    TNode *t = &m->nodes[m->n_nodes];
    for (int i = 0; i < NSO_NBASE; ++i) t->child[i] = -1;
    t->end = 0;
    return m->n_nodes++;
}

static uint64_t hash_bytes(const char *p, int n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 0x100000001b3ull; }
    return mix64(h);
}

static void gmap_init(GMap *g, int klen)
{
    g->cap = 1024; g->n = 0; g->klen = klen;
    g->slots = (GSlot *)calloc(g->cap, sizeof(GSlot));
}

static GSlot *gmap_find(const GMap *g, const char *key)
{
    uint64_t i = hash_bytes(key, g->klen) & (g->cap - 1);
    for (;;) {
        GSlot *s = &g->slots[i];
        if (!s->used) return s;
        if (memcmp(s->key, key, (size_t)g->klen) == 0) return s;
        i = (i + 1) & (g->cap - 1);
    }
}

static void gmap_add(GMap *g, const char *key)
{
    if ((g->n + 1) * 2 > g->cap) {
        GMap big = *g;
        big.cap = g->cap * 2;
        big.slots = (GSlot *)calloc(big.cap, sizeof(GSlot));
        for (uint64_t i = 0; i < g->cap; ++i)
            if (g->slots[i].used) *gmap_find(&big, g->slots[i].key) = g->slots[i];
        free(g->slots);
        *g = big;
    }
    GSlot *s = gmap_find(g, key);
    if (!s->used) {
        s->used = 1; s->count = 0;
        memset(s->key, 0, NSO_MAXORDER);
        memcpy(s->key, key, (size_t)g->klen);
        g->n++;
    }
    s->count++;
}

static int gmap_get(const GMap *g, const char *key)
{
    const GSlot *s = gmap_find(g, key);
    return s->used ? s->count : 0;
}

/* words: n_words NUL-terminated strings back to back, already stripped and
 * lower-cased, empties removed (scripts/ns_utils.py:59-65 does that).
 * corpus: n_texts NUL-terminated strings, already stripped and lower-cased.
 * floor_vsz: 1 = max(1,|V|) (ns_utils.py:148), 0 = |V| (in-file copies). */
NsoModel *nso_model_create(const char *words, int n_words, const char *corpus, int n_texts,
                           int order, int floor_vsz)
{
    NsoModel *m = (NsoModel *)calloc(1, sizeof(NsoModel));
    new_node(m);
    const char *p = words;
    for (int w = 0; w < n_words; ++w) {
        int cur = 0;
        size_t L = strlen(p);
        int complete = 1;
        for (size_t i = 0; i < L; ++i) {
            int b = base_index((unsigned char)p[i]);
            if (b < 0) { complete = 0; break; }   /* subtree below a non-base char is unreachable */
            if (m->nodes[cur].child[b] < 0) {
                int nn = new_node(m);
                m->nodes[cur].child[b] = nn;
            }
            cur = m->nodes[cur].child[b];
        }
        if (complete) m->nodes[cur].end = 1;
        p += L + 1;
    }
    m->order = order;
    if (order > 0) {
        gmap_init(&m->grams, order);
        gmap_init(&m->ctxs, order - 1 > 0 ? order - 1 : 0);
        const char *q = corpus;
        char *buf = NULL; size_t bufcap = 0;
        for (int t = 0; t < n_texts; ++t) {
            size_t L = strlen(q);
            if (L + 3 > bufcap) { bufcap = (L + 3) * 2; buf = (char *)realloc(buf, bufcap); }
            buf[0] = '^'; memcpy(buf + 1, q, L); buf[L + 1] = '$';
            long n = (long)L + 2;
            for (long i = 0; i + order <= n; ++i) {
                gmap_add(&m->grams, buf + i);
                gmap_add(&m->ctxs, buf + i);
                for (int k = 0; k < order; ++k) m->seen[(unsigned char)buf[i + k]] = 1;
            }
            q += L + 1;
        }
        free(buf);
        int v = 0;
        for (int i = 0; i < 256; ++i) v += m->seen[i];
        m->vsz = (floor_vsz && v < 1) ? 1 : v;
    }
    return m;
}

void nso_model_destroy(NsoModel *m)
{
    if (!m) return;
    free(m->nodes);
    if (m->order > 0) { free(m->grams.slots); free(m->ctxs.slots); }
    free(m);
}

int nso_model_nodes(const NsoModel *m) { return m->n_nodes; }
int nso_model_vsz(const NsoModel *m) { return m->vsz; }

/* CharNgramLM.logp (scripts/ns_utils.py:138-150); `text` has no NUL inside. */
double nso_lm_logp(const NsoModel *m, const char *text, int len)
{
    if (len == 0) return -1e9;
    int a = 0, b = len;
    while (a < b && text[a] == ' ') ++a;
    while (b > a && text[b - 1] == ' ') --b;
    char s[512];
    int n = 0;
    s[n++] = '^';
    for (int i = a; i < b && n < 510; ++i) s[n++] = text[i];
    s[n++] = '$';
    double lp = 0.0;
    int order = m->order;
    for (int i = 0; i + order <= n; ++i) {
        int c = gmap_get(&m->grams, s + i);
        int C = gmap_get(&m->ctxs, s + i);
        lp += log((double)(c + 1) / (double)(C + m->vsz + 1));
    }
    return lp;
}

/* ------------------------------------------------------------------ */
/* beam search                                                         */
/* ------------------------------------------------------------------ */
typedef struct {
    int variant;            /* 0 = U, 1 = I, 2 = E */
    int beam;
    int max_new_tokens;
    int use_lm;
    int allow_eos_anytime;
    int min_len_for_eos;
    double beta, gamma, kappa, alpha;
    float margin;
    int lsm_det;            /* must be 1: deterministic log-softmax */
} NsoParams;

typedef struct {
    double score;
    int n_ids;              /* including BOS */
    int text_len;
    int parent, token;
    uint64_t hist;
    uint8_t ids[132];
    char text[132];
} Hyp;

typedef struct { double key; int idx; } SortKey;

static int cmp_key(const void *a, const void *b)
{
    const SortKey *x = (const SortKey *)a, *y = (const SortKey *)b;
    if (x->key > y->key) return -1;
    if (x->key < y->key) return 1;
    return x->idx - y->idx;
}

static double norm_score(double lp, int n_ids, double alpha)
{
    int L = n_ids - 1 > 1 ? n_ids - 1 : 1;
    double denom = pow((double)(5 + L), alpha) / pow(6.0, alpha);
    return lp / denom;
}

/* walk the current word (characters after the last space) */
static int walk_word(const NsoModel *m, const char *text, int len, int *is_word)
{
    int start = len;
    while (start > 0 && text[start - 1] != ' ') --start;
    int cur = 0;
    for (int i = start; i < len; ++i) {
        int b = base_index((unsigned char)text[i]);
        if (b < 0 || m->nodes[cur].child[b] < 0) { *is_word = 0; return -1; }
        cur = m->nodes[cur].child[b];
    }
    *is_word = m->nodes[cur].end;
    return cur;
}

static void extend(Hyp *dst, const Hyp *src, double score, int tok, const char *text, int text_len, int parent)
{
    *dst = *src;
    dst->score = score;
    dst->ids[dst->n_ids++] = (uint8_t)tok;
    dst->hist = nso_hist_next(src->hist, tok);
    memcpy(dst->text, text, (size_t)text_len);
    dst->text_len = text_len;
    dst->parent = parent;
    dst->token = tok;
}

/* Decode one utterance.  Optional trace arrays, each [max_new_tokens][beam]:
 *   tr_parent/tr_token (token = -1 for a carried finished hypothesis),
 *   tr_score; tr_nbeams [max_new_tokens]; tr_mask [max_new_tokens][beam]
 *   (mask of parent rank r at that step, 0 if finished / fall-back).
 * Returns the number of ids written to out_ids (>= 1), or -1 when the
 * fall-back asks for more than V candidates in variants I/E (the reference
 * raises from torch.topk there). */
static int decode_one(const NsoModel *m, const NsoParams *P, uint64_t useed,
                      const int8_t *target, int target_len,
                      uint8_t *out_ids, double *out_score, int *out_steps,
                      int8_t *tr_parent, int8_t *tr_token, double *tr_score,
                      int32_t *tr_nbeams, uint64_t *tr_mask)
{
    const int B = P->beam;
    const int maxc = B * (NSO_V + 2);
    Hyp *hyps = (Hyp *)malloc(sizeof(Hyp) * (size_t)B);
    Hyp *cands = (Hyp *)malloc(sizeof(Hyp) * (size_t)maxc);
    SortKey *keys = (SortKey *)malloc(sizeof(SortKey) * (size_t)maxc);
    int nh = 1, err = 0, steps = 0;
    memset(&hyps[0], 0, sizeof(Hyp));
    hyps[0].ids[0] = NSO_BOS; hyps[0].n_ids = 1; hyps[0].hist = HIST0;

    for (int step = 0; step < P->max_new_tokens && !err; ++step) {
        int nc = 0;
        for (int p = 0; p < nh && !err; ++p) {
            const Hyp *h = &hyps[p];
            if (tr_mask) tr_mask[(size_t)step * B + p] = 0;
            if (h->ids[h->n_ids - 1] == NSO_EOS) {
                cands[nc] = *h; cands[nc].parent = p; cands[nc].token = -1; ++nc;
                continue;
            }
            float logits[NSO_V], lpf[NSO_V];
            int st = h->n_ids - 1;
            int tgt = st < target_len ? target[st] : NSO_EOS;
            nso_synth_row(useed, h->hist, st, tgt, P->margin, logits);
            nso_log_softmax(logits, NSO_V, lpf);

            if (P->variant == 0 && P->allow_eos_anytime)
                extend(&cands[nc++], h, h->score + (double)lpf[NSO_EOS], NSO_EOS, h->text, h->text_len, p);

            int whole = 0;
            int node = walk_word(m, h->text, h->text_len, &whole);
            int n_opts = 0;
            uint64_t mask = 0;
            char nt[132];
            memcpy(nt, h->text, (size_t)h->text_len);
            if (node >= 0) {
                for (int tok = 0; tok < NSO_NBASE; ++tok) {
                    int ok = (tok == NSO_SPACE) ? (whole || m->nodes[node].child[NSO_SPACE] >= 0)
                                                : (m->nodes[node].child[tok] >= 0);
                    if (!ok) continue;
                    mask |= 1ull << tok;
                    nt[h->text_len] = BASE_CHARS[tok];
                    double add = (double)lpf[tok];
                    if (P->use_lm) add += P->beta * nso_lm_logp(m, nt, h->text_len + 1);
                    if (tok == NSO_SPACE && !whole) add -= P->gamma;
                    if (P->variant == 2 && tok == NSO_SPACE && whole) add += P->kappa;
                    extend(&cands[nc++], h, h->score + add, tok, nt, h->text_len + 1, p);
                    ++n_opts;
                }
            }
            if (P->variant == 2 && h->text_len >= P->min_len_for_eos &&
                ((h->text_len > 0 && h->text[h->text_len - 1] == ' ') || whole)) {
                double add = (double)lpf[NSO_EOS];
                add -= 0.002 * (double)h->text_len;
                if (P->use_lm && h->text_len > 0) add += P->beta * nso_lm_logp(m, h->text, h->text_len);
                extend(&cands[nc++], h, h->score + add, NSO_EOS, h->text, h->text_len, p);
                ++n_opts;
            }
            if (n_opts == 0) {
                int k = B;
                if (P->variant == 0) { if (k > NSO_V) k = NSO_V; }
                else if (k > NSO_V) { err = 1; break; }
                SortKey sk[NSO_V];
                for (int i = 0; i < NSO_V; ++i) { sk[i].key = (double)lpf[i]; sk[i].idx = i; }
                qsort(sk, NSO_V, sizeof(SortKey), cmp_key);
                for (int q = 0; q < k; ++q) {
                    int i = sk[q].idx;
                    int tl = h->text_len;
                    memcpy(nt, h->text, (size_t)tl);
                    if (i < NSO_NBASE) nt[tl++] = BASE_CHARS[i];
                    /* .strip(): the surface never starts with a space, so only a trailing one can go */
                    int a = 0;
                    while (a < tl && nt[a] == ' ') ++a;
                    while (tl > a && nt[tl - 1] == ' ') --tl;
                    extend(&cands[nc++], h, h->score + ((double)lpf[i] - P->gamma), i, nt + a, tl - a, p);
                }
            }
            if (tr_mask) tr_mask[(size_t)step * B + p] = mask;
        }
        if (err) break;
        for (int i = 0; i < nc; ++i) {
            keys[i].idx = i;
            keys[i].key = (P->variant == 2) ? norm_score(cands[i].score, cands[i].n_ids, P->alpha) : cands[i].score;
        }
        qsort(keys, (size_t)nc, sizeof(SortKey), cmp_key);
        nh = nc < B ? nc : B;
        int all_done = 1;
        for (int r = 0; r < nh; ++r) {
            hyps[r] = cands[keys[r].idx];
            if (hyps[r].ids[hyps[r].n_ids - 1] != NSO_EOS) all_done = 0;
            if (tr_parent) {
                tr_parent[(size_t)step * B + r] = (int8_t)hyps[r].parent;
                tr_token[(size_t)step * B + r] = (int8_t)hyps[r].token;
                tr_score[(size_t)step * B + r] = hyps[r].score;
            }
        }
        if (tr_nbeams) tr_nbeams[step] = nh;
        steps = step + 1;
        if (all_done) break;
    }
    int n = -1;
    if (!err) {
        /* hyps are sorted by the pruning key, so the first maximal element is rank 0
         * (also true before any step: the single initial hypothesis) */
        n = hyps[0].n_ids;
        memcpy(out_ids, hyps[0].ids, (size_t)n);
        if (out_score) *out_score = hyps[0].score;
    }
    if (out_steps) *out_steps = steps;
    free(hyps); free(cands); free(keys);
    return n;
}

/* Decode n_utts utterances (OpenMP over utterances).
 * targets [n_utts][target_stride] int8, target_lens [n_utts].
 * out_ids [n_utts][max_new_tokens+1] uint8, out_lens [n_utts] (-1 = raised).
 * Trace pointers may be NULL; when given they are [n_utts][T][B] / [n_utts][T]. */
int nso_decode_synth(const NsoModel *m, const NsoParams *P, uint64_t base_seed, int64_t utt0, int n_utts,
                     const int8_t *targets, int target_stride, const int32_t *target_lens,
                     uint8_t *out_ids, int32_t *out_lens, double *out_scores, int32_t *out_steps,
                     int8_t *tr_parent, int8_t *tr_token, double *tr_score, int32_t *tr_nbeams, uint64_t *tr_mask)
{
    if (P->beam < 1 || P->beam > NSO_MAXB || P->max_new_tokens > 128 || !P->lsm_det) return -1;
    if (P->use_lm && (m->order < 1 || m->order > NSO_MAXORDER)) return -2;
    const int T = P->max_new_tokens, B = P->beam;
#pragma omp parallel for schedule(dynamic, 8)
    for (int u = 0; u < n_utts; ++u) {
        size_t tb = (size_t)u * T * B;
        int steps = 0;
        int n = decode_one(m, P, nso_utt_seed(base_seed, (uint64_t)(utt0 + u)),
                           targets + (size_t)u * target_stride, target_lens[u],
                           out_ids + (size_t)u * (T + 1), out_scores ? out_scores + u : NULL, &steps,
                           tr_parent ? tr_parent + tb : NULL, tr_token ? tr_token + tb : NULL,
                           tr_score ? tr_score + tb : NULL, tr_nbeams ? tr_nbeams + (size_t)u * T : NULL,
                           tr_mask ? tr_mask + tb : NULL);
        out_lens[u] = n;
        if (out_steps) out_steps[u] = steps;
    }
    return 0;
}

/* Frame argmax + repeat collapse + special removal
 * (inference_emg_llm.py:445-455).  logits [T][V]; returns count. */
int nso_ctc_greedy(const float *logits, int T, int V, uint8_t *out)
{
    int prev = -1, n = 0;
    for (int t = 0; t < T; ++t) {
        const float *row = logits + (size_t)t * V;
        int best = 0;
        for (int v = 1; v < V; ++v) if (row[v] > row[best]) best = v;
        if (best == prev) continue;
        prev = best;
        if (best == NSO_PAD || best == NSO_BOS || best == NSO_EOS) continue;
        out[n++] = (uint8_t)best;
    }
    return n;
}
