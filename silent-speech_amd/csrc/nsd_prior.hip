// Next-character prior of the n-gram tables and the training regulariser built on it.
//
// Layout: 8 consecutive lanes work on one context / one position, 5 consecutive vocabulary entries per
// lane (the layout of nsd_row_log_softmax5), so a 64-lane wave handles 8 positions and row sums are three
// DPP steps.
#include <hip/hip_runtime.h>
#include "nsd_device.h"

namespace {

NSD_D float row_sum8(float s)
{
    s = s + nsd_dpp_xor1(s);
    s = s + nsd_dpp_xor2(s);
    s = s + nsd_dpp_half_mirror(s);
    return s;
}

// Context key for a history whose last symbols are `tail` (packed 6-bit codes, most recent lowest) with
// `n` base characters available (n saturates at max(order-1, 1)); `bad` = one of them is outside the
// alphabet.  Returns false when no count can match.
NSD_D bool prior_key(const NsdTables &tab, int form, uint64_t tail, int n, bool bad, uint64_t &key)
{
    const int k = tab.order - 1;
    if (k <= 0) {                                  // order 1: the only context is the empty one
        key = 0;
        return form == NSD_PRIOR_U || n == 0;
    }
    if (bad) return false;
    if (n >= k) { key = tail & tab.ctx_sym_mask; return true; }
    if (n == k - 1) { key = (((uint64_t)NSD_SYM_HAT << (6 * n)) | tail) & tab.ctx_sym_mask; return true; }
    return false;                                  // two or more '^' (form U) or a short key (form T): never counted
}

// P(ch | context) for this lane's 5 vocabulary entries (0 for the three specials), renormalised over the
// 37 base characters as cond_dist does.  Counts are recovered exactly from the stored log-probabilities:
// denominator = round(exp(-backoff)), numerator = round(exp(logp) * denominator).
NSD_D void prior_row5(const NsdTables &tab, bool lookup, uint64_t key, int j, float (&p)[5])
{
    double den = __builtin_rint(exp(-tab.unseen));
    bool seen = false;
    if (lookup) {
        const NsdCtxSlot *s = nsd_ctx_find(tab, key);
        if (s) { den = __builtin_rint(exp(-s->backoff)); seen = true; }
    }
    float sum = 0.0f;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int v = j * 5 + q;
        float pv = 0.0f;
        if (v < NSD_NBASE) {
            double num = 1.0, lp;
            if (seen && nsd_gram_find(tab, (key << 6) | (uint64_t)(v + 1), lp)) num = __builtin_rint(exp(lp) * den);
            pv = (float)(num / den);
        }
        p[q] = pv;
        sum = (q == 0) ? pv : sum + pv;
    }
    sum = row_sum8(sum) + 1e-12f;
#pragma unroll
    for (int q = 0; q < 5; ++q) p[q] = p[q] / sum;
}

__global__ void __launch_bounds__(256)
nsd_cond_dist_kernel(NsdTables tab, const uint8_t *ctx_ids, const int32_t *ctx_len, int N, int stride, int form,
                     float *probs)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = gid >> 3, j = gid & 7;
    const bool act = row < N;
    const int r = act ? row : 0;
    const int want = tab.order - 1 > 1 ? tab.order - 1 : 1;
    const int len = act ? ctx_len[r] : 0;
    const uint8_t *s = ctx_ids + (size_t)r * stride;
    uint64_t tail = 0;
    int n = 0;
    bool bad = false;
    for (int i = len - 1; i >= 0 && n < want; --i, ++n) {
        const uint32_t c = s[i];
        if (c >= NSD_NBASE) bad = true;
        tail |= (uint64_t)((c + 1u) & 63u) << (6 * n);
    }
    uint64_t key = 0;
    const bool lookup = prior_key(tab, form, tail, n, bad, key);
    float p[5];
    prior_row5(tab, lookup, key, j, p);
    if (act) {
#pragma unroll
        for (int q = 0; q < 5; ++q)
            if (j * 5 + q < NSD_NBASE) probs[(size_t)row * NSD_NBASE + j * 5 + q] = p[q];
    }
}

// prior over the model vocabulary and log-softmax of the logits for position (b, t); returns false when
// the position is skipped
NSD_D bool position_terms(const NsdTables &tab, const float *logits, const int64_t *out_ids, int L, int form,
                          int pos, int j, bool in_range, float (&pv)[5], float (&lp)[5], float &psum)
{
    const int b = pos / L, t = pos % L;
    const int64_t *row = out_ids + (size_t)b * L;
    const bool valid = in_range && row[t] != -100;
    const int want = tab.order - 1 > 1 ? tab.order - 1 : 1;
    uint64_t tail = 0;
    int n = 0;
    if (valid) {
        // the context is made of the row's earlier targets that are base characters
        for (int i = t - 1; i >= 0 && n < want; --i) {
            const int64_t id = row[i];
            if (id >= 0 && id < NSD_NBASE) { tail |= (uint64_t)(id + 1) << (6 * n); ++n; }
        }
    }
    uint64_t key = 0;
    const bool lookup = valid && prior_key(tab, form, tail, n, false, key);
    prior_row5(tab, lookup, key, j, pv);
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 5; ++q) s = (q == 0) ? pv[q] : s + pv[q];
    s = row_sum8(s) + (float)(1e-8 * NSD_V);
    float s2 = 0.0f;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        pv[q] = (pv[q] + 1e-8f) / s;
        s2 = (q == 0) ? pv[q] : s2 + pv[q];
    }
    psum = row_sum8(s2);
    const float *x = logits + (size_t)(valid ? pos : 0) * NSD_V + j * 5;
#pragma unroll
    for (int q = 0; q < 5; ++q) lp[q] = valid ? x[q] : 0.0f;
    nsd_row_log_softmax5(lp);
    return valid;
}

__global__ void __launch_bounds__(256)
nsd_prior_ce_kernel(NsdTables tab, const float *logits, const int64_t *out_ids, int npos, int L, int form,
                    float *pos_loss)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int pos = gid >> 3, j = gid & 7;
    const bool in_range = pos < npos;
    float pv[5], lp[5], psum;
    const bool valid = position_terms(tab, logits, out_ids, L, form, in_range ? pos : 0, j, in_range, pv, lp, psum);
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 5; ++q) { const float m = pv[q] * lp[q]; s = (q == 0) ? m : s + m; }
    s = row_sum8(s);
    if (in_range && j == 0) pos_loss[pos] = valid ? -s : 0.0f;
}

// one block: fixed-order fp64 reduction of the per-position terms
__global__ void __launch_bounds__(256)
nsd_prior_reduce_kernel(const float *pos_loss, const int64_t *out_ids, int npos, double *stats)
{
    __shared__ double ssum[256];
    __shared__ double scnt[256];
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < npos; i += 256) {
        a += (double)pos_loss[i];
        c += out_ids[i] != -100 ? 1.0 : 0.0;
    }
    ssum[threadIdx.x] = a;
    scnt[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            ssum[threadIdx.x] += ssum[threadIdx.x + off];
            scnt[threadIdx.x] += scnt[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] = ssum[0];
        stats[1] = scnt[0];
        stats[2] = scnt[0] > 0.0 ? ssum[0] / scnt[0] : 0.0;
    }
}

__global__ void __launch_bounds__(256)
nsd_prior_ce_grad_kernel(NsdTables tab, const float *logits, const int64_t *out_ids, int npos, int L, int form,
                         const double *stats, const float *grad_out, float *grad_logits)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int pos = gid >> 3, j = gid & 7;
    const bool in_range = pos < npos;
    float pv[5], lp[5], psum;
    const bool valid = position_terms(tab, logits, out_ids, L, form, in_range ? pos : 0, j, in_range, pv, lp, psum);
    const double count = stats[1];
    const float scale = count > 0.0 ? (float)((double)grad_out[0] / count) : 0.0f;
    if (in_range) {
        float *g = grad_logits + (size_t)pos * NSD_V + j * 5;
#pragma unroll
        for (int q = 0; q < 5; ++q) g[q] = valid ? scale * (nsd_det_expf(lp[q]) * psum - pv[q]) : 0.0f;
    }
}

} // namespace

extern "C" int nsd_cond_dist(const NsdTables *t, const uint8_t *ctx_ids, const int32_t *ctx_len, int32_t N,
                             int32_t stride, int32_t form, float *probs, void *stream)
{
    if (!t) return NSD_E_NULL;
    if (t->order < 1) return NSD_E_ORDER;
    if (form != NSD_PRIOR_U && form != NSD_PRIOR_T) return NSD_E_VARIANT;
    if (N < 0 || stride < 0) return NSD_E_SIZE;
    if (N == 0) return NSD_OK;
    if (!ctx_len || !probs || (stride > 0 && !ctx_ids)) return NSD_E_NULL;
    const long long threads = (long long)N * 8;
    hipLaunchKernelGGL(nsd_cond_dist_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       *t, ctx_ids, ctx_len, N, stride, form, probs);
    return (int)hipGetLastError();
}

static int prior_args(const NsdTables *t, int32_t B, int32_t L, int32_t V, int32_t form)
{
    if (!t) return NSD_E_NULL;
    if (t->order < 1) return NSD_E_ORDER;
    if (V != NSD_V) return NSD_E_VOCAB;
    if (form != NSD_PRIOR_U && form != NSD_PRIOR_T) return NSD_E_VARIANT;
    if (B < 0 || L < 0 || (long long)B * L > (1ll << 27)) return NSD_E_SIZE;
    return NSD_OK;
}

extern "C" int nsd_lm_prior_ce(const NsdTables *t, const float *logits, const int64_t *out_ids, int32_t B, int32_t L,
                               int32_t V, int32_t form, float *pos_loss, double *stats, void *stream)
{
    const int rc = prior_args(t, B, L, V, form);
    if (rc != NSD_OK) return rc;
    if (!stats) return NSD_E_NULL;
    const int npos = B * L;
    if (npos > 0 && (!logits || !out_ids || !pos_loss)) return NSD_E_NULL;
    if (npos > 0) {
        const long long threads = (long long)npos * 8;
        hipLaunchKernelGGL(nsd_prior_ce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, *t, logits, out_ids, npos, L, form, pos_loss);
    }
    hipLaunchKernelGGL(nsd_prior_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pos_loss, out_ids, npos, stats);
    return (int)hipGetLastError();
}

extern "C" int nsd_lm_prior_ce_grad(const NsdTables *t, const float *logits, const int64_t *out_ids, int32_t B,
                                    int32_t L, int32_t V, int32_t form, const double *stats, const float *grad_out,
                                    float *grad_logits, void *stream)
{
    const int rc = prior_args(t, B, L, V, form);
    if (rc != NSD_OK) return rc;
    const int npos = B * L;
    if (npos == 0) return NSD_OK;
    if (!logits || !out_ids || !stats || !grad_out || !grad_logits) return NSD_E_NULL;
    const long long threads = (long long)npos * 8;
    hipLaunchKernelGGL(nsd_prior_ce_grad_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, *t, logits, out_ids, npos, L, form, stats, grad_out, grad_logits);
    return (int)hipGetLastError();
}
