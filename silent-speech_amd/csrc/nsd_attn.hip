// The non-GEMM part of a KV-cached, batched decode step of the AR head (gfx950): what EMGTextModel.step
// (inference_emg_llm.py:83-88) needs around its dense contractions when all N utterances x B hypotheses of a
// beam step advance together and the generated keys / values are never moved:
//   nsd_ancestors       hypothesis m's ancestor slot at every generated position, from the scorer's back-pointers
//   nsd_add_rmsnorm     residual add + RMSNorm (one wave per row)
//   nsd_rope_kv_append  rotary embedding of q and k, k / v written to the step's slab of the generated cache
//   nsd_attn_step       attention of one new token per hypothesis over [context keys shared by the utterance ;
//                       generated keys addressed through the ancestor table], online softmax, fp32 accumulation
//   nsd_swiglu          silu(gate) * up
// Element types: fp32 or bf16 storage (dtype 0 / 1), arithmetic in fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nsd_device.h"

namespace {

struct BF16 { uint16_t b; };

template <class T> NSD_D float to_f(T v);
template <> NSD_D float to_f<float>(float v) { return v; }
template <> NSD_D float to_f<BF16>(BF16 v) { return __uint_as_float((uint32_t)v.b << 16); }

template <class T> NSD_D T from_f(float v);
template <> NSD_D float from_f<float>(float v) { return v; }
template <> NSD_D BF16 from_f<BF16>(float v)
{
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return BF16{(uint16_t)((u >> 16) | 0x40u)};     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                   // round to nearest even
    return BF16{(uint16_t)(u >> 16)};
}

// sixteen bytes of a row as floats: 4 (fp32) or 8 (bf16) elements
template <class T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    NSD_D static void unpack(const uint4 &v, float (&o)[8])
    {
        o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
    }
    NSD_D static void load(const float *p, float (&o)[8]) { unpack(*reinterpret_cast<const uint4 *>(p), o); }
};
template <> struct Vec16<BF16> {
    static constexpr int N = 8;
    NSD_D static void unpack(const uint4 &v, float (&o)[8])
    {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    NSD_D static void load(const BF16 *p, float (&o)[8]) { unpack(*reinterpret_cast<const uint4 *>(p), o); }
};

// Wave-wide maximum / sum, the same value in every lane.  Inside a row of 16 lanes the butterfly runs on DPP register
// moves (after the distance-1 and -2 steps a quad is uniform, so the mirrored lane of the other quad / half carries
// what the lane at distance 4 / 8 carries); the four row results meet through scalar registers.  No LDS-crossbar
// permutes: they cost an address, an LDS instruction and its latency per step.
NSD_D float nsd_dpp_row_mirror(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
}
NSD_D float row_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
NSD_D float wave_max_f(float v)
{
    v = fmaxf(v, nsd_dpp_xor1(v));
    v = fmaxf(v, nsd_dpp_xor2(v));
    v = fmaxf(v, nsd_dpp_half_mirror(v));
    v = fmaxf(v, nsd_dpp_row_mirror(v));
    return fmaxf(fmaxf(row_lane(v, 0), row_lane(v, 16)), fmaxf(row_lane(v, 32), row_lane(v, 48)));
}
NSD_D float wave_sum_f(float v)
{
    v = v + nsd_dpp_xor1(v);
    v = v + nsd_dpp_xor2(v);
    v = v + nsd_dpp_half_mirror(v);
    v = v + nsd_dpp_row_mirror(v);
    return (row_lane(v, 0) + row_lane(v, 16)) + (row_lane(v, 32) + row_lane(v, 48));
}

// ---------------------------------------------------------------------
// ancestors: one wave per utterance, the utterance's back-pointers of steps 0..t-1 staged in LDS
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(64)
nsd_ancestors_kernel(const uint8_t *backptr, int t, int M, int B, uint8_t *anc)
{
    extern __shared__ unsigned char bp[];
    const int u = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < t * B; i += 64) bp[i] = backptr[(size_t)(i / B) * M + (size_t)u * B + (i % B)];
    __syncthreads();
    if (lane >= B) return;
    int a = lane;
    for (int s = t - 1; s >= 0; --s) {
        int b = bp[s * B + a];
        if (b >= B) b = 0;                         // slots the beam does not use: any in-range ancestor
        a = b;
        anc[(size_t)s * M + (size_t)u * B + lane] = (uint8_t)a;
    }
}

// ---------------------------------------------------------------------
// residual add + RMSNorm: x_out = x (+ delta); h = w * (x_out * rsqrt(mean(x_out^2) + eps)), one wave per row,
// sixteen bytes per lane and access (hidden must be a multiple of the 4 / 8 elements that makes)
// ---------------------------------------------------------------------
template <class T> struct Store16;
template <> struct Store16<float> {
    NSD_D static void store(float *p, const float (&v)[8]) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Store16<BF16> {
    NSD_D static void store(BF16 *p, const float (&v)[8])
    {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)from_f<BF16>(v[2 * i]).b | ((uint32_t)from_f<BF16>(v[2 * i + 1]).b << 16);
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

template <class T>
__global__ void __launch_bounds__(256)
nsd_add_rmsnorm_kernel(const T *x, const T *delta, const T *w, float eps, int rows, int hidden, T *x_out, T *h)
{
    constexpr int VN = Vec16<T>::N;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T *xr = x + (size_t)row * hidden;
    const T *dr = delta ? delta + (size_t)row * hidden : nullptr;
    T *xo = x_out + (size_t)row * hidden;
    float ss = 0.0f;
    for (int i = lane * VN; i < hidden; i += 64 * VN) {
        float v[8];
        Vec16<T>::load(xr + i, v);
        if (dr) {
            float d[8];
            Vec16<T>::load(dr + i, d);
#pragma unroll
            for (int e = 0; e < VN; ++e) v[e] = to_f(from_f<T>(v[e] + d[e]));     // the sum is stored in the element type
            Store16<T>::store(xo + i, v);
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) ss += v[e] * v[e];
    }
    ss = wave_sum_f(ss);
    const float inv = rsqrtf(ss / (float)hidden + eps);
    const T *src = dr ? xo : xr;
    for (int i = lane * VN; i < hidden; i += 64 * VN) {
        float v[8], g[8];
        Vec16<T>::load(src + i, v);
        Vec16<T>::load(w + i, g);
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] = g[e] * to_f(from_f<T>(v[e] * inv));
        Store16<T>::store(h + (size_t)row * hidden + i, v);
    }
}

// The same for a row that fits the registers of one workgroup: one workgroup of W waves per row, every lane holds NV
// 16-byte pieces of it from the first read to the last write (the row is read once), the sum of squares crosses the
// waves through LDS.  A decode step has about a thousand rows: one wave per row leaves three quarters of the wave
// slots empty and each wave with a long dependent loop.
template <class T, int NV>
__global__ void __launch_bounds__(1024)
nsd_add_rmsnorm_row_kernel(const T *x, const T *delta, const T *w, float eps, int hidden, T *x_out, T *h)
{
    constexpr int VN = Vec16<T>::N;
    __shared__ float part[16];
    const int row = blockIdx.x, nvec = hidden / VN;
    const T *xr = x + (size_t)row * hidden;
    const T *dr = delta ? delta + (size_t)row * hidden : nullptr;
    float v[NV][8];
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = (int)threadIdx.x + k * (int)blockDim.x;
        if (i < nvec) {
            Vec16<T>::load(xr + (size_t)i * VN, v[k]);
            if (dr) {
                float d[8];
                Vec16<T>::load(dr + (size_t)i * VN, d);
#pragma unroll
                for (int e = 0; e < VN; ++e) v[k][e] = to_f(from_f<T>(v[k][e] + d[e]));   // the sum is stored in the element type
                Store16<T>::store(x_out + (size_t)row * hidden + (size_t)i * VN, v[k]);
            }
#pragma unroll
            for (int e = 0; e < VN; ++e) ss += v[k][e] * v[k][e];
        }
    }
    ss = wave_sum_f(ss);
    const int waves = (int)blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
    for (int q = 0; q < waves; ++q) tot += part[q];
    const float inv = rsqrtf(tot / (float)hidden + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = (int)threadIdx.x + k * (int)blockDim.x;
        if (i < nvec) {
            float g[8];
            Vec16<T>::load(w + (size_t)i * VN, g);
#pragma unroll
            for (int e = 0; e < VN; ++e) v[k][e] = g[e] * to_f(from_f<T>(v[k][e] * inv));
            Store16<T>::store(h + (size_t)row * hidden + (size_t)i * VN, v[k]);
        }
    }
}

template <class T>
void launch_add_rmsnorm(const T *x, const T *delta, const T *w, float eps, int rows, int hidden, T *x_out, T *h, hipStream_t s)
{
    // two 16-byte pieces per lane (one or four measure the same or worse at 1 024 x 3 072 bf16: 8.1 / 8.0 / 8.8 us)
    constexpr int VN = Vec16<T>::N, NV = 2;
    const int nvec = hidden / VN;
    const int waves = (nvec + 64 * NV - 1) / (64 * NV);
    if (waves <= 16)
        hipLaunchKernelGGL((nsd_add_rmsnorm_row_kernel<T, NV>), dim3((unsigned)rows), dim3(64 * waves), 0, s,
                           x, delta, w, eps, hidden, x_out, h);
    else
        hipLaunchKernelGGL(nsd_add_rmsnorm_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,
                           x, delta, w, eps, rows, hidden, x_out, h);
}

// ---------------------------------------------------------------------
// rotary embedding (rotate-half form) of q and k; k and v go to the step's slab of the generated cache
// ---------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256)
nsd_rope_kv_append_kernel(const T *qkv, const float *cs, const float *sn, int M, int heads, int kvh, int hd,
                          T *q_out, T *k_out, T *v_out)
{
    const int half = hd >> 1;
    const int per_row = (heads + 2 * kvh) * half;              // one thread per pair of elements
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)M * per_row) return;
    const int m = (int)(gid / per_row), rem = (int)(gid % per_row);
    const int head = rem / half, i = rem % half;
    const T *src = qkv + ((size_t)m * (heads + 2 * kvh) + head) * hd;
    const float x1 = to_f(src[i]), x2 = to_f(src[i + half]);
    if (head >= heads + kvh) {                                   // value rows are copied
        T *dst = v_out + ((size_t)m * kvh + (head - heads - kvh)) * hd;
        dst[i] = src[i];
        dst[i + half] = src[i + half];
        return;
    }
    // x * cos + rotate_half(x) * sin, each product rounded before the sum
    const float a1 = x1 * cs[i], b1 = x2 * sn[i];
    const float a2 = x2 * cs[i + half], b2 = x1 * sn[i + half];
    const float o1 = a1 - b1, o2 = a2 + b2;
    T *dst = head < heads ? q_out + ((size_t)m * heads + head) * hd : k_out + ((size_t)m * kvh + (head - heads)) * hd;
    dst[i] = from_f<T>(o1);
    dst[i + half] = from_f<T>(o2);
}

// The same with 16-byte pieces: a thread takes VN consecutive elements of the first half of a head and the VN
// elements they are rotated with (hd / 2 a multiple of VN, rows 16-byte aligned: checked by the launcher).
template <class T>
__global__ void __launch_bounds__(256)
nsd_rope_kv_append_vec_kernel(const T *qkv, const float *cs, const float *sn, int M, int heads, int kvh, int hd,
                              T *q_out, T *k_out, T *v_out)
{
    constexpr int VN = Vec16<T>::N;
    const int half = hd >> 1, hv = half / VN;
    const int per_row = (heads + 2 * kvh) * hv;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)M * per_row) return;
    const int m = (int)(gid / per_row), rem = (int)(gid % per_row);
    const int head = rem / hv, i = (rem % hv) * VN;
    const T *src = qkv + ((size_t)m * (heads + 2 * kvh) + head) * hd;
    float x1[8], x2[8];
    Vec16<T>::load(src + i, x1);
    Vec16<T>::load(src + i + half, x2);
    T *dst;
    if (head >= heads + kvh) {                                   // value rows are copied
        dst = v_out + ((size_t)m * kvh + (head - heads - kvh)) * hd;
    } else {
        dst = head < heads ? q_out + ((size_t)m * heads + head) * hd : k_out + ((size_t)m * kvh + (head - heads)) * hd;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            // x * cos + rotate_half(x) * sin, each product rounded before the sum
            const float a1 = x1[e] * cs[i + e], b1 = x2[e] * sn[i + e];
            const float a2 = x2[e] * cs[i + e + half], b2 = x1[e] * sn[i + e + half];
            x1[e] = a1 - b1;
            x2[e] = a2 + b2;
        }
    }
    Store16<T>::store(dst + i, x1);
    Store16<T>::store(dst + i + half, x2);
}

// One 16-byte vector of gate and of up per thread; the grid's second dimension walks the rows, so no thread divides
// its index.  (Two or four vectors per thread with their loads requested together are slower at a thousand rows:
// 12.9 / 15.0 against 11.7 us; the launch wants as many waves as it can get.)
template <class T>
__global__ void __launch_bounds__(256)
nsd_swiglu_kernel(const T *gu, long long rows, int inter, T *out)
{
    constexpr int VN = Vec16<T>::N;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) * VN;
    if (i >= inter) return;
    for (long long r = blockIdx.y; r < rows; r += gridDim.y) {
        float g[8], u[8];
        Vec16<T>::load(gu + r * 2 * inter + i, g);
        Vec16<T>::load(gu + r * 2 * inter + inter + i, u);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            // silu, stored in the element type; hardware exp2 and reciprocal (about 1 ulp each, below what the element
            // type keeps): the precise expf and division made this kernel arithmetic-bound
            const float s = to_f(from_f<T>(g[e] * __frcp_rn(1.0f + __expf(-g[e]))));
            g[e] = s * u[e];
        }
        Store16<T>::store(out + r * inter + i, g);
    }
}

// ---------------------------------------------------------------------
// attention of one new token per hypothesis.  One wave per (hypothesis, kv head): its G = heads / kv_heads
// query heads share every key / value row.  Keys in chunks of 64 positions: lane = position for q . k (the
// queries are broadcast from LDS), lane = head dimension for the weighted sum of the values; running maximum
// and sum of an online softmax.  The waves of a workgroup are the hypotheses of one utterance and one kv
// head, so the context rows they share stay in that CU's caches.
// ---------------------------------------------------------------------
struct AttnArgs {
    const void *q, *kctx, *vctx, *kgen, *vgen;
    const uint8_t *anc;
    void *out;
    int N, B, heads, kvh, hd, S, t, M;
    float scale;
    // when given: softmax state of the context part (nsd_attn_ctx): unnormalised weighted value sums, running
    // maxima and sums per query row; this kernel then starts at the first generated position
    const float *part_o, *part_m, *part_l;
};

// q . k over one 16-byte piece of a key row: the queries lie in LDS in their storage type (every lane reads the
// same address: a broadcast), bf16 pairs go through the packed dot-product instruction
template <class T, int G> struct QDot;
template <int G> struct QDot<float, G> {
    NSD_D static void run(const float *qs, int hd, int d, const float *kr, float (&sc)[G])
    {
        const float4 k = *reinterpret_cast<const float4 *>(kr + d);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 q = *reinterpret_cast<const float4 *>(qs + g * hd + d);
            sc[g] = __builtin_fmaf(q.x, k.x, sc[g]);
            sc[g] = __builtin_fmaf(q.y, k.y, sc[g]);
            sc[g] = __builtin_fmaf(q.z, k.z, sc[g]);
            sc[g] = __builtin_fmaf(q.w, k.w, sc[g]);
        }
    }
};
typedef __bf16 nsd_bf2 __attribute__((ext_vector_type(2)));
template <int G> struct QDot<BF16, G> {
    NSD_D static void run(const BF16 *qs, int hd, int d, const BF16 *kr, float (&sc)[G])
    {
        const uint4 k = *reinterpret_cast<const uint4 *>(kr + d);
        const uint32_t kw[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint4 q = *reinterpret_cast<const uint4 *>(qs + g * hd + d);
            const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sc[g] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(nsd_bf2, qw[e]), __builtin_bit_cast(nsd_bf2, kw[e]),
                                                        sc[g], false);
        }
    }
};

template <class T, int G>
__global__ void __launch_bounds__(1024)
nsd_attn_step_kernel(AttnArgs a)
{
    constexpr int VN = Vec16<T>::N;                             // elements of a 16-byte piece
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int u = blockIdx.x / a.kvh, h = blockIdx.x % a.kvh;
    const int r = blockIdx.y * waves + w;
    if (r >= a.B) return;
    const int hd = a.hd, M = a.M, S = a.S, t = a.t;
    const int m = u * a.B + r;
    const size_t q_bytes = ((size_t)G * hd * sizeof(T) + 15) & ~(size_t)15;
    const size_t per_wave = q_bytes + (size_t)G * 64 * 4 + 64 * 8;
    T *qs = reinterpret_cast<T *>(smem + (size_t)w * per_wave);               // [G][hd]
    float *ps = reinterpret_cast<float *>(smem + (size_t)w * per_wave + q_bytes);   // [G][64] softmax numerators
    const T **vrow = reinterpret_cast<const T **>(ps + G * 64);                // [64] value row of each position
    const T *q = static_cast<const T *>(a.q) + ((size_t)m * a.heads + (size_t)h * G) * hd;
    for (int i = lane; i < G * hd / VN; i += 64)               // 16-byte pieces (hd is a multiple of VN)
        reinterpret_cast<uint4 *>(qs)[i] = reinterpret_cast<const uint4 *>(q)[i];
    nsd_wave_sync();

    // value sum: LP lanes share a position (16 bytes of the row each), 64 / LP positions advance together
    const int LP = hd / VN;                                     // a power of two up to 64
    const int PPAR = 64 / LP;
    const int vd = (lane % LP) * VN, vp0 = lane / LP;
    float mx[G], den[G], acc[G][VN];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        mx[g] = -__builtin_inff(); den[g] = 0.0f;
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[g][e] = 0.0f;
    }
    const T *kctx = static_cast<const T *>(a.kctx) + ((size_t)u * a.kvh + h) * (size_t)S * hd;
    const T *vctx = static_cast<const T *>(a.vctx) + ((size_t)u * a.kvh + h) * (size_t)S * hd;
    const T *kgen = static_cast<const T *>(a.kgen), *vgen = static_cast<const T *>(a.vgen);
    const int L = S + t + 1;
    int first = 0;
    if (a.part_o) {
        first = S;
        const size_t row0 = (size_t)m * a.heads + (size_t)h * G;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            mx[g] = a.part_m[row0 + g];
            den[g] = a.part_l[row0 + g];
            if (vp0 == 0) {
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[g][e] = a.part_o[(row0 + g) * hd + vd + e];
            }
        }
    }
    for (int c0 = first; c0 < L; c0 += 64) {
        const int p = c0 + lane;
        const bool valid = p < L;
        const T *kr = kgen, *vr = vgen;
        if (valid) {
            if (p < S) {
                kr = kctx + (size_t)p * hd;
                vr = vctx + (size_t)p * hd;
            } else {
                const int s = p - S;
                const int slot = s == t ? m : u * a.B + (int)a.anc[(size_t)s * M + m];
                const size_t off = (((size_t)s * M + slot) * a.kvh + h) * hd;
                kr = kgen + off;
                vr = vgen + off;
            }
        }
        float sc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) sc[g] = 0.0f;
        if (valid) {
#pragma unroll 4
            for (int d = 0; d < hd; d += VN) QDot<T, G>::run(qs, hd, d, kr, sc);
        }
        vrow[lane] = vr;
        float alpha[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float sv = sc[g] * a.scale;
            const float cm = wave_max_f(valid ? sv : -__builtin_inff());
            const float mn = fmaxf(mx[g], cm);                 // finite: every chunk holds a valid position
            alpha[g] = expf(mx[g] - mn);
            const float pe = valid ? expf(sv - mn) : 0.0f;
            den[g] = den[g] * alpha[g] + wave_sum_f(pe);
            mx[g] = mn;
            ps[g * 64 + lane] = pe;
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[g][e] *= alpha[g];
        }
        nsd_wave_sync();
        const int n = L - c0 < 64 ? L - c0 : 64;
        // (requesting the first eight value pieces before the dot products, to take their round trip off the end of
        // the chunk, costs registers and with them resident waves: 33.9 -> 53.6 us per launch, not kept)
#pragma unroll 2
        for (int pp = vp0; pp < n; pp += PPAR) {
            float vf[8];
            Vec16<T>::load(vrow[pp] + vd, vf);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float pw = ps[g * 64 + pp];
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[g][e] = __builtin_fmaf(pw, vf[e], acc[g][e]);
            }
        }
        nsd_wave_sync();
    }
    // the position groups' partial sums meet in the lanes of group 0 (row exchanges by permlane16/32_swap instead of
    // these permutes were measured: 33.9 -> 37.8 us per launch, not kept)
    for (int d = LP; d < 64; d <<= 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[g][e] += __shfl_xor(acc[g][e], d);
        }
    }
    if (vp0 == 0) {
        T *o = static_cast<T *>(a.out) + ((size_t)m * a.heads + (size_t)h * G) * hd + vd;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float ov[8];
#pragma unroll
            for (int e = 0; e < VN; ++e) ov[e] = acc[g][e] / den[g];
            Store16<T>::store(o + (size_t)g * hd, ov);
        }
    }
}

// ---------------------------------------------------------------------
// context part of the attention on the matrix cores (bf16).  The B hypotheses of an utterance and the G query
// heads of a kv head share the S context keys: [B*G x hd] x [hd x S] and [B*G x S] x [S x hd] are small dense
// products.  One wave per tile of 16 query rows; v_mfma_f32_16x16x32_bf16 with the keys as the A operand, so that
// the scores come out transposed - a lane holds, for ONE query row, four keys of every 16-key tile - which is the
// A-operand layout of the second product (weights x values) with no lane movement; the values are stored
// transposed ([hd, S], keys contiguous) so that their B fragments are 8-byte reads.  Keys are padded to a multiple
// of 32 with zero rows.  Output: unnormalised sums, maxima and sums per query row, which nsd_attn_step continues.
// ---------------------------------------------------------------------
typedef __bf16 nsd_bf8 __attribute__((ext_vector_type(8)));
typedef float nsd_f4 __attribute__((ext_vector_type(4)));

struct CtxArgs {
    const BF16 *q, *kpad, *vt;
    float *po, *pm, *pl;
    int N, B, heads, kvh, S, S32;
    float scale;
};

template <int NP, int HD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
nsd_attn_ctx_kernel(CtxArgs a)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int u = blockIdx.x / a.kvh, h = blockIdx.x % a.kvh;
    const int G = a.heads / a.kvh, R = a.B * G;                 // query rows of this (utterance, kv head)
    const int c16 = lane & 15, g4 = lane >> 4;
    const BF16 *K = a.kpad + ((size_t)u * a.kvh + h) * (size_t)a.S32 * HD;
    const BF16 *Vt = a.vt + ((size_t)u * a.kvh + h) * (size_t)HD * a.S32;
    for (int r0 = w * 16; r0 < R; r0 += waves * 16) {
        // this lane's query row (B operand of the first product, row of the softmax)
        const bool rv = r0 + c16 < R;
        const int rho = rv ? r0 + c16 : 0;
        const size_t qrow = (size_t)(u * a.B + rho / G) * a.heads + (size_t)h * G + (size_t)(rho % G);
        const BF16 *qp = a.q + qrow * HD + 8 * g4;
        nsd_bf8 qf[HD / 32];
#pragma unroll
        for (int kc = 0; kc < HD / 32; ++kc) qf[kc] = __builtin_bit_cast(nsd_bf8, *reinterpret_cast<const uint4 *>(qp + 32 * kc));
        // scores, transposed: st[nt][reg] = key (16 nt + 4 g4 + reg) . query row c16
        nsd_f4 st[2 * NP];
#pragma unroll
        for (int nt = 0; nt < 2 * NP; ++nt) {
            nsd_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            const BF16 *kp = K + (size_t)(nt * 16 + c16) * HD + 8 * g4;
#pragma unroll
            for (int kc = 0; kc < HD / 32; ++kc)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(nsd_bf8, *reinterpret_cast<const uint4 *>(kp + 32 * kc)), qf[kc], acc, 0, 0, 0);
            st[nt] = acc;
        }
        // The value fragments of the first DTB head-dimension tiles are requested here, before the scores are
        // reduced, so that their round trip runs under the softmax (one round trip per tile otherwise); the launch
        // has few waves (1.5 per SIMD at 64 utterances x 8 kv heads) and may use the registers of those that are
        // not there.  Splitting the head dimension over more workgroups was measured and is slower (18.9 -> 22.8 ->
        // 33.4 us for 1 / 2 / 4 workgroups); staging keys and values once per workgroup in LDS changes nothing
        // (19.8 us), and without its output stores the kernel still takes 15.8 us.
        constexpr int DTN = HD / 16, DTB = NP <= 4 ? DTN : 4;
        uint2 vlo[DTB][NP], vhi[DTB][NP];
        auto fetch_v = [&](int dt0) {
#pragma unroll
            for (int d = 0; d < DTB; ++d) {
                const BF16 *vp = Vt + (size_t)((dt0 + d) * 16 + c16) * a.S32 + 4 * g4;
#pragma unroll
                for (int np = 0; np < NP; ++np) {
                    vlo[d][np] = *reinterpret_cast<const uint2 *>(vp + 32 * np);
                    vhi[d][np] = *reinterpret_cast<const uint2 *>(vp + 32 * np + 16);
                }
            }
        };
        fetch_v(0);
        float mrow = -__builtin_inff();
#pragma unroll
        for (int nt = 0; nt < 2 * NP; ++nt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = nt * 16 + 4 * g4 + e;
                const float v = key < a.S ? st[nt][e] * a.scale : -__builtin_inff();
                st[nt][e] = v;
                mrow = fmaxf(mrow, v);
            }
        }
        mrow = fmaxf(mrow, __shfl_xor(mrow, 16));
        mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
        float lrow = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 2 * NP; ++nt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pe = expf(st[nt][e] - mrow);       // 0 for the padded keys
                st[nt][e] = pe;
                lrow += pe;
            }
        }
        lrow += __shfl_xor(lrow, 16);
        lrow += __shfl_xor(lrow, 32);
        // weights as A fragments: k-slot 8 g4 + j of pair np is key 32 np + 4 g4 + j (j < 4), 32 np + 16 + 4 g4 + j - 4
        nsd_bf8 pf[NP];
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            uint32_t wv[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wv[i] = (uint32_t)from_f<BF16>(st[2 * np][2 * i]).b | ((uint32_t)from_f<BF16>(st[2 * np][2 * i + 1]).b << 16);
                wv[2 + i] = (uint32_t)from_f<BF16>(st[2 * np + 1][2 * i]).b |
                            ((uint32_t)from_f<BF16>(st[2 * np + 1][2 * i + 1]).b << 16);
            }
            pf[np] = __builtin_bit_cast(nsd_bf8, make_uint4(wv[0], wv[1], wv[2], wv[3]));
        }
#pragma unroll
        for (int dt0 = 0; dt0 < DTN; dt0 += DTB) {
            if (dt0) fetch_v(dt0);
#pragma unroll
            for (int d = 0; d < DTB; ++d) {
                const int dt = dt0 + d;
                nsd_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int np = 0; np < NP; ++np)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        pf[np], __builtin_bit_cast(nsd_bf8, make_uint4(vlo[d][np].x, vlo[d][np].y, vhi[d][np].x, vhi[d][np].y)),
                        acc, 0, 0, 0);
                // acc[reg] = query row r0 + 4 g4 + reg, head dimension 16 dt + c16
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rr = r0 + 4 * g4 + e;
                    if (rr < R) {
                        const size_t orow = (size_t)(u * a.B + rr / G) * a.heads + (size_t)h * G + (size_t)(rr % G);
                        a.po[orow * HD + dt * 16 + c16] = acc[e];
                    }
                }
            }
        }
        if (g4 == 0 && rv) {
            a.pm[qrow] = mrow;
            a.pl[qrow] = lrow;
        }
    }
}

template <int HD>
int launch_ctx_hd(const CtxArgs &a, hipStream_t s)
{
    const int R = a.B * (a.heads / a.kvh);
    const int waves = R >= 64 ? 4 : (R + 15) / 16;
    const dim3 grid((unsigned)(a.N * a.kvh)), block(64 * waves);
    switch (a.S32 / 32) {
    case 1: hipLaunchKernelGGL((nsd_attn_ctx_kernel<1, HD>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((nsd_attn_ctx_kernel<2, HD>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((nsd_attn_ctx_kernel<3, HD>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((nsd_attn_ctx_kernel<4, HD>), grid, block, 0, s, a); break;
    case 5: hipLaunchKernelGGL((nsd_attn_ctx_kernel<5, HD>), grid, block, 0, s, a); break;
    case 6: hipLaunchKernelGGL((nsd_attn_ctx_kernel<6, HD>), grid, block, 0, s, a); break;
    case 7: hipLaunchKernelGGL((nsd_attn_ctx_kernel<7, HD>), grid, block, 0, s, a); break;
    case 8: hipLaunchKernelGGL((nsd_attn_ctx_kernel<8, HD>), grid, block, 0, s, a); break;
    default: return NSD_E_SIZE;
    }
    return (int)hipGetLastError();
}

template <class T, int G>
int launch_attn_g(const AttnArgs &a, hipStream_t s)
{
    const size_t q_bytes = ((size_t)G * a.hd * sizeof(T) + 15) & ~(size_t)15;
    const size_t per_wave = q_bytes + (size_t)G * 64 * 4 + 64 * 8;
    int waves = a.B < 16 ? a.B : 16;
    while (waves > 1 && per_wave * waves > 65536) --waves;
    if (per_wave > 65536) return NSD_E_SIZE;
    const dim3 grid((unsigned)(a.N * a.kvh), (unsigned)((a.B + waves - 1) / waves));
    hipLaunchKernelGGL((nsd_attn_step_kernel<T, G>), grid, dim3(64 * waves), per_wave * waves, s, a);
    return (int)hipGetLastError();
}

template <class T>
int launch_attn(const AttnArgs &a, hipStream_t s)
{
    switch (a.heads / a.kvh) {
    case 1: return launch_attn_g<T, 1>(a, s);
    case 2: return launch_attn_g<T, 2>(a, s);
    case 3: return launch_attn_g<T, 3>(a, s);
    case 4: return launch_attn_g<T, 4>(a, s);
    case 8: return launch_attn_g<T, 8>(a, s);
    default: return NSD_E_SIZE;
    }
}

} // namespace

extern "C" int nsd_ancestors(const uint8_t *backptr, int32_t t, int32_t N, int32_t B, uint8_t *anc, void *stream)
{
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (N < 0 || t < 0 || t > NSD_MAX_STEPS) return NSD_E_SIZE;
    if (N == 0 || t == 0) return NSD_OK;
    if (!backptr || !anc) return NSD_E_NULL;
    hipLaunchKernelGGL(nsd_ancestors_kernel, dim3((unsigned)N), dim3(64), (size_t)t * B, (hipStream_t)stream,
                       backptr, t, N * B, B, anc);
    return (int)hipGetLastError();
}

// every pointer given is a multiple of 16 (null pointers pass): the kernels move rows in 16-byte pieces
template <class... P>
static bool aligned16(P... p) { return ((((uintptr_t)p) | ...) & 15u) == 0; }

extern "C" int nsd_add_rmsnorm(const void *x, const void *delta, const void *w, float eps, int32_t rows, int32_t hidden,
                               int32_t dtype, void *x_out, void *h, void *stream)
{
    if (rows < 0 || hidden < 1 || (dtype != 0 && dtype != 1) || hidden % (dtype ? 8 : 4)) return NSD_E_SIZE;
    if (rows == 0) return NSD_OK;
    if (!x || !w || !h || (delta && !x_out)) return NSD_E_NULL;
    if (!aligned16(x, delta, w, x_out, h)) return NSD_E_ALIGN;
    if (dtype == 0)
        launch_add_rmsnorm<float>((const float *)x, (const float *)delta, (const float *)w, eps, rows, hidden,
                                  (float *)x_out, (float *)h, (hipStream_t)stream);
    else
        launch_add_rmsnorm<BF16>((const BF16 *)x, (const BF16 *)delta, (const BF16 *)w, eps, rows, hidden,
                                 (BF16 *)x_out, (BF16 *)h, (hipStream_t)stream);
    return (int)hipGetLastError();
}

extern "C" int nsd_rope_kv_append(const void *qkv, const float *cos_t, const float *sin_t, int32_t M, int32_t heads,
                                  int32_t kvh, int32_t hd, int32_t dtype, void *q_out, void *k_out, void *v_out,
                                  void *stream)
{
    if (M < 0 || heads < 1 || kvh < 1 || hd < 2 || (hd & 1) || (dtype != 0 && dtype != 1)) return NSD_E_SIZE;
    if (M == 0) return NSD_OK;
    if (!qkv || !cos_t || !sin_t || !q_out || !k_out || !v_out) return NSD_E_NULL;
    const int vn = dtype ? 8 : 4;
    const bool aligned = ((((uintptr_t)qkv) | ((uintptr_t)q_out) | ((uintptr_t)k_out) | ((uintptr_t)v_out)) & 15u) == 0;
    if ((hd / 2) % vn == 0 && aligned) {
        const long long n = (long long)M * (heads + 2 * kvh) * (hd / 2 / vn);
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        if (dtype == 0)
            hipLaunchKernelGGL(nsd_rope_kv_append_vec_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)qkv,
                               cos_t, sin_t, M, heads, kvh, hd, (float *)q_out, (float *)k_out, (float *)v_out);
        else
            hipLaunchKernelGGL(nsd_rope_kv_append_vec_kernel<BF16>, grid, block, 0, (hipStream_t)stream, (const BF16 *)qkv,
                               cos_t, sin_t, M, heads, kvh, hd, (BF16 *)q_out, (BF16 *)k_out, (BF16 *)v_out);
        return (int)hipGetLastError();
    }
    const long long n = (long long)M * (heads + 2 * kvh) * (hd / 2);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL(nsd_rope_kv_append_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)qkv,
                           cos_t, sin_t, M, heads, kvh, hd, (float *)q_out, (float *)k_out, (float *)v_out);
    else
        hipLaunchKernelGGL(nsd_rope_kv_append_kernel<BF16>, grid, block, 0, (hipStream_t)stream, (const BF16 *)qkv,
                           cos_t, sin_t, M, heads, kvh, hd, (BF16 *)q_out, (BF16 *)k_out, (BF16 *)v_out);
    return (int)hipGetLastError();
}

extern "C" int nsd_swiglu(const void *gate_up, int64_t rows, int32_t inter, int32_t dtype, void *out, void *stream)
{
    if (rows < 0 || inter < 1 || (dtype != 0 && dtype != 1) || inter % (dtype ? 8 : 4)) return NSD_E_SIZE;
    if (rows == 0) return NSD_OK;
    if (!gate_up || !out) return NSD_E_NULL;
    if (!aligned16(gate_up, out)) return NSD_E_ALIGN;
    const int per_row = inter / (dtype ? 8 : 4);
    const dim3 grid((unsigned)((per_row + 255) / 256), (unsigned)(rows < 65535 ? rows : 65535)), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL(nsd_swiglu_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)gate_up,
                           (long long)rows, inter, (float *)out);
    else
        hipLaunchKernelGGL(nsd_swiglu_kernel<BF16>, grid, block, 0, (hipStream_t)stream, (const BF16 *)gate_up,
                           (long long)rows, inter, (BF16 *)out);
    return (int)hipGetLastError();
}

extern "C" int nsd_attn_ctx(const void *q, const void *kctx_pad, const void *vctx_t, int32_t N, int32_t B, int32_t heads,
                            int32_t kvh, int32_t hd, int32_t S, int32_t S32, float scale, float *part_o, float *part_m,
                            float *part_l, void *stream)
{
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (N < 0 || heads < 1 || kvh < 1 || heads % kvh || S < 1 || S32 < S || S32 % 32 || S32 > 256 || (hd != 64 && hd != 128))
        return NSD_E_SIZE;
    if (N == 0) return NSD_OK;
    if (!q || !kctx_pad || !vctx_t || !part_o || !part_m || !part_l) return NSD_E_NULL;
    if (!aligned16(q, kctx_pad, vctx_t, part_o)) return NSD_E_ALIGN;
    CtxArgs a;
    a.q = (const BF16 *)q; a.kpad = (const BF16 *)kctx_pad; a.vt = (const BF16 *)vctx_t;
    a.po = part_o; a.pm = part_m; a.pl = part_l;
    a.N = N; a.B = B; a.heads = heads; a.kvh = kvh; a.S = S; a.S32 = S32; a.scale = scale;
    return hd == 64 ? launch_ctx_hd<64>(a, (hipStream_t)stream) : launch_ctx_hd<128>(a, (hipStream_t)stream);
}

extern "C" int nsd_attn_step(const void *q, const void *kctx, const void *vctx, const void *kgen, const void *vgen,
                             const uint8_t *anc, void *out, int32_t N, int32_t B, int32_t heads, int32_t kvh, int32_t hd,
                             int32_t S, int32_t t, float scale, int32_t dtype, const float *part_o, const float *part_m,
                             const float *part_l, void *stream)
{
    if (B < 1 || B > NSD_MAX_BEAM) return NSD_E_BEAM;
    if (N < 0 || heads < 1 || kvh < 1 || heads % kvh || S < 0 || t < 0 || (dtype != 0 && dtype != 1)) return NSD_E_SIZE;
    // rows are read in 16-byte pieces, hd / piece lanes per row in the value sum: a power of two of pieces
    {
        const int vn = dtype ? 8 : 4;
        if (hd < vn || hd > 256 || (hd & (hd - 1))) return NSD_E_SIZE;
    }
    if (N == 0) return NSD_OK;
    const bool part = part_o != nullptr;
    if (part && (!part_m || !part_l)) return NSD_E_NULL;
    if (!q || !kgen || !vgen || !out || (S > 0 && !part && (!kctx || !vctx)) || (t > 0 && !anc)) return NSD_E_NULL;
    if (!aligned16(q, kctx, vctx, kgen, vgen, out, part_o)) return NSD_E_ALIGN;
    AttnArgs a;
    a.part_o = part_o; a.part_m = part_m; a.part_l = part_l;
    a.q = q; a.kctx = kctx; a.vctx = vctx; a.kgen = kgen; a.vgen = vgen; a.anc = anc; a.out = out;
    a.N = N; a.B = B; a.heads = heads; a.kvh = kvh; a.hd = hd; a.S = S; a.t = t; a.M = N * B; a.scale = scale;
    return dtype == 0 ? launch_attn<float>(a, (hipStream_t)stream) : launch_attn<BF16>(a, (hipStream_t)stream);
}
