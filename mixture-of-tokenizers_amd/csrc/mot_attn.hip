// mot_attn.hip -- the cross-attention byte mixin (scaled-pre-train/train_gpt.py:243-300 CrossAttention,
// 446-464 ByteMixinCrossAttn), forward, fp32.
//
// Per token t and head h the reference projects q from the token embedding and k, v from every one of the
// T*bpt byte embeddings (two (T*bpt x D x D) GEMMs), rms-norms q and k per head, applies RoPE with the
// position in each one's own sequence, and takes a softmax over the token's bpt keys.
//
// MI355X formulation:
//   * With one id tensor, xkv[j] = norm(E_b[id_j]) takes only byte_rows (458) distinct values, and so do
//     k = norm_head(W_k xkv) and lambda * v = lambda * W_v xkv BEFORE RoPE.  They are projected once per
//     byte-table ROW (a 458-row GEMM instead of T*bpt rows: 2300x fewer flops at 64 k tokens x 16) into two
//     L2-resident tables; RoPE depends on the position and is applied where the key is used.
//     With two id tensors (norm(E[padded] + E[pulled]), train_gpt.py:378) a key depends on the id PAIR: that
//     case materialises xkv and projects all T*bpt rows, like the reference.
//   * q = W_q norm(E_t[tok]) and out = W_proj y run through the fused gather+MFMA kernel of mot_linear.hip
//     (dense-row mode), so the token rows are gathered and normalised inside the GEMM's A staging.
//   * cross_attn_kernel: one wave per (token, head); a lane owns elements i and i+64 of the 128-wide head, which
//     are exactly the pair RoPE rotates (Rotary.forward: halves x1 | x2), so norm, RoPE, scores and the weighted
//     sum are lane-local plus one wave reduction per key; online softmax over the bpt keys.  HBM traffic is
//     q in, y out and the cos/sin rows; keys and values come from L2.
//   * k and v are VIEWED as (H, T, bpt, hd) by the reference (lines 283-284) -- a reshape of (T*bpt, H, hd)
//     memory, not the transpose its comment names.  head_layout 0 reproduces that (key c of (h, t) is flat row
//     (h*T + t)*bpt + c -> position r / H, head r % H); head_layout 1 is the commented intent.
#include <float.h>
#include <string.h>

#include "mot_mix.hpp"

namespace mot {

constexpr int kHd = 128;  // head_dim of every CrossAttention the reference builds (train_gpt.py:459)

// a value every lane of the wave computed alike, moved into scalar registers: loop counters, positions and row bases derived from
// the wave's (token, head) then run on the scalar unit -- the attention kernels are bound by VALU issue (950 vector instructions
// per (token, head) in the forward before this: counters SQ_INSTS_VALU / SQ_WAVE_CYCLES, profiles/r03_cross_attn_pmc.txt)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v) {
    return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v));
}

struct AttnArgs {
    const float *q;       // [T, HD] projected queries
    float *y;             // [T, HD]
    __bf16 *y16;          // optional: y once more in bf16 (the row operand of c_proj on the bf16 MFMA: no narrowing pass)
    const float *kt, *vt; // [rows, HD]: per byte-table row (ids != null) or per kv position (ids == null)
    const __bf16 *kt16, *vt16;   // optional: the same two tables in bf16 (bf16 products: norm(k) and lambda v are bf16 tensors in the reference)
    const int64_t *ids;   // [T*bpt] byte ids or null
    int64_t rows;
    int64_t T;
    int bpt, H, layout;
    const float *cos_q, *sin_q, *cos_k, *sin_k;  // [len, 64]
    float eps;
    uint32_t *status;
};

// k <- rms_norm over each head; v <- lambda * v   (train_gpt.py:278, 280), one wave per (row, head)
__global__ __launch_bounds__(kThreads) void kv_finish_kernel(float *__restrict__ k, float *__restrict__ v, int64_t rows, int H,
                                                             const float *__restrict__ lambda, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (w >= rows * H) return;
    float *kp = k + w * kHd, *vp = v + w * kHd;
    const float k0 = kp[lane], k1 = kp[64 + lane];
    const float r = rms_scale(wave_sum(k0 * k0 + k1 * k1), kHd, eps);
    kp[lane] = k0 * r;
    kp[64 + lane] = k1 * r;
    const float lam = *lambda;
    vp[lane] = lam * vp[lane];
    vp[64 + lane] = lam * vp[64 + lane];
}

__global__ __launch_bounds__(kThreads) void cross_attn_kernel(const AttnArgs A) {
#pragma clang fp contract(fast)   // fused multiply-adds: the parity bars of this mixin are tolerances against float64, not bit patterns
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * kWaves + uni(threadIdx.x >> 6);
    if (w >= A.T * A.H) return;
    const int64_t t = uni64(w / A.H);
    const int h = uni((int)(w - t * A.H));
    const int HD = A.H * kHd;
    // q: per-head rms norm, then RoPE at position t (lines 278-279)
    const float *qp = A.q + t * HD + h * kHd;
    float q0 = qp[lane], q1 = qp[64 + lane];
    const float rq = rms_scale(wave_sum(q0 * q0 + q1 * q1), kHd, A.eps);
    q0 *= rq;
    q1 *= rq;
    {
        const float c = A.cos_q[t * 64 + lane], s = A.sin_q[t * 64 + lane];
        const float a = q0 * c + q1 * s, b = q0 * (-s) + q1 * c;
        q0 = a;
        q1 = b;
    }
    const float inv_sqrt_hd = 1.0f / sqrtf((float)kHd);   // line 286 divides by sqrt(head_dim): kHd = 128, the product differs by <= 1 ulp
    // flat row r of the (T*bpt, H, hd) key/value memory = pos * H + hk; walked incrementally (one division per wave)
    int64_t pos0 = t * A.bpt;
    int hk0 = h;
    if (A.layout == 0) {
        const int64_t r0 = ((int64_t)h * A.T + t) * A.bpt;
        pos0 = uni64(r0 / A.H);
        hk0 = uni((int)(r0 - pos0 * A.H));
    }
    auto advance = [&](int64_t &pos, int &hk) {
        if (A.layout == 0) { if (++hk == A.H) { hk = 0; ++pos; } } else ++pos;
    };
    // lane c < bpt looks up the table row of key c once (the per-key loads below then start from a scalar row index instead
    // of waiting for an id load each)
    int rowv = 0;
    if (lane < A.bpt) {
        int64_t p = pos0 + lane;
        if (A.layout == 0) p = (((int64_t)h * A.T + t) * A.bpt + lane) / A.H;
        int64_t row = p;
        if (A.ids) {
            row = A.ids[p];
            if ((uint64_t)row >= (uint64_t)A.rows) {
                if (A.status) atomicOr(A.status, kStatusByteOor);
                row = 0;
            }
        }
        rowv = (int)row;
    }
    // pass 1: scores, lane c keeps s_c; then the softmax over the bpt keys across lanes (one exp for all of them, line 287)
    float sc = -FLT_MAX;
    {
        // Two keys per trip: their eight loads are in flight together and their two wave sums interleave (a DPP step has to wait two
        // cycles for its operand -- alone, the chain is padded with s_nop on the scalar unit, which all four SIMDs of a CU share and
        // which was as busy as the vector unit here: 740 scalar instructions per (token, head)).
        int64_t pos = pos0, pos_cs = -1; int hk = hk0;
        float ck = 0.f, sk = 0.f;
        auto partial = [&](int c) {   // this lane's part of q . rope(k_c); moves (pos, hk) on to the next key
            const float *kp = A.kt + (int64_t)__builtin_amdgcn_readlane(rowv, c) * HD + hk * kHd;
            const float k0 = kp[lane], k1 = kp[64 + lane];
            if (pos != pos_cs) { ck = A.cos_k[pos * 64 + lane]; sk = A.sin_k[pos * 64 + lane]; pos_cs = pos; }   // as_viewed walks H heads per position
            const float ka = k0 * ck + k1 * sk, kb = k0 * (-sk) + k1 * ck;
            advance(pos, hk);
            return q0 * ka + q1 * kb;
        };
        int c = 0;
        for (; c + 1 < A.bpt; c += 2) {
            const float pa = partial(c), pb = partial(c + 1);
            const float sa = wave_sum(pa) * inv_sqrt_hd, sb = wave_sum(pb) * inv_sqrt_hd;
            if (lane == c) sc = sa;
            if (lane == c + 1) sc = sb;
        }
        if (c < A.bpt) {
            const float s = wave_sum(partial(c)) * inv_sqrt_hd;
            if (lane == c) sc = s;
        }
    }
    const float mx = wave_max(sc);
    float p = lane < A.bpt ? expf(sc - mx) : 0.f;
    p /= wave_sum(p);
    // pass 2: y = sum_c p_c v_c   (line 289)
    float y0 = 0.f, y1 = 0.f;
    {
        int hk = hk0;
        int64_t pos = pos0;
        auto value = [&](int c, float &v0, float &v1) {
            const float *vp = A.vt + (int64_t)__builtin_amdgcn_readlane(rowv, c) * HD + hk * kHd;
            v0 = vp[lane]; v1 = vp[64 + lane];
            advance(pos, hk);
            return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), c));
        };
        int c = 0;
        for (; c + 1 < A.bpt; c += 2) {   // (two keys per trip, as above; the sums keep their order)
            float a0, a1, b0, b1;
            const float pa = value(c, a0, a1), pb = value(c + 1, b0, b1);
            y0 += pa * a0; y1 += pa * a1;
            y0 += pb * b0; y1 += pb * b1;
        }
        if (c < A.bpt) {
            float a0, a1;
            const float pa = value(c, a0, a1);
            y0 += pa * a0; y1 += pa * a1;
        }
    }
    float *yp = A.y + t * HD + h * kHd;
    yp[lane] = y0;
    yp[64 + lane] = y1;
    if (A.y16) {
        A.y16[t * HD + h * kHd + lane] = (__bf16)y0;
        A.y16[t * HD + h * kHd + 64 + lane] = (__bf16)y1;
    }
}


// The same attention for bpt <= 16, a lane per (key, quarter of the head) instead of a loop over the keys: lane 4c + j holds the
// dims 16i + 4j .. + 3 (i = 0..7) of key c -- every 16-byte load of a wave-instruction takes 64 contiguous bytes of each of 16 rows
// -- rotates them with its own position's cos / sin and multiplies with the query, which the wave normalises and rotates once in
// the one-lane-per-dim form and passes through 512 bytes of LDS.  A score is then a sum over 32 products in a lane and two quad
// permutes (the loop: 2 products and a 6-step wave sum PER KEY, 950 vector and 400 scalar instructions per (token, head); this
// form about 300 and 60).  The values are summed a lane per (key parity, 4 dims): 8 row-wide loads, weights by v_readlane.
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_move<kDppQuadXor1, 0xf>(0.f, v);
    v += dpp_move<kDppQuadXor2, 0xf>(0.f, v);
    return v;
}
constexpr int kQuadPos = 4;         // positions whose cos / sin rows a wave stages in LDS
constexpr int kQuadPosStride = 80;  // floats between them: 64 + 16, so that the quads of keys on different positions read different banks
// The 32 dims of a head a lane of the lane-per-(key, quarter) kernels holds, as 16 pairs (d, d + 64) -- the pairs RoPE rotates -- n = 0..15:
//   fp32 rows: d = 16 i + 4 j + e, n = 4 i + e  (16-byte pieces of 4 floats);   bf16 rows (L8): d = 32 i + 8 j + e, n = 8 i + e  (16-byte
//   pieces of 8 bf16: the texture addresser takes a quad of lanes per cycle whatever the width of their pieces, so a row of half the
//   bytes only costs half when it comes in half as many 16-byte pieces -- 8-byte pieces in the fp32 mapping were measured: no gain).
template <bool L8>
__device__ __forceinline__ void load16(const float *base, int j, float (&o)[16]) {   // o[n] = base[d(n)], base in LDS or global memory
    if constexpr (L8) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 a = *(const float4 *)(base + 32 * i + 8 * j), b = *(const float4 *)(base + 32 * i + 8 * j + 4);
            o[8 * i] = a.x; o[8 * i + 1] = a.y; o[8 * i + 2] = a.z; o[8 * i + 3] = a.w; o[8 * i + 4] = b.x; o[8 * i + 5] = b.y; o[8 * i + 6] = b.z; o[8 * i + 7] = b.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 a = *(const float4 *)(base + 16 * i + 4 * j);
            o[4 * i] = a.x; o[4 * i + 1] = a.y; o[4 * i + 2] = a.z; o[4 * i + 3] = a.w;
        }
    }
}
template <bool L8>
__device__ __forceinline__ void store16(float *base, int j, const float (&o)[16]) {
    if constexpr (L8) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(float4 *)(base + 32 * i + 8 * j) = make_float4(o[8 * i], o[8 * i + 1], o[8 * i + 2], o[8 * i + 3]);
            *(float4 *)(base + 32 * i + 8 * j + 4) = make_float4(o[8 * i + 4], o[8 * i + 5], o[8 * i + 6], o[8 * i + 7]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(float4 *)(base + 16 * i + 4 * j) = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    }
}
// one head slice of a table row (element offset off, a multiple of 128) into the two halves: fp32 rows 8 x 16 bytes, bf16 rows 4 x 16 bytes
template <bool L8>
__device__ __forceinline__ void load_row32(const void *table, int64_t off4, int j, float (&lo)[16], float (&hi)[16]) {
    if constexpr (L8) {
        const uint4 *p = (const uint4 *)table + (off4 >> 1) + j;
        uint4 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = p[4 * i];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t a[4] = {r[i].x, r[i].y, r[i].z, r[i].w}, b[4] = {r[i + 2].x, r[i + 2].y, r[i + 2].z, r[i + 2].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[8 * i + 2 * e] = __uint_as_float(a[e] << 16); lo[8 * i + 2 * e + 1] = __uint_as_float(a[e] & 0xffff0000u);
                hi[8 * i + 2 * e] = __uint_as_float(b[e] << 16); hi[8 * i + 2 * e + 1] = __uint_as_float(b[e] & 0xffff0000u);
            }
        }
    } else {
        const float4 *p = (const float4 *)table + off4 + j;
        float4 r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = p[4 * i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[4 * i] = r[i].x; lo[4 * i + 1] = r[i].y; lo[4 * i + 2] = r[i].z; lo[4 * i + 3] = r[i].w;
            hi[4 * i] = r[i + 4].x; hi[4 * i + 1] = r[i + 4].y; hi[4 * i + 2] = r[i + 4].z; hi[4 * i + 3] = r[i + 4].w;
        }
    }
}
__device__ __forceinline__ float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <bool KV16>   // KV16: keys and values from the bf16 tables
__global__ __launch_bounds__(kThreads) void cross_attn_quad_kernel(const AttnArgs A) {
#pragma clang fp contract(fast)
    __shared__ __attribute__((aligned(16))) float qs[kWaves][kHd];
    __shared__ __attribute__((aligned(16))) float cs[kWaves][2][kQuadPos * kQuadPosStride];
    const int lane = threadIdx.x & 63, wv = uni(threadIdx.x >> 6);
    const int64_t w = (int64_t)blockIdx.x * kWaves + wv;
    if (w >= A.T * A.H) return;
    // which (token, head): head-major for the viewed layout -- the keys of (h, t) are the flat rows (h T + t) bpt + c, so waves that
    // follow each other share positions (cos / sin rows) and, all being on one head, gather from one 128-wide slice of the two
    // tables (458 x 512 bytes each, a few dozen rows of it hot: L1-sized); token-major for the per-head layout, whose H heads of a
    // token share their bpt positions
    int64_t t; int h;
    if (A.layout == 0) { h = uni((int)(w / A.T)); t = w - (int64_t)h * A.T; }
    else { t = uni64(w / A.H); h = uni((int)(w - t * A.H)); }
    const int HD = A.H * kHd;
    {   // q: per-head rms norm, then RoPE at position t (lines 278-279), one lane per pair of dims
        const float *qp = A.q + t * HD + h * kHd;
        float q0 = qp[lane], q1 = qp[64 + lane];
        const float rq = rms_scale(wave_sum(q0 * q0 + q1 * q1), kHd, A.eps);
        q0 *= rq;
        q1 *= rq;
        const float c = A.cos_q[t * 64 + lane], s = A.sin_q[t * 64 + lane];
        qs[wv][lane] = q0 * c + q1 * s;
        qs[wv][64 + lane] = q0 * (-s) + q1 * c;
    }
    // this lane's key: flat row r of the (T*bpt, H, hd) key / value memory = pos * H + hk
    const int c = lane >> 2, j = lane & 3;
    const bool live = c < A.bpt;
    const int ce = live ? c : A.bpt - 1;   // lanes past the last key repeat it (valid addresses) with weight 0
    int64_t pos = t * A.bpt + ce;
    int hk = h, dpos = 0;
    bool staged = false;
    if (A.layout == 0) {
        const int64_t r0 = ((int64_t)h * A.T + t) * A.bpt;
        const int64_t pos0 = uni64(r0 / A.H);
        const int hk0 = uni((int)(r0 - pos0 * A.H)), hc = hk0 + ce;
        dpos = hc / A.H;
        pos = pos0 + dpos;
        hk = hc - dpos * A.H;
        // the bpt keys sit on (hk0 + bpt - 1) / H + 1 positions (4 of them for 16 keys of 6 heads): their cos / sin rows are loaded once,
        // a lane per pair of dims, and handed to the lanes of their keys through LDS -- the texture addresser, 64 bytes a cycle, is what
        // this kernel keeps busy (TA_BUSY 85 % with every lane loading the 128 bytes of its own key's rows)
        const int np = (hk0 + A.bpt - 1) / A.H + 1;
        staged = np <= kQuadPos;
        if (staged) {
            for (int n = 0; n < np; ++n) {
                cs[wv][0][n * kQuadPosStride + lane] = A.cos_k[(pos0 + n) * 64 + lane];
                cs[wv][1][n * kQuadPosStride + lane] = A.sin_k[(pos0 + n) * 64 + lane];
            }
        }
    }
    int64_t row = pos;
    if (A.ids) {
        row = A.ids[pos];
        if ((uint64_t)row >= (uint64_t)A.rows) {
            if (A.status) atomicOr(A.status, kStatusByteOor);
            row = 0;
        }
    }
    const int off4 = (int)((row * HD + hk * kHd) >> 2);   // this key's head slice, in 16-byte units (the host checks rows * HD < 2^33)
    float klo[16], khi[16], ck[16], sk[16];
    load_row32<KV16>(KV16 ? (const void *)A.kt16 : (const void *)A.kt, off4, j, klo, khi);
    __builtin_amdgcn_wave_barrier();   // (LDS operations of one wave complete in order: the rotated query and the cos / sin rows are there)
    if (staged) {
        load16<KV16>(&cs[wv][0][dpos * kQuadPosStride], j, ck);
        load16<KV16>(&cs[wv][1][dpos * kQuadPosStride], j, sk);
    } else {
        load16<KV16>(A.cos_k + pos * 64, j, ck);
        load16<KV16>(A.sin_k + pos * 64, j, sk);
    }
    float s = 0.f;
    {
        float qa[16], qb[16];
        load16<KV16>(&qs[wv][0], j, qa);
        load16<KV16>(&qs[wv][64], j, qb);
#pragma unroll
        for (int n = 0; n < 16; ++n) s += qa[n] * (klo[n] * ck[n] + khi[n] * sk[n]) + qb[n] * (khi[n] * ck[n] - klo[n] * sk[n]);
    }
    const float inv_sqrt_hd = 1.0f / sqrtf((float)kHd);
    const float sc = live ? quad_sum(s) * inv_sqrt_hd : -FLT_MAX;
    const float mx = wave_max(sc);
    float p = live ? expf(sc - mx) : 0.f;
    p /= wave_sum(j == 0 ? p : 0.f);
    if constexpr (KV16) {
        // y = sum_c p_c v_c: four groups of 16 lanes, group g the keys 4 cc + g, a lane 8 dims (one 16-byte piece of the bf16 row)
        const int g = lane >> 4, m = lane & 15;
        const uint4 *vt8 = (const uint4 *)A.vt16 + m;
        const int off8 = off4 >> 1;
        float y[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto four_keys = [&](int cc) {
            const int src = 4 * (4 * cc + g);   // a lane of that key's quad
            const float pc = __shfl(p, src, 64);
            const uint4 r = vt8[__shfl(off8, src, 64)];
            const uint32_t u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { y[2 * e] += pc * __uint_as_float(u[e] << 16); y[2 * e + 1] += pc * __uint_as_float(u[e] & 0xffff0000u); }
        };
        if (A.bpt > 12) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) four_keys(cc);
        } else {
            for (int cc = 0; 4 * cc < A.bpt; ++cc) four_keys(cc);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { y[e] += __shfl_xor(y[e], 16, 64); y[e] += __shfl_xor(y[e], 32, 64); }
        const int64_t o = t * HD + h * kHd + 8 * m;
        if (g == 0) {
            *(float4 *)(A.y + o) = make_float4(y[0], y[1], y[2], y[3]);
            *(float4 *)(A.y + o + 4) = make_float4(y[4], y[5], y[6], y[7]);
        } else if (g == 1 && A.y16) {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8 b;
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] = (__bf16)y[e];
            *(bf16x8 *)(A.y16 + o) = b;
        }
        return;
    }
    // y = sum_c p_c v_c: lanes 0..31 take the even keys, 32..63 the odd ones, 4 dims each
    const int g = lane >> 5, m = lane & 31;
    const float4 *vt4 = (const float4 *)A.vt + m;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    auto two_keys = [&](int cc) {
        const int l0 = 8 * cc, l1 = 8 * cc + 4;
        const int o0 = __builtin_amdgcn_readlane(off4, l0), o1 = __builtin_amdgcn_readlane(off4, l1);
        const float p0 = lane_value(p, l0), p1 = lane_value(p, l1);
        const float4 v = vt4[g ? o1 : o0];
        const float pc = g ? p1 : p0;
        y.x += pc * v.x; y.y += pc * v.y; y.z += pc * v.z; y.w += pc * v.w;
    };
    if (A.bpt > 14) {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) two_keys(cc);
    } else {
        for (int cc = 0; 2 * cc < A.bpt; ++cc) two_keys(cc);
    }
    y.x += __shfl_xor(y.x, 32, 64); y.y += __shfl_xor(y.y, 32, 64); y.z += __shfl_xor(y.z, 32, 64); y.w += __shfl_xor(y.w, 32, 64);
    const int64_t o = t * HD + h * kHd + 4 * m;
    if (g == 0) *(float4 *)(A.y + o) = y;
    else if (A.y16) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 b; b[0] = (__bf16)y.x; b[1] = (__bf16)y.y; b[2] = (__bf16)y.z; b[3] = (__bf16)y.w;
        *(bf16x4 *)(A.y16 + o) = b;
    }
}


// the lane-per-(key, quarter) kernels: at most 16 keys, 16-byte loads of rows they address in 16-byte units with 32 bits
static bool quad_form(int bpt, int64_t rows, int HD, const void *k, const void *v, const void *cos_k, const void *sin_k) {
    bool ok = bpt <= 16 && (uint64_t)rows * HD < ((uint64_t)1 << 33) && !(((uintptr_t)k | (uintptr_t)v | (uintptr_t)cos_k | (uintptr_t)sin_k) & 15);
#ifdef MOT_DEV_ABLATION
    if (getenv("MOT_ATTN_LOOP")) ok = false;
#endif
    return ok;
}
static int launch_attention(const AttnArgs &A, hipStream_t stream) {
    const int64_t waves = A.T * A.H;
    const dim3 grid((unsigned)((waves + kWaves - 1) / kWaves));
    if (quad_form(A.bpt, A.rows, A.H * kHd, A.kt, A.vt, A.cos_k, A.sin_k) && !(((uintptr_t)A.y | (uintptr_t)A.y16) & 15))
    {
        if (A.kt16 && A.vt16) hipLaunchKernelGGL(cross_attn_quad_kernel<true>, grid, dim3(kThreads), 0, stream, A);
        else hipLaunchKernelGGL(cross_attn_quad_kernel<false>, grid, dim3(kThreads), 0, stream, A);
    } else hipLaunchKernelGGL(cross_attn_kernel, grid, dim3(kThreads), 0, stream, A);
    return check_launch("cross_attn_kernel");
}

// ------------------------------------------------------------------------------------------ host
// workspace, in floats: [q: T*HD][y: T*HD][kt: R*HD][vt: R*HD][xkv: R*D (dual: R = T*bpt)][xq: D > HD ? T*D : 0]
struct AttnLayout { size_t q, y, kt, vt, xkv, xq, a16, w16, part, part_n, kt16, vt16, total; int64_t R; };
// matmul_dtype == MOT_BF16: the two products over the tokens (q = W_q xq, out = c_proj y) run on the bf16 MFMA
// (launch_gemm_rows_bf16, fp32 accumulation and fp32 results): their row operands are rounded to bf16 first -- the reference's own
// rounding points in the production cast (xq is a bf16 tensor out of norm(), y one out of the attention, train_gpt.py:277, 292-293;
// the weights are cast where they are used, 185-186, 277-278) -- everything else stays as it is.
static bool mm16(const MotCrossAttnDesc &d) { return d.matmul_dtype == MOT_BF16; }
// out[n][Nc] = A[n][R] . W[Nc][R]^T: A, W fp32 -> bf16 copies in a16 / w16 -> bf16 MFMA, fp32 out
static int dense_rows_bf16(const float *A, int64_t n, int R, const float *W, int Nc, float *out, void *a16, void *w16, hipStream_t stream) {
    int rc;
    if ((rc = launch_narrow(A, n * R, a16, stream))) return rc;
    if ((rc = launch_narrow(W, (int64_t)Nc * R, w16, stream))) return rc;
    return launch_gemm_rows_bf16(a16, R, n, w16, R, R, Nc, out, Nc, false, nullptr, stream);
}

static AttnLayout attn_layout(const MotCrossAttnDesc &d) {
    AttnLayout L;
    const size_t T = (size_t)d.n_tokens, HD = (size_t)d.n_heads * kHd, D = (size_t)d.dim;
    const bool dual = d.ids_b != nullptr;
    L.R = dual ? (int64_t)(T * d.bpt) : d.byte_rows;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    L.q = take(T * HD); L.y = take(T * HD); L.kt = take((size_t)L.R * HD); L.vt = take((size_t)L.R * HD);
    L.xkv = take(dual ? T * d.bpt * D : (size_t)L.R * D);   // the (normalised) key/value source rows: per kv position, or per byte-table row
    L.xq = take(D > HD ? T * D : 0);   // the gathered query rows live in y's place until the attention writes y
    const size_t wide = D > HD ? D : HD;
    // bf16 copies: the row operand of a product, its weight (two id tensors: the key / value source rows of every kv position, kv_w)
    const size_t a16 = dual && T * d.bpt * D > T * wide ? T * d.bpt * D : T * wide;
    L.a16 = take(mm16(d) ? (a16 + 1) / 2 : 0);
    L.w16 = take(mm16(d) ? ((dual ? 2 : 1) * HD * D + 1) / 2 : 0);
    L.part_n = gemm_rows_sliced_floats(L.R, (int)D, (int)HD);   // the key / value projections of the few byte-table rows, cut along dim
    L.part = take(L.part_n);
    // bf16 products, one id tensor: the two tables once more in bf16 for the attention kernel (per kv position they would cost a pass
    // over 2 x T*bpt x HD to make, what reading them in bf16 saves)
    const size_t t16 = mm16(d) && !dual ? ((size_t)L.R * HD + 1) / 2 : 0;
    L.kt16 = take(t16); L.vt16 = take(t16);
    L.total = o;
    return L;
}

size_t cross_attn_workspace_bytes(const MotCrossAttnDesc &d) { return attn_layout(d).total * 4; }

__global__ void rows_norm_kernel(const float *__restrict__ table, int64_t rows, int D, int norm, float eps, float *__restrict__ xn);

int launch_cross_attn(const MotCrossAttnDesc &d, hipStream_t stream) {
    const int64_t T = d.n_tokens;
    const int H = d.n_heads, HD = H * kHd, D = d.dim;
    const AttnLayout L = attn_layout(d);
    if (!d.workspace || d.workspace_bytes < L.total * 4)
        return set_error(MOT_EWORKSPACE, "cross_attn: needs %zu workspace bytes, got %zu", L.total * 4, d.workspace_bytes);
    float *ws = (float *)d.workspace;
    float *q = ws + L.q, *y = ws + L.y, *kt = ws + L.kt, *vt = ws + L.vt, *xkv = ws + L.xkv;
    const float eps = d.eps > 0.f ? d.eps : FLT_EPSILON;
    const bool dual = d.ids_b != nullptr;
    if (d.saved_qy) {   // kept for the backward
        q = (float *)d.saved_qy;
        y = q + (size_t)T * HD;
    }
    const bool kv_cached = !dual && d.kv_tables != nullptr;
    if (kv_cached) {   // caller-kept tables: built by this call unless it says they are current
        kt = (float *)d.kv_tables;
        vt = kt + (size_t)L.R * HD;
    }
    int rc;
    // 1. q = W_q norm?(E_t[tok])          (train_gpt.py:348-377 + 277): seam gather, then the plain dense MFMA kernel
    float *xq = D > HD ? ws + L.xq : y;
    if (mm16(d) && d.tok_table_bf16) {   // the caller's bf16 table: the normalised rows come out in bf16 directly (the same roundings, two passes less)
        if ((rc = launch_gather_rows(d.tokens, nullptr, 4, T, d.tok_table_bf16, d.tok_rows, (int)D, d.norm_tok, eps, nullptr, ws + L.a16, d.status, MOT_BF16,
                                     stream))) return rc;
        if ((rc = launch_narrow((const float *)d.q_w, (int64_t)HD * D, ws + L.w16, stream))) return rc;
        rc = launch_gemm_rows_bf16(ws + L.a16, D, T, ws + L.w16, D, D, HD, q, HD, false, nullptr, stream);
    } else {
        if ((rc = launch_gather_rows(d.tokens, nullptr, 4, T, d.tok_table, d.tok_rows, (int)D, d.norm_tok, eps, nullptr, xq, d.status, MOT_F32, stream)))
            return rc;
        if (mm16(d)) rc = dense_rows_bf16(xq, T, D, (const float *)d.q_w, HD, q, ws + L.a16, ws + L.w16, stream);
        else rc = launch_gemm_rows(xq, (int)D, T, (const float *)d.q_w, (int)D, (int)D, (int)HD, q, (int)HD, true, stream);
    }
    if (rc) return rc;
    // 2. key/value rows: per byte-table row, or per kv position when the embedding is norm(E[a] + E[b])
    const int kv_norm = d.norm_byte;
    if (dual) {
        if ((rc = launch_gather_rows(d.ids_a, d.ids_b, 8, T * d.bpt, d.byte_table, d.byte_rows, D, d.norm_byte, eps, nullptr, xkv, d.status,
                                     MOT_F32, stream))) return rc;
    }
    if (!(kv_cached && d.kv_tables_ready)) {
        const float *kv_w = (const float *)d.kv_w;
        if (!dual) {   // rows of the byte table, normalised once (458 rows): the plain dense kernel then projects them
            hipLaunchKernelGGL(rows_norm_kernel, dim3((unsigned)((L.R + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, (const float *)d.byte_table, L.R,
                               (int)D, kv_norm, eps, xkv);
            if ((rc = check_launch("rows_norm_kernel"))) return rc;
        }
        if (dual && mm16(d)) {   // a row per kv position: these two are 2 bpt times the flops of q -- on the bf16 MFMA like it (xkv is a bf16 tensor in the reference)
            __bf16 *a16 = (__bf16 *)(ws + L.a16), *w16 = (__bf16 *)(ws + L.w16);
            if ((rc = launch_narrow(xkv, L.R * D, a16, stream))) return rc;
            if ((rc = launch_narrow(kv_w, (int64_t)2 * HD * D, w16, stream))) return rc;
            if ((rc = launch_gemm_rows_bf16(a16, D, L.R, w16, D, D, HD, kt, HD, false, nullptr, stream))) return rc;
            if ((rc = launch_gemm_rows_bf16(a16, D, L.R, w16 + (size_t)HD * D, D, D, HD, vt, HD, false, nullptr, stream))) return rc;
        } else {
            // (one id tensor: 458 rows -- 24 output blocks of the plain kernel; cut along dim they fill the chip: 72 -> 15 us each)
            if ((rc = launch_gemm_rows_sliced(xkv, (int)D, L.R, kv_w, (int)D, (int)D, (int)HD, kt, (int)HD, true, ws + L.part, L.part_n, stream))) return rc;
            if ((rc = launch_gemm_rows_sliced(xkv, (int)D, L.R, kv_w + (size_t)HD * D, (int)D, (int)D, (int)HD, vt, (int)HD, true, ws + L.part, L.part_n, stream)))
                return rc;
        }
        const int64_t kvw = L.R * H;
        hipLaunchKernelGGL(kv_finish_kernel, dim3((unsigned)((kvw + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, kt, vt, L.R, H,
                           d.lambda_factor, eps);
        if ((rc = check_launch("kv_finish_kernel"))) return rc;
    }
    // 3. attention of every token over its own bpt keys
    AttnArgs A;
    A.q = q; A.y = y; A.kt = kt; A.vt = vt; A.ids = dual ? nullptr : d.ids_a; A.rows = L.R; A.T = T; A.bpt = d.bpt; A.H = H;
    A.layout = d.head_layout; A.cos_q = d.cos_q; A.sin_q = d.sin_q; A.cos_k = d.cos_k; A.sin_k = d.sin_k; A.eps = eps; A.status = d.status;
    A.y16 = mm16(d) ? (__bf16 *)(ws + L.a16) : nullptr;   // (the row operand of q has been consumed: stream order)
    A.kt16 = A.vt16 = nullptr;
    if (mm16(d) && !dual && d.bpt <= 16) {
        if ((rc = launch_narrow(kt, L.R * HD, ws + L.kt16, stream))) return rc;
        if ((rc = launch_narrow(vt, L.R * HD, ws + L.vt16, stream))) return rc;
        A.kt16 = (const __bf16 *)(ws + L.kt16); A.vt16 = (const __bf16 *)(ws + L.vt16);
    }
    if ((rc = launch_attention(A, stream))) return rc;
    // 4. out = c_proj y                   (line 293): out[t][c] = sum_r y[t][r] * proj_w[c][r]
    if (mm16(d)) {
        if ((rc = launch_narrow((const float *)d.proj_w, (int64_t)D * HD, ws + L.w16, stream))) return rc;
        return launch_gemm_rows_bf16(ws + L.a16, HD, T, ws + L.w16, HD, HD, D, d.out, D, d.io_dtype == MOT_BF16, nullptr, stream);
    }
    return launch_gemm_rows(y, (int)HD, T, (const float *)d.proj_w, (int)HD, (int)HD, (int)D, (float *)d.out, (int)D, true, stream);
}


// ==========================================================================================
// Backward (loss.backward() through ByteMixinCrossAttn + FlexibleEmbedding, train_gpt.py:1319).  One id tensor; with two
// (add_padded_and_pulled: xkv = norm?(E[a] + E[b]) per kv position, train_gpt.py:364-372) the "table" below has one row per kv
// position -- same kernels, identity grouping, and the byte-table gradient is the norm's backward per position followed by the plain
// embedding backward once per id tensor.
//   forward recompute   q_pre = W_q xq;  k_pre, v_pre per byte-table row;  k_n = norm_head(k_pre);  y (attention)
//   dW_p += g^T y                      gemm_tn            dy = g W_p            dense GEMM (prebuilt k-major operand)
//   cross_attn_bwd_kernel, one wave per (token, head): softmax weights p_c across lanes (lane c holds key c),
//       keeps p_c, ds_c [T, H, bpt] and the rotated queries [T, HD]; per key slot (flat row r = pos*H + head, owned by exactly one query)
//       dV_l[r] = p_c dy,  dk_n[r] = rope^T(ds_c q_r)  -- formed and summed per byte id by attn_kv_rows_bwd_kernel, never stored
//       dq_pre  = norm_head^T(rope^T(sum_c ds_c k_r))        -> dq [T, HD]
//   per byte-table row: dkn_tab / dvl_tab = sum of those slot gradients over the kv positions with that id (positions grouped by
//       launch_group_positions, attn_kv_rows_bwd_kernel)
//   kv_table_bwd_kernel: d lambda, d v_pre = lambda dV_l, d k_pre = norm_head^T(dk_n)   -> dkv [R, 2 HD]
//   dW_kv += dkv^T xkv_tab (gemm_tn, R rows);  dxkv_tab = dkv W_kv (dense GEMM);  byte_rows_bwd_kernel: norm^T, d_byte += (rows ARE table rows)
//   dW_q += dq^T xq (gemm_tn);  dxq = dq W_q (dense GEMM);  token-table gradient = launch_embed_mix_bwd (NOOP, norm_tok) on dxq
// ==========================================================================================
struct AttnBwdArgs {
    const float *q_pre, *dy;   // [T, HD]
    const float *kn, *vpre;    // [rows, HD]
    const __bf16 *kn16, *vl16; // optional: norm(k) and lambda v in bf16, as the forward read them
    const float *lambda;
    const int64_t *ids;
    int64_t rows, T;
    int bpt, H, layout;
    const float *cos_q, *sin_q, *cos_k, *sin_k;
    float eps;
    float *dq;
    __bf16 *dq16;      // when given, dq is written here in bf16 INSTEAD (it is only ever the row operand of two bf16 products)
    float *pw, *dsw;   // [T, H, bpt]: softmax weight and score gradient of every (token, head, key)
    float *qrot;       // [T, HD]: the normalised, rotated queries
};

__global__ __launch_bounds__(kThreads) void cross_attn_bwd_kernel(const AttnBwdArgs A) {
#pragma clang fp contract(fast)
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * kWaves + uni(threadIdx.x >> 6);
    if (w >= A.T * A.H) return;
    const int64_t t = uni64(w / A.H);
    const int h = uni((int)(w - t * A.H));
    const int HD = A.H * kHd;
    const float lam = *A.lambda;
    const float inv_sqrt = 1.0f / sqrtf((float)kHd);
    // q: head norm + RoPE, as in the forward
    const float *qp = A.q_pre + t * HD + h * kHd;
    const float qp0 = qp[lane], qp1 = qp[64 + lane];
    const float rq = rms_scale(wave_sum(qp0 * qp0 + qp1 * qp1), kHd, A.eps);
    const float qn0 = qp0 * rq, qn1 = qp1 * rq;
    const float cq = A.cos_q[t * 64 + lane], sq = A.sin_q[t * 64 + lane];
    const float q0 = qn0 * cq + qn1 * sq, q1 = qn0 * (-sq) + qn1 * cq;
    const float dy0 = A.dy[t * HD + h * kHd + lane], dy1 = A.dy[t * HD + h * kHd + 64 + lane];
    int64_t pos0 = t * A.bpt;
    int hk0 = h;
    if (A.layout == 0) {
        const int64_t r0 = ((int64_t)h * A.T + t) * A.bpt;
        pos0 = uni64(r0 / A.H);
        hk0 = uni((int)(r0 - pos0 * A.H));
    }
    auto advance = [&](int64_t &pos, int &hk) {
        if (A.layout == 0) { if (++hk == A.H) { hk = 0; ++pos; } } else ++pos;
    };
    // lane c < bpt looks up the table row of key c once; the passes below start from that scalar row index
    int rowv = 0;
    if (lane < A.bpt) {
        int64_t p = pos0 + lane;
        if (A.layout == 0) p = (((int64_t)h * A.T + t) * A.bpt + lane) / A.H;
        int64_t row = p;   // two id tensors: the key / value rows are per kv position
        if (A.ids) {
            row = A.ids[p];
            if ((uint64_t)row >= (uint64_t)A.rows) row = 0;   // flagged by the forward
        }
        rowv = (int)row;
    }
    auto row_at = [&](int c) { return (int64_t)__builtin_amdgcn_readlane(rowv, c); };
    // pass 1: scores; lane c keeps s_c
    float sc = -FLT_MAX;
    {
        int64_t pos = pos0, pos_cs = -1; int hk = hk0;
        float ck = 0.f, sk = 0.f;
        auto partial = [&](int c) {   // two keys per trip in every pass, as in the forward kernel
            const float *kp = A.kn + row_at(c) * HD + hk * kHd;
            const float k0 = kp[lane], k1 = kp[64 + lane];
            if (pos != pos_cs) { ck = A.cos_k[pos * 64 + lane]; sk = A.sin_k[pos * 64 + lane]; pos_cs = pos; }
            const float ka = k0 * ck + k1 * sk, kb = k0 * (-sk) + k1 * ck;
            advance(pos, hk);
            return q0 * ka + q1 * kb;
        };
        int c = 0;
        for (; c + 1 < A.bpt; c += 2) {
            const float pa = partial(c), pb = partial(c + 1);
            const float sa = wave_sum(pa) * inv_sqrt, sb = wave_sum(pb) * inv_sqrt;
            if (lane == c) sc = sa;
            if (lane == c + 1) sc = sb;
        }
        if (c < A.bpt) {
            const float s = wave_sum(partial(c)) * inv_sqrt;
            if (lane == c) sc = s;
        }
    }
    const float mx = wave_max(sc);
    float p = lane < A.bpt ? expf(sc - mx) : 0.f;
    p /= wave_sum(p);
    // pass 2: dp_c = dy . v_c; dV_l rows
    float dp = 0.f;
    {
        int64_t pos = pos0; int hk = hk0;
        auto partial = [&](int c) {
            const float *vp = A.vpre + row_at(c) * HD + hk * kHd;
            const float v0 = lam * vp[lane], v1 = lam * vp[64 + lane];
            advance(pos, hk);
            return dy0 * v0 + dy1 * v1;
        };
        int c = 0;
        for (; c + 1 < A.bpt; c += 2) {
            const float pa = partial(c), pb = partial(c + 1);
            const float da = wave_sum(pa), db = wave_sum(pb);
            if (lane == c) dp = da;
            if (lane == c + 1) dp = db;
        }
        if (c < A.bpt) {
            const float d = wave_sum(partial(c));
            if (lane == c) dp = d;
        }
    }
    const float dot = wave_sum(p * dp);
    const float ds = p * (dp - dot) * inv_sqrt;   // lane c
    // what the per-table-row sums need of this (token, head): p_c and ds_c of its keys, and its rotated query
    if (lane < A.bpt) {
        A.pw[w * A.bpt + lane] = p;
        A.dsw[w * A.bpt + lane] = ds;
    }
    A.qrot[t * HD + h * kHd + lane] = q0;
    A.qrot[t * HD + h * kHd + 64 + lane] = q1;
    // pass 3: dq_r += ds_c k_r;  dk_n = rope^T(ds_c q_r)
    float dq0 = 0.f, dq1 = 0.f;
    {
        int64_t pos = pos0, pos_cs = -1; int hk = hk0;
        float ck = 0.f, sk = 0.f;
        auto rotated = [&](int c, float &ka, float &kb) {
            const float *kp = A.kn + row_at(c) * HD + hk * kHd;
            const float k0 = kp[lane], k1 = kp[64 + lane];
            if (pos != pos_cs) { ck = A.cos_k[pos * 64 + lane]; sk = A.sin_k[pos * 64 + lane]; pos_cs = pos; }
            ka = k0 * ck + k1 * sk; kb = k0 * (-sk) + k1 * ck;
            advance(pos, hk);
            return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ds), c));
        };
        int c = 0;
        for (; c + 1 < A.bpt; c += 2) {
            float a0, a1, b0, b1;
            const float da = rotated(c, a0, a1), db = rotated(c + 1, b0, b1);
            dq0 += da * a0; dq1 += da * a1;
            dq0 += db * b0; dq1 += db * b1;
        }
        if (c < A.bpt) {
            float a0, a1;
            const float da = rotated(c, a0, a1);
            dq0 += da * a0; dq1 += da * a1;
        }
    }
    // q: rope^T, head-norm^T
    const float dn0 = dq0 * cq - dq1 * sq, dn1 = dq0 * sq + dq1 * cq;
    const float m = wave_sum(dn0 * qn0 + dn1 * qn1) / (float)kHd;
    if (A.dq16) {
        A.dq16[t * HD + h * kHd + lane] = (__bf16)(rq * (dn0 - qn0 * m));
        A.dq16[t * HD + h * kHd + 64 + lane] = (__bf16)(rq * (dn1 - qn1 * m));
        return;
    }
    float *o = A.dq + t * HD + h * kHd;
    o[lane] = rq * (dn0 - qn0 * m);
    o[64 + lane] = rq * (dn1 - qn1 * m);
}

// The same for bpt <= 16 in the forward's lane-per-(key, quarter) form (cross_attn_quad_kernel): the keys are loaded and rotated
// once and stay in registers for the query gradient, whose sum over the keys runs over the lanes of a 16-lane row by two DPP
// rotates and over the four rows through 2 KB of LDS.
template <bool KV16>   // KV16: norm(k) and lambda v from their bf16 copies (kn16, vl16), in the forward's bf16-row lane layout
__global__ __launch_bounds__(kThreads) void cross_attn_bwd_quad_kernel(const AttnBwdArgs A) {
#pragma clang fp contract(fast)
    __shared__ __attribute__((aligned(16))) float qs[kWaves][kHd], dys[kWaves][kHd], red[kWaves][4][kHd];
    __shared__ __attribute__((aligned(16))) float cs[kWaves][2][kQuadPos * kQuadPosStride];
    const int lane = threadIdx.x & 63, wv = uni(threadIdx.x >> 6);
    const int64_t w = (int64_t)blockIdx.x * kWaves + wv;
    if (w >= A.T * A.H) return;
    int64_t t; int h;   // (head-major for the viewed layout, as in the forward)
    if (A.layout == 0) { h = uni((int)(w / A.T)); t = w - (int64_t)h * A.T; }
    else { t = uni64(w / A.H); h = uni((int)(w - t * A.H)); }
    const int HD = A.H * kHd;
    const int64_t th = t * A.H + h, o128 = t * HD + h * kHd;
    const float lam = *A.lambda;
    const float inv_sqrt = 1.0f / sqrtf((float)kHd);
    // q: head norm + RoPE, a lane per pair of dims; kept for the way back
    const float qp0 = A.q_pre[o128 + lane], qp1 = A.q_pre[o128 + 64 + lane];
    const float rq = rms_scale(wave_sum(qp0 * qp0 + qp1 * qp1), kHd, A.eps);
    const float qn0 = qp0 * rq, qn1 = qp1 * rq;
    const float cq = A.cos_q[t * 64 + lane], sq = A.sin_q[t * 64 + lane];
    {
        const float q0 = qn0 * cq + qn1 * sq, q1 = qn0 * (-sq) + qn1 * cq;
        qs[wv][lane] = q0; qs[wv][64 + lane] = q1;
        A.qrot[o128 + lane] = q0; A.qrot[o128 + 64 + lane] = q1;   // (the per-table-row sums need it)
        dys[wv][lane] = A.dy[o128 + lane]; dys[wv][64 + lane] = A.dy[o128 + 64 + lane];
    }
    const int c = lane >> 2, j = lane & 3;
    const bool live = c < A.bpt;
    const int ce = live ? c : A.bpt - 1;
    int64_t pos = t * A.bpt + ce;
    int hk = h, dpos = 0;
    bool staged = false;
    if (A.layout == 0) {
        const int64_t r0 = ((int64_t)h * A.T + t) * A.bpt;
        const int64_t pos0 = uni64(r0 / A.H);
        const int hk0 = uni((int)(r0 - pos0 * A.H)), hc = hk0 + ce;
        dpos = hc / A.H;
        pos = pos0 + dpos;
        hk = hc - dpos * A.H;
        const int np = (hk0 + A.bpt - 1) / A.H + 1;
        staged = np <= kQuadPos;
        if (staged) {
            for (int n = 0; n < np; ++n) {
                cs[wv][0][n * kQuadPosStride + lane] = A.cos_k[(pos0 + n) * 64 + lane];
                cs[wv][1][n * kQuadPosStride + lane] = A.sin_k[(pos0 + n) * 64 + lane];
            }
        }
    }
    int64_t row = pos;   // two id tensors: the key / value rows are per kv position
    if (A.ids) {
        row = A.ids[pos];
        if ((uint64_t)row >= (uint64_t)A.rows) row = 0;   // flagged by the forward
    }
    const int off4 = (int)((row * HD + hk * kHd) >> 2);
    float4 k[8], v[8];
    float krl[16], krh[16];   // KV16: the rotated key, dims d(n) and d(n) + 64
    float s = 0.f, dpv = 0.f;
    if constexpr (KV16) {
        float vlo[16], vhi[16], ck[16], sk[16];
        load_row32<true>(A.kn16, off4, j, krl, krh);
        load_row32<true>(A.vl16, off4, j, vlo, vhi);
        __builtin_amdgcn_wave_barrier();
        if (staged) {
            load16<true>(&cs[wv][0][dpos * kQuadPosStride], j, ck);
            load16<true>(&cs[wv][1][dpos * kQuadPosStride], j, sk);
        } else {
            load16<true>(A.cos_k + pos * 64, j, ck);
            load16<true>(A.sin_k + pos * 64, j, sk);
        }
        {
            float qa[16], qb[16];
            load16<true>(&qs[wv][0], j, qa);
            load16<true>(&qs[wv][64], j, qb);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const float x = krl[n], y = krh[n];
                krl[n] = x * ck[n] + y * sk[n];
                krh[n] = y * ck[n] - x * sk[n];
                s += qa[n] * krl[n] + qb[n] * krh[n];
            }
        }
        {
            float da[16], db[16];
            load16<true>(&dys[wv][0], j, da);
            load16<true>(&dys[wv][64], j, db);
#pragma unroll
            for (int n = 0; n < 16; ++n) dpv += da[n] * vlo[n] + db[n] * vhi[n];
        }
    } else {
    const float4 *kp = (const float4 *)A.kn + off4 + j, *vp = (const float4 *)A.vpre + off4 + j;
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = kp[4 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = vp[4 * i];
    __builtin_amdgcn_wave_barrier();
    const float *cl = &cs[wv][0][dpos * kQuadPosStride + 4 * j], *sl = &cs[wv][1][dpos * kQuadPosStride + 4 * j];
    const float4 *cg = (const float4 *)(A.cos_k + pos * 64) + j, *sg = (const float4 *)(A.sin_k + pos * 64) + j;
    // scores (the keys rotated in place) and dp_c = dy . v_c
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 qa = *(const float4 *)&qs[wv][16 * i + 4 * j], qb = *(const float4 *)&qs[wv][64 + 16 * i + 4 * j];
        const float4 da = *(const float4 *)&dys[wv][16 * i + 4 * j], db = *(const float4 *)&dys[wv][64 + 16 * i + 4 * j];
        const float4 a = k[i], b = k[i + 4];
        const float4 cc = staged ? *(const float4 *)(cl + 16 * i) : cg[4 * i], ss = staged ? *(const float4 *)(sl + 16 * i) : sg[4 * i];
        k[i].x = a.x * cc.x + b.x * ss.x; k[i + 4].x = b.x * cc.x - a.x * ss.x;
        k[i].y = a.y * cc.y + b.y * ss.y; k[i + 4].y = b.y * cc.y - a.y * ss.y;
        k[i].z = a.z * cc.z + b.z * ss.z; k[i + 4].z = b.z * cc.z - a.z * ss.z;
        k[i].w = a.w * cc.w + b.w * ss.w; k[i + 4].w = b.w * cc.w - a.w * ss.w;
        s += qa.x * k[i].x + qb.x * k[i + 4].x;
        s += qa.y * k[i].y + qb.y * k[i + 4].y;
        s += qa.z * k[i].z + qb.z * k[i + 4].z;
        s += qa.w * k[i].w + qb.w * k[i + 4].w;
        dpv += da.x * v[i].x + db.x * v[i + 4].x;
        dpv += da.y * v[i].y + db.y * v[i + 4].y;
        dpv += da.z * v[i].z + db.z * v[i + 4].z;
        dpv += da.w * v[i].w + db.w * v[i + 4].w;

    }
    }
    const float sc = live ? quad_sum(s) * inv_sqrt : -FLT_MAX;
    const float mx = wave_max(sc);
    float p = live ? expf(sc - mx) : 0.f;
    p /= wave_sum(j == 0 ? p : 0.f);
    const float dp = (KV16 ? 1.f : lam) * quad_sum(dpv);   // (vl16 is lambda v already)
    const float dot = wave_sum(j == 0 ? p * dp : 0.f);
    const float ds = p * (dp - dot) * inv_sqrt;
    if (live && j == 0) {
        A.pw[th * A.bpt + c] = p;
        A.dsw[th * A.bpt + c] = ds;
    }
    // dq_r = sum_c ds_c k_r: over the four keys of a 16-lane row by DPP, over the four rows through LDS
    if constexpr (KV16) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            krl[n] *= ds; krh[n] *= ds;
            krl[n] += dpp_move<kDppRowRor4, 0xf>(0.f, krl[n]); krh[n] += dpp_move<kDppRowRor4, 0xf>(0.f, krh[n]);
            krl[n] += dpp_move<kDppRowRor8, 0xf>(0.f, krl[n]); krh[n] += dpp_move<kDppRowRor8, 0xf>(0.f, krh[n]);
        }
        if ((lane & 12) == 0) {
            store16<true>(&red[wv][lane >> 4][0], j, krl);
            store16<true>(&red[wv][lane >> 4][64], j, krh);
        }
    } else
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float4 x = make_float4(ds * k[i].x, ds * k[i].y, ds * k[i].z, ds * k[i].w);
        x.x += dpp_move<kDppRowRor4, 0xf>(0.f, x.x); x.y += dpp_move<kDppRowRor4, 0xf>(0.f, x.y);
        x.z += dpp_move<kDppRowRor4, 0xf>(0.f, x.z); x.w += dpp_move<kDppRowRor4, 0xf>(0.f, x.w);
        x.x += dpp_move<kDppRowRor8, 0xf>(0.f, x.x); x.y += dpp_move<kDppRowRor8, 0xf>(0.f, x.y);
        x.z += dpp_move<kDppRowRor8, 0xf>(0.f, x.z); x.w += dpp_move<kDppRowRor8, 0xf>(0.f, x.w);
        if ((lane & 12) == 0) *(float4 *)&red[wv][lane >> 4][16 * i + 4 * j] = x;
    }
    __builtin_amdgcn_wave_barrier();
    const float dq0 = (red[wv][0][lane] + red[wv][1][lane]) + (red[wv][2][lane] + red[wv][3][lane]);
    const float dq1 = (red[wv][0][64 + lane] + red[wv][1][64 + lane]) + (red[wv][2][64 + lane] + red[wv][3][64 + lane]);
    // q: rope^T, head-norm^T
    const float dn0 = dq0 * cq - dq1 * sq, dn1 = dq0 * sq + dq1 * cq;
    const float m = wave_sum(dn0 * qn0 + dn1 * qn1) / (float)kHd;
    if (A.dq16) {
        A.dq16[o128 + lane] = (__bf16)(rq * (dn0 - qn0 * m));
        A.dq16[o128 + 64 + lane] = (__bf16)(rq * (dn1 - qn1 * m));
        return;
    }
    A.dq[o128 + lane] = rq * (dn0 - qn0 * m);
    A.dq[o128 + 64 + lane] = rq * (dn1 - qn1 * m);
}

// Gradients of the per-row key / value tables: dkn_tab[r] = sum over the kv positions with byte id r of rope^T(ds * q_rot),
// dvl_tab[r] = sum of p * dy -- an embedding backward whose "gradient rows" are never written: positions come grouped by byte
// id (launch_group_positions), a wave walks a stretch of that order for ONE head slice, rebuilds each position's slice from
// the query that owns it (p, ds: one scalar each; dy and q_rot: 512 bytes each, all of it L2 / Infinity Cache resident) and
// keeps the running sums of the current id in registers.  (The first version wrote the two [T*bpt, HD] position arrays,
// 6.4 GB, and read them back in two embedding-backward calls.)
struct AttnRowsArgs {
    const float *pw, *dsw, *dy, *qrot, *cos_k, *sin_k;
    const int32_t *pos_sorted, *id_sorted;
    int64_t P, T;
    int bpt, H, layout;
    float *dkn_tab, *dvl_tab;
};
constexpr int kRowsSeg = 256;   // sorted positions per wave
__global__ __launch_bounds__(kThreads) void attn_kv_rows_bwd_kernel(const AttnRowsArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * kWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hk = blockIdx.y;
    const int HD = A.H * kHd;
    const int64_t s_begin = w * kRowsSeg, s_end = min(A.P, s_begin + kRowsSeg);
    float dv0 = 0.f, dv1 = 0.f, dk0 = 0.f, dk1 = 0.f;
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;
        float *ok = A.dkn_tab + (int64_t)cur * HD + hk * kHd, *ov = A.dvl_tab + (int64_t)cur * HD + hk * kHd;
        atomicAdd(ok + lane, dk0); atomicAdd(ok + 64 + lane, dk1);
        atomicAdd(ov + lane, dv0); atomicAdd(ov + 64 + lane, dv1);
    };
    for (int64_t s0 = s_begin; s0 < s_end; s0 += 64) {
        const int cnt = (int)min((int64_t)64, s_end - s0);
        int vpos = 0, vid = 0;
        if (lane < cnt) { vpos = A.pos_sorted[s0 + lane]; vid = A.id_sorted[s0 + lane]; }
        for (int k = 0; k < cnt; ++k) {
            const int64_t pos = __builtin_amdgcn_readlane(vpos, k);
            const int id = __builtin_amdgcn_readlane(vid, k);
            if (id != cur) { flush(); cur = id; dv0 = dv1 = dk0 = dk1 = 0.f; }
            // the query (tq, hq) that owns key slot (pos, hk), and which of its keys this is
            int64_t tq; int hq, c;
            if (A.layout == 0) {
                const int64_t r = pos * A.H + hk, qi = r / A.bpt;
                c = (int)(r - qi * A.bpt);
                hq = (int)(qi / A.T);
                tq = qi - (int64_t)hq * A.T;
            } else {
                tq = pos / A.bpt;
                c = (int)(pos - tq * A.bpt);
                hq = hk;
            }
            const int64_t sidx = (tq * A.H + hq) * A.bpt + c, row = tq * HD + hq * kHd;
            const float p = A.pw[sidx], ds = A.dsw[sidx];
            const float ck = A.cos_k[pos * 64 + lane], sk = A.sin_k[pos * 64 + lane];
            dv0 += p * A.dy[row + lane];
            dv1 += p * A.dy[row + 64 + lane];
            const float g0 = ds * A.qrot[row + lane], g1 = ds * A.qrot[row + 64 + lane];   // d k_r
            dk0 += g0 * ck - g1 * sk;                                                        // rope^T
            dk1 += g0 * sk + g1 * ck;
        }
    }
    flush();
}

// The same sums with the head count as a template parameter: a wave then forms all HH head slices of a kv position in one
// visit -- the rotary row of the position is read once instead of once per head, and the owner (tq, hq, c) of slot
// (pos, hk + 1) follows from that of (pos, hk) without a division.
template <int HH>
__global__ __launch_bounds__(kThreads) void attn_kv_rows_bwd_h_kernel(const AttnRowsArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * kWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int HD = HH * kHd;
    const int64_t s_begin = w * kRowsSeg, s_end = min(A.P, s_begin + kRowsSeg);
    float dv0[HH], dv1[HH], dk0[HH], dk1[HH];
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;
#pragma unroll
        for (int hk = 0; hk < HH; ++hk) {
            float *ok = A.dkn_tab + (int64_t)cur * HD + hk * kHd, *ov = A.dvl_tab + (int64_t)cur * HD + hk * kHd;
            atomicAdd(ok + lane, dk0[hk]); atomicAdd(ok + 64 + lane, dk1[hk]);
            atomicAdd(ov + lane, dv0[hk]); atomicAdd(ov + 64 + lane, dv1[hk]);
        }
    };
    for (int64_t s0 = s_begin; s0 < s_end; s0 += 64) {
        const int cnt = (int)min((int64_t)64, s_end - s0);
        int vpos = 0, vid = 0;
        if (lane < cnt) { vpos = A.pos_sorted[s0 + lane]; vid = A.id_sorted[s0 + lane]; }
        // owner (query token, head, key index) of slot (pos, 0), a lane per position: the two 64-bit divisions are vector work once
        // per 64 positions (on the scalar unit, once per position, they made this kernel scalar-bound: 230 scalar instructions per
        // position, the CU's one scalar unit 80 % busy)
        int vtq, vhq = 0, vc;
        if (A.layout == 0) {
            const int64_t r = (int64_t)vpos * HH, qi = r / A.bpt;
            vc = (int)(r - qi * A.bpt);
            vhq = (int)(qi / A.T);
            vtq = (int)(qi - (int64_t)vhq * A.T);
        } else {
            vtq = vpos / A.bpt;
            vc = vpos - vtq * A.bpt;
        }
        for (int k = 0; k < cnt; ++k) {
            const int64_t pos = __builtin_amdgcn_readlane(vpos, k);
            const int id = __builtin_amdgcn_readlane(vid, k);
            if (id != cur) {
                flush();
                cur = id;
#pragma unroll
                for (int hk = 0; hk < HH; ++hk) dv0[hk] = dv1[hk] = dk0[hk] = dk1[hk] = 0.f;
            }
            const float ck = A.cos_k[pos * 64 + lane], sk = A.sin_k[pos * 64 + lane];
            // the following slots advance the key index c, then the query index
            int64_t tq = __builtin_amdgcn_readlane(vtq, k);
            int c = __builtin_amdgcn_readlane(vc, k), hq = __builtin_amdgcn_readlane(vhq, k);
            if (A.layout == 0 && A.bpt >= HH) {
                // the HH slots of a position are keys c .. c + HH - 1 of ONE query, or run on into the next query's first keys: its
                // dy and rotated-query slices are loaded once (twice), not once per slot (26 wave-loads per position -> 6 or 10)
                const int64_t rowa = tq * HD + hq * kHd, sa = (tq * HH + hq) * A.bpt + c;
                const float ya0 = A.dy[rowa + lane], ya1 = A.dy[rowa + 64 + lane], xa0 = A.qrot[rowa + lane], xa1 = A.qrot[rowa + 64 + lane];
                float yb0 = 0.f, yb1 = 0.f, xb0 = 0.f, xb1 = 0.f;
                int64_t sb = 0;
                const int na = min(HH, A.bpt - c);   // slots of the first query
                if (na < HH) {   // the next query of the (H, T) grid
                    int64_t tb = tq + 1; int hb = hq;
                    if (tb == A.T) { tb = 0; ++hb; }
                    const int64_t rowb = tb * HD + hb * kHd;
                    sb = (tb * HH + hb) * A.bpt - na;
                    yb0 = A.dy[rowb + lane]; yb1 = A.dy[rowb + 64 + lane]; xb0 = A.qrot[rowb + lane]; xb1 = A.qrot[rowb + 64 + lane];
                }
#pragma unroll
                for (int hk = 0; hk < HH; ++hk) {
                    const bool first = hk < na;
                    const int64_t sidx = (first ? sa : sb) + hk;
                    const float p = A.pw[sidx], ds = A.dsw[sidx];
                    const float y0 = first ? ya0 : yb0, y1 = first ? ya1 : yb1, x0 = first ? xa0 : xb0, x1 = first ? xa1 : xb1;
                    dv0[hk] += p * y0;
                    dv1[hk] += p * y1;
                    const float g0 = ds * x0, g1 = ds * x1;
                    dk0[hk] += g0 * ck - g1 * sk;
                    dk1[hk] += g0 * sk + g1 * ck;
                }
                continue;
            }
#pragma unroll
            for (int hk = 0; hk < HH; ++hk) {
                if (A.layout != 0) hq = hk;
                const int64_t sidx = (tq * HH + hq) * A.bpt + c, row = tq * HD + hq * kHd;
                const float p = A.pw[sidx], ds = A.dsw[sidx];
                dv0[hk] += p * A.dy[row + lane];
                dv1[hk] += p * A.dy[row + 64 + lane];
                const float g0 = ds * A.qrot[row + lane], g1 = ds * A.qrot[row + 64 + lane];
                dk0[hk] += g0 * ck - g1 * sk;
                dk1[hk] += g0 * sk + g1 * ck;
                if (A.layout == 0 && ++c == A.bpt) {   // next flat row belongs to the next query of the (H, T) grid
                    c = 0;
                    if (++tq == A.T) { tq = 0; ++hq; }
                }
            }
        }
    }
    flush();
}

// k_n = norm_head(k_pre), v_l = lambda v_pre    (the forward's kv_finish, out of place)
__global__ __launch_bounds__(kThreads) void kv_norm_kernel(const float *__restrict__ kpre, const float *__restrict__ vpre, int64_t rows, int H,
                                                           const float *__restrict__ lambda, float eps, float *__restrict__ kn, float *__restrict__ vl) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (w >= rows * H) return;
    const float k0 = kpre[w * kHd + lane], k1 = kpre[w * kHd + 64 + lane];
    const float r = rms_scale(wave_sum(k0 * k0 + k1 * k1), kHd, eps);
    kn[w * kHd + lane] = k0 * r;
    kn[w * kHd + 64 + lane] = k1 * r;
    const float lam = *lambda;
    vl[w * kHd + lane] = lam * vpre[w * kHd + lane];
    vl[w * kHd + 64 + lane] = lam * vpre[w * kHd + 64 + lane];
}

// per (table row, head): d lambda += dV_l . v_pre;  dkv[row][HD + ..] = lambda dV_l;  dkv[row][..] = norm_head^T(dk_n)
__global__ __launch_bounds__(kThreads) void kv_table_bwd_kernel(const float *__restrict__ dkn, const float *__restrict__ dvl, const float *__restrict__ kpre,
                                                                const float *__restrict__ vpre, int64_t rows, int H, const float *__restrict__ lambda,
                                                                float eps, float *__restrict__ dkv, float *__restrict__ d_lambda) {
    // a wave walks (row, head) pairs with the grid's stride and adds its share of d lambda ONCE: with two id tensors there is a pair
    // per kv position and head (6 M at 65 536 tokens x 16), and one atomic each on the same word took 80 ms
    const int lane = threadIdx.x & 63;
    const int HD = H * kHd;
    const float lam = *lambda;
    float dl = 0.f;
    for (int64_t w = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); w < rows * H; w += (int64_t)gridDim.x * kWaves) {
        const int64_t row = w / H;
        const int h = (int)(w - row * H);
        const float dv0 = dvl[w * kHd + lane], dv1 = dvl[w * kHd + 64 + lane];
        dl += dv0 * vpre[w * kHd + lane] + dv1 * vpre[w * kHd + 64 + lane];
        const float k0 = kpre[w * kHd + lane], k1 = kpre[w * kHd + 64 + lane];
        const float r = rms_scale(wave_sum(k0 * k0 + k1 * k1), kHd, eps);
        const float kn0 = k0 * r, kn1 = k1 * r, g0 = dkn[w * kHd + lane], g1 = dkn[w * kHd + 64 + lane];
        const float m = wave_sum(g0 * kn0 + g1 * kn1) / (float)kHd;
        float *o = dkv + row * 2 * HD + h * kHd;
        o[lane] = r * (g0 - kn0 * m);
        o[64 + lane] = r * (g1 - kn1 * m);
        o[HD + lane] = lam * dv0;
        o[HD + 64 + lane] = lam * dv1;
    }
    dl = wave_sum(dl);
    if (lane == 0 && d_lambda) atomicAdd(d_lambda, dl);
}

// two id tensors: dxkv[p] <- norm^T(dxkv[p]) with x = E[a_p] + E[b_p], in place (the gradient of the un-normalised sum, which both
// table rows receive), one wave per kv position
__global__ __launch_bounds__(kThreads) void dual_rows_norm_bwd_kernel(const float *__restrict__ table, const int32_t *__restrict__ ids_a,
                                                                      const int32_t *__restrict__ ids_b, int64_t n, int D, float eps, float *__restrict__ dxn) {
    const int lane = threadIdx.x & 63;
    for (int64_t p = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); p < n; p += (int64_t)gridDim.x * kWaves) {
        const float *ra = table + (int64_t)ids_a[p] * D, *rb = table + (int64_t)ids_b[p] * D;
        float ss = 0.f, dot = 0.f;
        for (int j = lane; j < D; j += 64) { const float v = ra[j] + rb[j]; ss += v * v; dot += dxn[p * D + j] * v; }
        const float rs = rms_scale(wave_sum(ss), D, eps);
        const float m = wave_sum(dot) * rs / (float)D;      // mean(dxn * xn), xn = x * rs
        for (int j = lane; j < D; j += 64) dxn[p * D + j] = rs * (dxn[p * D + j] - (ra[j] + rb[j]) * rs * m);
    }
}

// xn[row] = norm?(table[row]) for every table row (the operand the K/V GEMMs saw), one wave per row
__global__ __launch_bounds__(kThreads) void rows_norm_kernel(const float *__restrict__ table, int64_t rows, int D, int norm, float eps, float *__restrict__ xn) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= rows) return;
    float ss = 0.f;
    for (int j = lane; j < D; j += 64) { const float v = table[r * D + j]; ss += v * v; }
    const float rs = norm ? rms_scale(wave_sum(ss), D, eps) : 1.0f;
    for (int j = lane; j < D; j += 64) xn[r * D + j] = table[r * D + j] * rs;
}

// d_byte[row] += norm^T(dxn[row]): the K/V table rows ARE the byte-table rows, so no scatter
__global__ __launch_bounds__(kThreads) void byte_rows_bwd_kernel(const float *__restrict__ table, const float *__restrict__ dxn, int64_t rows, int D, int norm,
                                                                 float eps, float *__restrict__ d_table) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= rows) return;
    if (!norm) {
        for (int j = lane; j < D; j += 64) atomicAdd(d_table + r * D + j, dxn[r * D + j]);
        return;
    }
    float ss = 0.f, dot = 0.f;
    for (int j = lane; j < D; j += 64) { const float v = table[r * D + j]; ss += v * v; dot += dxn[r * D + j] * v; }
    const float rs = rms_scale(wave_sum(ss), D, eps);
    const float m = wave_sum(dot) * rs / (float)D;      // mean(dxn * xn), xn = x * rs
    for (int j = lane; j < D; j += 64) atomicAdd(d_table + r * D + j, rs * (dxn[r * D + j] - table[r * D + j] * rs * m));
}

__global__ __launch_bounds__(kThreads) void ids_to_i32_kernel(const int64_t *__restrict__ ids, int64_t n, int64_t rows, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const int64_t v = ids[i];
        out[i] = (uint64_t)v < (uint64_t)rows ? (int32_t)v : 0;
    }
}


__global__ __launch_bounds__(kThreads) void iota_i32_kernel(int32_t *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) out[i] = (int32_t)i;
}

// workspace of the backward, in floats
struct AttnBwdLayout {
    size_t q, y, kpre, vpre, kn, vl, dy, dq, pw, dsw, qrot, grp, dkn_tab, dvl_tab, dkv, xkv, dxkv, xq, dxq, ids32, ids32b, emb, emb_bytes, b0, b1, w16, x16, dq16, part, part_n, kn16, vl16, wt, total;
};
static void noop_bwd_desc(MotEmbedMixDesc &e, const void *tokens, int64_t n, const void *table, int64_t rows, int dim, int norm, float eps, uint32_t *status) {
    memset(&e, 0, sizeof(e));
    e.struct_size = sizeof(e); e.dtype = MOT_F32; e.n_rows = 1; e.tokens_per_row = n; e.mode = MOT_MIX_NOOP;
    e.tokens = (const int32_t *)tokens; e.tok_table = table; e.tok_rows = rows; e.tok_dim = dim; e.model_dim = dim;
    e.norm_tok = norm; e.eps = eps; e.status = status;
}
static AttnBwdLayout attn_bwd_layout(const MotCrossAttnDesc &d) {
    AttnBwdLayout L;
    const size_t T = (size_t)d.n_tokens, HD = (size_t)d.n_heads * kHd, D = (size_t)d.dim, P = T * d.bpt;
    const bool dual = d.ids_b != nullptr;
    const size_t R = dual ? P : (size_t)d.byte_rows;   // key / value rows: per byte-table row, or per kv position
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    L.q = take(T * HD); L.y = take(T * HD); L.kpre = take(R * HD); L.vpre = take(R * HD); L.kn = take(R * HD); L.vl = take(R * HD);
    L.dy = take(T * HD); L.dq = take(T * HD); L.pw = take(P * d.n_heads); L.dsw = take(P * d.n_heads); L.qrot = take(T * HD);
    // byte ids of the kv positions, grouped -- or, per position, the identity order (2 P ints)
    L.grp = take(dual ? 2 * P : group_positions_ws_ints((int64_t)P, (int64_t)d.byte_rows));
    L.dkn_tab = take(R * HD); L.dvl_tab = take(R * HD); L.dkv = take(R * 2 * HD); L.xkv = take(R * D); L.dxkv = take(R * D);
    L.xq = take(T * D); L.dxq = take(T * D); L.ids32 = take(P); L.ids32b = take(dual ? P : 0);
    MotEmbedMixDesc e;   // scratch of the token-table embedding backward and of the two byte-table backwards of the two-id embedding
    noop_bwd_desc(e, nullptr, (int64_t)T, nullptr, d.tok_rows, (int)D, d.norm_tok, 0.f, nullptr);
    size_t a = embed_mix_bwd_workspace_bytes(e), b = 0;
    if (dual) {
        noop_bwd_desc(e, nullptr, (int64_t)P, nullptr, d.byte_rows, (int)D, 0, 0.f, nullptr);
        b = embed_mix_bwd_workspace_bytes(e);
    }
    L.emb_bytes = a > b ? a : b;
    L.emb = take((L.emb_bytes + 3) / 4);
    const size_t wide = D > HD ? D : HD;   // matmul_dtype == MOT_BF16: two bf16 row operands and one (transposed) weight at a time
    // (two id tensors: + the key / value source rows of every kv position; b1 also takes dkv, w16 also kv_w)
    const size_t b1n = dual && P * 2 * HD > T * wide ? P * 2 * HD : T * wide;
    L.b0 = take(mm16(d) ? (T * wide + 1) / 2 : 0); L.b1 = take(mm16(d) ? (b1n + 1) / 2 : 0); L.w16 = take(mm16(d) ? ((dual ? 2 : 1) * HD * D + 1) / 2 : 0);
    L.x16 = take(mm16(d) && dual ? (P * D + 1) / 2 : 0);
    L.dq16 = take(mm16(d) ? (T * HD + 1) / 2 : 0);   // dq in bf16, written by the attention backward (b1 is reused for dkv in between)
    // partial blocks of the few-row products (key / value projections; dxkv = dkv W_kv), see launch_gemm_rows_sliced
    const size_t pa = gemm_rows_sliced_floats((int64_t)R, (int)D, (int)HD), pb = gemm_rows_sliced_floats((int64_t)R, (int)(2 * HD), (int)D);
    L.part_n = pa > pb ? pa : pb;
    L.part = take(L.part_n);
    const size_t t16 = mm16(d) && !dual ? (R * HD + 1) / 2 : 0;   // norm(k), lambda v in bf16 (as in the forward)
    L.kn16 = take(t16); L.vl16 = take(t16);
    L.wt = take(mm16(d) ? 0 : 2 * HD * D);   // a transposed fp32 weight (c_proj, q_w, kv_w) for the k-major products over many rows
    L.total = o;
    return L;
}

size_t cross_attn_bwd_workspace_bytes(const MotCrossAttnDesc &d) { return attn_bwd_layout(d).total * 4; }

// out[n][Nout] = rows[n][Kc] . W[Kc][Nout]  (W with the reduction index as its row index): the plain dense MFMA kernel
// (wt: scratch of Kc * Nout floats -- over many rows the weight is transposed into it and the product runs on the LDS-DMA kernel, whose
//  B operand is row-major along the reduction: 625 against 730 us at 65 536 x 768 x 768)
static int dense_gemm_kmajor(const float *rows, int64_t n, int Kc, const float *W, int Nout, float *out, hipStream_t stream, float *part = nullptr,
                             size_t part_n = 0, float *wt = nullptr) {
    if (wt && gemm_rows_f32_256_usable(rows, Kc, n, wt, Kc, Kc, Nout)) {   // (512 rows and more)
        if (int rc = launch_transpose_f32(W, Kc, Nout, wt, stream)) return rc;
        return launch_gemm_rows_f32_256(rows, Kc, n, wt, Kc, Kc, Nout, out, Nout, nullptr, false, stream);
    }
    return launch_gemm_rows_sliced(rows, Kc, n, W, Nout, Kc, Nout, out, Nout, false, part, part_n, stream);
}

int launch_cross_attn_bwd(const MotCrossAttnDesc &d, const MotCrossAttnGrads &gr, hipStream_t stream) {
    const int64_t T = d.n_tokens, P = T * d.bpt;
    const bool dual = d.ids_b != nullptr;          // kv embedding norm?(E[a] + E[b]): key / value rows per kv position, not per table row
    const int64_t R = dual ? P : d.byte_rows;
    const int H = d.n_heads, HD = H * kHd, D = d.dim;
    const AttnBwdLayout L = attn_bwd_layout(d);
    if (!d.workspace || d.workspace_bytes < L.total * 4)
        return set_error(MOT_EWORKSPACE, "cross_attn_bwd: needs %zu workspace bytes, got %zu", L.total * 4, d.workspace_bytes);
    if (dual && R * (int64_t)HD >= 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "cross_attn_bwd: two id tensors at %lld kv positions are not built", (long long)P);
    float *ws = (float *)d.workspace;
    float *q = ws + L.q, *y = ws + L.y, *kpre = ws + L.kpre, *vpre = ws + L.vpre, *kn = ws + L.kn, *vl = ws + L.vl, *dy = ws + L.dy, *dq = ws + L.dq;
    float *pw = ws + L.pw, *dsw = ws + L.dsw, *qrot = ws + L.qrot, *dkn_tab = ws + L.dkn_tab, *dvl_tab = ws + L.dvl_tab, *dkv = ws + L.dkv;
    float *xkv = ws + L.xkv, *dxkv = ws + L.dxkv, *xq = ws + L.xq, *dxq = ws + L.dxq;
    int32_t *ids32 = (int32_t *)(ws + L.ids32);
    void *emb_ws = ws + L.emb;
    const float eps = d.eps > 0.f ? d.eps : FLT_EPSILON;
    const float *g_out = (const float *)gr.grad_out;
    int rc;
    hipLaunchKernelGGL(ids_to_i32_kernel, dim3(1024), dim3(kThreads), 0, stream, d.ids_a, P, (int64_t)d.byte_rows, ids32);
    if ((rc = check_launch("ids_to_i32"))) return rc;
    if ((rc = launch_zero_words(dkn_tab, (int64_t)R * HD, stream))) return rc;
    if ((rc = launch_zero_words(dvl_tab, (int64_t)R * HD, stream))) return rc;
    // ---- forward recompute (the queries and the attention output come from the forward when it kept them)
    bool have_xq = false;   // the gathered (normalised) token rows are already in xq
    if (d.saved_qy) {
        q = (float *)d.saved_qy;
        y = q + (size_t)T * HD;
    } else {
        if ((rc = launch_gather_rows(d.tokens, nullptr, 4, T, d.tok_table, d.tok_rows, D, d.norm_tok, eps, nullptr, xq, d.status, MOT_F32, stream))) return rc;
        if (mm16(d)) rc = dense_rows_bf16(xq, T, D, (const float *)d.q_w, HD, q, ws + L.b0, ws + L.w16, stream);
        else rc = launch_gemm_rows(xq, D, T, (const float *)d.q_w, D, D, HD, q, HD, true, stream);
        if (rc) return rc;
        have_xq = true;
    }
    const float *kv_w = (const float *)d.kv_w;
    // the normalised byte-table rows -- or the normalised per-position sums E[a] + E[b] -- (also the operand of dW_kv below),
    // projected by the plain dense kernel
    if (dual) {
        if ((rc = launch_gather_rows(d.ids_a, d.ids_b, 8, P, d.byte_table, d.byte_rows, D, d.norm_byte, eps, nullptr, xkv, d.status, MOT_F32, stream))) return rc;
    } else {
        hipLaunchKernelGGL(rows_norm_kernel, dim3((unsigned)((R + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, (const float *)d.byte_table, R, D, d.norm_byte, eps, xkv);
        if ((rc = check_launch("rows_norm_kernel"))) return rc;
    }
    const bool kv16 = dual && mm16(d);   // the products over the kv positions on the bf16 MFMA too (forward: launch_cross_attn)
    __bf16 *x16 = (__bf16 *)(ws + L.x16);
    if (kv16) {
        __bf16 *w16k = (__bf16 *)(ws + L.w16);
        if ((rc = launch_narrow(xkv, R * D, x16, stream))) return rc;
        if ((rc = launch_narrow(kv_w, (int64_t)2 * HD * D, w16k, stream))) return rc;
        if ((rc = launch_gemm_rows_bf16(x16, D, R, w16k, D, D, HD, kpre, HD, false, nullptr, stream))) return rc;
        if ((rc = launch_gemm_rows_bf16(x16, D, R, w16k + (size_t)HD * D, D, D, HD, vpre, HD, false, nullptr, stream))) return rc;
    } else {
        if ((rc = launch_gemm_rows_sliced(xkv, D, R, kv_w, D, D, HD, kpre, HD, true, ws + L.part, L.part_n, stream))) return rc;
        if ((rc = launch_gemm_rows_sliced(xkv, D, R, kv_w + (size_t)HD * D, D, D, HD, vpre, HD, true, ws + L.part, L.part_n, stream))) return rc;
    }
    const int64_t kvw = R * H;
    hipLaunchKernelGGL(kv_norm_kernel, dim3((unsigned)((kvw + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, kpre, vpre, R, H, d.lambda_factor, eps, kn, vl);
    if ((rc = check_launch("kv_norm_kernel"))) return rc;
    AttnArgs A;
    A.q = q; A.y = y; A.kt = kn; A.vt = vl; A.ids = dual ? nullptr : d.ids_a; A.rows = R; A.T = T; A.bpt = d.bpt; A.H = H;
    A.layout = d.head_layout; A.cos_q = d.cos_q; A.sin_q = d.sin_q; A.cos_k = d.cos_k; A.sin_k = d.sin_k; A.eps = eps; A.status = d.status;
    A.y16 = nullptr;
    A.kt16 = A.vt16 = nullptr;
    if (mm16(d) && !dual && d.bpt <= 16) {   // norm(k) and lambda v in bf16, as the forward's attention read them
        if ((rc = launch_narrow(kn, R * HD, ws + L.kn16, stream))) return rc;
        if ((rc = launch_narrow(vl, R * HD, ws + L.vl16, stream))) return rc;
        A.kt16 = (const __bf16 *)(ws + L.kn16); A.vt16 = (const __bf16 *)(ws + L.vl16);
    }
    const int64_t waves = T * H;
    if (!d.saved_qy && (rc = launch_attention(A, stream))) return rc;
    // ---- c_proj:  dW_p += g^T y;  dy = g W_p   (proj_w [D, HD] is the k-major operand of g[T, D] -> dy[T, HD])
    // (matmul_dtype == MOT_BF16: the four products over the tokens -- dW_p, dy, dW_q, dxq -- on the bf16 MFMA, row operands rounded
    //  to bf16 as the reference's bf16 autograd has them, fp32 sums; b0 = g, then xq; b1 = y, then dq; w16 = the k-major weight)
    __bf16 *b0 = (__bf16 *)(ws + L.b0), *b1 = (__bf16 *)(ws + L.b1), *w16 = (__bf16 *)(ws + L.w16);
    if (mm16(d)) {
        const __bf16 *g16 = b0;
        if (d.io_dtype == MOT_BF16) g16 = (const __bf16 *)gr.grad_out;   // the caller's bf16 gradient as it is
        else if ((rc = launch_narrow(g_out, T * D, b0, stream))) return rc;
        if (gr.d_proj_w) {
            if ((rc = launch_narrow(y, T * HD, b1, stream))) return rc;
            if ((rc = launch_gemm_tn_bf16(g16, D, D, b1, HD, HD, T, (float *)gr.d_proj_w, HD, stream))) return rc;
        }
        if ((rc = launch_narrow_transpose((const float *)d.proj_w, D, HD, w16, stream))) return rc;   // [HD][D]
        if ((rc = launch_gemm_rows_bf16(g16, D, T, w16, D, D, HD, dy, HD, false, nullptr, stream))) return rc;
    } else {
        if (gr.d_proj_w && (rc = launch_gemm_tn(g_out, D, D, y, HD, HD, T, (float *)gr.d_proj_w, HD, stream))) return rc;
        if ((rc = dense_gemm_kmajor(g_out, T, D, (const float *)d.proj_w, HD, dy, stream, nullptr, 0, ws + L.wt))) return rc;
    }
    // ---- attention
    AttnBwdArgs B;
    B.q_pre = q; B.dy = dy; B.kn = kn; B.vpre = vpre; B.lambda = d.lambda_factor; B.ids = dual ? nullptr : d.ids_a; B.rows = R; B.T = T; B.bpt = d.bpt; B.H = H;
    B.layout = d.head_layout; B.cos_q = d.cos_q; B.sin_q = d.sin_q; B.cos_k = d.cos_k; B.sin_k = d.sin_k; B.eps = eps;
    B.dq = dq; B.pw = pw; B.dsw = dsw; B.qrot = qrot;
    B.dq16 = mm16(d) ? (__bf16 *)(ws + L.dq16) : nullptr;
    B.kn16 = A.kt16; B.vl16 = A.vt16;
    if (quad_form(B.bpt, R, HD, B.kn, B.vpre, B.cos_k, B.sin_k)) {
        if (B.kn16 && B.vl16) hipLaunchKernelGGL(cross_attn_bwd_quad_kernel<true>, dim3((unsigned)((waves + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, B);
        else hipLaunchKernelGGL(cross_attn_bwd_quad_kernel<false>, dim3((unsigned)((waves + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, B);
    } else hipLaunchKernelGGL(cross_attn_bwd_kernel, dim3((unsigned)((waves + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, B);
    if ((rc = check_launch("cross_attn_bwd_kernel"))) return rc;
    // ---- per byte-table row: the sums over the kv positions of every byte id, from the grouped positions
    {
        const int32_t *pos_sorted, *id_sorted;
        if (dual) {   // every kv position is its own row: the identity order, runs of one
            int32_t *iota = (int32_t *)(ws + L.grp);
            hipLaunchKernelGGL(iota_i32_kernel, dim3(1024), dim3(kThreads), 0, stream, iota, P);
            if ((rc = check_launch("iota_i32_kernel"))) return rc;
            pos_sorted = id_sorted = iota;
        } else if ((rc = launch_group_positions(ids32, P, R, (int32_t *)(ws + L.grp), &pos_sorted, &id_sorted, nullptr, stream))) return rc;
        AttnRowsArgs Rw;
        Rw.pw = pw; Rw.dsw = dsw; Rw.dy = dy; Rw.qrot = qrot; Rw.cos_k = d.cos_k; Rw.sin_k = d.sin_k;
        Rw.pos_sorted = pos_sorted; Rw.id_sorted = id_sorted; Rw.P = P; Rw.T = T; Rw.bpt = d.bpt; Rw.H = H; Rw.layout = d.head_layout;
        Rw.dkn_tab = dkn_tab; Rw.dvl_tab = dvl_tab;
        const int64_t nw = (P + kRowsSeg - 1) / kRowsSeg;
        const dim3 g1((unsigned)((nw + kWaves - 1) / kWaves));
        switch (H) {
            case 1: hipLaunchKernelGGL(attn_kv_rows_bwd_h_kernel<1>, g1, dim3(kThreads), 0, stream, Rw); break;
            case 2: hipLaunchKernelGGL(attn_kv_rows_bwd_h_kernel<2>, g1, dim3(kThreads), 0, stream, Rw); break;
            case 3: hipLaunchKernelGGL(attn_kv_rows_bwd_h_kernel<3>, g1, dim3(kThreads), 0, stream, Rw); break;
            case 4: hipLaunchKernelGGL(attn_kv_rows_bwd_h_kernel<4>, g1, dim3(kThreads), 0, stream, Rw); break;
            case 6: hipLaunchKernelGGL(attn_kv_rows_bwd_h_kernel<6>, g1, dim3(kThreads), 0, stream, Rw); break;
            case 8: hipLaunchKernelGGL(attn_kv_rows_bwd_h_kernel<8>, g1, dim3(kThreads), 0, stream, Rw); break;
            default: hipLaunchKernelGGL(attn_kv_rows_bwd_kernel, dim3(g1.x, (unsigned)H), dim3(kThreads), 0, stream, Rw);   // one head slice per wave
        }
        if ((rc = check_launch("attn_kv_rows_bwd_kernel"))) return rc;
    }
    MotEmbedMixDesc ed;
    MotEmbedMixGrads eg;
    memset(&eg, 0, sizeof(eg));
    eg.struct_size = sizeof(eg);
    {
        int64_t nb = (kvw + kWaves - 1) / kWaves;
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(kv_table_bwd_kernel, dim3((unsigned)nb), dim3(kThreads), 0, stream, dkn_tab, dvl_tab, kpre, vpre, R, H, d.lambda_factor, eps, dkv,
                           gr.d_lambda);
    }
    if ((rc = check_launch("kv_table_bwd_kernel"))) return rc;
    // ---- kv_w and the byte table (xkv: the normalised table rows built for the recompute above)
    if (kv16 && (gr.d_kv_w || gr.d_byte_table) && (rc = launch_narrow(dkv, R * 2 * HD, b1, stream))) return rc;
    if (gr.d_kv_w) {
        if (kv16) rc = launch_gemm_tn_bf16(b1, 2 * HD, 2 * HD, x16, D, D, R, (float *)gr.d_kv_w, D, stream);
        else rc = launch_gemm_tn(dkv, 2 * HD, 2 * HD, xkv, D, D, R, (float *)gr.d_kv_w, D, stream);
        if (rc) return rc;
    }
    if (gr.d_byte_table && dual) {
        // d(E[a] + E[b]) = norm^T(dxkv), added to the table at rows a AND b: the norm's backward in place, then the plain embedding
        // backward twice (positions grouped by id, run sums, one row add per run and wave).  (Round 2 used the SUM front-end's
        // backward with the table in both roles: its byte side is built for narrow byte rows in LDS, and a 768-wide "byte" row per
        // position became 800 M float atomics on 458 rows -- 226 ms of the 418 this backward took at 65 536 tokens x 16.)
        if (kv16) {
            if ((rc = launch_narrow_transpose(kv_w, 2 * HD, D, w16, stream))) return rc;   // [D][2 HD]
            if ((rc = launch_gemm_rows_bf16(b1, 2 * HD, R, w16, 2 * HD, 2 * HD, D, dxkv, D, false, nullptr, stream))) return rc;
        } else if ((rc = dense_gemm_kmajor(dkv, R, 2 * HD, kv_w, D, dxkv, stream, ws + L.part, L.part_n, mm16(d) ? nullptr : ws + L.wt))) return rc;
        int32_t *ids32b = (int32_t *)(ws + L.ids32b);
        hipLaunchKernelGGL(ids_to_i32_kernel, dim3(1024), dim3(kThreads), 0, stream, d.ids_b, P, (int64_t)d.byte_rows, ids32b);
        if ((rc = check_launch("ids_to_i32"))) return rc;
        if (d.norm_byte) {
            hipLaunchKernelGGL(dual_rows_norm_bwd_kernel, dim3(8192), dim3(kThreads), 0, stream, (const float *)d.byte_table, ids32, ids32b, P, D, eps, dxkv);
            if ((rc = check_launch("dual_rows_norm_bwd_kernel"))) return rc;
        }
        for (const int32_t *ids : {(const int32_t *)ids32, (const int32_t *)ids32b}) {
            noop_bwd_desc(ed, ids, P, d.byte_table, d.byte_rows, D, 0, eps, d.status);
            ed.workspace = emb_ws; ed.workspace_bytes = L.emb_bytes;
            memset(&eg, 0, sizeof(eg));
            eg.struct_size = sizeof(eg);
            eg.grad_out = dxkv; eg.d_tok_table = gr.d_byte_table;
            if ((rc = launch_embed_mix_bwd(ed, eg, stream))) return rc;
        }
        memset(&eg, 0, sizeof(eg));
        eg.struct_size = sizeof(eg);
    } else if (gr.d_byte_table) {
        if ((rc = dense_gemm_kmajor(dkv, R, 2 * HD, kv_w, D, dxkv, stream, ws + L.part, L.part_n, mm16(d) ? nullptr : ws + L.wt))) return rc;   // kv_w as [2 HD, D]
        hipLaunchKernelGGL(byte_rows_bwd_kernel, dim3((unsigned)((R + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, (const float *)d.byte_table, dxkv, R, D,
                           d.norm_byte, eps, (float *)gr.d_byte_table);
        if ((rc = check_launch("byte_rows_bwd_kernel"))) return rc;
    }
    // ---- q_w and the token table
    // (mm16: dq arrived in bf16, straight from the attention backward)
    if (gr.d_q_w) {
        if (mm16(d) && d.tok_table_bf16 && !have_xq) {   // the normalised token rows in bf16 directly from the caller's bf16 table
            if ((rc = launch_gather_rows(d.tokens, nullptr, 4, T, d.tok_table_bf16, d.tok_rows, D, d.norm_tok, eps, nullptr, b0, d.status, MOT_BF16, stream)))
                return rc;
        } else {
            if (!have_xq && (rc = launch_gather_rows(d.tokens, nullptr, 4, T, d.tok_table, d.tok_rows, D, d.norm_tok, eps, nullptr, xq, d.status, MOT_F32,
                                                     stream))) return rc;
            if (mm16(d) && (rc = launch_narrow(xq, T * D, b0, stream))) return rc;
        }
        if (mm16(d)) rc = launch_gemm_tn_bf16((const __bf16 *)(ws + L.dq16), HD, HD, b0, D, D, T, (float *)gr.d_q_w, D, stream);
        else rc = launch_gemm_tn(dq, HD, HD, xq, D, D, T, (float *)gr.d_q_w, D, stream);
        if (rc) return rc;
    }
    if (gr.d_tok_table) {
        if (mm16(d)) {
            if ((rc = launch_narrow_transpose((const float *)d.q_w, HD, D, w16, stream))) return rc;   // [D][HD]
            if ((rc = launch_gemm_rows_bf16(ws + L.dq16, HD, T, w16, HD, HD, D, dxq, D, false, nullptr, stream))) return rc;
        } else if ((rc = dense_gemm_kmajor(dq, T, HD, (const float *)d.q_w, D, dxq, stream, nullptr, 0, ws + L.wt))) return rc;   // q_w [HD, D]
        noop_bwd_desc(ed, d.tokens, T, d.tok_table, d.tok_rows, D, d.norm_tok, eps, d.status);
        ed.workspace = emb_ws; ed.workspace_bytes = L.emb_bytes;
        eg.grad_out = dxq; eg.d_tok_table = gr.d_tok_table;
        if ((rc = launch_embed_mix_bwd(ed, eg, stream))) return rc;
    }
    return MOT_OK;
}

}  // namespace mot
