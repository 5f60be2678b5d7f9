// mot_swa.hip -- the Llama character mixer of BASELINE config 5 (gfx950): sliding-window token <- character attention
// (inference/inference.py:146-224, TokenMixByCharBMM) on top of the two embedding gathers (323-327), with the residuals of
// TokenMixByCharBMMBlock.forward (260-267).  fp32; with matmul_dtype = MOT_BF16 (bf16 tables and weights widened by the caller) the
// outputs of wq / wk / wv / wo are kept as the bf16 tensors they are in a bf16 cast of the module and the result can leave in bf16.
//
//   xn = RMSNorm_a(E_tok[t])                 attention_norm, eps = norm_eps, learned weight           (lines 126-132, 261-267)
//   cn = RMSNorm_c(E_char[c])                char_norm
//   q  = wq xn;   k = wk cn;   v = wv cn     per token / per character embedding                     (199-200)
//   keys of token t = the c_v characters of each of the tokens t-7 .. t of its batch row, zero vectors in front of the row
//                                            swa_transform: key j = w * c_v + c, w = 0 the oldest    (174-179)
//   RoPE over the first head_dim / 2 elements of q AND of every key, at the QUERY's position t        (209-217)
//   p  = softmax_j(q . k_j / sqrt(head_dim));   y = sum_j p_j v_j;   out = wo y                       (219-238)
//   h  = out (+ lambda_tok E_tok[t] + lambda_char mean_c E_char[c])                                   (260-267)
//
// What the kernels exploit:
//   * k and v depend on the CHARACTER ID only (132 ids): they are projected once per character-table row into two
//     L2-resident tables [132, n_heads * head_dim] instead of once per (token, slot) -- 500 x fewer flops at 65 536 tokens;
//   * q and every key of a query are rotated by the same angle (both sit at position t in the reference's layout, lines
//     209-217), and a rotation applied to both factors leaves q . k unchanged: the rotation is skipped.  The float64
//     checker of the parity tests does rotate, per the published algorithm of rotary-embedding-torch, so the tests check
//     exactly this identity (to fp32 rounding);
//   * zero-padded keys keep their place in the softmax (score 0, value 0), as in the reference.
#include <string.h>

#include "mot_mix.hpp"

namespace mot {

// out[n] = weight * (x * rsqrt(mean(x^2) + eps)), x = table[ids ? ids[n] : n]   (RMSNorm.forward, inference.py:126-132).
// One wave per row, 16-byte lanes.
template <typename IdT>
__global__ __launch_bounds__(kThreads) void rows_rmsnorm_w_kernel(const IdT *__restrict__ ids, int64_t n, const float *__restrict__ table, int64_t rows,
                                                                  int dim, const float *__restrict__ weight, float eps, float *__restrict__ out,
                                                                  uint32_t *status, uint32_t oor_flag, __bf16 *__restrict__ out16 = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= n) return;
    int64_t id = ids ? (int64_t)ids[r] : r;
    if ((uint64_t)id >= (uint64_t)rows) {
        if (status && lane == 0) atomicOr(status, oor_flag);
        id = 0;
    }
    const float *p = table + id * dim;
    float *o = out + r * dim;
    const int nv = dim >> 2;
    float ss = 0.f;
    for (int j = lane; j < nv; j += 64) {
        const float4v v = *(const float4v *)(p + 4 * j);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)dim + eps);       // torch.rsqrt(x.pow(2).mean(-1) + eps), correctly rounded pieces
    for (int j = lane; j < nv; j += 64) {
        const float4v v = *(const float4v *)(p + 4 * j), w = *(const float4v *)(weight + 4 * j);
        const float4v x = (v * rs) * w;
        if (out16) Elem<__bf16>::store4_nt(out16 + r * dim + 4 * j, x);   // the row operand of a bf16 product: written in bf16 INSTEAD
        else *(float4v *)(o + 4 * j) = x;
    }
}

// h[n] += lambda_tok * E_tok[tok[n]] (+ lambda_char * mean_c E_char[cid[n][c]]) on a bf16 h (inference.py:264 / 267 with the attention
// output a bf16 tensor, as `self.wo(...)` is in a bf16 cast): fp32 sum of the widened h and the fp32 rows, one rounding.  One wave per
// token, 16-byte lanes; c_v = 0: the token term alone.
__global__ __launch_bounds__(kThreads) void swa_residual_bf16_kernel(const int32_t *__restrict__ tokens, const int64_t *__restrict__ char_ids, int64_t n,
                                                                     const float *__restrict__ tok_table, int64_t tok_rows, const float *__restrict__ char_table,
                                                                     int64_t char_rows, int c_v, int dim, const float *__restrict__ lambda_tok,
                                                                     const float *__restrict__ lambda_char, __bf16 *__restrict__ h, uint32_t *status) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (r >= n) return;
    int64_t id = tokens[r];
    if ((uint64_t)id >= (uint64_t)tok_rows) {
        if (status && lane == 0) atomicOr(status, kStatusTokenOor);
        id = 0;
    }
    int my = 0;   // lane c < c_v: the id of character c
    if (lane < c_v) {
        const int64_t c = char_ids[r * c_v + lane];
        my = (int)c;
        if ((uint64_t)c >= (uint64_t)char_rows) {
            if (status) atomicOr(status, kStatusByteOor);
            my = 0;
        }
    }
    const float lt = lambda_tok ? *lambda_tok : 1.f, lc = (lambda_char ? *lambda_char : 1.f) / (float)(c_v > 0 ? c_v : 1);
    const float *tp = tok_table + id * dim;
    __bf16 *hp = h + r * dim;
    for (int j = lane; j < (dim >> 2); j += 64) {
        float4v x = Elem<__bf16>::load4(hp + 4 * j) + lt * *(const float4v *)(tp + 4 * j);
        float4v m = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < c_v; ++c) m += *(const float4v *)(char_table + (int64_t)__builtin_amdgcn_readlane(my, c) * dim + 4 * j);
        x += lc * m;
        Elem<__bf16>::store4_nt(hp + 4 * j, x);
    }
}

// The attention core.  Workgroup = (head h, tile of TT tokens); K_h and V_h of all character rows live in LDS for the whole
// tile (132 x 64 x 2 floats = 68 KB at head_dim 64: two workgroups per CU); one wave per token.
//   scores: lane = key (window * c_v <= 64 of them).  The query row's address is wave-uniform, so q arrives through the scalar
//           cache into SGPRs and every product is ONE v_fmac (SGPR x the lane's key element from LDS): no cross-lane traffic.
//   softmax: two wave reductions; p and the key's character id go to a per-wave LDS strip as (p, id) pairs.
//   y:      lane = (key group g, four consecutive elements of the head): 64 / (HD / 4) groups walk the keys g, g + groups, ...,
//           one 8-byte LDS read for (p, id) and one 16-byte read of the value row piece per key, four fmas; the groups' partial
//           sums meet through two (one at head_dim 128) cross-lane exchanges.  (The first version handed p and id out with two
//           readlanes per key and element-per-lane value reads: 4.7 ms per 65 536 tokens x 32 heads.)
constexpr int kSwaThreads = 512, kSwaWaves = kSwaThreads / 64;   // 8 waves share one copy of K_h / V_h: two workgroups = 16 waves per CU

// (KV16: the key / value rows sit in LDS as bf16 and `q` points to bf16 queries -- with bf16 tables and weights q = wq xn, k = wk cn and
//  v = wv cn ARE bf16 tensors in the reference, and the LDS reads of the key / value rows are what bounds the kernel: 33 KB per
//  (token, head) in fp32, all four SIMDs on one LDS)
__device__ __forceinline__ uint32_t swa_pack2(float a, float b) {
    return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)a) | ((uint32_t)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
__device__ __forceinline__ float swa_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float swa_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

template <int HDL, bool KV16>   // head_dim = 64 * HDL
__global__ __launch_bounds__(kSwaThreads) void char_swa_kernel(const float *__restrict__ q, const float *__restrict__ ktab, const float *__restrict__ vtab,
                                                            const int64_t *__restrict__ char_ids, int64_t n0, int64_t n_tok, int64_t T, int c_v, int window,
                                                            int char_rows, int n_heads, int tile_tokens, float *__restrict__ y, uint32_t *status,
                                                            __bf16 *__restrict__ y16) {   // y16: y is written there in bf16 instead (row operand of wo on the bf16 MFMA)
    constexpr int HD = 64 * HDL;
    constexpr int KS = (KV16 ? HD / 2 : HD) + 4, VS = KV16 ? HD / 2 : HD;   // row strides in dwords; key rows padded by 16 bytes: lanes reading different rows spread over the banks
    constexpr int QUADS = HD / 4, GROUPS = 64 / QUADS;           // 16 x 4 at head_dim 64, 32 x 2 at 128
    extern __shared__ __attribute__((aligned(16))) float lds_kv[];
    float *lk = lds_kv, *lv = lds_kv + (size_t)char_rows * KS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *pid = (float2 *)(lv + (size_t)char_rows * VS) + wave * 64;   // this wave's (p, id) strip
    const int h = blockIdx.y;
    const int HDIM = n_heads * HD;
    for (int i = tid; i < char_rows * (HD / 4); i += kSwaThreads) {
        const int r = i / (HD / 4), c = (i - r * (HD / 4)) * 4;
        const float4v kk = *(const float4v *)(ktab + (int64_t)r * HDIM + h * HD + c), vv = *(const float4v *)(vtab + (int64_t)r * HDIM + h * HD + c);
        if (KV16) {
            *(uint2 *)(lk + r * KS + c / 2) = uint2{swa_pack2(kk.x, kk.y), swa_pack2(kk.z, kk.w)};
            *(uint2 *)(lv + r * VS + c / 2) = uint2{swa_pack2(vv.x, vv.y), swa_pack2(vv.z, vv.w)};
        } else {
            *(float4v *)(lk + r * KS + c) = kk;
            *(float4v *)(lv + r * VS + c) = vv;
        }
    }
    __syncthreads();
    const int nkeys = window * c_v;
    const int w = lane / c_v, c = lane - w * c_v;                 // this lane's key: window slot w (0 = oldest), character slot c
    const float scale = 1.0f / sqrtf((float)HD);                 // qk / self.head_dim ** .5, line 220
    const int g = lane / QUADS, dq = lane - g * QUADS;            // value pass: key group, element quad
    const int64_t t_lo = (int64_t)blockIdx.x * tile_tokens, t_hi = min(n_tok, t_lo + tile_tokens);
    // scores of this lane's key against q[tl, h, :] (scalar addresses) from a key row in registers or in LDS
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    constexpr int KR = KV16 ? HD / 8 : HD / 4;                   // 16-byte pieces of a key row
    auto dot_row = [&](int64_t tl, auto piece) -> float {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;           // four chains: a single one is 64 dependent fmas deep
        if constexpr (KV16) {
            // (the query is a bf16 tensor too -- written so by its product -- and arrives as scalar pairs: v_dot2c_f32_bf16, one
            //  instruction per two dims where unpacking the key pairs for fp32 multiply-adds took four)
            const uint32_t *q2 = (const uint32_t *)q + ((tl * HDIM + h * HD) >> 1);
#pragma unroll
            for (int d8 = 0; d8 < KR; ++d8) {
                const uint4 u = piece(d8);
                s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q2[4 * d8 + 0]), __builtin_bit_cast(bf16x2, u.x), s0, false);
                s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q2[4 * d8 + 1]), __builtin_bit_cast(bf16x2, u.y), s1, false);
                s2 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q2[4 * d8 + 2]), __builtin_bit_cast(bf16x2, u.z), s2, false);
                s3 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q2[4 * d8 + 3]), __builtin_bit_cast(bf16x2, u.w), s3, false);
            }
        } else {
            const float *qrow = q + tl * HDIM + h * HD;
#pragma unroll
            for (int d4 = 0; d4 < KR; ++d4) {
                const uint4 u = piece(d4);
                s0 += qrow[4 * d4 + 0] * __uint_as_float(u.x);
                s1 += qrow[4 * d4 + 1] * __uint_as_float(u.y);
                s2 += qrow[4 * d4 + 2] * __uint_as_float(u.z);
                s3 += qrow[4 * d4 + 3] * __uint_as_float(u.w);
            }
        }
        return ((s0 + s1) + (s2 + s3)) * scale;
    };
    // softmax over the nkeys keys (padding keys included, score 0), then y = sum_j p_j v_j through the wave's (p, id) strip
    auto finish = [&](int64_t tl, float s_real, bool real, int id) {
        const bool is_key = lane < nkeys;
        const float s = real ? s_real : 0.f;                     // not real: a zero vector of the padding (lines 175-176)
        const float m = wave_max(is_key ? s : -INFINITY);
        const float e = is_key ? expf(s - m) : 0.f;
        const float p = e / wave_sum(e);
        pid[lane] = float2{real ? p : 0.f, __int_as_float(id)};  // a padding key's value is the zero vector; lanes past the keys: p = 0
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int j = g; j < 64; j += GROUPS) {
            const float2 pi = pid[j];
            if (KV16) {
                const uint2 u = *(const uint2 *)(lv + __float_as_int(pi.y) * VS + 2 * dq);
                acc += pi.x * float4v{swa_lo(u.x), swa_hi(u.x), swa_lo(u.y), swa_hi(u.y)};
            } else {
                const float4v vv = *(const float4v *)(lv + __float_as_int(pi.y) * VS + 4 * dq);
                acc += pi.x * vv;
            }
        }
#pragma unroll
        for (int o = QUADS; o < 64; o <<= 1) {
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
            acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
        }
        if (lane < QUADS) {
            if (y16) Elem<__bf16>::store4_nt(y16 + tl * HDIM + h * HD + 4 * dq, acc);
            else *(float4v *)(y + tl * HDIM + h * HD + 4 * dq) = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the strip is rewritten for the wave's next token
        __builtin_amdgcn_wave_barrier();
    };
    auto checked = [&](int64_t idq) -> int {
        if ((uint64_t)idq >= (uint64_t)char_rows) { if (status) atomicOr(status, kStatusByteOor); return 0; }
        return (int)idq;
    };
    if constexpr (HDL == 1) {
        // head_dim 64: a wave takes CONSECUTIVE tokens and keeps its keys' rows in registers.  Lane (w, c) holds character c of the one
        // token of the window whose position in its batch row is w modulo the window: from a query to the next, only the lanes whose
        // slot the new token takes change their key -- c_v of 64 lanes re-read a row from LDS instead of all of them (the LDS reads of
        // key and value rows are what bounds this kernel; the softmax and the value pass do not care which lane holds which key).
        const int64_t per_wave = (tile_tokens + kSwaWaves - 1) / kSwaWaves;
        const int64_t w_lo = min(t_hi, t_lo + wave * per_wave), w_hi = min(t_hi, w_lo + per_wave);
        if (w_lo >= w_hi) return;                                // (no block barrier below)
        const bool is_key = lane < nkeys;
        uint4 kreg[KR];
#pragma unroll
        for (int d = 0; d < KR; ++d) kreg[d] = uint4{0u, 0u, 0u, 0u};
        auto load_row = [&](int id) {
#pragma unroll
            for (int d = 0; d < KR; ++d) kreg[d] = *(const uint4 *)(lk + id * KS + 4 * d);
        };
        int id = 0;
        bool real = false;
        int64_t tr = (n0 + w_lo) % T;                            // position of the query in its batch row (wave-uniform)
        {   // first query of the stretch: every lane finds the token of its slot
            const int back = (int)(((tr % window) - w + window) % window);
            const int64_t src = tr - back;
            if (is_key && src >= 0) { real = true; id = checked(char_ids[(n0 + w_lo - back) * c_v + c]); load_row(id); }
        }
        // the characters of the NEXT query's own token, for the lanes of its slot: requested one query ahead
        auto own_chars = [&](int64_t tl, int64_t tr_) -> int64_t {
            return (tl < w_hi && is_key && w == (int)(tr_ % window)) ? char_ids[(n0 + tl) * c_v + c] : -1;
        };
        int64_t tr_nx = tr + 1 == T ? 0 : tr + 1;
        int64_t id_nx = own_chars(w_lo + 1, tr_nx);
        for (int64_t tl = w_lo; tl < w_hi; ++tl) {
            const float s = dot_row(tl, [&](int d) { return kreg[d]; });
            const int64_t idq = id_nx;
            const int64_t tr_cur_nx = tr_nx;
            tr_nx = tr_nx + 1 == T ? 0 : tr_nx + 1;
            id_nx = own_chars(tl + 2, tr_nx);
            finish(tl, s, real, id);
            // the next query: its own token takes the slot of the oldest one; at the start of a batch row every other slot is padding
            if (tr_cur_nx == 0 && w != 0) real = false;
            if (idq >= 0) { real = true; id = checked(idq); load_row(id); }
        }
        return;
    }
    // the character id of this lane's key for token tl (-1: a padding key or no key): requested one token ahead of its use
    auto key_id = [&](int64_t tl) -> int64_t {
        const int64_t n = n0 + tl, tr = n % T, src_t = tr - (window - 1) + w;   // the window may not leave the token's batch row
        return (tl < t_hi && lane < nkeys && src_t >= 0) ? char_ids[(n - tr + src_t) * c_v + c] : -1;
    };
    int64_t id_nx = key_id(t_lo + wave);
    for (int64_t tl = t_lo + wave; tl < t_hi; tl += kSwaWaves) {
        const int64_t idq = id_nx;
        id_nx = key_id(tl + kSwaWaves);
        const bool real = idq >= 0;
        const int id = real ? checked(idq) : 0;
        const float *krow = lk + id * KS;
        const float s = dot_row(tl, [&](int d) { return *(const uint4 *)(krow + 4 * d); });
        finish(tl, s, real, id);
    }
}

static size_t swa_align(size_t n) { return (n + 63) & ~(size_t)63; }
constexpr int64_t kSwaSlab = 65536;    // tokens per pass: bounds the workspace (normalised rows, queries, attention output)

size_t char_swa_workspace_bytes(const MotCharSwaDesc &d) {
    const int64_t N = d.n_rows * d.tokens_per_row, slab = N < kSwaSlab ? N : kSwaSlab;
    const size_t hdim = (size_t)d.n_heads * d.head_dim;
    // [cn: char_rows x dim][K, V: char_rows x hdim each][xn | y: slab x max(dim, hdim)][q: slab x hdim][byte_rnorm scratch of the residual call]
    size_t fl = swa_align((size_t)d.char_rows * d.dim) + 2 * swa_align((size_t)d.char_rows * hdim) +
                swa_align((size_t)slab * (d.dim > (int)hdim ? d.dim : hdim)) + swa_align((size_t)slab * hdim) + swa_align(d.char_rows) +
                swa_align(gemm_rows_sliced_floats(d.char_rows, d.dim, (int)hdim));   // partial blocks of the key / value projections
    if (d.matmul_dtype == MOT_BF16)   // bf16 copies: the row operand of a product (xn, then y), wq, wo
        fl += swa_align(((size_t)slab * (d.dim > (int)hdim ? d.dim : hdim) + 1) / 2) + 2 * swa_align((hdim * (size_t)d.dim + 1) / 2);
    return fl * sizeof(float);
}

int launch_char_swa(const MotCharSwaDesc &d, hipStream_t stream) {
    const int64_t N = d.n_rows * d.tokens_per_row;
    const int hdim = d.n_heads * d.head_dim;
    const size_t need = char_swa_workspace_bytes(d);
    if (!d.workspace || d.workspace_bytes < need) return set_error(MOT_EWORKSPACE, "char_swa: needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
    const int64_t slab = N < kSwaSlab ? N : kSwaSlab;
    float *cn = (float *)d.workspace;
    float *kt = cn + swa_align((size_t)d.char_rows * d.dim), *vt = kt + swa_align((size_t)d.char_rows * hdim);
    float *xn = vt + swa_align((size_t)d.char_rows * hdim), *yb = xn;   // the attention output reuses the normalised rows' buffer
    float *qb = xn + swa_align((size_t)slab * (d.dim > hdim ? d.dim : hdim));
    float *rn_scratch = qb + swa_align((size_t)slab * hdim);
    const size_t part_n = gemm_rows_sliced_floats(d.char_rows, d.dim, hdim);
    float *part = rn_scratch + swa_align(d.char_rows);
    // matmul_dtype == MOT_BF16: the two products over the tokens on the bf16 MFMA (fp32 sums and results), their row operands and
    // weights narrowed to bf16 first (the weights once per call)
    const bool mm16 = d.matmul_dtype == MOT_BF16;
    float *a16 = part + swa_align(part_n);
    float *wq16 = a16 + swa_align(((size_t)slab * (d.dim > hdim ? d.dim : hdim) + 1) / 2), *wo16 = wq16 + swa_align(((size_t)hdim * d.dim + 1) / 2);
    const float eps = d.norm_eps > 0.f ? d.norm_eps : 1e-5f;   // ModelArgs.norm_eps default, inference.py:43
    int rc;
    if (mm16) {
        if ((rc = launch_narrow((const float *)d.wq, (int64_t)hdim * d.dim, wq16, stream))) return rc;
        if ((rc = launch_narrow((const float *)d.wo, (int64_t)hdim * d.dim, wo16, stream))) return rc;
    }
    if (d.kv_tables) {   // caller-kept key / value tables: built by this call unless it says they are current
        kt = (float *)d.kv_tables;
        vt = kt + (size_t)d.char_rows * hdim;
    }
    // ---- per character-table row: normalise, project to keys and values
    if (!(d.kv_tables && d.kv_tables_ready)) {
        hipLaunchKernelGGL(rows_rmsnorm_w_kernel<int64_t>, dim3((unsigned)((d.char_rows + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream,
                           (const int64_t *)nullptr, (int64_t)d.char_rows, (const float *)d.char_table, (int64_t)d.char_rows, d.dim,
                           (const float *)d.char_norm_w, eps, cn, d.status, kStatusByteOor);
        if ((rc = check_launch("rows_rmsnorm_w_kernel"))) return rc;
        // (132 rows: 32 output blocks of the plain kernel, each 128 steps deep; cut along dim they fill the chip)
        if ((rc = launch_gemm_rows_sliced(cn, d.dim, d.char_rows, (const float *)d.wk, d.dim, d.dim, hdim, kt, hdim, true, part, part_n, stream))) return rc;
        if ((rc = launch_gemm_rows_sliced(cn, d.dim, d.char_rows, (const float *)d.wv, d.dim, d.dim, hdim, vt, hdim, true, part, part_n, stream))) return rc;
    }
    // (matmul_dtype == MOT_BF16: the key / value rows are bf16 tensors in the reference's bf16 cast and are kept as such in LDS)
    const int row_dw = mm16 ? d.head_dim / 2 : d.head_dim;
    const size_t lds = ((size_t)d.char_rows * (row_dw + 4) + (size_t)d.char_rows * row_dw + 2 * 64 * kSwaWaves) * sizeof(float);
    if (lds > 160 * 1024) return set_error(MOT_EUNSUPPORTED, "char_swa: %d character rows x head_dim %d need %zu B of LDS (> 160 KiB)", d.char_rows, d.head_dim, lds);
    for (int64_t n0 = 0; n0 < N; n0 += slab) {
        const int64_t nn = N - n0 < slab ? N - n0 : slab;
        // (io_dtype == MOT_BF16: the last product writes its result -- a bf16 tensor in the reference's bf16 cast -- in bf16 and the residuals
        //  are added to it in place; adding them inside the product from an fp32 buffer cost it twice its time: 985 against 500 us)
        const bool out16 = d.io_dtype == MOT_BF16;
        float *out = (float *)d.out + n0 * d.dim;
        // ---- queries: gather + RMSNorm, then the projection
        hipLaunchKernelGGL(rows_rmsnorm_w_kernel<int32_t>, dim3((unsigned)((nn + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, d.tokens + n0, nn,
                           (const float *)d.tok_table, d.tok_rows, d.dim, (const float *)d.attn_norm_w, eps, xn, d.status, kStatusTokenOor,
                           mm16 ? (__bf16 *)a16 : nullptr);
        if ((rc = check_launch("rows_rmsnorm_w_kernel"))) return rc;
        if (mm16) {   // (the normalised rows and, below, the attention output arrive in bf16: no narrowing passes; q leaves its product in bf16)
            if ((rc = launch_gemm_rows_bf16(a16, d.dim, nn, wq16, d.dim, d.dim, hdim, qb, hdim, true, nullptr, stream))) return rc;
        } else if ((rc = launch_gemm_rows(xn, d.dim, nn, (const float *)d.wq, d.dim, d.dim, hdim, qb, hdim, true, stream))) return rc;
        // ---- attention
        const int tile = 256;
        const dim3 grid((unsigned)((nn + tile - 1) / tile), (unsigned)d.n_heads);
#define MOT_SWA_LAUNCH(HDL, KV16)                                                                                                             \
    do {                                                                                                                                      \
        static std::atomic<uint64_t> lds_ok{0};                                                                                               \
        if ((rc = ensure_max_dyn_lds((const void *)char_swa_kernel<HDL, KV16>, lds_ok, "char_swa_kernel"))) return rc;                         \
        hipLaunchKernelGGL((char_swa_kernel<HDL, KV16>), grid, dim3(kSwaThreads), lds, stream, qb, kt, vt, d.char_ids, n0, nn, d.tokens_per_row, \
                           d.c_v, d.window, d.char_rows, d.n_heads, tile, yb, d.status, mm16 ? (__bf16 *)a16 : nullptr);                      \
    } while (0)
        if (d.head_dim == 64) { if (mm16) MOT_SWA_LAUNCH(1, true); else MOT_SWA_LAUNCH(1, false); }
        else { if (mm16) MOT_SWA_LAUNCH(2, true); else MOT_SWA_LAUNCH(2, false); }
#undef MOT_SWA_LAUNCH
        if ((rc = check_launch("char_swa_kernel"))) return rc;
        // ---- residuals first (they overwrite `out`), then out += wo y
        bool accumulate = false;
        if (out16) {
            __bf16 *h16 = (__bf16 *)d.out + n0 * d.dim;
            if ((rc = launch_gemm_rows_bf16(a16, hdim, nn, wo16, hdim, hdim, d.dim, h16, d.dim, true, nullptr, stream))) return rc;
            MotEmbedMixDesc r;   // two_residual at the sizes of the LDS-table MEAN kernel: its read-modify-write form (the eight character rows of a
            memset(&r, 0, sizeof(r));   // token come out of LDS there, out of L2 in the plain kernel below: 200 against 370 us at 65 536 x 2048)
            r.struct_size = sizeof(r); r.dtype = MOT_F32; r.n_rows = 1; r.tokens_per_row = nn; r.tokens = d.tokens + n0;
            r.tok_table = d.tok_table; r.tok_rows = d.tok_rows; r.tok_dim = d.dim; r.model_dim = d.dim; r.out = h16; r.status = d.status;
            r.mode = MOT_MIX_MEAN; r.bpt = d.c_v; r.id_source = MOT_IDS_GIVEN; r.ids_a = d.char_ids + n0 * d.c_v;
            r.byte_table = d.char_table; r.byte_rows = d.char_rows; r.byte_dim = d.dim; r.scale_tok = d.lambda_tok; r.scale_byte = d.lambda_char;
            if (d.version == MOT_SWA_TWO_RESIDUAL && embed_mix_mean_takes_add16(r)) {
                if ((rc = launch_embed_mix(r, stream, h16))) return rc;
            } else if (d.version != MOT_SWA_NO_RESIDUAL) {
                const bool two = d.version == MOT_SWA_TWO_RESIDUAL;
                hipLaunchKernelGGL(swa_residual_bf16_kernel, dim3((unsigned)((nn + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, d.tokens + n0,
                                   d.char_ids + n0 * d.c_v, nn, (const float *)d.tok_table, d.tok_rows, (const float *)d.char_table, (int64_t)d.char_rows,
                                   two ? d.c_v : 0, d.dim, two ? (const float *)d.lambda_tok : nullptr, two ? (const float *)d.lambda_char : nullptr, h16,
                                   d.status);
                if ((rc = check_launch("swa_residual_bf16_kernel"))) return rc;
            }
            continue;
        }
        if (d.version != MOT_SWA_NO_RESIDUAL) {
            MotEmbedMixDesc r;
            memset(&r, 0, sizeof(r));
            r.struct_size = sizeof(r);
            r.dtype = MOT_F32;
            r.n_rows = 1; r.tokens_per_row = nn;                  // no pull: rows are independent, a slab is one flat row
            r.tokens = d.tokens + n0;
            r.tok_table = d.tok_table; r.tok_rows = d.tok_rows; r.tok_dim = d.dim; r.model_dim = d.dim;
            r.out = out; r.status = d.status;
            if (d.version == MOT_SWA_TWO_RESIDUAL) {              // + lambda_tok * toks + lambda_char * chars.mean(dim=-2), line 267
                r.mode = MOT_MIX_MEAN; r.bpt = d.c_v; r.id_source = MOT_IDS_GIVEN; r.ids_a = d.char_ids + n0 * d.c_v;
                r.byte_table = d.char_table; r.byte_rows = d.char_rows; r.byte_dim = d.dim;
                r.scale_tok = d.lambda_tok; r.scale_byte = d.lambda_char;
                r.workspace = rn_scratch; r.workspace_bytes = (size_t)d.char_rows * sizeof(float);
            } else {                                               // + toks, line 264
                r.mode = MOT_MIX_NOOP;
            }
            // two_residual at the sizes of the LDS-table MEAN kernel: the product first, plain, and the residuals added to its rows by that
            // kernel's out += form (a streaming read-modify-write) -- the product's own C += epilogue reads C element by element with
            // nothing to overlap it: 4.53 against 4.10 ms at 65 536 x 2048 x 2048
            if (d.version == MOT_SWA_TWO_RESIDUAL && embed_mix_mean_takes_add16(r)) {
                if (mm16) rc = launch_gemm_rows_bf16(a16, hdim, nn, wo16, hdim, hdim, d.dim, out, d.dim, false, nullptr, stream);
                else rc = launch_gemm_rows(yb, hdim, nn, (const float *)d.wo, hdim, hdim, d.dim, out, d.dim, true, stream);
                if (rc) return rc;
                if ((rc = launch_embed_mix(r, stream, nullptr, true))) return rc;
                continue;
            }
            if ((rc = launch_embed_mix(r, stream))) return rc;
            accumulate = true;
        }
        if (mm16) {
            if ((rc = launch_gemm_rows_bf16(a16, hdim, nn, wo16, hdim, hdim, d.dim, out, d.dim, false, nullptr, stream, accumulate))) return rc;
        } else if ((rc = launch_gemm_rows(yb, hdim, nn, (const float *)d.wo, hdim, hdim, d.dim, out, d.dim, true, stream, nullptr, accumulate))) return rc;
    }
    return MOT_OK;
}

}  // namespace mot
