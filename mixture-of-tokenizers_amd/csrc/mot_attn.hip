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

struct AttnArgs {
    const float *q;       // [T, HD] projected queries
    float *y;             // [T, HD]
    const float *kt, *vt; // [rows, HD]: per byte-table row (ids != null) or per kv position (ids == null)
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
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (w >= A.T * A.H) return;
    const int64_t t = w / A.H;
    const int h = (int)(w - t * A.H);
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
    const float sqrt_hd = sqrtf((float)kHd);
    float m = -FLT_MAX, l = 0.f, y0 = 0.f, y1 = 0.f;
    // flat row r of the (T*bpt, H, hd) key/value memory = pos * H + hk; walked incrementally (one division per wave)
    int64_t pos = t * A.bpt;
    int hk = h;
    if (A.layout == 0) {
        const int64_t r0 = ((int64_t)h * A.T + t) * A.bpt;
        pos = r0 / A.H;
        hk = (int)(r0 - pos * A.H);
    }
    // software pipeline: key c+1's row, rotary row and value are requested before key c's score chain (exp, max) runs;
    // the rotary row is re-read only when the kv position changes (as_viewed walks H heads per position)
    auto row_of = [&](int64_t p) {
        int64_t row = p;
        if (A.ids) {
            row = A.ids[p];
            if ((uint64_t)row >= (uint64_t)A.rows) {
                if (A.status && lane == 0) atomicOr(A.status, kStatusByteOor);
                row = 0;
            }
        }
        return row;
    };
    float k0n, k1n, v0n, v1n, ckn, skn;
    {
        const int64_t row = row_of(pos);
        const float *kp = A.kt + row * HD + hk * kHd, *vp = A.vt + row * HD + hk * kHd;
        k0n = kp[lane]; k1n = kp[64 + lane]; v0n = vp[lane]; v1n = vp[64 + lane];
        ckn = A.cos_k[pos * 64 + lane]; skn = A.sin_k[pos * 64 + lane];
    }
    for (int c = 0; c < A.bpt; ++c) {
        const float k0 = k0n, k1 = k1n, v0 = v0n, v1 = v1n, ck = ckn, sk = skn;
        if (c + 1 < A.bpt) {
            const int64_t pos_prev = pos;
            if (A.layout == 0) {
                if (++hk == A.H) { hk = 0; ++pos; }
            } else {
                ++pos;
            }
            const int64_t row = row_of(pos);
            const float *kp = A.kt + row * HD + hk * kHd, *vp = A.vt + row * HD + hk * kHd;
            k0n = kp[lane]; k1n = kp[64 + lane]; v0n = vp[lane]; v1n = vp[64 + lane];
            if (pos != pos_prev) { ckn = A.cos_k[pos * 64 + lane]; skn = A.sin_k[pos * 64 + lane]; }
        }
        const float ka = k0 * ck + k1 * sk, kb = k0 * (-sk) + k1 * ck;
        const float s = wave_sum(q0 * ka + q1 * kb) / sqrt_hd;   // line 286
        const float mn = fmaxf(m, s);
        const float scale = expf(m - mn), p = expf(s - mn);
        l = l * scale + p;
        y0 = y0 * scale + p * v0;
        y1 = y1 * scale + p * v1;
        m = mn;
    }
    float *yp = A.y + t * HD + h * kHd;
    yp[lane] = y0 / l;
    yp[64 + lane] = y1 / l;
}

__global__ __launch_bounds__(kThreads) void iota32_kernel(int32_t *p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) p[i] = (int32_t)i;
}

// ------------------------------------------------------------------------------------------ host
// workspace, in floats: [q: T*HD][y: T*HD][kt: R*HD][vt: R*HD][xkv: dual ? T*bpt*D : 0][iota: max(T, R) int32][byte0: 4][lin: ...]
struct AttnLayout { size_t q, y, kt, vt, xkv, iota, byte0, lin, lin_floats, total; int64_t R; };

static void dense_desc(MotEmbedMixDesc &g, const int32_t *iota, const float *byte0, const void *rows, int64_t n_rows_tab, int64_t n, int K,
                       const void *weight, int Dm, void *out, uint32_t *status, void *ws, size_t ws_bytes) {
    memset(&g, 0, sizeof(g));
    g.struct_size = sizeof(g); g.dtype = MOT_F32;
    g.n_rows = 1; g.tokens_per_row = n; g.bpt = 0; g.mode = MOT_MIX_CONCAT_LINEAR;
    g.tokens = iota; g.id_source = MOT_IDS_GIVEN; g.ids_a = (const int64_t *)iota;  // never read with bpt == 0
    g.tok_table = rows; g.tok_rows = n_rows_tab; g.tok_dim = K;
    g.byte_table = byte0; g.byte_rows = 1; g.byte_dim = 4;
    g.weight = weight; g.model_dim = Dm; g.out = out; g.status = status;
    g.workspace = ws; g.workspace_bytes = ws_bytes;
}

static AttnLayout attn_layout(const MotCrossAttnDesc &d) {
    AttnLayout L;
    const size_t T = (size_t)d.n_tokens, HD = (size_t)d.n_heads * kHd, D = (size_t)d.dim;
    const bool dual = d.ids_b != nullptr;
    L.R = dual ? (int64_t)(T * d.bpt) : d.byte_rows;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    L.q = take(T * HD); L.y = take(T * HD); L.kt = take((size_t)L.R * HD); L.vt = take((size_t)L.R * HD);
    L.xkv = take(dual ? T * d.bpt * D : 0);
    L.iota = take(T > (size_t)L.R ? T : (size_t)L.R); L.byte0 = take(4);
    MotEmbedMixDesc g;     // the widest of the GEMMs decides the transposed-weight scratch
    const int64_t max_rows = d.tok_rows > L.R ? d.tok_rows : L.R;   // the normalising GEMMs keep a per-row rms table in the scratch
    dense_desc(g, nullptr, nullptr, nullptr, max_rows, 1, (int)D, nullptr, (int)HD, nullptr, nullptr, nullptr, 0);
    g.norm_tok = 1;
    size_t a = embed_mix_linear_workspace_bytes(g);
    dense_desc(g, nullptr, nullptr, nullptr, max_rows, 1, (int)HD, nullptr, (int)D, nullptr, nullptr, nullptr, 0);
    g.norm_tok = 1;
    size_t b = embed_mix_linear_workspace_bytes(g);
    L.lin_floats = ((a > b ? a : b) + 3) / 4;
    L.lin = take(L.lin_floats);
    L.total = o;
    return L;
}

size_t cross_attn_workspace_bytes(const MotCrossAttnDesc &d) { return attn_layout(d).total * 4; }

int launch_cross_attn(const MotCrossAttnDesc &d, hipStream_t stream) {
    const int64_t T = d.n_tokens;
    const int H = d.n_heads, HD = H * kHd, D = d.dim;
    const AttnLayout L = attn_layout(d);
    if (!d.workspace || d.workspace_bytes < L.total * 4)
        return set_error(MOT_EWORKSPACE, "cross_attn: needs %zu workspace bytes, got %zu", L.total * 4, d.workspace_bytes);
    float *ws = (float *)d.workspace;
    float *q = ws + L.q, *y = ws + L.y, *kt = ws + L.kt, *vt = ws + L.vt, *xkv = ws + L.xkv, *byte0 = ws + L.byte0, *lin = ws + L.lin;
    int32_t *iota = (int32_t *)(ws + L.iota);
    const size_t lin_bytes = L.lin_floats * 4;
    const float eps = d.eps > 0.f ? d.eps : FLT_EPSILON;
    const bool dual = d.ids_b != nullptr;
    const int64_t n_iota = T > L.R ? T : L.R;
    int rc;
    hipLaunchKernelGGL(iota32_kernel, dim3(256), dim3(kThreads), 0, stream, iota, n_iota);
    if ((rc = check_launch("iota32_kernel"))) return rc;
    hipError_t e = hipMemsetAsync(byte0, 0, 16, stream);
    if (e != hipSuccess) return set_error(MOT_EHIP, "cross_attn: hipMemsetAsync: %s", hipGetErrorString(e));
    MotEmbedMixDesc g;
    // 1. q = W_q norm?(E_t[tok])          (train_gpt.py:348-377 + 277)
    dense_desc(g, d.tokens, byte0, d.tok_table, d.tok_rows, T, D, d.q_w, HD, q, d.status, lin, lin_bytes);
    g.norm_tok = d.norm_tok; g.eps = eps;
    if ((rc = launch_embed_mix_linear(g, stream))) return rc;
    // 2. key/value rows: per byte-table row, or per kv position when the embedding is norm(E[a] + E[b])
    const void *kv_rows = d.byte_table;
    int kv_norm = d.norm_byte;
    if (dual) {
        if ((rc = launch_gather_rows(d.ids_a, d.ids_b, 8, T * d.bpt, d.byte_table, d.byte_rows, D, d.norm_byte, eps, nullptr, xkv, d.status,
                                     MOT_F32, stream))) return rc;
        kv_rows = xkv;
        kv_norm = 0;
    }
    const float *kv_w = (const float *)d.kv_w;
    dense_desc(g, iota, byte0, kv_rows, L.R, L.R, D, kv_w, HD, kt, d.status, lin, lin_bytes);
    g.norm_tok = kv_norm; g.eps = eps;
    if ((rc = launch_embed_mix_linear(g, stream))) return rc;
    dense_desc(g, iota, byte0, kv_rows, L.R, L.R, D, kv_w + (size_t)HD * D, HD, vt, d.status, lin, lin_bytes);
    g.norm_tok = kv_norm; g.eps = eps;
    if ((rc = launch_embed_mix_linear(g, stream))) return rc;
    const int64_t kvw = L.R * H;
    hipLaunchKernelGGL(kv_finish_kernel, dim3((unsigned)((kvw + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, kt, vt, L.R, H,
                       d.lambda_factor, eps);
    if ((rc = check_launch("kv_finish_kernel"))) return rc;
    // 3. attention of every token over its own bpt keys
    AttnArgs A;
    A.q = q; A.y = y; A.kt = kt; A.vt = vt; A.ids = dual ? nullptr : d.ids_a; A.rows = L.R; A.T = T; A.bpt = d.bpt; A.H = H;
    A.layout = d.head_layout; A.cos_q = d.cos_q; A.sin_q = d.sin_q; A.cos_k = d.cos_k; A.sin_k = d.sin_k; A.eps = eps; A.status = d.status;
    const int64_t waves = T * H;
    hipLaunchKernelGGL(cross_attn_kernel, dim3((unsigned)((waves + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, A);
    if ((rc = check_launch("cross_attn_kernel"))) return rc;
    // 4. out = c_proj y                   (line 293)
    dense_desc(g, iota, byte0, y, T, T, HD, d.proj_w, D, d.out, d.status, lin, lin_bytes);
    return launch_embed_mix_linear(g, stream);
}

}  // namespace mot
