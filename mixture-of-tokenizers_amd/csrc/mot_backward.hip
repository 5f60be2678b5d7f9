// mot_backward.hip -- backward of the fused front-end for the gather + sum family (SUM, NOOP):
// dL/dx -> dL/d{token table, byte table, learned scalars}, what autograd computes for
// norm(embed_tokens(tok) + concat_k embed_bytes(byte_k)) and its variants
// (modded-nanogpt/runs/71_*.py:227-230, 312-314; 71041: 311-313; 71081: 302-315;
//  scaled-pre-train/train_gpt.py:342-348 tokens-only mode), called from loss.backward()
// (train_gpt.py:1319; mathblations/main.py:304).
//
// Per token (one wave): re-gather the rows, recompute the forward scalars (rms factors), push the
// upstream gradient back through the norms
//     x = y*r, r = rsqrt(mean(y^2)+eps)   =>   dy = r*(g - x*mean(g*x))
// and scatter-add:
//   * token table: global float atomics.  Lane l owns elements l, l+64, ... of the row, so every
//     atomic wave-instruction covers 256 contiguous bytes -- the shape that runs at the chip-wide
//     atomic rate (~1.3 TB/s); float4-per-lane would spread each instruction over 1 KiB.
//   * byte table (458 rows hit 8.4 M times per step): privatised in LDS per workgroup
//     (ds_add_f32), flushed once with contiguous global atomics.
// 512-thread workgroups, one per CU (the LDS copy of the byte-table gradient is ~88 KB), persistent
// over tokens.  Byte ids are taken as given (the forward returns them), so no tile machinery here.
// Float atomics make the sums order-dependent in the last bits, like the reference's own GPU
// embedding backward; the parity tests state the tolerance they use against a float64 evaluation.
#include "mot_mix.hpp"

namespace mot {

constexpr int kBwdThreads = 512;  // 8 waves, 2 per SIMD: a 256-register budget per lane
constexpr int kBwdWaves = kBwdThreads / 64;

struct BwdArgs {
    const int32_t *tokens;
    int64_t n_tokens;
    int bpt;
    const int64_t *ids_a, *ids_b;
    const float *tok_table;
    int64_t tok_rows;
    int D;
    const float *byte_table;
    int64_t byte_rows;
    int Db;
    int norm_tok, norm_byte, norm_out;
    float eps;
    const float *scale_tok, *scale_byte;
    const float *byte_rnorm;
    const float *grad_out;
    float *d_tok, *d_byte, *d_scale_tok, *d_scale_byte;
    uint32_t *status;
    int privatize;  // byte-table gradient accumulated in LDS
};

template <int MODE, int NE>
__global__ __launch_bounds__(kBwdThreads) void embed_mix_bwd_kernel(const BwdArgs A) {
    extern __shared__ float lds_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbyte = A.privatize ? (int)A.byte_rows * A.Db : 0;
    float *dbyte_l = lds_f;                           // [byte_rows*Db]
    float *seg = lds_f + nbyte + wave * kMaxBpt;      // per-wave per-slot dot products
    for (int i = tid; i < nbyte; i += kBwdThreads) dbyte_l[i] = 0.f;
    __syncthreads();

    const int D = A.D;
    const float inv_db = MODE == MOT_MIX_SUM ? 1.0f / (float)A.Db : 0.f;
    // element e = lane + 64*j of a row lives in byte slot e / Db; (e + 0.5) * (1/Db) floors exactly for e < 2048
    auto slot_of = [&](int e) { return MODE == MOT_MIX_SUM ? __float2int_rd(((float)e + 0.5f) * inv_db) : 0; };
    const float s_tok = A.scale_tok ? *A.scale_tok : 1.0f;
    const float s_byte = A.scale_byte ? *A.scale_byte : 1.0f;
    float ds_t = 0.f, ds_b = 0.f;
    float *dbyte_dst = A.privatize ? dbyte_l : A.d_byte;

    for (int64_t n = (int64_t)blockIdx.x * kBwdWaves + wave; n < A.n_tokens; n += (int64_t)gridDim.x * kBwdWaves) {
        int tok = A.tokens[n];
        if ((uint64_t)(uint32_t)tok >= (uint64_t)A.tok_rows) {
            if (A.status && lane == 0) atomicOr(A.status, kStatusTokenOor);
            tok = 0;
        }
        const float *trow = A.tok_table + (int64_t)tok * D;
        const float *grow = A.grad_out + n * D;
        float an[NE], bn[NE], dy[NE];
        int id1[NE];
        // ---- gather (the same rows the forward read)
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int e = lane + 64 * j;
            const bool act = e < D;
            an[j] = act ? trow[e] : 0.f;
            dy[j] = act ? grow[e] : 0.f;  // holds g until the norm backward below
            bn[j] = 0.f;
            id1[j] = 0;
            if (MODE == MOT_MIX_SUM && act) {
                const int sl = slot_of(e), wi = e - sl * A.Db;
                int64_t ia = A.ids_a[n * A.bpt + sl];
                if ((uint64_t)ia >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); ia = 0; }
                id1[j] = (int)ia;
                float v = A.byte_table[ia * A.Db + wi];
                if (A.ids_b) {
                    int64_t ib = A.ids_b[n * A.bpt + sl];
                    if ((uint64_t)ib >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); ib = 0; }
                    v += A.byte_table[ib * A.Db + wi];
                }
                if (A.norm_byte) v *= A.byte_rnorm[ia];
                bn[j] = v;  // normalised, unscaled
            }
        }
        // ---- forward scalars
        float ra = 1.f;
        if (A.norm_tok) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) ss += an[j] * an[j];
            ra = rms_scale(wave_sum(ss), D, A.eps);
#pragma unroll
            for (int j = 0; j < NE; ++j) an[j] *= ra;
        }
        if (A.norm_out) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const float y = an[j] * s_tok + (MODE == MOT_MIX_SUM ? bn[j] * s_byte : 0.f);
                ss += y * y;
            }
            const float ry = rms_scale(wave_sum(ss), D, A.eps);
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) m += dy[j] * ((an[j] * s_tok + (MODE == MOT_MIX_SUM ? bn[j] * s_byte : 0.f)) * ry);
            m = wave_sum(m) / (float)D;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const float x = (an[j] * s_tok + (MODE == MOT_MIX_SUM ? bn[j] * s_byte : 0.f)) * ry;
                dy[j] = ry * (dy[j] - x * m);
            }
        }
        // ---- token side
        {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) dot += dy[j] * an[j];
            ds_t += dot;  // d scale_tok = sum dy * a_n
            float mt = 0.f;
            if (A.norm_tok) mt = wave_sum(dot * s_tok) / (float)D;
            float *drow = A.d_tok + (int64_t)tok * D;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const float da = dy[j] * s_tok;
                const float v = A.norm_tok ? ra * (da - an[j] * mt) : da;
                if (lane + 64 * j < D) atomicAdd(drow + lane + 64 * j, v);
            }
        }
        // ---- byte side
        if (MODE == MOT_MIX_SUM) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) dot += dy[j] * bn[j];
            ds_b += dot;
            if (A.norm_byte) {  // per-slot mean(db * b_n): slots are ragged lane groups -> LDS accumulators
                if (lane < A.bpt) seg[lane] = 0.f;
                __threadfence_block();
#pragma unroll
                for (int j = 0; j < NE; ++j)
                    if (lane + 64 * j < D) atomicAdd(&seg[slot_of(lane + 64 * j)], dy[j] * s_byte * bn[j]);
                __threadfence_block();
            }
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int e = lane + 64 * j;
                if (e >= D) continue;
                const int sl = slot_of(e), wi = e - sl * A.Db;
                const float db = dy[j] * s_byte;
                float v = db;
                if (A.norm_byte) v = A.byte_rnorm[id1[j]] * (db - bn[j] * (seg[sl] / (float)A.Db));
                atomicAdd(dbyte_dst + (int64_t)id1[j] * A.Db + wi, v);
                if (A.ids_b) {
                    int64_t ib = A.ids_b[n * A.bpt + sl];
                    if ((uint64_t)ib >= (uint64_t)A.byte_rows) ib = 0;
                    atomicAdd(dbyte_dst + ib * A.Db + wi, v);
                }
            }
            if (A.norm_byte) __threadfence_block();  // seg is rewritten by the next token
        }
    }
    // ---- flush
    if (A.d_scale_tok) { ds_t = wave_sum(ds_t); if (lane == 0) atomicAdd(A.d_scale_tok, ds_t); }
    if (MODE == MOT_MIX_SUM && A.d_scale_byte) { ds_b = wave_sum(ds_b); if (lane == 0) atomicAdd(A.d_scale_byte, ds_b); }
    if (A.privatize) {
        __syncthreads();
        for (int i = tid; i < nbyte; i += kBwdThreads) {
            const float v = dbyte_l[i];
            if (v != 0.f) atomicAdd(A.d_byte + i, v);
        }
    }
}

template <int MODE, int NE>
static int launch_bwd(const BwdArgs &A, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)embed_mix_bwd_kernel<MODE, NE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return set_error(MOT_EHIP, "hipFuncSetAttribute(embed_mix_bwd_kernel): %s", hipGetErrorString(e));
        attr_set = true;
    }
    int64_t blocks = (A.n_tokens + kBwdWaves - 1) / kBwdWaves;
    if (blocks > 256) blocks = 256;  // one persistent workgroup per CU
    hipLaunchKernelGGL((embed_mix_bwd_kernel<MODE, NE>), dim3((unsigned)blocks), dim3(kBwdThreads), lds, stream, A);
    return check_launch("embed_mix_bwd_kernel");
}

template <int MODE>
static int dispatch_ne(const BwdArgs &A, size_t lds, hipStream_t stream) {
    const int ne = (A.D + 63) / 64;
    if (ne <= 1) return launch_bwd<MODE, 1>(A, lds, stream);
    if (ne <= 2) return launch_bwd<MODE, 2>(A, lds, stream);
    if (ne <= 4) return launch_bwd<MODE, 4>(A, lds, stream);
    if (ne <= 8) return launch_bwd<MODE, 8>(A, lds, stream);
    if (ne <= 12) return launch_bwd<MODE, 12>(A, lds, stream);
    if (ne <= 16) return launch_bwd<MODE, 16>(A, lds, stream);
    if (ne <= 24) return launch_bwd<MODE, 24>(A, lds, stream);
    if (ne <= 32) return launch_bwd<MODE, 32>(A, lds, stream);
    return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: model_dim %d > 2048 is not built", A.D);
}

size_t embed_mix_bwd_workspace_bytes(const MotEmbedMixDesc &d) {
    return (d.mode == MOT_MIX_SUM && d.norm_byte) ? (size_t)d.byte_rows * sizeof(float) : 0;
}

int launch_embed_mix_bwd(const MotEmbedMixDesc &d, const MotEmbedMixGrads &gr, hipStream_t stream) {
    if (d.mode != MOT_MIX_SUM && d.mode != MOT_MIX_NOOP)
        return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: only the SUM and NOOP modes are built (mode %d)", d.mode);
    if (d.mode == MOT_MIX_SUM && d.id_source != MOT_IDS_GIVEN)
        return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: pass the byte ids the forward returned (MOT_IDS_GIVEN)");
    BwdArgs A;
    A.tokens = d.tokens; A.n_tokens = d.n_rows * d.tokens_per_row; A.bpt = d.bpt;
    A.ids_a = d.ids_a; A.ids_b = d.ids_b;
    A.tok_table = (const float *)d.tok_table; A.tok_rows = d.tok_rows; A.D = d.tok_dim;
    A.byte_table = (const float *)d.byte_table; A.byte_rows = d.byte_rows; A.Db = d.byte_dim;
    A.norm_tok = d.norm_tok; A.norm_byte = d.norm_byte; A.norm_out = d.norm_out;
    A.eps = d.eps > 0.f ? d.eps : FLT_EPSILON;
    A.scale_tok = d.scale_tok; A.scale_byte = d.scale_byte; A.byte_rnorm = nullptr;
    A.grad_out = (const float *)gr.grad_out;
    A.d_tok = (float *)gr.d_tok_table; A.d_byte = (float *)gr.d_byte_table;
    A.d_scale_tok = gr.d_scale_tok; A.d_scale_byte = gr.d_scale_byte;
    A.status = d.status;
    size_t lds = (size_t)kBwdWaves * kMaxBpt * sizeof(float);
    A.privatize = 0;
    if (d.mode == MOT_MIX_SUM) {
        const size_t tab = (size_t)d.byte_rows * d.byte_dim * sizeof(float);
        if (tab + lds <= 150 * 1024) { A.privatize = 1; lds += tab; }
        if (d.norm_byte) {
            const size_t need = (size_t)d.byte_rows * sizeof(float);
            if (!d.workspace || d.workspace_bytes < need)
                return set_error(MOT_EWORKSPACE, "embed_mix_bwd: norm_byte needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
            int rc = launch_rows_rnorm(A.byte_table, d.byte_rows, d.byte_dim, A.eps, (float *)d.workspace, stream);
            if (rc) return rc;
            A.byte_rnorm = (const float *)d.workspace;
        }
        return dispatch_ne<MOT_MIX_SUM>(A, lds, stream);
    }
    return dispatch_ne<MOT_MIX_NOOP>(A, lds, stream);
}

}  // namespace mot
