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
//   * token table: FineWeb-shaped ids are heavily skewed (the most frequent id takes ~3 % of all
//     positions), and thousands of float atomics on one row serialise (measured: 4.5 ms per 524 k
//     tokens, 28 % of the atomic ceiling).  So the positions are first counting-sorted by token id
//     (histogram -> scan -> scatter, three tiny kernels); each wave then walks a window of kWindow
//     SORTED positions, keeps the running gradient row of the current token in registers and issues
//     one atomic row-add per run of equal tokens.  Lane l owns elements l, l+64, ... of the row, so
//     every atomic wave-instruction covers 256 contiguous bytes (the shape that runs at the chip-wide
//     atomic rate); float4-per-lane would spread each instruction over 1 KiB.
//   * byte table (458 rows hit 8.4 M times per step): privatised in LDS per workgroup
//     (ds_add_f32), flushed once with contiguous global atomics.
// 512-thread workgroups, one per CU (the LDS copy of the byte-table gradient is ~88 KB), persistent
// over tokens.  Byte ids are taken as given (the forward returns them), so no tile machinery here.
// Float atomics make the sums order-dependent in the last bits, like the reference's own GPU
// embedding backward; the parity tests state the tolerance they use against a float64 evaluation.
#include <stdlib.h>

#include "mot_mix.hpp"

namespace mot {

constexpr int kBwdThreads = 512;  // 8 waves, 2 per SIMD: a 256-register budget per lane
constexpr int kBwdWaves = kBwdThreads / 64;
constexpr int kWindow = 64;  // sorted positions per wave work item

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
    const int32_t *pos_sorted;  // token positions ordered by token id
    int abl;  // dev-only timing ablations (MOT_DEV_ABLATION builds): 1 no LDS byte adds, 2 no token-row flush, 4 no wave sums
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

    float acc[NE];  // running d(token row) of the current run of equal tokens
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0 || (A.abl & 2)) return;
        float *drow = A.d_tok + (int64_t)cur * D;
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (lane + 64 * j < D) atomicAdd(drow + lane + 64 * j, acc[j]);
    };
    // Each position needs position -> token -> rows and position -> byte ids -> byte rows: up to four
    // dependent round trips.  The index side (position, token, byte ids) of position i+1 is therefore
    // fetched while position i's rows are in flight, leaving one row-fetch latency per position.
    auto load_ids = [&](int64_t n, int (&ids)[NE]) {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            ids[j] = 0;
            const int e = lane + 64 * j;
            if (MODE == MOT_MIX_SUM && e < D) {
                int64_t ia = A.ids_a[n * A.bpt + slot_of(e)];
                if ((uint64_t)ia >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); ia = 0; }
                ids[j] = (int)ia;
            }
        }
    };
    auto load_tok = [&](int64_t n) {
        int t = A.tokens[n];
        if ((uint64_t)(uint32_t)t >= (uint64_t)A.tok_rows) {
            if (A.status && lane == 0) atomicOr(A.status, kStatusTokenOor);
            t = 0;
        }
        return t;
    };
    const int64_t nwin = (A.n_tokens + kWindow - 1) / kWindow;
    for (int64_t w = (int64_t)blockIdx.x * kBwdWaves + wave; w < nwin; w += (int64_t)gridDim.x * kBwdWaves) {
    const int64_t s_end = min(A.n_tokens, (w + 1) * kWindow);
    int64_t n_nx = A.pos_sorted[w * kWindow];
    int tok_nx = load_tok(n_nx);
    int id_nx[NE];
    load_ids(n_nx, id_nx);
    for (int64_t si = w * kWindow; si < s_end; ++si) {
        const int64_t n = n_nx;
        const int tok = tok_nx;
        int id1[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) id1[j] = id_nx[j];
        // index of the next position first (oldest outstanding load), then this position's rows
        const int64_t s_nx = min(si + 1, s_end - 1);
        n_nx = A.pos_sorted[s_nx];
        if (tok != cur) {
            flush();
            cur = tok;
#pragma unroll
            for (int j = 0; j < NE; ++j) acc[j] = 0.f;
        }
        const float *trow = A.tok_table + (int64_t)tok * D;
        const float *grow = A.grad_out + n * D;
        float an[NE], bn[NE], dy[NE];
        // ---- gather (the same rows the forward read)
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int e = lane + 64 * j;
            const bool act = e < D;
            an[j] = act ? trow[e] : 0.f;
            dy[j] = act ? grow[e] : 0.f;  // holds g until the norm backward below
            bn[j] = 0.f;
            if (MODE == MOT_MIX_SUM && act) {
                const int sl = slot_of(e), wi = e - sl * A.Db;
                float v = A.byte_table[(int64_t)id1[j] * A.Db + wi];
                if (A.ids_b) {
                    int64_t ib = A.ids_b[n * A.bpt + sl];
                    if ((uint64_t)ib >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); ib = 0; }
                    v += A.byte_table[ib * A.Db + wi];
                }
                if (A.norm_byte) v *= A.byte_rnorm[id1[j]];
                bn[j] = v;  // normalised, unscaled
            }
        }
        tok_nx = load_tok(n_nx);
        load_ids(n_nx, id_nx);
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
            const float ry = rms_scale((A.abl & 4) ? ss : wave_sum(ss), D, A.eps);
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) m += dy[j] * ((an[j] * s_tok + (MODE == MOT_MIX_SUM ? bn[j] * s_byte : 0.f)) * ry);
            m = ((A.abl & 4) ? m : wave_sum(m)) / (float)D;
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
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const float da = dy[j] * s_tok;
                acc[j] += A.norm_tok ? ra * (da - an[j] * mt) : da;
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
                // two explicit address spaces (ds_add_f32 / global_atomic_add_f32): a pointer that may be
                // either would become a flat atomic, which faults on the LDS aperture
                const int i1 = id1[j] * A.Db + wi;
                if (A.abl & 1) continue;
                if (A.privatize) atomicAdd(dbyte_l + i1, v); else atomicAdd(A.d_byte + i1, v);
                if (A.ids_b) {
                    int64_t ib = A.ids_b[n * A.bpt + sl];
                    if ((uint64_t)ib >= (uint64_t)A.byte_rows) ib = 0;
                    const int i2 = (int)ib * A.Db + wi;
                    if (A.privatize) atomicAdd(dbyte_l + i2, v); else atomicAdd(A.d_byte + i2, v);
                }
            }
            if (A.norm_byte) __threadfence_block();  // seg is rewritten by the next token
        }
    }
    }
    // ---- flush
    flush();
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

// ---- counting sort of the token positions by (clamped) token id
__global__ __launch_bounds__(kThreads) void bwd_hist_kernel(const int32_t *__restrict__ tokens, int64_t n, int64_t rows,
                                                            int32_t *__restrict__ counts) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        int t = tokens[i];
        if ((uint64_t)(uint32_t)t >= (uint64_t)rows) t = 0;
        atomicAdd(&counts[t], 1);
    }
}

// exclusive scan of counts[0..rows) into starts (one 1024-thread workgroup; rows <= a few 100 k)
__global__ __launch_bounds__(1024) void bwd_scan_kernel(const int32_t *__restrict__ counts, int64_t rows,
                                                        int32_t *__restrict__ starts) {
    __shared__ int32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (rows + 1023) / 1024, lo = min(rows, tid * per), hi = min(rows, lo + per);
    int32_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += counts[i];
    const int32_t incl = wave_incl_add(s, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t off = incl - s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    for (int64_t i = lo; i < hi; ++i) { starts[i] = off; off += counts[i]; }
}

__global__ __launch_bounds__(kThreads) void bwd_scatter_kernel(const int32_t *__restrict__ tokens, int64_t n, int64_t rows,
                                                               const int32_t *__restrict__ starts, int32_t *__restrict__ cursor,
                                                               int32_t *__restrict__ pos_sorted) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        int t = tokens[i];
        if ((uint64_t)(uint32_t)t >= (uint64_t)rows) t = 0;
        pos_sorted[starts[t] + atomicAdd(&cursor[t], 1)] = (int32_t)i;
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
    int64_t blocks = ((A.n_tokens + kWindow - 1) / kWindow + kBwdWaves - 1) / kBwdWaves;
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

// workspace: [byte_rnorm: byte_rows f32][counts: tok_rows][cursor: tok_rows][starts: tok_rows][pos_sorted: N] (int32)
static size_t bwd_rnorm_floats(const MotEmbedMixDesc &d) { return d.mode == MOT_MIX_SUM ? ((size_t)d.byte_rows + 3) & ~(size_t)3 : 0; }
size_t embed_mix_bwd_workspace_bytes(const MotEmbedMixDesc &d) {
    return (bwd_rnorm_floats(d) + 3 * (size_t)d.tok_rows + (size_t)(d.n_rows * d.tokens_per_row)) * 4;
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
    if (A.n_tokens > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: more than 2^31 tokens");
    const size_t need = embed_mix_bwd_workspace_bytes(d);
    if (!d.workspace || d.workspace_bytes < need)
        return set_error(MOT_EWORKSPACE, "embed_mix_bwd: needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
    float *rn = (float *)d.workspace;
    int32_t *counts = (int32_t *)(rn + bwd_rnorm_floats(d)), *cursor = counts + d.tok_rows, *starts = cursor + d.tok_rows;
    int32_t *pos_sorted = starts + d.tok_rows;
    {
        hipError_t e = hipMemsetAsync(counts, 0, 2 * (size_t)d.tok_rows * sizeof(int32_t), stream);  // counts + cursor
        if (e != hipSuccess) return set_error(MOT_EHIP, "embed_mix_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
        int64_t hb = (A.n_tokens + kThreads - 1) / kThreads;
        if (hb > 2048) hb = 2048;
        hipLaunchKernelGGL(bwd_hist_kernel, dim3((unsigned)hb), dim3(kThreads), 0, stream, A.tokens, A.n_tokens, A.tok_rows, counts);
        hipLaunchKernelGGL(bwd_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, A.tok_rows, starts);
        hipLaunchKernelGGL(bwd_scatter_kernel, dim3((unsigned)hb), dim3(kThreads), 0, stream, A.tokens, A.n_tokens, A.tok_rows,
                           starts, cursor, pos_sorted);
        int rc = check_launch("embed_mix_bwd sort kernels");
        if (rc) return rc;
    }
    A.pos_sorted = pos_sorted;
    A.abl = 0;
#ifdef MOT_DEV_ABLATION
    if (getenv("MOT_BWD_ABL")) A.abl = atoi(getenv("MOT_BWD_ABL"));
    if (getenv("MOT_BWD_SORT_ONLY")) return MOT_OK;  // dev: inspect the sort prologue's workspace from the host
#endif
    size_t lds = (size_t)kBwdWaves * kMaxBpt * sizeof(float);
    A.privatize = 0;
    if (d.mode == MOT_MIX_SUM) {
        const size_t tab = (size_t)d.byte_rows * d.byte_dim * sizeof(float);
        if (tab + lds <= 150 * 1024) { A.privatize = 1; lds += tab; }
        if (d.norm_byte) {
            int rc = launch_rows_rnorm(A.byte_table, d.byte_rows, d.byte_dim, A.eps, rn, MOT_F32, stream);
            if (rc) return rc;
            A.byte_rnorm = rn;
        }
        return dispatch_ne<MOT_MIX_SUM>(A, lds, stream);
    }
    return dispatch_ne<MOT_MIX_NOOP>(A, lds, stream);
}

}  // namespace mot
