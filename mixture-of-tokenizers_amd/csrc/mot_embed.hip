// mot_embed.hip -- float path kernels (gfx950): the fused gather + mix forward for the
// SUM / MEAN / NOOP modes, the seam gather (materialised FlexibleEmbedding outputs) and the
// byte-table inverse-rms prologue.  fp32 throughout (the reference's CPU parity mode).
//
// Fused kernel (embed_mix_kernel): every WAVE works alone on a unit of 16 or 32 consecutive tokens of one row -- no workgroup
// barrier anywhere (the 256-thread tile version of round 1, seven barriers per tile, is gone from this file; the tile machinery
// of mot_tile.hpp still serves the standalone index kernels and the one-launch concat + linear kernels):
//   index pass  the unit's byte ids into wave-private LDS: token->byte-table rows in registers, valid counts and EOT segments
//               by DPP wave scans, branch-free compaction (mot_wave.hpp), or a coalesced copy of precomputed int64 ids.
//               Byte ids never touch HBM.
//   streaming   U tokens in flight per wave: the token row is read with 16 B/lane coalesced loads from a SCALAR base (lane l
//               owns float4 chunks l, l+64, ...; the token id is handed out with v_readlane), the byte rows that line up with
//               those chunks come from the (L2-resident, <= 1.4 MB) byte table, rms-norm reductions are DPP wave reductions
//               (row_shr / row_bcast: VALU, not LDS shuffles), the mixed row is written once with non-temporal 16 B stores.
//               Algorithmic traffic: 4 + 2*bpt + 4*Dt B read and 4*Dm B written per token; HBM-bound.
#include <stdlib.h>

#include "mot_wave.hpp"

namespace mot {

// ------------------------------------------------------------------------------------------ fused kernel
// MODE: MOT_MIX_NOOP / SUM / MEAN.   NCH: 16-byte chunks per lane (covers Dm <= 64*NCH*VEC).   U: tokens in flight per wave.
// Every WAVE is on its own (mot_wave.hpp): it owns a unit of A.unit consecutive tokens of one row, produces their byte ids
// in wave-private LDS and streams them; the four waves of a workgroup share nothing and never meet at a barrier.
// (105 VGPRs at fp32 / 768 columns: four waves per SIMD.  Forcing the fifth with a launch bound spills into the streaming loop:
// 471 us instead of 435 at 524 288 tokens.)
template <int MODE, int NCH, int U, typename T, bool DUAL>
__global__ __launch_bounds__(kThreads) void embed_mix_kernel(const MixArgs A) {
    const T *tok_table = (const T *)A.tok_table, *byte_table = (const T *)A.byte_table;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_wave[];
    constexpr bool has_ids = MODE != MOT_MIX_NOOP;
    constexpr bool dual = DUAL;                   // two id tensors: emb(padded) + emb(pulled)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t unit_id = (int64_t)blockIdx.x * kWaves + wave;
    if (unit_id >= A.n_units) return;             // no barrier anywhere below: a wave may leave on its own
    const int64_t row = unit_id / A.units_per_row;
    const int64_t u0 = (unit_id - row * A.units_per_row) * A.unit;
    const int ntok = (int)min((int64_t)A.unit, A.T - u0);
    const int stream_eb = (has_ids && A.id_source == MOT_IDS_FROM_TTB && A.pull_dir != kPullNone) ? A.ttb_elem : 0;
    const WaveLds W = wave_lds_carve(lds_wave + (size_t)wave * A.wave_lds, A.unit, has_ids ? A.bpt : 0, dual, stream_eb);

    typedef typename Elem<T>::vec vec_t;          // VEC floats: one 16-byte lane load (4 fp32 / 8 bf16)
    typedef typename Elem<T>::raw raw_t;
    constexpr int VEC = Elem<T>::kVec;
    const int Dm = A.Dt, nchunk = Dm / VEC;
    bool act[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) act[i] = lane + 64 * i < nchunk;
    // token rows of a batch of U tokens: lane unit_lane0 + j of `tokv` holds the unit's token j.  The first batch is requested
    // as soon as the token ids are known -- BEFORE the byte-index pass, which then runs under the rows' flight time instead of
    // in front of the first HBM request of every wave of a small launch.
    int tokv = 0, unit_lane0 = 0;
    raw_t ar[U][NCH];
    auto request_tok_rows = [&](int tb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = min(tb + u, ntok - 1);  // the tail re-reads the last token; its store is skipped
            int tok = __builtin_amdgcn_readlane(tokv, unit_lane0 + t);   // wave-uniform: the row's address is scalar
            if ((uint64_t)(uint32_t)tok >= (uint64_t)A.tok_rows) {
                if (A.status && lane == 0) atomicOr(A.status, kStatusTokenOor);
                tok = 0;
            }
            const T *trow = tok_table + (int64_t)tok * Dm;
#pragma unroll
            for (int i = 0; i < NCH; ++i) ar[u][i] = Elem<T>::load_raw(trow + VEC * (act[i] ? lane + 64 * i : 0));
        }
    };
    // ---- phase 1: the unit's byte ids into wave-private LDS.  Request order: token ids, their token->byte rows, the first
    // batch's token rows; the index pass waits for the table rows only (older in the queue than the token rows).
    auto from_ttb = [&](auto indexer) {
        tokv = indexer.tokens();
        unit_lane0 = indexer.unit_lane0;
        indexer.load_rows();
#ifndef MOT_VAR_NOPREFETCH      // dev A/B (tools/variants.sh): rows requested after the index pass cost 1.6 us at 65 536 tokens
        request_tok_rows(0);
#endif
        indexer.finish();
#ifdef MOT_VAR_NOPREFETCH
        request_tok_rows(0);
#endif
    };
    if (has_ids && A.id_source == MOT_IDS_FROM_TTB) {
        if (A.ttb_elem == 2) {
            if (A.pull_dir == kPullLeft) from_ttb(WaveIndexer<kPullLeft, int16_t>(A, W, row, u0, ntok, dual));
            else if (A.pull_dir == kPullRight) from_ttb(WaveIndexer<kPullRight, int16_t>(A, W, row, u0, ntok, dual));
            else from_ttb(WaveIndexer<kPullNone, int16_t>(A, W, row, u0, ntok, dual));
        } else {
            if (A.pull_dir == kPullLeft) from_ttb(WaveIndexer<kPullLeft, int32_t>(A, W, row, u0, ntok, dual));
            else if (A.pull_dir == kPullRight) from_ttb(WaveIndexer<kPullRight, int32_t>(A, W, row, u0, ntok, dual));
            else from_ttb(WaveIndexer<kPullNone, int32_t>(A, W, row, u0, ntok, dual));
        }
    } else {
        tokv = lane < ntok ? A.tokens[row * A.T + u0 + lane] : 0;
        request_tok_rows(0);
        if (has_ids) wave_ids_given(A, W, row, u0, ntok);
        else if (A.counters && lane == 0) atomicAdd((unsigned long long *)A.counters, (unsigned long long)ntok);
    }

    // ---- phase 2
    const int sv = A.bpt | 1;
    int slot[NCH], within[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        const int cc = act[i] ? c : 0;
        if (MODE == MOT_MIX_SUM) {
            slot[i] = (VEC * cc) / A.Db;            // concat_k: column j belongs to slot j / Db
            within[i] = VEC * cc - slot[i] * A.Db;
        } else {
            slot[i] = 0;
            within[i] = VEC * cc;
        }
    }
    const float s_tok = A.scale_tok ? *A.scale_tok : 1.0f;
    const float s_byte = A.scale_byte ? *A.scale_byte : 1.0f;
    const bool scale_t = A.scale_tok != nullptr, scale_b = A.scale_byte != nullptr;
    T *orow = (T *)A.out + (row * A.T + u0) * (int64_t)Dm;
    auto sumsq = [](const vec_t &v) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += v[e] * v[e];
        return s;
    };

    for (int tb = 0; tb < ntok; tb += U) {
        raw_t br[U][NCH], br2[DUAL ? U : 1][NCH];  // rows exactly as loaded (bf16 stays packed until it is used)
        vec_t bm[U][NCH];                           // MEAN accumulates while loading
        int idr[U][NCH];
        // ---- issue every load of the U tokens before touching any of them (the first batch's token rows are already in flight)
        if (tb > 0) request_tok_rows(tb);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = min(tb + u, ntok - 1);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (MODE == MOT_MIX_SUM) {
                    const int id = W.ids[t * sv + slot[i]];
                    idr[u][i] = id;
                    br[u][i] = Elem<T>::load_raw(byte_table + (int64_t)id * A.Db + within[i]);
                    if (dual) br2[DUAL ? u : 0][i] = Elem<T>::load_raw(byte_table + (int64_t)W.ids2[t * sv + slot[i]] * A.Db + within[i]);
                } else if (MODE == MOT_MIX_MEAN) {
                    vec_t acc = (vec_t)(0.f);
                    for (int k = 0; k < A.bpt; ++k) {  // chars.mean(dim=-2), inference.py:267
                        const int id = W.ids[t * sv + k];
                        vec_t v = Elem<T>::loadv(byte_table + (int64_t)id * A.Db + within[i]);
                        if (dual) {
                            const int id2 = W.ids2[t * sv + k];
                            v += Elem<T>::loadv(byte_table + (int64_t)id2 * A.Db + within[i]);
                        }
                        if (A.norm_byte) v *= A.byte_rnorm[id];
                        acc += v;
                    }
                    bm[u][i] = acc / (float)A.bpt;
                }
            }
        }
        // ---- mix, normalise, store
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tb + u;
            vec_t a[1][NCH], b[1][NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                a[0][i] = Elem<T>::widen(ar[u][i]);   // lanes past the row end hold a copy of the row's first chunk: kept out of the sums below, never stored
                if (MODE == MOT_MIX_SUM) {
                    vec_t v = Elem<T>::widen(br[u][i]);
                    if (dual) v += Elem<T>::widen(br2[DUAL ? u : 0][i]);  // emb(padded) + emb(pulled), train_gpt.py:378
                    if (A.norm_byte) v *= A.byte_rnorm[idr[u][i]];
                    b[0][i] = v;
                } else if (MODE == MOT_MIX_MEAN) {
                    b[0][i] = bm[u][i];
                }
            }
            if (A.norm_tok) {
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < NCH; ++i) ss += act[i] ? sumsq(a[0][i]) : 0.f;
                const float r = rms_scale(wave_sum(ss), Dm, A.eps);
#pragma unroll
                for (int i = 0; i < NCH; ++i) a[0][i] *= r;
            }
            if (scale_t) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) a[0][i] *= s_tok;
            }
            vec_t x[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (MODE == MOT_MIX_NOOP) x[i] = a[0][i];
                else x[i] = a[0][i] + (scale_b ? b[0][i] * s_byte : b[0][i]);
            }
            if (A.norm_out) {
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < NCH; ++i) ss += act[i] ? sumsq(x[i]) : 0.f;
                const float r = rms_scale(wave_sum(ss), Dm, A.eps);
#pragma unroll
                for (int i = 0; i < NCH; ++i) x[i] *= r;
            }
            if (t < ntok) {
#pragma unroll
                for (int i = 0; i < NCH; ++i)
                    if (act[i]) Elem<T>::storev_nt(orow + (int64_t)t * Dm + VEC * (lane + 64 * i), x[i]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ prologue
// byte_rnorm[r] = rsqrt(mean(byte_table[r]^2) + eps): rms-norm of a gathered byte row depends on
// the row only, so the per-slot reduction of norm(embed_bytes(ids)) (train_gpt.py:357,368)
// collapses to one multiply in the fused kernel.  One wave per table row.
template <typename T>
__global__ __launch_bounds__(kThreads) void rows_rnorm_kernel(const T *__restrict__ table, int64_t rows, int dim,
                                                              float eps, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= rows) return;
    float ss = 0.f;
    for (int j = lane; j < dim; j += 64) {
        const float v = Elem<T>::load1(table + r * dim + j);
        ss += v * v;
    }
    ss = wave_sum(ss);
    if (lane == 0) out[r] = rms_scale(ss, dim, eps);
}

// ------------------------------------------------------------------------------------------ seam gather
// out[n] = scale * rms_norm?(table[idsA[n]] (+ table[idsB[n]])): the tensors FlexibleEmbedding.forward
// returns (train_gpt.py:342-379).  G lanes cooperate on a row (G = 8..64 by row length).
template <int G, typename IdT, typename T>
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const IdT *__restrict__ ids_a, const IdT *__restrict__ ids_b,
                                                               int64_t n, const T *__restrict__ table, int64_t rows,
                                                               int dim, int rms, float eps, const float *scale,
                                                               T *__restrict__ out, uint32_t *status, int group, int64_t out_ld,
                                                               uint32_t oor_flag) {
    // row r lands at out + (r / group) * out_ld + (r % group) * dim: group 1, out_ld dim = a dense [n, dim] tensor;
    // group bpt, out_ld K = the byte part of the concat operand [tokens, K]
    const int g = threadIdx.x % G;
    const int64_t per_block = kThreads / G;
    const float s = scale ? *scale : 1.0f;
    const bool vec = (dim & 3) == 0;
    for (int64_t r = (int64_t)blockIdx.x * per_block + threadIdx.x / G; r < n; r += (int64_t)gridDim.x * per_block) {
        int64_t ia = (int64_t)ids_a[r], ib = ids_b ? (int64_t)ids_b[r] : 0;
        if ((uint64_t)ia >= (uint64_t)rows || (uint64_t)ib >= (uint64_t)rows) {
            if (status) atomicOr(status, oor_flag);
            if ((uint64_t)ia >= (uint64_t)rows) ia = 0;
            if ((uint64_t)ib >= (uint64_t)rows) ib = 0;
        }
        const T *pa = table + ia * dim, *pb = table + ib * dim;
        T *po = group == 1 ? out + r * out_ld : out + (r / group) * out_ld + (r % group) * dim;
        float mult = s;
        if (rms) {
            float ss = 0.f;
            if (vec) {
                for (int j = g; j < (dim >> 2); j += G) {
                    float4v v = Elem<T>::load4(pa + 4 * j);
                    if (ids_b) v += Elem<T>::load4(pb + 4 * j);
                    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                }
            } else {
                for (int j = g; j < dim; j += G) {
                    float v = Elem<T>::load1(pa + j);
                    if (ids_b) v += Elem<T>::load1(pb + j);
                    ss += v * v;
                }
            }
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, G);
            mult = rms_scale(ss, dim, eps);
        }
        if (vec) {
            for (int j = g; j < (dim >> 2); j += G) {
                float4v v = Elem<T>::load4(pa + 4 * j);
                if (ids_b) v += Elem<T>::load4(pb + 4 * j);
                if (rms) v *= mult;
                if (scale) v *= s;
                Elem<T>::store4_nt(po + 4 * j, v);
            }
        } else {
            for (int j = g; j < dim; j += G) {
                float v = Elem<T>::load1(pa + j);
                if (ids_b) v += Elem<T>::load1(pb + j);
                if (rms) v *= mult;
                if (scale) v *= s;
                Elem<T>::store1(po + j, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ launchers
template <int G, typename T>
static int launch_gather_g(const void *ids_a, const void *ids_b, int ids_elem, int64_t n, const void *table,
                           int64_t rows, int dim, int rms, float eps, const float *scale, void *out, uint32_t *status,
                           int group, int64_t out_ld, uint32_t oor_flag, hipStream_t stream) {
    const int64_t per_block = kThreads / G;
    int64_t blocks = (n + per_block - 1) / per_block;
    if (blocks > 256 * 128) blocks = 256 * 128;   // one row per lane group and pass: short dependent chains (id -> row), so many groups in flight
    if (ids_elem == 8)
        hipLaunchKernelGGL((gather_rows_kernel<G, int64_t, T>), dim3((unsigned)blocks), dim3(kThreads), 0, stream,
                           (const int64_t *)ids_a, (const int64_t *)ids_b, n, (const T *)table, rows, dim, rms, eps, scale, (T *)out, status, group, out_ld, oor_flag);
    else
        hipLaunchKernelGGL((gather_rows_kernel<G, int32_t, T>), dim3((unsigned)blocks), dim3(kThreads), 0, stream,
                           (const int32_t *)ids_a, (const int32_t *)ids_b, n, (const T *)table, rows, dim, rms, eps, scale, (T *)out, status, group, out_ld, oor_flag);
    return check_launch("gather_rows_kernel");
}

template <typename T>
static int launch_gather_t(const void *ids_a, const void *ids_b, int ids_elem, int64_t n, const void *table, int64_t rows,
                           int dim, int rms_norm, float eps, const float *scale, void *out, uint32_t *status, int group, int64_t out_ld,
                           uint32_t oor_flag, hipStream_t stream) {
    const int lanes = (dim & 3) == 0 ? dim / 4 : dim;
    if (lanes <= 8) return launch_gather_g<8, T>(ids_a, ids_b, ids_elem, n, table, rows, dim, rms_norm, eps, scale, out, status, group, out_ld, oor_flag, stream);
    if (lanes <= 16) return launch_gather_g<16, T>(ids_a, ids_b, ids_elem, n, table, rows, dim, rms_norm, eps, scale, out, status, group, out_ld, oor_flag, stream);
    if (lanes <= 32) return launch_gather_g<32, T>(ids_a, ids_b, ids_elem, n, table, rows, dim, rms_norm, eps, scale, out, status, group, out_ld, oor_flag, stream);
    return launch_gather_g<64, T>(ids_a, ids_b, ids_elem, n, table, rows, dim, rms_norm, eps, scale, out, status, group, out_ld, oor_flag, stream);
}

int launch_gather_rows_placed(const void *ids_a, const void *ids_b, int ids_elem, int64_t n, const void *table, int64_t rows,
                              int dim, int rms_norm, float eps, const float *scale, void *out, int group, int64_t out_ld,
                              uint32_t *status, uint32_t oor_flag, int dtype, hipStream_t stream) {
    if (n == 0) return MOT_OK;
    if (eps <= 0.f) eps = dtype == MOT_BF16 ? 0.0078125f : FLT_EPSILON;
    if (dtype == MOT_BF16)
        return launch_gather_t<__bf16>(ids_a, ids_b, ids_elem, n, table, rows, dim, rms_norm, eps, scale, out, status, group, out_ld, oor_flag, stream);
    return launch_gather_t<float>(ids_a, ids_b, ids_elem, n, table, rows, dim, rms_norm, eps, scale, out, status, group, out_ld, oor_flag, stream);
}

int launch_gather_rows(const void *ids_a, const void *ids_b, int ids_elem, int64_t n, const void *table, int64_t rows,
                       int dim, int rms_norm, float eps, const float *scale, void *out, uint32_t *status, int dtype,
                       hipStream_t stream) {
    return launch_gather_rows_placed(ids_a, ids_b, ids_elem, n, table, rows, dim, rms_norm, eps, scale, out, 1, dim, status, kStatusByteOor, dtype,
                                     stream);
}

// ------------------------------------------------------------------------------------------ concat operand
// u[t] = [ rms_norm?(tok_table[tokens[t]]) | rms_norm?(byte_table[ids[t, k]]) for k < bpt ]  (or bytes first): the operand of the
// concat + linear mixin, one wave per token, written as one contiguous row of K elements.  One id tensor; the byte rows' rms
// factors come from a per-row table (rows_rnorm_kernel), the token row's from a wave reduction.  16-byte vectors throughout
// (Dt, Db multiples of the vector length).
template <typename T>
__global__ __launch_bounds__(kThreads) void concat_rows_kernel(const int32_t *__restrict__ tokens, const int64_t *__restrict__ ids, int64_t n,
                                                               const T *__restrict__ tok_table, int64_t tok_rows, int Dt,
                                                               const T *__restrict__ byte_table, int64_t byte_rows, int Db, int bpt, int norm_tok,
                                                               const float *__restrict__ byte_rnorm, float eps, T *__restrict__ u, int K, int tok_lo,
                                                               int byte_lo, uint32_t *status) {
    using vec_t = typename Elem<T>::vec;
    constexpr int VEC = Elem<T>::kVec;
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (t >= n) return;
    int tok = tokens[t];
    if ((uint64_t)(uint32_t)tok >= (uint64_t)tok_rows) {
        if (status && lane == 0) atomicOr(status, kStatusTokenOor);
        tok = 0;
    }
    const T *trow = tok_table + (int64_t)tok * Dt;
    T *urow = u + t * K;
    const int nvt = Dt / VEC;
    float rs = 1.0f;
    if (norm_tok) {
        float ss = 0.f;
        for (int j = lane; j < nvt; j += 64) {
            const vec_t v = Elem<T>::loadv(trow + VEC * j);
#pragma unroll
            for (int e = 0; e < VEC; ++e) ss += v[e] * v[e];
        }
        rs = rms_scale(wave_sum(ss), Dt, eps);
    }
    for (int j = lane; j < nvt; j += 64) Elem<T>::storev_nt(urow + tok_lo + VEC * j, Elem<T>::loadv(trow + VEC * j) * rs);
    const int vps = Db / VEC, nvb = bpt * vps;   // vectors per slot, per token
    for (int j = lane; j < nvb; j += 64) {
        const int slot = j / vps, within = (j - slot * vps) * VEC;
        int64_t id = ids[t * bpt + slot];
        if ((uint64_t)id >= (uint64_t)byte_rows) {
            if (status) atomicOr(status, kStatusByteOor);
            id = 0;
        }
        vec_t v = Elem<T>::loadv(byte_table + id * Db + within);
        if (byte_rnorm) v *= byte_rnorm[id];
        Elem<T>::storev_nt(urow + byte_lo + slot * Db + within, v);
    }
}

int launch_concat_rows(const int32_t *tokens, const int64_t *ids, int64_t n, const void *tok_table, int64_t tok_rows, int Dt, const void *byte_table,
                       int64_t byte_rows, int Db, int bpt, int norm_tok, const float *byte_rnorm, float eps, void *u, int K, int tok_lo, int byte_lo,
                       uint32_t *status, int dtype, hipStream_t stream) {
    if (n == 0) return MOT_OK;
    const unsigned blocks = (unsigned)((n + kWaves - 1) / kWaves);
    if (dtype == MOT_BF16)
        hipLaunchKernelGGL(concat_rows_kernel<__bf16>, dim3(blocks), dim3(kThreads), 0, stream, tokens, ids, n, (const __bf16 *)tok_table, tok_rows, Dt,
                           (const __bf16 *)byte_table, byte_rows, Db, bpt, norm_tok, byte_rnorm, eps, (__bf16 *)u, K, tok_lo, byte_lo, status);
    else
        hipLaunchKernelGGL(concat_rows_kernel<float>, dim3(blocks), dim3(kThreads), 0, stream, tokens, ids, n, (const float *)tok_table, tok_rows, Dt,
                           (const float *)byte_table, byte_rows, Db, bpt, norm_tok, byte_rnorm, eps, (float *)u, K, tok_lo, byte_lo, status);
    return check_launch("concat_rows_kernel");
}

int launch_rows_rnorm(const void *table, int64_t rows, int dim, float eps, float *out, int dtype, hipStream_t stream) {
    const int64_t rb = (rows + kWaves - 1) / kWaves;
    if (dtype == MOT_BF16)
        hipLaunchKernelGGL(rows_rnorm_kernel<__bf16>, dim3((unsigned)rb), dim3(kThreads), 0, stream, (const __bf16 *)table, rows, dim, eps, out);
    else
        hipLaunchKernelGGL(rows_rnorm_kernel<float>, dim3((unsigned)rb), dim3(kThreads), 0, stream, (const float *)table, rows, dim, eps, out);
    return check_launch("rows_rnorm_kernel");
}

size_t embed_mix_workspace_bytes(const MotEmbedMixDesc &d) {
    if (d.mode == MOT_MIX_CONCAT_LINEAR) return d.dtype == MOT_BF16 ? embed_mix_linear_bf16_workspace_bytes(d) : embed_mix_linear_workspace_bytes(d);
    if (d.mode != MOT_MIX_NOOP && d.norm_byte) return (size_t)d.byte_rows * sizeof(float);
    return 0;
}

template <int MODE, int NCH, int U, typename T>
static int launch_mix(const MixArgs &A, int64_t blocks, size_t lds, hipStream_t stream) {
    const bool dual = MODE != MOT_MIX_NOOP && (A.id_source == MOT_IDS_FROM_TTB ? A.add_padded != 0 : A.ids_b != nullptr);
    if (dual)
        hipLaunchKernelGGL((embed_mix_kernel<MODE, NCH, U, T, true>), dim3((unsigned)blocks), dim3(kThreads), lds, stream, A);
    else
        hipLaunchKernelGGL((embed_mix_kernel<MODE, NCH, U, T, false>), dim3((unsigned)blocks), dim3(kThreads), lds, stream, A);
    return check_launch("embed_mix_kernel");
}

// NCH = 16-byte chunks per lane: ceil(D / (64 * VEC)).  U keeps ~12 independent 16 B loads per lane in flight.
template <int MODE>
static int dispatch_nch(const MixArgs &A, int dtype, int64_t blocks, size_t lds, hipStream_t stream) {
    if (dtype == MOT_BF16) {
        switch ((A.Dt / 8 + 63) / 64) {
            case 1: return launch_mix<MODE, 1, 4, __bf16>(A, blocks, lds, stream);
            case 2: return launch_mix<MODE, 2, 4, __bf16>(A, blocks, lds, stream);
            case 3: return launch_mix<MODE, 3, 2, __bf16>(A, blocks, lds, stream);
            case 4: return launch_mix<MODE, 4, 2, __bf16>(A, blocks, lds, stream);
            default: return set_error(MOT_EUNSUPPORTED, "embed_mix: model_dim %d > 2048 is not built", A.Dt);
        }
    }
    switch ((A.Dt / 4 + 63) / 64) {
        case 1: return launch_mix<MODE, 1, 4, float>(A, blocks, lds, stream);
        case 2: return launch_mix<MODE, 2, 4, float>(A, blocks, lds, stream);
        case 3: return launch_mix<MODE, 3, 2, float>(A, blocks, lds, stream);
        case 4: return launch_mix<MODE, 4, 2, float>(A, blocks, lds, stream);
        case 5:
        case 6: return launch_mix<MODE, 6, 1, float>(A, blocks, lds, stream);
        case 7:
        case 8: return launch_mix<MODE, 8, 1, float>(A, blocks, lds, stream);
        default: return set_error(MOT_EUNSUPPORTED, "embed_mix: model_dim %d > 2048 is not built", A.Dt);
    }
}


// ------------------------------------------------------------------------------------------ MEAN with the character table in LDS
// x = s_t * E_tok[t] + s_c * mean_k E_char[c_k]   (inference/inference.py:266-267, 323-327; config 5: 128 k token vocab, 132
// characters, d 2048, 8 character slots).  With whole rows per token, the eight 8 KB character rows of every token come out
// of L2 -- 64 KB of L2 reads against 16 KB of HBM traffic, and the kernel ran at the L2 rate (35 % of the HBM roofline).
// The character table is small (132 x 2048) but does not fit LDS whole; a COLUMN SLICE of it does (132 rows x 256 fp32
// columns = 135 KB of the 160 KB).  So a workgroup owns one slice, keeps it in LDS for its whole life, and streams the same
// slice of the token rows and of the output: the character rows are read from HBM/L2 once per workgroup instead of eight
// times per token.  1024 threads (16 waves share the slice: one workgroup per CU by LDS), one wave per token, U tokens in flight.
// BPT: character slots per token as a compile-time constant (8: the reference's create_char_matrix, inference.py:290) or 0 = A.bpt at
// run time; NORMB: the per-character rms factor is applied.  With both known the inner loop is straight-line: per character one
// v_readlane of the row's byte offset (a constant lane), one v_add, one ds_read_b128, two v_pk_add_f32 -- round 2's loop over a
// run-time bpt with its scalar multiplies, branches and per-flag selects issued ~100 instructions per token and wave, and without
// its output stores the kernel still took 2.7 of its 6.0 ms: instruction issue, 16 waves per CU.
template <typename T, int U, int BPT, bool NORMB, int ADD = 0>   // ADD (T = float): 1 = MixArgs.add16 (bf16 rows), 2 = the fp32 rows of out: out += result
__global__ __launch_bounds__(1024) void embed_mean_lds_kernel(const MixArgs A, int slice_cols, int nslices, int64_t tokens_per_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *tab = (T *)lds_raw;                                   // [byte_rows][slice_cols]
    constexpr int VEC = Elem<T>::kVec;
    typedef typename Elem<T>::vec vec_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (The nslices workgroups of a token partition are consecutive block ids, i.e. on nslices DIFFERENT XCDs.  Placing them on one
    //  XCD, default-policy stores and non-temporal loads of the token rows were all measured in round 3 and all lost:
    //  profiles/r03_c5_variants.txt, built from commit 0312ce5 and its successors with -DMOT_C5_MAP / _STORE / _NTLOAD.)
    const int slice = blockIdx.x % nslices;
    const int64_t part = blockIdx.x / nslices;
    const int col0 = slice * slice_cols;
    const int D = A.Dt;
    const T *byte_table = (const T *)A.byte_table, *tok_table = (const T *)A.tok_table;
    const int pieces_per_row = slice_cols / VEC;              // 16-byte pieces
    for (int64_t q = tid; q < A.byte_rows * pieces_per_row; q += 1024) {
        const int r = (int)(q / pieces_per_row), c = (int)(q - (int64_t)r * pieces_per_row) * VEC;
        *(typename Elem<T>::raw *)(tab + (size_t)r * slice_cols + c) = Elem<T>::load_raw(byte_table + (int64_t)r * D + col0 + c);
    }
    __syncthreads();
    const float s_tok = A.scale_tok ? *A.scale_tok : 1.0f;
    const float s_byte = A.scale_byte ? *A.scale_byte : 1.0f;
    const float inv_bpt = 1.0f / (float)A.bpt;
    const int64_t n_all = A.T;                                // tokens are addressed flat: A.T = n_rows * tokens_per_row here
    const int64_t n0 = part * tokens_per_part, n1 = min(n_all, n0 + tokens_per_part);
    T *out = (T *)A.out;
    const int c = lane * VEC;                                 // one 16-byte chunk per lane: slice_cols == 64 * VEC (launcher)
    // A wave takes a contiguous stretch of the part's tokens.  Memory operations of a wave complete in issue order, stores
    // included: a batch whose loads were issued BEHIND the previous batch's stores (load, compute, store, load, ...) waited for
    // those stores to reach HBM before it could touch its own data, and the token-row address came from a second dependent
    // load -- 7 us per batch of four 1 KB row slices, 58 % of the roofline.  Now the token ids arrive 64 at a time, a lane each
    // (handed out with readlane: the row address is scalar), a batch's character ids are ONE 8-byte load per lane, and batch
    // b + 1 is requested before batch b is computed and stored, so that the wait for its data leaves the stores in flight.
    const int64_t per_wave = (((n1 - n0 + 15) / 16) + U - 1) / U * U;
    const int64_t w_lo = min(n1, n0 + wave * per_wave), w_hi = min(n1, w_lo + per_wave);
    if (w_lo >= w_hi) return;                                 // no barrier below
    const int id_lanes = U * A.bpt;                           // <= 64 (launcher)
    int tokv_nx = w_lo + lane < w_hi ? A.tokens[w_lo + lane] : 0;
    for (int64_t chunk = w_lo; chunk < w_hi; chunk += 64) {
        const int tokv = tokv_nx;
        if (chunk + 64 < w_hi) tokv_nx = chunk + 64 + lane < w_hi ? A.tokens[chunk + 64 + lane] : 0;
        const int nb = (int)min((int64_t)64, w_hi - chunk);
        typename Elem<T>::raw ar_nx[U];
        uint2 old_nx[U];   // ADD == 1: the bf16 values the batch's results are added to, requested with the batch's rows
        float4v old32_nx[U];   // ADD == 2: the fp32 values
        int64_t id_nx = 0;
        auto request = [&](int b) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int tok = __builtin_amdgcn_readlane(tokv, min(b + u, nb - 1));   // the tail re-reads the last token; its store is skipped
                if ((uint64_t)(uint32_t)tok >= (uint64_t)A.tok_rows) {
                    if (A.status && lane == 0) atomicOr(A.status, kStatusTokenOor);
                    tok = 0;
                }
                ar_nx[u] = Elem<T>::load_raw(tok_table + (int64_t)tok * D + col0 + c);
                if constexpr (ADD == 1) old_nx[u] = *(const uint2 *)(A.add16 + (chunk + min(b + u, nb - 1)) * D + col0 + c);
                if constexpr (ADD == 2) old32_nx[u] = *(const float4v *)(A.out + (chunk + min(b + u, nb - 1)) * D + col0 + c);
            }
            const int64_t at = (chunk + b) * A.bpt + lane;    // lane = (token of the batch, slot)
            id_nx = (lane < id_lanes && at < n_all * A.bpt) ? A.ids_a[at] : 0;
        };
        request(0);
        for (int b = 0; b < nb; b += U) {
            typename Elem<T>::raw ar[U];
            uint2 old[U];
            float4v old32[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ar[u] = ar_nx[u];
                if constexpr (ADD == 1) old[u] = old_nx[u];
                if constexpr (ADD == 2) old32[u] = old32_nx[u];
            }
            const int64_t idq = id_nx;
            if (b + U < nb) request(b + U);
            // range check of the batch's ids, where they are used (one compare per lane; a bad id is flagged and reads row 0); the
            // lane keeps the BYTE OFFSET of its character's row inside the LDS slice
            const bool bad = (uint64_t)idq >= (uint64_t)A.byte_rows;
            if (bad && A.status) atomicOr(A.status, kStatusByteOor);
            const int idv = bad ? 0 : (int)idq;
            const uint32_t row_bytes = (uint32_t)slice_cols * (uint32_t)sizeof(T);
            const uint32_t offv = (uint32_t)idv * row_bytes;
            const unsigned char *tabc = (const unsigned char *)tab + (size_t)c * sizeof(T);
            const int bpt = BPT ? BPT : A.bpt;
            // s_tok / s_byte are 1.0f when the scalars are absent (x * 1.0f is exact); the mean's 1 / bpt is folded into the character
            // scalar when it is a power of two (exact), as for the 8 slots of the reference
            const bool pow2 = (bpt & (bpt - 1)) == 0;
            const float cb = pow2 ? inv_bpt * s_byte : s_byte;
            T *orow = out + (chunk + b) * D + col0 + c;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // chars.mean(dim=-2), inference.py:267: independent LDS reads, two accumulators keep the adds off one chain.  A slot's
                // row offset is handed out with v_readlane (the slot's lane is a constant when BPT is)
                vec_t acc0 = (vec_t)(0.f), acc1 = (vec_t)(0.f);
                const int l0 = u * bpt;
                auto one = [&](int k, vec_t &acc) {
                    const uint32_t so = (uint32_t)__builtin_amdgcn_readlane((int)offv, l0 + k);
                    vec_t v = Elem<T>::widen(*(const typename Elem<T>::raw *)(tabc + so));
                    if (NORMB) v *= A.byte_rnorm[__builtin_amdgcn_readlane(idv, l0 + k)];
                    acc += v;
                };
                if constexpr (BPT > 0) {
#pragma unroll
                    for (int k = 0; k < BPT; ++k) one(k, (k & 1) ? acc1 : acc0);
                } else {
                    int k = 0;
                    for (; k + 2 <= bpt; k += 2) { one(k, acc0); one(k + 1, acc1); }
                    if (k < bpt) one(k, acc0);
                }
                vec_t acc = acc0 + acc1;
                if (!pow2) acc = acc * inv_bpt;               // within an ulp of acc / bpt for the other slot counts
                const vec_t x = Elem<T>::widen(ar[u]) * s_tok + acc * cb;
                if constexpr (ADD == 2) {   // (character mixer, fp32: h = wo y is there already; the product's own C += epilogue cost it 0.43 of 4.5 ms)
                    if (b + u < nb) Elem<float>::storev_nt((float *)orow + (int64_t)u * D, old32[u] + x);
                    continue;
                }
                if constexpr (ADD == 1) {   // (character mixer with a bf16 result: h += residuals on the bf16 output of wo, mot_swa.hip)
                    if (b + u < nb) {
                        const float4v h = {__uint_as_float(old[u].x << 16), __uint_as_float(old[u].x & 0xffff0000u), __uint_as_float(old[u].y << 16),
                                           __uint_as_float(old[u].y & 0xffff0000u)};
                        Elem<__bf16>::store4_nt(A.add16 + (chunk + b + u) * D + col0 + c, h + x);
                    }
                    continue;
                }
#ifdef MOT_C5_NOSTORE   // dev, timing only: everything but the output stores (tools/variants.sh c5_nostore:"-DMOT_C5_NOSTORE")
                if (b + u < nb && x[0] == 123.456f)
#else
                if (b + u < nb)
#endif
                    Elem<T>::storev_nt(orow + (int64_t)u * D, x);
            }
        }
    }
}

template <typename T>
static int launch_mean_lds(MixArgs A, const MotEmbedMixDesc &d, int slice_cols, hipStream_t stream) {
    const int nslices = d.tok_dim / slice_cols;
    const int64_t N = d.n_rows * d.tokens_per_row;
    int64_t parts = 256 / nslices;                            // one persistent workgroup per CU
    if (parts < 1) parts = 1;
    const int64_t per = ((N + parts - 1) / parts + 63) & ~(int64_t)63;
    parts = (N + per - 1) / per;
    A.T = N;                                                  // flat token addressing (rows are independent without a pull)
    const size_t lds = (size_t)d.byte_rows * slice_cols * sizeof(T);
    static std::atomic<uint64_t> lds_ok{0};   // per-device bits
    constexpr int kMeanU = 4;   // token-row slices in flight per wave (8 spilled before the loop was pipelined; no effect since)
#define MOT_MEAN_LAUNCH(BPT, NORMB)                                                                                                          \
    do {                                                                                                                                    \
        static std::atomic<uint64_t> ok_{0};   /* per-device bits */                                                                        \
        if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mean_lds_kernel<T, kMeanU, BPT, NORMB>, ok_, "embed_mean_lds_kernel")) return rc_lds; \
        hipLaunchKernelGGL((embed_mean_lds_kernel<T, kMeanU, BPT, NORMB>), dim3((unsigned)(parts * nslices)), dim3(1024), lds, stream, A, slice_cols, \
                           nslices, per);                                                                                                   \
    } while (0)
    (void)lds_ok;
    if constexpr (sizeof(T) == 4) {
        if (A.add16) {   // (the read-modify-write form on bf16 rows, see MixArgs)
            static std::atomic<uint64_t> ok_{0};
            if (d.norm_byte) return set_error(MOT_EUNSUPPORTED, "embed_mix: add16 without norm_byte");
            if (d.bpt == 8) {
                if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mean_lds_kernel<T, kMeanU, 8, false, 1>, ok_, "embed_mean_lds_kernel")) return rc_lds;
                hipLaunchKernelGGL((embed_mean_lds_kernel<T, kMeanU, 8, false, 1>), dim3((unsigned)(parts * nslices)), dim3(1024), lds, stream, A, slice_cols, nslices, per);
            } else {
                static std::atomic<uint64_t> ok0_{0};
                if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mean_lds_kernel<T, kMeanU, 0, false, 1>, ok0_, "embed_mean_lds_kernel")) return rc_lds;
                hipLaunchKernelGGL((embed_mean_lds_kernel<T, kMeanU, 0, false, 1>), dim3((unsigned)(parts * nslices)), dim3(1024), lds, stream, A, slice_cols, nslices, per);
            }
            return check_launch("embed_mean_lds_kernel");
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (A.add_out) {   // out += result (see MixArgs)
            if (d.norm_byte) return set_error(MOT_EUNSUPPORTED, "embed_mix: out += without norm_byte");
            if (d.bpt == 8) {
                static std::atomic<uint64_t> ok_{0};
                if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mean_lds_kernel<T, kMeanU, 8, false, 2>, ok_, "embed_mean_lds_kernel")) return rc_lds;
                hipLaunchKernelGGL((embed_mean_lds_kernel<T, kMeanU, 8, false, 2>), dim3((unsigned)(parts * nslices)), dim3(1024), lds, stream, A, slice_cols, nslices, per);
            } else {
                static std::atomic<uint64_t> ok0_{0};
                if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mean_lds_kernel<T, kMeanU, 0, false, 2>, ok0_, "embed_mean_lds_kernel")) return rc_lds;
                hipLaunchKernelGGL((embed_mean_lds_kernel<T, kMeanU, 0, false, 2>), dim3((unsigned)(parts * nslices)), dim3(1024), lds, stream, A, slice_cols, nslices, per);
            }
            return check_launch("embed_mean_lds_kernel");
        }
    }
    if (d.bpt == 8) { if (d.norm_byte) MOT_MEAN_LAUNCH(8, true); else MOT_MEAN_LAUNCH(8, false); }
    else { if (d.norm_byte) MOT_MEAN_LAUNCH(0, true); else MOT_MEAN_LAUNCH(0, false); }
#undef MOT_MEAN_LAUNCH
    return check_launch("embed_mean_lds_kernel");
}

// slice width (columns) for the LDS-table MEAN kernel, or 0 when the shape does not qualify
static int mean_lds_slice(const MotEmbedMixDesc &d) {
    if (d.mode != MOT_MIX_MEAN || d.id_source != MOT_IDS_GIVEN || d.ids_b || d.norm_tok || d.norm_out || d.counters || d.out_ids_padded ||
        d.out_ids_pulled || (d.flags & MOT_FLAG_MEAN_GENERIC))
        return 0;
    const int esize = d.dtype == MOT_BF16 ? 2 : 4, chunk = 64 * (16 / esize);   // columns one wave covers with 16-byte lanes
    if (d.tok_dim % chunk || d.n_rows * d.tokens_per_row < 16384 || d.bpt * 4 > 64) return 0;   // (a batch's ids: 4 tokens x bpt lanes)
    const size_t budget = 144 * 1024;
    if ((size_t)d.byte_rows * chunk * esize > budget) return 0;
    return chunk;   // one chunk per slice: the most slices, the smallest table image
}

// Tokens per wave.  16 below 131 072 tokens: the shard a GPU gets when config 4 is split over 8 GPUs is 65 536 tokens, and 16 puts
// 16 waves on every CU there (measured 58.8 us = 86 % of the roofline, 60.4 us with 32); 32 from there on: the index pass over
// the 64-token window is then shared by twice the tokens (434.6 us at 524 288 tokens against 438.3).
static int pick_unit(int64_t n_tokens) { return n_tokens >= 131072 ? 32 : 16; }

bool embed_mix_mean_takes_add16(const MotEmbedMixDesc &d) { return d.dtype == MOT_F32 && mean_lds_slice(d) != 0; }

int launch_embed_mix(const MotEmbedMixDesc &d, hipStream_t stream, __bf16 *add16, bool add_out) {
    MixArgs A;
    fill_mix_args(A, d);
    A.add16 = add16;
    A.add_out = add_out ? 1 : 0;
    if ((add16 || add_out) && (d.dtype != MOT_F32 || !mean_lds_slice(d)))
        return set_error(MOT_EUNSUPPORTED, "embed_mix: add16 / out += are modes of the LDS-table MEAN kernel with fp32 tables");

    const bool has_ids = d.mode != MOT_MIX_NOOP;
    const bool dual = has_ids && (d.id_source == MOT_IDS_FROM_TTB ? d.add_padded != 0 : d.ids_b != nullptr);
    A.unit = pick_unit(d.n_rows * d.tokens_per_row);
#ifdef MOT_DEV_ABLATION
    if (getenv("MOT_UNIT") && atoi(getenv("MOT_UNIT")) > 0) A.unit = atoi(getenv("MOT_UNIT"));
#endif
    A.units_per_row = (d.tokens_per_row + A.unit - 1) / A.unit;
    A.n_units = d.n_rows * A.units_per_row;
    const int64_t blocks = (A.n_units + kWaves - 1) / kWaves;
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "embed_mix: too many units");
    const int stream_eb = (has_ids && d.id_source == MOT_IDS_FROM_TTB && d.pull_dir != MOT_PULL_NONE) ? d.ttb_elem_bytes : 0;
    A.wave_lds = (int)wave_lds_bytes(A.unit, has_ids ? d.bpt : 0, dual, stream_eb);
    const size_t lds = (size_t)A.wave_lds * kWaves;

    if (d.mode != MOT_MIX_NOOP && d.norm_byte) {
        const size_t need = (size_t)d.byte_rows * sizeof(float);
        if (!d.workspace || d.workspace_bytes < need)
            return set_error(MOT_EWORKSPACE, "embed_mix: norm_byte needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
        float *rn = (float *)d.workspace;
        int rc = launch_rows_rnorm(d.byte_table, d.byte_rows, d.byte_dim, A.eps, rn, d.dtype, stream);
        if (rc) return rc;
        A.byte_rnorm = rn;
    }
    if (const int sc = mean_lds_slice(d)) {
        return d.dtype == MOT_BF16 ? launch_mean_lds<__bf16>(A, d, sc, stream) : launch_mean_lds<float>(A, d, sc, stream);
    }
    switch (d.mode) {
        case MOT_MIX_NOOP: return dispatch_nch<MOT_MIX_NOOP>(A, d.dtype, blocks, lds, stream);
        case MOT_MIX_SUM: return dispatch_nch<MOT_MIX_SUM>(A, d.dtype, blocks, lds, stream);
        case MOT_MIX_MEAN: return dispatch_nch<MOT_MIX_MEAN>(A, d.dtype, blocks, lds, stream);
        default: return set_error(MOT_EINVAL, "embed_mix: bad mode %d", d.mode);
    }
}

}  // namespace mot
