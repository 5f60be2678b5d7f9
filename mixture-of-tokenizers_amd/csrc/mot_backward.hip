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
//     (bwd_rank_kernel -> bwd_scan_kernel -> bwd_place_kernel, three small kernels); each wave then
//     walks a contiguous stretch of the SORTED positions, keeps the running gradient row of the current
//     token in registers and issues one atomic row-add per run of equal tokens.  Lane l owns elements
//     l, l+64, ... of the row, so every atomic wave-instruction covers 256 contiguous bytes (the shape
//     that runs at the chip-wide atomic rate); float4-per-lane would spread each instruction over 1 KiB.
//   * byte table (458 rows hit 8.4 M times per step): privatised in LDS per workgroup as 64-bit fixed
//     point (ds_add_u64; ds_add_f32 issues one lane at a time on gfx950), flushed once with contiguous
//     global float atomics.
// 512-thread workgroups, one per CU (the LDS copy of the byte-table gradient takes up to 150 KB).
// Byte ids are taken as given (the forward returns them), so no tile machinery here.
// Float atomics make the sums order-dependent in the last bits, like the reference's own GPU
// embedding backward; the parity tests state the tolerance they use against a float64 evaluation.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "mot_mix.hpp"

namespace mot {

// Ordering of one wave's traffic on its per-slot LDS accumulators (seg / segr / seg_q): a plain zeroing store by every lane,
// LDS atomics from other lanes, a plain read-back.  DS instructions of one wave execute in issue order; what has to be pinned is
// the COMPILER's order across lanes: a release fence, a wave barrier (lanes are separate threads to the memory model: the
// barrier is what orders lane A's store before lane B's atomic) and an acquire fence.  No instruction beyond the waits the
// fences imply.
__device__ __forceinline__ void seg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

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
    // layout of one gradient row of D elements: token part [tok_lo, tok_lo+Dt), byte part [byte_lo, byte_lo+bpt*Db).
    // SUM: both parts span the whole row (x = a + concat b); CONCAT_LINEAR: they are the two halves of du = dy.W
    int Dt, tok_lo, byte_lo, nbk;
    // byte-table gradient privatised in LDS as 64-bit fixed point: rows [0, priv_lo) and [priv_hi0, byte_rows) have a slot
    // (everything when the table fits; otherwise the raw byte values and the trailing specials such as pad / eot)
    int priv_lo, priv_hi0, priv_rows;
    const int32_t *pos_sorted;  // token positions ordered by token id
    const int32_t *tok_sorted;  // their (clamped) token ids
    int in_bf16;  // tables and grad_out are bf16 (gradients are accumulated and returned in fp32 either way)
    // lane-contiguous kernel only (the two halves of the CONCAT_LINEAR scatter): elements between gradient rows when they are columns of
    // a wider matrix (0: D), and "no token table" (SUM over byte slots only: nothing is read from or added to a token table)
    int g_ld, no_tok;
    int slot0;    // first byte slot of this pass (its ids are ids[n * bpt + slot0 + ...]): the byte part taken in column blocks
    int abl;  // dev-only timing ablations (MOT_DEV_ABLATION builds): 1 no LDS byte adds, 2 no token-row flush, 4 no wave sums
};

// element i of a float or bf16 array (uniform choice per launch)
__device__ __forceinline__ float ld_in(const float *base, int64_t i, int bf16) {
    return bf16 ? (float)((const __bf16 *)base)[i] : base[i];
}

template <int MODE, int NE>
__global__ __launch_bounds__(kBwdThreads) void embed_mix_bwd_kernel(const BwdArgs A) {
    extern __shared__ unsigned long long lds_q[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbyte = A.priv_rows * A.Db;
    unsigned long long *dbyte_q = lds_q;              // [priv_rows*Db] fixed-point sums (two's complement)
    float *lds_f = (float *)(lds_q + nbyte);
    float *seg = lds_f + wave * kMaxBpt;              // per-wave per-slot dot products
    float *segr = lds_f + (kBwdWaves + wave) * kMaxBpt;  // per-wave per-slot rms factor (two-id-tensor norm)
    uint32_t *fx_bits = (uint32_t *)(lds_f + 2 * kBwdWaves * kMaxBpt);  // max |v| over the waves' first tokens (float bits)
    for (int i = tid; i < nbyte; i += kBwdThreads) dbyte_q[i] = 0ull;
    if (tid == 0) *fx_bits = 0u;
    __syncthreads();
    // LDS float atomics run one lane at a time on gfx950 (ds_add_f32: ~190 cycles per wave-instruction, ds_add_u64: ~20,
    // tools/ubench/lds_atomic.hip), so the privatised sums are 64-bit fixed point: v * 2^fx_k, fx_k chosen per workgroup so
    // that the largest |v| of the waves' first tokens lands at 2^28.  Terms outside [2^12, 2^40) after scaling (too coarse /
    // too close to the 2^62 budget of <= 2^22 adds) and rows without a slot take the exact global float atomic instead,
    // so the result never depends on the choice of scale -- only the speed does.
    int fx_k = 0;
    bool fx_pending = MODE != MOT_MIX_NOOP;
    auto byte_slot = [&](int id) { return id < A.priv_lo ? id : (id >= A.priv_hi0 ? id - A.priv_hi0 + A.priv_lo : -1); };
    auto add_byte = [&](int id, int wi, float v) {
        const int sl = byte_slot(id);
        const float x = rintf(ldexpf(v, fx_k)), ax = fabsf(x);
        if (sl >= 0 && ax < 0x1p40f && (ax >= 0x1p12f || v == 0.f)) {
            if (v != 0.f) atomicAdd(dbyte_q + sl * A.Db + wi, (unsigned long long)(long long)x);
        } else {
            atomicAdd(A.d_byte + id * A.Db + wi, v);   // exact path (also carries inf / nan through)
        }
    };

    const int D = A.D;
    const float inv_db = MODE != MOT_MIX_NOOP ? 1.0f / (float)A.Db : 0.f;
    // element e = lane + 64*j of a row lives in byte slot e / Db; (e + 0.5) * (1/Db) floors exactly for e < 2048
    auto slot_of = [&](int eb) { return MODE != MOT_MIX_NOOP ? __float2int_rd(((float)eb + 0.5f) * inv_db) : 0; };
    const int Dt = A.Dt;
    auto tok_off = [&](int e) { const int o = e - A.tok_lo; return (e < D && (unsigned)o < (unsigned)Dt) ? o : -1; };
    auto byte_off = [&](int e) { const int o = e - A.byte_lo; return (MODE != MOT_MIX_NOOP && e < D && (unsigned)o < (unsigned)A.nbk) ? o : -1; };
    // norm over the sum of two embeddings only exists in front of the concat mixin (train_gpt.py:378, 443)
    const bool pair_norm = MODE == MOT_MIX_CONCAT_LINEAR && A.norm_byte && A.ids_b;
    const float s_tok = A.scale_tok ? *A.scale_tok : 1.0f;
    const float s_byte = A.scale_byte ? *A.scale_byte : 1.0f;
    float ds_t = 0.f, ds_b = 0.f;

    float acc[NE];  // running d(token row) of the current run of equal tokens
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0 || (A.abl & 2)) return;
        float *drow = A.d_tok + (int64_t)cur * Dt;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int o = tok_off(lane + 64 * j);
            if (o >= 0) atomicAdd(drow + o, acc[j]);
        }
    };
    // Each position needs position -> token -> rows and position -> byte ids -> byte rows: up to four
    // dependent round trips.  The index side (position, token, byte ids) of position i+1 is therefore
    // fetched while position i's rows are in flight, leaving one row-fetch latency per position.
    auto load_ids = [&](int64_t n, int (&ids)[NE]) {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            ids[j] = 0;
            const int eb = byte_off(lane + 64 * j);
            if (eb >= 0) {
                int64_t ia = A.ids_a[n * A.bpt + slot_of(eb)];
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
        const int64_t trow = (int64_t)tok * Dt, grow = n * (int64_t)(A.g_ld ? A.g_ld : D);   // (g_ld: gradient rows that are columns of a wider matrix)
        float an[NE], bn[NE], dy[NE];
        // ---- gather (the same rows the forward read)
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int e = lane + 64 * j;
            const int et = tok_off(e), eb = byte_off(e);
            an[j] = et >= 0 ? ld_in(A.tok_table, trow + et, A.in_bf16) : 0.f;
            dy[j] = e < D ? ld_in(A.grad_out, grow + e, A.in_bf16) : 0.f;  // holds g until the norm backward below
            bn[j] = 0.f;
            if (eb >= 0) {
                const int sl = slot_of(eb), wi = eb - sl * A.Db;
                float v = ld_in(A.byte_table, (int64_t)id1[j] * A.Db + wi, A.in_bf16);
                if (A.ids_b) {
                    int64_t ib = A.ids_b[n * A.bpt + sl];
                    if ((uint64_t)ib >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); ib = 0; }
                    v += ld_in(A.byte_table, ib * A.Db + wi, A.in_bf16);
                }
                if (A.norm_byte && !pair_norm) v *= A.byte_rnorm[id1[j]];
                bn[j] = v;  // normalised, unscaled
            }
        }
        if (pair_norm) {
            // norm(emb(padded) + emb(pulled)) (train_gpt.py:378): the rms factor belongs to the (token, slot) pair,
            // not to a table row -- reduce sum(b^2) per slot through the wave's LDS accumulators
            segr[lane] = 0.f;   // all 64 lanes (kMaxBpt entries): a `lane < bpt` guard lets the compiler reorder the adds around it
            seg_sync();
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int eb = byte_off(lane + 64 * j);
                if (eb >= 0) atomicAdd(&segr[slot_of(eb)], bn[j] * bn[j]);
            }
            seg_sync();
            segr[lane] = rms_scale(segr[lane], A.Db, A.eps);
            seg_sync();
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int eb = byte_off(lane + 64 * j);
                if (eb >= 0) bn[j] *= segr[slot_of(eb)];
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
            ra = rms_scale(wave_sum(ss), Dt, A.eps);
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
            if (A.norm_tok) mt = wave_sum(dot * s_tok) / (float)Dt;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const float da = dy[j] * s_tok;
                acc[j] += A.norm_tok ? ra * (da - an[j] * mt) : da;
            }
        }
        // ---- byte side
        if (MODE != MOT_MIX_NOOP) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) dot += dy[j] * bn[j];
            ds_b += dot;
            if (A.norm_byte) {  // per-slot mean(db * b_n): slots are ragged lane groups -> LDS accumulators
                seg[lane] = 0.f;
                seg_sync();
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int eb = byte_off(lane + 64 * j);
                    if (eb >= 0) atomicAdd(&seg[slot_of(eb)], dy[j] * s_byte * bn[j]);
                }
                seg_sync();
            }
            float vmax = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) {   // dy[j] <- gradient w.r.t. the (un-normalised) byte-table element
                const int eb = byte_off(lane + 64 * j);
                if (eb < 0) continue;
                const int sl = slot_of(eb);
                const float db = dy[j] * s_byte;
                float v = db;
                if (A.norm_byte) v = (pair_norm ? segr[sl] : A.byte_rnorm[id1[j]]) * (db - bn[j] * (seg[sl] / (float)A.Db));
                dy[j] = v;
                vmax = fmaxf(vmax, fabsf(v));
            }
            if (fx_pending) {   // once per wave: agree on the workgroup's fixed-point scale
                vmax = wave_max(vmax);
                if (lane == 0 && vmax > 0.f && vmax < INFINITY) atomicMax(fx_bits, __float_as_uint(vmax));
                __syncthreads();
                const float m = __uint_as_float(*fx_bits);
                fx_k = m > 0.f ? 28 - ilogbf(m) : 0;
                fx_pending = false;
            }
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int eb = byte_off(lane + 64 * j);
                if (eb < 0 || (A.abl & 1)) continue;
                const int sl = slot_of(eb), wi = eb - sl * A.Db;
                add_byte(id1[j], wi, dy[j]);
                if (A.ids_b) {
                    int64_t ib = A.ids_b[n * A.bpt + sl];
                    if ((uint64_t)ib >= (uint64_t)A.byte_rows) ib = 0;
                    add_byte((int)ib, wi, dy[j]);
                }
            }
            if (A.norm_byte) seg_sync();  // seg is rewritten by the next token
        }
    }
    }
    // ---- flush
    flush();
    if (A.d_scale_tok) { ds_t = wave_sum(ds_t); if (lane == 0) atomicAdd(A.d_scale_tok, ds_t); }
    if (MODE != MOT_MIX_NOOP && A.d_scale_byte) { ds_b = wave_sum(ds_b); if (lane == 0) atomicAdd(A.d_scale_byte, ds_b); }
    if (MODE != MOT_MIX_NOOP) {
        if (fx_pending) {   // a wave without work still meets the others at the scale barrier
            __syncthreads();
            const float m = __uint_as_float(*fx_bits);
            fx_k = m > 0.f ? 28 - ilogbf(m) : 0;
        }
        __syncthreads();
        for (int i = tid; i < nbyte; i += kBwdThreads) {
            const long long q = (long long)dbyte_q[i];
            if (q == 0) continue;
            const int sl = i / A.Db, wi = i - sl * A.Db;
            const int row = sl < A.priv_lo ? sl : sl - A.priv_lo + A.priv_hi0;
            atomicAdd(A.d_byte + row * A.Db + wi, (float)ldexp((double)q, -fx_k));
        }
    }
}

// ------------------------------------------------------------------------------------------
// Lean variant for the layouts where a row is "full": SUM / NOOP with D == 64 * NE (the headline shapes: 768, 1024,
// 2048, 256 ...) and the CONCAT_LINEAR scatter whose row splits into a token part and a byte part on 64-element
// boundaries.  Same algorithm as embed_mix_bwd_kernel, minus everything the general layout needs per element: the slot
// and within-slot index of a lane's elements are computed once, position and token id are wave-uniform scalars (rows
// are addressed as SGPR base + immediates), a token's byte ids are loaded once by lanes < bpt and handed out with
// ds_bpermute, the token row is read once per run of equal tokens, and the byte-table adds are straight-line 64-bit
// fixed-point LDS atomics.  ~5x fewer instructions per position than the general kernel.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long to_fixed(float v, int k) {
    // round(v * 2^k) as two's complement in 64 bits, |v * 2^k| < 2^51: add 1.5 * 2^52 and read the mantissa
    const double d = ldexp((double)v, k) + 6755399441055744.0;
    return (unsigned long long)(__double_as_longlong(d) - 0x4338000000000000ll);
}


template <int MODE, int NE, bool BF>   // BF: tables and grad_out are bf16 (SUM / NOOP; the CONCAT_LINEAR path widens its operands first)
__global__ __launch_bounds__(kBwdThreads) void embed_mix_bwd_full_kernel(const BwdArgs A) {
    constexpr int D = 64 * NE;
    extern __shared__ unsigned long long lds_q[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler too: scalar loop control below
    const int nbyte = A.priv_rows * A.Db;
    unsigned long long *dbyte_q = lds_q;
    unsigned long long *seg_q = lds_q + nbyte + wave * kMaxBpt;                      // per-wave per-slot sums (byte-norm backward)
    uint32_t *fx_bits = (uint32_t *)(lds_q + nbyte + kBwdWaves * kMaxBpt);
    for (int i = tid; i < nbyte; i += kBwdThreads) dbyte_q[i] = 0ull;
    if (tid == 0) *fx_bits = 0u;
    // SPLIT: the gradient row is du of CONCAT_LINEAR -- [token part | byte part] (or the reverse), every 64-element
    // chunk j belonging wholly to one part (host-checked); SUM / NOOP rows are token part and byte part at once.
    constexpr bool SPLIT = MODE == MOT_MIX_CONCAT_LINEAR;
    constexpr bool BYTES = MODE != MOT_MIX_NOOP;
    uint32_t tmask = 0, bmask = 0;   // bit j: chunk j carries token-row / byte-row elements
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e0 = 64 * j;
        if (!SPLIT || (e0 >= A.tok_lo && e0 < A.tok_lo + A.Dt)) tmask |= 1u << j;
        if (MODE == MOT_MIX_SUM || (SPLIT && e0 >= A.byte_lo && e0 < A.byte_lo + A.nbk)) bmask |= 1u << j;
    }
    const bool dual = BYTES && A.ids_b != nullptr;
    const bool pair_norm = SPLIT && dual && A.norm_byte;   // norm(emb(padded) + emb(pulled)), train_gpt.py:378
    const float s_tok = A.scale_tok ? *A.scale_tok : 1.0f;
    const float s_byte = A.scale_byte ? *A.scale_byte : 1.0f;
    float ds_t = 0.f, ds_b = 0.f;
    // element lane + 64 j of a row: ds_bpermute address of its byte slot's lane (slot * 4) and byte offset of the element
    // within a byte-table row, for 4-byte elements (x2 in the 64-bit LDS copy, /2 for bf16 tables)
    // Register slot j of a lane holds row element lane + 64 j (fp32: one dword per lane per load), or, for bf16 rows,
    // 2 lane + 128 (j / 2) + (j % 2): a lane loads PAIRS of adjacent bf16 as one dword (2-byte lane loads run at a
    // fraction of the dword rate), and a pair shares its byte slot (Db is even, host-checked), so one ds_bpermute and one
    // dword gather serve both.
    auto elem = [&](int j) { return BF ? 2 * lane + 128 * (j >> 1) + (j & 1) : lane + 64 * j; };
    int sl4[NE], wi4[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int eb = max(elem(j) - A.byte_lo, 0), sl = BYTES ? eb / A.Db : 0;
        sl4[j] = sl * 4;
        wi4[j] = (eb - sl * A.Db) * 4;
    }
    constexpr uint32_t esz = BF ? 2u : 4u;
    const uint32_t lane4 = (uint32_t)lane * 4u;   // byte offset of a lane's first dword in a row (one fp32 / two bf16 elements)
    const uint32_t row8 = (uint32_t)A.Db * 8u;
    // Rows are addressed as (wave-uniform base pointer) + (32-bit byte offset): the position and its token id are made
    // scalar with readfirstlane, so a row's 64-bit address lives in SGPRs and the per-element offsets are immediates.
    // A row is first requested RAW (kRaw dwords per lane: one fp32 element or two bf16 each) and turned into floats later,
    // after everything else the position needs has been requested too -- unpacking bf16 pairs right at the load makes the
    // compiler wait for the row before it issues the byte-row gathers.
    constexpr int kRaw = BF ? (NE + 1) / 2 : NE;   // (bf16 kernels are launched with even NE only)
    auto load_raw = [&](const char *rbase, uint32_t (&raw)[kRaw], uint32_t mask) {   // mask: 64-element chunks present (fp32 split rows)
#pragma unroll
        for (int g = 0; g < kRaw; ++g) raw[g] = (BF || (mask >> g & 1)) ? *(const uint32_t *)(rbase + (lane4 + 256u * g)) : 0u;
    };
    auto unpack = [&](const uint32_t (&raw)[kRaw], float (&dst)[NE]) {
#pragma unroll
        for (int j = 0; j < NE; ++j)
            dst[j] = __uint_as_float(BF ? ((j & 1) ? raw[j >> 1] & 0xffff0000u : raw[j >> 1] << 16) : raw[BF ? 0 : j]);
    };
    // fixed-point scale of the privatised byte-table sums: chosen per workgroup from the upstream gradient rows of the
    // waves' first positions, v * 2^fx_k with the sample's max |g * scale_byte| at 2^27, so that a term converts through
    // a 32-bit integer.  Terms outside [2^12, 2^31) after scaling -- up to 16x the sample's maximum, down to 2^-15 of
    // it -- non-finite terms and rows without a slot take the exact global float atomic in a separate, rarely entered
    // block; the common path is straight-line (a term that must not be added there is replaced by zero).
    int fx_k = 0;
    uint32_t fx_lo_bits = 0, fx_span = 0;
    bool fx_known = !BYTES;
    auto byte_slot = [&](int id) { return id < A.priv_lo ? id : (id >= A.priv_hi0 ? id - A.priv_hi0 + A.priv_lo : -1); };
    auto in_range = [&](float v) { return ((__float_as_uint(v) & 0x7fffffffu) - fx_lo_bits) < fx_span; };
    // Within its LDS row, element wi sits at wi (fp32) or, for bf16, at wi / 2 + (wi odd ? Db / 2 : 0): a lane then holds the
    // elements 2 lane, 2 lane + 1, and with the even and the odd ones stored apart one ds_add_u64 touches consecutive
    // 8-byte words (stride-16 addresses would double the bank conflicts).  The flush at the end undoes the permutation.
    auto add_fixed = [&](int lrow, int j, float v) {   // lrow: byte offset of the row in dbyte_q (>= 0)
        const int q = __float2int_rn(ldexpf(v, fx_k));
        const int within = BF ? ((j & 1) ? wi4[j - 1] + (int)(row8 >> 1) : wi4[j]) : 2 * wi4[j];
        atomicAdd((unsigned long long *)((char *)dbyte_q + (uint32_t)(lrow + within)), (unsigned long long)(long long)q);
    };

    float acc[NE], an[NE];   // the current run: gradient of the token's table row so far, its (normalised) table row
    float ra = 1.f;          // 1 / rms of that row
    const float inv_dt = 1.0f / (float)A.Dt;
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0 || (A.abl & 2)) return;
        char *drow = (char *)(A.d_tok + ((int64_t)cur * A.Dt - A.tok_lo));
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (tmask >> j & 1) atomicAdd((float *)(drow + (BF ? 2u * lane4 + 512u * (j >> 1) + 4u * (j & 1) : lane4 + 256u * j)), acc[j]);
    };
    auto load_id = [&](const int64_t *ids, int64_t n) {   // lanes < bpt: the token's byte ids, clamped once
        int64_t v = 0;
        if (lane < A.bpt) {
            v = *(const int64_t *)((const char *)(ids + n * A.bpt) + (uint32_t)lane * 8u);
            if ((uint64_t)v >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); v = 0; }
        }
        return (int)v;
    };
    // ---- positions come grouped by token id (A.pos_sorted / A.tok_sorted, the counting sort above): a workgroup takes a
    // contiguous share of the sorted order and a wave a contiguous eighth of that, so a token's positions form a run
    // inside a wave -- its table row is read once and its gradient row leaves with one atomic row-add per run (plus one
    // where a run crosses a wave boundary).  A wave reads 64 (position, token) pairs at a time and hands them out with
    // readlane, keeping the loop control scalar.
    const int64_t per_wg = (A.n_tokens + gridDim.x - 1) / gridDim.x;
    const int64_t wg_lo = min(A.n_tokens, (int64_t)blockIdx.x * per_wg), wg_hi = min(A.n_tokens, wg_lo + per_wg);
    const int64_t per_wave = (wg_hi - wg_lo + kBwdWaves - 1) / kBwdWaves;
    const int64_t s_begin = min(wg_hi, wg_lo + wave * per_wave), s_end = min(wg_hi, s_begin + per_wave);
    __syncthreads();   // orders the table zeroing above
    if (!fx_known) {   // uniform: every wave takes part, once per workgroup
        float gmax = 0.f;
        if (s_begin < s_end) {
            const char *grow = (const char *)A.grad_out + (int64_t)A.pos_sorted[s_begin] * D * (int64_t)esz;
            uint32_t gr[kRaw];
            float gs[NE];
            load_raw(grow, gr, 0xffffffffu);
            unpack(gr, gs);
#pragma unroll
            for (int j = 0; j < NE; ++j) gmax = fmaxf(gmax, fabsf(gs[j]));
            gmax = wave_max(gmax) * fabsf(s_byte);
            if (lane == 0 && gmax > 0.f && gmax < INFINITY) atomicMax(fx_bits, __float_as_uint(gmax));
        }
        __syncthreads();
        const float m = __uint_as_float(*fx_bits);
        fx_k = m > 0.f ? min(27 - ilogbf(m), 100) : 0;
        fx_lo_bits = __float_as_uint(ldexpf(1.0f, 12 - fx_k));
        fx_span = __float_as_uint(ldexpf(1.0f, 31 - fx_k)) - fx_lo_bits;
        fx_known = true;
    }
    for (int64_t s0 = s_begin; s0 < s_end; s0 += 64) {
        const int cnt = (int)min((int64_t)64, s_end - s0);
        int vpos = 0, vtok = 0;
        if (lane < cnt) { vpos = A.pos_sorted[s0 + lane]; vtok = A.tok_sorted[s0 + lane]; }
        int n_nx = __builtin_amdgcn_readfirstlane(vpos);
        int ida_nx = 0, idb_nx = 0;
        if (BYTES) {
            ida_nx = load_id(A.ids_a, n_nx);
            if (dual) idb_nx = load_id(A.ids_b, n_nx);
        }
        for (int k = 0; k < cnt; ++k) {
            const int64_t n = n_nx;
            const int tok = __builtin_amdgcn_readlane(vtok, k), ida = ida_nx, idb = idb_nx;
            const bool more = k + 1 < cnt;
            if (more) n_nx = __builtin_amdgcn_readlane(vpos, k + 1);

            float bn[NE], dy[NE];
            int lra[NE], lrb[NE];   // byte offset of the element's row in the LDS copy (negative: no slot), first / second id tensor
            const char *grow = (const char *)A.grad_out + n * D * (int64_t)esz;
            // The two element types get their own copy of the request / gather part: the bf16 one keeps raw dwords until
            // everything is in flight; folding both into one body cost the fp32 kernel 5 % (scheduling, not instruction count).
            if constexpr (BF) {
                uint32_t graw[kRaw], traw[kRaw], braw[kRaw], braw2[kRaw];
                load_raw(grow, graw, 0xffffffffu);
                if constexpr (!BF) unpack(graw, dy);
                const bool newrun = tok != cur;   // the previous token's gradient row leaves, this token's row comes in
                auto begin_run = [&]() {   // raw token row -> floats, its gradient accumulator, its rms factor
                    unpack(traw, an);
#pragma unroll
                    for (int j = 0; j < NE; ++j) acc[j] = 0.f;
                    ra = 1.f;
                    if (A.norm_tok) {
                        float ss = 0.f;
#pragma unroll
                        for (int j = 0; j < NE; ++j) ss += an[j] * an[j];
                        ra = rms_scale(wave_sum(ss), A.Dt, A.eps);
#pragma unroll
                        for (int j = 0; j < NE; ++j) an[j] *= ra;
                    }
                };
                if (newrun) {
                    flush();
                    cur = tok;
                    load_raw((const char *)A.tok_table + ((int64_t)tok * A.Dt - A.tok_lo) * (int64_t)esz, traw, tmask);
                    if constexpr (!BF) begin_run();   // fp32: nothing to unpack, the compiler places the waits at the first use
                }
                if (BYTES) {
                    // lanes < bpt: gather offset of the slot's byte row and offset of its LDS row (or -1)
                    const int ga = ida * A.Db * (int)esz, gb = idb * A.Db * (int)esz;
                    const int sa = byte_slot(ida), sb = byte_slot(idb);
                    const int la = sa >= 0 ? sa * (int)row8 : -1, lb = sb >= 0 ? sb * (int)row8 : -1;
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        if (BF && (j & 1)) continue;   // a bf16 pair (j, j + 1) lies in one byte row: one permute, one dword gather
                        lra[j] = 0; lrb[j] = 0;
                        if constexpr (BF) {
                            const int g = j >> 1;
                            braw[g] = 0u; braw2[g] = 0u;
                            if (bmask >> j & 1) {
                                const uint32_t w = (uint32_t)wi4[j] >> 1;
                                braw[g] = *(const uint32_t *)((const char *)A.byte_table + ((uint32_t)__builtin_amdgcn_ds_bpermute(sl4[j], ga) + w));
                                lra[j] = __builtin_amdgcn_ds_bpermute(sl4[j], la);
                                if (dual) {
                                    braw2[g] = *(const uint32_t *)((const char *)A.byte_table + ((uint32_t)__builtin_amdgcn_ds_bpermute(sl4[j], gb) + w));
                                    lrb[j] = __builtin_amdgcn_ds_bpermute(sl4[j], lb);
                                }
                            }
                            if (j + 1 < NE) { lra[j + 1] = lra[j]; lrb[j + 1] = lrb[j]; }
                        } else {
                            bn[j] = 0.f;
                            if (!(bmask >> j & 1)) continue;
                            const uint32_t w = (uint32_t)wi4[j];
                            float v = *(const float *)((const char *)A.byte_table + ((uint32_t)__builtin_amdgcn_ds_bpermute(sl4[j], ga) + w));
                            lra[j] = __builtin_amdgcn_ds_bpermute(sl4[j], la);
                            if (dual) {
                                v += *(const float *)((const char *)A.byte_table + ((uint32_t)__builtin_amdgcn_ds_bpermute(sl4[j], gb) + w));
                                lrb[j] = __builtin_amdgcn_ds_bpermute(sl4[j], lb);
                            }
                            bn[j] = v;
                        }
                    }
                    if (more) {   // the next position's byte ids while this one's rows are in flight
                        ida_nx = load_id(A.ids_a, n_nx);
                        if (dual) idb_nx = load_id(A.ids_b, n_nx);
                    }
                }
                if constexpr (BF) {   // everything is requested: raw words -> floats
                    unpack(graw, dy);
                    if (BYTES) {
                        unpack(braw, bn);
                        if (dual) {
                            float b2[NE];
                            unpack(braw2, b2);
#pragma unroll
                            for (int j = 0; j < NE; ++j) bn[j] += b2[j];
                        }
                    }
                    if (newrun) begin_run();
                }
            } else {
#pragma unroll
                for (int j = 0; j < NE; ++j) dy[j] = *(const float *)(grow + (lane4 + 256u * j));
                if (tok != cur) {   // a new run: the previous token's gradient row leaves, this token's (normalised) row comes in
                    flush();
                    cur = tok;
                    const char *trow = (const char *)A.tok_table + ((int64_t)tok * A.Dt - A.tok_lo) * (int64_t)esz;
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        acc[j] = 0.f;
                        an[j] = (tmask >> j & 1) ? *(const float *)(trow + (lane4 + 256u * j)) : 0.f;
                    }
                    ra = 1.f;
                    if (A.norm_tok) {
                        float ss = 0.f;
#pragma unroll
                        for (int j = 0; j < NE; ++j) ss += an[j] * an[j];
                        ra = rms_scale(wave_sum(ss), A.Dt, A.eps);
#pragma unroll
                        for (int j = 0; j < NE; ++j) an[j] *= ra;
                    }
                }
                if (BYTES) {
                    // lanes < bpt: gather offset of the slot's byte row and offset of its LDS row (or -1)
                    const int ga = ida * A.Db * (int)esz, gb = idb * A.Db * (int)esz;
                    const int sa = byte_slot(ida), sb = byte_slot(idb);
                    const int la = sa >= 0 ? sa * (int)row8 : -1, lb = sb >= 0 ? sb * (int)row8 : -1;
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        bn[j] = 0.f; lra[j] = 0; lrb[j] = 0;
                        if (!(bmask >> j & 1)) continue;
                        const uint32_t w = (uint32_t)wi4[j];
                        float v = *(const float *)((const char *)A.byte_table + ((uint32_t)__builtin_amdgcn_ds_bpermute(sl4[j], ga) + w));
                        lra[j] = __builtin_amdgcn_ds_bpermute(sl4[j], la);
                        if (dual) {
                            v += *(const float *)((const char *)A.byte_table + ((uint32_t)__builtin_amdgcn_ds_bpermute(sl4[j], gb) + w));
                            lrb[j] = __builtin_amdgcn_ds_bpermute(sl4[j], lb);
                        }
                        bn[j] = v;
                    }
                    if (more) {   // the next position's byte ids while this one's rows are in flight
                        ida_nx = load_id(A.ids_a, n_nx);
                        if (dual) idb_nx = load_id(A.ids_b, n_nx);
                    }
                }
            }
            float rnb = 1.f;   // lanes < bpt: 1/rms of the slot's byte row
            if (BYTES && A.norm_byte) {
                if (pair_norm) {   // the rms factor belongs to the (token, slot) pair: per-slot sum of squares in fixed point
                    float pmax = 0.f;
#pragma unroll
                    for (int j = 0; j < NE; ++j) pmax = fmaxf(pmax, bn[j] * bn[j]);
                    pmax = wave_max(pmax);
                    const int kk = (pmax > 0.f && pmax < INFINITY) ? 40 - ilogbf(pmax) : 0;
                    seg_q[lane] = 0ull;
                    seg_sync();
#pragma unroll
                    for (int j = 0; j < NE; ++j)
                        if (bmask >> j & 1) atomicAdd(seg_q + (sl4[j] >> 2), to_fixed(bn[j] * bn[j], kk));
                    seg_sync();
                    rnb = rms_scale((float)ldexp((double)(long long)seg_q[lane], -kk), A.Db, A.eps);
                    seg_sync();
                } else {
                    rnb = A.byte_rnorm[ida];
                }
#pragma unroll
                for (int j = 0; j < NE; ++j) bn[j] *= __int_as_float(__builtin_amdgcn_ds_bpermute(sl4[j], __float_as_int(rnb)));
            }
            // ---- back through the output norm
            if (A.norm_out) {
                float ss = 0.f, m = 0.f;
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const float y = an[j] * s_tok + (MODE == MOT_MIX_SUM ? bn[j] * s_byte : 0.f);
                    ss += y * y;
                    m += dy[j] * y;
                }
                const float ry = rms_scale(wave_sum(ss), D, A.eps);
                m = wave_sum(m) * ry * (1.0f / (float)D);       // mean(g * x), x = y * ry   (a multiply, not an IEEE divide: <= 1 ulp)
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
                ds_t += dot;
                float mt = 0.f;
                if (A.norm_tok) mt = wave_sum(dot * s_tok) * inv_dt;
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const float da = dy[j] * s_tok;
                    acc[j] += A.norm_tok ? ra * (da - an[j] * mt) : da;
                }
            }
            // ---- byte side
            if (BYTES) {
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < NE; ++j) { dot += dy[j] * bn[j]; dy[j] *= s_byte; }
                ds_b += dot;
                if (A.norm_byte) {   // v = r_row * (db - b_n * mean_slot(db * b_n)): per-slot sums in fixed point
                    float pmax = 0.f;
#pragma unroll
                    for (int j = 0; j < NE; ++j) pmax = fmaxf(pmax, fabsf(dy[j] * bn[j]));
                    pmax = wave_max(pmax);
                    const int kk = (pmax > 0.f && pmax < INFINITY) ? 40 - ilogbf(pmax) : 0;
                    // every lane takes part in the zeroing and the read-back (kMaxBpt = 64 entries): with a
                    // `lane < bpt` guard the compiler sinks the atomics into both sides of the branch and the
                    // unguarded lanes' adds run BEFORE the zeroing store
                    seg_q[lane] = 0ull;
                    seg_sync();
#pragma unroll
                    for (int j = 0; j < NE; ++j)
                        if (bmask >> j & 1) atomicAdd(seg_q + (sl4[j] >> 2), to_fixed(dy[j] * bn[j], kk));
                    seg_sync();
                    float sg = (float)ldexp((double)(long long)seg_q[lane], -kk) / (float)A.Db;
                    if (!(pmax < INFINITY)) sg = NAN;   // non-finite gradients stay non-finite
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        const float rj = __int_as_float(__builtin_amdgcn_ds_bpermute(sl4[j], __float_as_int(rnb)));
                        const float sj = __int_as_float(__builtin_amdgcn_ds_bpermute(sl4[j], __float_as_int(sg)));
                        dy[j] = rj * (dy[j] - bn[j] * sj);
                    }
                    seg_sync();
                }
                if (!(A.abl & 1)) {
                    bool slow = false;   // some element of this lane needs the exact path
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        if (!(bmask >> j & 1)) continue;
                        const bool inr = in_range(dy[j]), nz = (__float_as_uint(dy[j]) << 1) != 0u;
                        const bool oka = inr && lra[j] >= 0;
                        add_fixed(max(lra[j], 0), j, oka ? dy[j] : 0.f);
                        slow |= nz && !oka;
                        if (dual) {
                            const bool okb = inr && lrb[j] >= 0;
                            add_fixed(max(lrb[j], 0), j, okb ? dy[j] : 0.f);
                            slow |= nz && !okb;
                        }
                    }
                    if (__any(slow)) {
#pragma unroll
                        for (int j = 0; j < NE; ++j) {
                            if (!(bmask >> j & 1)) continue;
                            const bool inr = in_range(dy[j]), nz = (__float_as_uint(dy[j]) << 1) != 0u;
                            const int wi = wi4[j] >> 2;
                            const int ia = __builtin_amdgcn_ds_bpermute(sl4[j], ida);
                            if (nz && !(inr && lra[j] >= 0)) atomicAdd(A.d_byte + ia * A.Db + wi, dy[j]);
                            if (dual) {
                                const int ib = __builtin_amdgcn_ds_bpermute(sl4[j], idb);
                                if (nz && !(inr && lrb[j] >= 0)) atomicAdd(A.d_byte + ib * A.Db + wi, dy[j]);
                            }
                        }
                    }
                }
            }
        }
    }
    flush();
    if (A.d_scale_tok) { ds_t = wave_sum(ds_t); if (lane == 0) atomicAdd(A.d_scale_tok, ds_t); }
    if (BYTES && A.d_scale_byte) { ds_b = wave_sum(ds_b); if (lane == 0) atomicAdd(A.d_scale_byte, ds_b); }
    if (BYTES) {
        __syncthreads();
        for (int i = tid; i < nbyte; i += kBwdThreads) {
            const long long q = (long long)dbyte_q[i];
            if (q == 0) continue;
            const int sl = i / A.Db, at = i - sl * A.Db, half = A.Db >> 1;
            const int wi = BF ? (at < half ? 2 * at : 2 * (at - half) + 1) : at;
            const int row = sl < A.priv_lo ? sl : sl - A.priv_lo + A.priv_hi0;
            atomicAdd(A.d_byte + row * A.Db + wi, (float)ldexp((double)q, -fx_k));
        }
    }
}

// ------------------------------------------------------------------------------------------
// Lane-contiguous variant for the headline shapes: SUM / NOOP, fp32, D = 64 NE with NE a multiple of 4, and (SUM) a byte slot
// made of whole lanes (Db a multiple of NE, i.e. 64 / bpt lanes per slot: bpt a power of two).  A lane owns NE CONSECUTIVE
// elements of a row, so
//   * a gradient row, a token row and the lane's piece of a byte row are NE / 4 16-byte loads each (the strided layout of
//     embed_mix_bwd_full_kernel needs NE 4-byte loads: four times the vector-memory instructions per position, and the
//     position rate of that kernel was what its address unit could issue);
//   * a lane belongs to ONE byte slot: one id, one row offset, one rms factor per lane, no cross-lane hand-out; the per-slot
//     sums of the byte-norm backward are a lane-local sum plus a log2(64 / bpt)-step exchange inside the slot's lane group;
//   * 12 waves per workgroup (one workgroup per CU: the privatised byte-table gradient takes most of the LDS), each with the
//     NEXT position's gradient row already requested while it works on the current one: the kernel streams 1.6 GB of gradient
//     rows, and one 3 KB row in flight per wave (16 waves: 48 KB per CU) covered 60 % of what HBM latency x bandwidth asks for;
//   * the privatised table's rows are Db + 1 sums apart: with a stride of Db = 48 64-bit words every row starts on one of two
//     bank offsets and a wave's 64 adds met on 8 bank positions (SQ_LDS_BANK_CONFLICT was a quarter of the kernel's cycles).
// The token-table flush wants 256 contiguous bytes per atomic wave-instruction (lane-contiguous pieces would spread each
// instruction over 3 KB: ~12x the memory-side atomic requests), so a run's gradient row is transposed through a per-wave LDS
// buffer once per run (a run of equal tokens is ~17 positions long on FineWeb-shaped ids).
// ------------------------------------------------------------------------------------------
constexpr int kLcThreads = 768, kLcWaves = kLcThreads / 64;   // 12 waves, 3 per SIMD: ~170 registers per lane, room for the next position's gradient row

// T: the element type of the tables and of grad_out (float, or __bf16: rows widened as they are loaded; every sum stays fp32)
template <int MODE, int NE, bool DUAL, typename T>
__global__ __launch_bounds__(kLcThreads) void embed_mix_bwd_lc_kernel(const BwdArgs A) {
#pragma clang fp contract(fast)   // fused multiply-adds here: the sums below are compared with float64 at 2e-5, not bit for bit
    constexpr int D = 64 * NE, NV = NE / 4;
    constexpr bool BYTES = MODE != MOT_MIX_NOOP;
    extern __shared__ unsigned long long lds_q[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qs = A.Db + 1;                                      // row stride of the privatised sums, in 64-bit words
    const int nbyte = (A.priv_rows * qs + 1) & ~1;
    unsigned long long *dbyte_q = lds_q;
    uint32_t *fx_bits = (uint32_t *)(lds_q + nbyte);
    float *xp = (float *)(lds_q + nbyte + 2) + wave * D;          // this wave's transposition buffer (16-byte aligned: nbyte is even)
    for (int i = tid; i < nbyte; i += kLcThreads) dbyte_q[i] = 0ull;
    if (tid == 0) *fx_bits = 0u;
    const float s_tok = A.scale_tok ? *A.scale_tok : 1.0f;
    const float s_byte = A.scale_byte ? *A.scale_byte : 1.0f;
    const float inv_d = 1.0f / (float)D;
    // the lane's place: byte slot, first element inside the slot's row
    const int lps = BYTES ? A.Db / NE : 64;                       // lanes per slot
    const int slot = BYTES ? lane / lps : 0;
    const int wi0 = BYTES ? (lane - slot * lps) * NE : 0;
    const uint32_t lane_off = (uint32_t)lane * (NE * (uint32_t)sizeof(T));
    const T *grad_out = (const T *)A.grad_out, *tok_table = (const T *)A.tok_table, *byte_table = (const T *)A.byte_table;
    const int64_t gld = A.g_ld ? A.g_ld : D;                      // elements between gradient rows
    const bool no_tok = A.no_tok != 0;                            // byte slots only: no token row is read, none is written
    auto load_row = [&](const char *rbase, float (&dst)[NE]) {   // NE consecutive elements at rbase + lane_off
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4v w = Elem<T>::load4((const T *)(rbase + (lane_off + 4u * (uint32_t)sizeof(T) * v)));
            dst[4 * v] = w.x; dst[4 * v + 1] = w.y; dst[4 * v + 2] = w.z; dst[4 * v + 3] = w.w;
        }
    };
    auto load_byte_row = [&](int id, float (&dst)[NE]) { load_row((const char *)(byte_table + (int64_t)id * A.Db + wi0) - lane_off, dst); };
    auto slot_sum = [&](float v) {                                // sum over the lanes of this lane's slot (lps a power of two)
        for (int o = 1; o < lps; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    int fx_k = 0;
    bool fx_on = false;                                           // a non-zero finite sample exists: without one NOTHING is converted
    float fx_mul = 1.f;                                           // s_byte * 2^fx_k: a byte-row term's fixed-point value is dy * fx_mul
    auto byte_slot = [&](int id) { return id < A.priv_lo ? id : (id >= A.priv_hi0 ? id - A.priv_hi0 + A.priv_lo : -1); };

    float acc[NE], an[NE];
    float ra = 1.f, ds_t = 0.f, ds_b = 0.f;
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0 || no_tok) return;
#pragma unroll
        for (int v = 0; v < NV; ++v) *(float4v *)(xp + lane * NE + 4 * v) = float4v{acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]};
        seg_sync();
        float *drow = A.d_tok + (int64_t)cur * D;
#pragma unroll
        for (int j = 0; j < NE; ++j) atomicAdd(drow + lane + 64 * j, xp[lane + 64 * j]);
        seg_sync();
    };
    const int64_t per_wg = (A.n_tokens + gridDim.x - 1) / gridDim.x;
    const int64_t wg_lo = min(A.n_tokens, (int64_t)blockIdx.x * per_wg), wg_hi = min(A.n_tokens, wg_lo + per_wg);
    const int64_t per_wave = (wg_hi - wg_lo + kLcWaves - 1) / kLcWaves;
    const int64_t s_begin = min(wg_hi, wg_lo + wave * per_wave), s_end = min(wg_hi, s_begin + per_wave);
    __syncthreads();   // orders the table zeroing above
    if (BYTES) {       // fixed-point scale of the privatised sums, as in embed_mix_bwd_full_kernel: every wave takes part
        float gmax = 0.f;
        if (s_begin < s_end) {
            float gs[NE];
            load_row((const char *)(grad_out + (int64_t)A.pos_sorted[s_begin] * gld), gs);
#pragma unroll
            for (int j = 0; j < NE; ++j) gmax = fmaxf(gmax, fabsf(gs[j]));
            gmax = wave_max(gmax) * fabsf(s_byte);
            if (lane == 0 && gmax > 0.f && gmax < INFINITY) atomicMax(fx_bits, __float_as_uint(gmax));
        }
        __syncthreads();
        const float m = __uint_as_float(*fx_bits);
        fx_on = m > 0.f;
        fx_k = fx_on ? max(-100, min(27 - ilogbf(m), 100)) : 0;
        fx_mul = ldexpf(s_byte, fx_k);
    }
    // the id of this lane's slot: requested raw, range-checked where it is first used (a check right behind the load would make
    // the wave wait for everything it has in flight)
    auto load_id = [&](const int64_t *ids, int64_t n) { return *(const int64_t *)((const char *)(ids + n * A.bpt + A.slot0) + (uint32_t)slot * 8u); };
    auto clamp_id = [&](int64_t v) {
        if ((uint64_t)v >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); v = 0; }
        return (int)v;
    };
    // Three positions are in flight per wave: k is being worked on, the rows of k + 1 (gradient row, byte rows) and the byte
    // ids of k + 2 have been requested.  Positions come 64 (position, token) pairs at a time, handed out with readlane.
    for (int64_t s0 = s_begin; s0 < s_end; s0 += 64) {
        const int cnt = (int)min((int64_t)64, s_end - s0);
        int vpos = 0, vtok = 0;
        if (lane < cnt) { vpos = A.pos_sorted[s0 + lane]; vtok = A.tok_sorted[s0 + lane]; }
        const int n0 = __builtin_amdgcn_readfirstlane(vpos), n1 = __builtin_amdgcn_readlane(vpos, cnt > 1 ? 1 : 0);
        float g_nx[NE], b_nx[NE], rn_nx = 1.f;
        int ida_nx = 0, idb_nx = 0;
        int64_t ida_n2 = 0, idb_n1 = 0;                           // raw: the first id tensor's entry two positions ahead, the second's one ahead
        load_row((const char *)(grad_out + (int64_t)n0 * gld), g_nx);
        if (BYTES) {
            ida_nx = clamp_id(load_id(A.ids_a, n0));
            if (DUAL) idb_nx = clamp_id(load_id(A.ids_b, n0));
            ida_n2 = load_id(A.ids_a, n1);
            if (DUAL) idb_n1 = load_id(A.ids_b, n1);
            load_byte_row(ida_nx, b_nx);
            if (A.norm_byte) rn_nx = A.byte_rnorm[ida_nx];
        }
        for (int k = 0; k < cnt; ++k) {
            const int tok = __builtin_amdgcn_readlane(vtok, k), ida = ida_nx, idb = idb_nx;
            const bool more = k + 1 < cnt, newrun = tok != cur;
            float dy[NE], bn[NE];
            const float rnb = rn_nx;
#pragma unroll
            for (int j = 0; j < NE; ++j) { dy[j] = g_nx[j]; bn[j] = BYTES ? b_nx[j] : 0.f; }
            // Requests are issued oldest-needed first (a wave's loads return in order): what THIS position still lacks -- a new
            // run's token row, the second id tensor's byte row -- goes out before the next position's rows, so that waiting for
            // it leaves those in flight.
            float b2[NE];
            if constexpr (DUAL) load_byte_row(idb, b2);
            if (newrun) {   // the previous token's gradient row leaves, this token's row comes in
                flush();
                cur = tok;
                if (no_tok) {
#pragma unroll
                    for (int j = 0; j < NE; ++j) an[j] = 0.f;
                } else load_row((const char *)(tok_table + (int64_t)tok * D), an);
            }
            if (more) {   // the next position's rows, the byte ids of the one after
                const int nn = __builtin_amdgcn_readlane(vpos, k + 1);
                load_row((const char *)(grad_out + (int64_t)nn * gld), g_nx);
                if (BYTES) {
                    ida_nx = clamp_id(ida_n2);
                    if (DUAL) idb_nx = clamp_id(idb_n1);
                    load_byte_row(ida_nx, b_nx);
                    if (A.norm_byte) rn_nx = A.byte_rnorm[ida_nx];
                    if (k + 2 < cnt) {
                        const int n2 = __builtin_amdgcn_readlane(vpos, k + 2);
                        ida_n2 = load_id(A.ids_a, n2);
                        if (DUAL) idb_n1 = load_id(A.ids_b, n2);
                    }
                }
            }
            if constexpr (DUAL) {
#pragma unroll
                for (int j = 0; j < NE; ++j) bn[j] += b2[j];
            }
            if (newrun) {
#pragma unroll
                for (int j = 0; j < NE; ++j) acc[j] = 0.f;
                ra = 1.f;
                if (A.norm_tok) {
                    float ss = 0.f;
#pragma unroll
                    for (int j = 0; j < NE; ++j) ss += an[j] * an[j];
                    ra = rms_scale(wave_sum(ss), D, A.eps);
#pragma unroll
                    for (int j = 0; j < NE; ++j) an[j] *= ra;
                }
            }
            if (BYTES && A.norm_byte) {
#pragma unroll
                for (int j = 0; j < NE; ++j) bn[j] *= rnb;
            }
            // ---- back through the output norm: x = ry y, dy = ry (g - x mean(g x)) = ry g - (ry^3 sum(g y) / D) y
            if (A.norm_out) {
                float y[NE], ss = 0.f, m = 0.f;
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    y[j] = BYTES ? an[j] * s_tok + bn[j] * s_byte : an[j] * s_tok;
                    ss += y[j] * y[j];
                    m += dy[j] * y[j];
                }
                const float ry = rms_scale(wave_sum(ss), D, A.eps);
                const float c2 = wave_sum(m) * inv_d * ry * ry * ry;
#pragma unroll
                for (int j = 0; j < NE; ++j) dy[j] = ry * dy[j] - c2 * y[j];
            }
            // ---- token side
            {
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < NE; ++j) dot += dy[j] * an[j];
                ds_t += dot;
                if (A.norm_tok) {
                    const float c = wave_sum(dot) * s_tok * inv_d, rs = ra * s_tok;   // ra (s_t dy - a_n mean(s_t dy . a_n))
#pragma unroll
                    for (int j = 0; j < NE; ++j) acc[j] += rs * dy[j] - (ra * c) * an[j];
                } else {
#pragma unroll
                    for (int j = 0; j < NE; ++j) acc[j] += dy[j] * s_tok;
                }
            }
            // ---- byte side
            if (BYTES) {
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < NE; ++j) dot += dy[j] * bn[j];
                ds_b += dot;
                if (A.norm_byte) {   // v = r_row (db - b_n mean_slot(db b_n)), db = s_b dy: the slot is this lane's group
                    const float sg = slot_sum(dot) / (float)A.Db;
#pragma unroll
                    for (int j = 0; j < NE; ++j) dy[j] = rnb * (dy[j] - bn[j] * sg);
                }
                // Fixed-point sums in LDS: x = dy s_b 2^fx_k rounded to a 32-bit integer, added into 64-bit words.  A lane takes
                // this path with all its NE terms or not at all: it must have an LDS row, the workgroup must have a scale (a
                // non-zero finite sample: with all sampled rows zero, 2^0 would round every |term| < 0.5 away), and its largest |x|
                // must lie in [2^12, 2^31) -- or be zero, which adds nothing.  The upper bound is what converts (the comparison is
                // made on the |x| BIT PATTERNS, which order NaN above infinity: v_max_f32 would drop a NaN); the lower bound keeps
                // rows far below the sample (masked or down-weighted positions: largest term <= 2^-15 of the sampled maximum) on
                // the float atomics with their full relative precision.  Inside a converted lane a term's absolute rounding error
                // is <= 2^-(fx_k + 1), i.e. 2^-28 of the sampled |g s_b| maximum -- below what a float atomic loses on a row's
                // running sum.  Lanes that fail add exact global float atomics instead.
                const int sa = byte_slot(ida), sb = DUAL ? byte_slot(idb) : 0;
                uint32_t abits = 0u;
#pragma unroll
                for (int j = 0; j < NE; ++j) { dy[j] *= fx_mul; abits = max(abits, __float_as_uint(dy[j]) & 0x7fffffffu); }
                const bool fits = fx_on && abits < 0x4f000000u /* 2^31 */ && abits >= 0x45800000u /* 2^12 */, zero = abits == 0u;
                const bool oka = zero || (fits && sa >= 0), okb = DUAL ? (zero || (fits && sb >= 0)) : true;
                if (__all(oka && okb)) {   // (an all-zero lane may have no LDS row: it adds zeros to row 0)
                    unsigned long long *ra_q = dbyte_q + (size_t)max(sa, 0) * qs + wi0, *rb_q = dbyte_q + (size_t)max(sb, 0) * qs + wi0;
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        const unsigned long long q = (unsigned long long)(long long)__float2int_rn(dy[j]);
                        atomicAdd(ra_q + j, q);
                        if (DUAL) atomicAdd(rb_q + j, q);
                    }
                } else {
                    const float back = ldexpf(1.0f, -fx_k);       // x 2^-fx_k: the term itself, exactly
                    unsigned long long *ra_q = dbyte_q + (size_t)max(sa, 0) * qs + wi0, *rb_q = dbyte_q + (size_t)max(sb, 0) * qs + wi0;
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        const unsigned long long q = (unsigned long long)(long long)__float2int_rn(fits ? dy[j] : 0.f);
                        const float v = dy[j] * back;
                        if (oka) atomicAdd(ra_q + j, q); else if (v != 0.f) atomicAdd(A.d_byte + (int64_t)ida * A.Db + wi0 + j, v);
                        if (DUAL) { if (okb) atomicAdd(rb_q + j, q); else if (v != 0.f) atomicAdd(A.d_byte + (int64_t)idb * A.Db + wi0 + j, v); }
                    }
                }
            }
        }
    }
    flush();
    if (A.d_scale_tok) { ds_t = wave_sum(ds_t); if (lane == 0) atomicAdd(A.d_scale_tok, ds_t); }
    if (BYTES && A.d_scale_byte) { ds_b = wave_sum(ds_b); if (lane == 0) atomicAdd(A.d_scale_byte, ds_b); }
    if (BYTES) {
        __syncthreads();
        for (int i = tid; i < A.priv_rows * A.Db; i += kLcThreads) {
            const int sl = i / A.Db, wi = i - sl * A.Db;
            const long long q = (long long)dbyte_q[sl * qs + wi];
            if (q == 0) continue;
            const int row = sl < A.priv_lo ? sl : sl - A.priv_lo + A.priv_hi0;
            atomicAdd(A.d_byte + row * A.Db + wi, (float)ldexp((double)q, -fx_k));
        }
    }
}

template <int MODE>
static bool lc_layout(const BwdArgs &A) {
    if ((A.D & 255) || A.Dt != A.D || A.tok_lo != 0 || A.D > 1024) return false;
    const int ne = A.D / 64;
    if (MODE == MOT_MIX_SUM) {
        if (A.byte_lo != 0 || A.nbk != A.D || A.Db % ne) return false;
        const int lps = A.Db / ne;
        if (lps & (lps - 1)) return false;                        // the slot sums exchange inside power-of-two lane groups
        if (A.ids_b && ne > 12) return false;                     // two id tensors at 1024 columns would spill
    }
    return ne == 4 || ne == 8 || ne == 12 || ne == 16;   // 1024-thread workgroups cap a lane at 128 registers: NE 24 / 32 would spill
}

template <int MODE, int NE, bool DUAL, typename T>
static int launch_bwd_lc_tt(const BwdArgs &A, size_t lds, hipStream_t stream);
template <int MODE, int NE, bool DUAL>
static int launch_bwd_lc_t(const BwdArgs &A, size_t lds, hipStream_t stream) {
    return A.in_bf16 ? launch_bwd_lc_tt<MODE, NE, DUAL, __bf16>(A, lds, stream) : launch_bwd_lc_tt<MODE, NE, DUAL, float>(A, lds, stream);
}
template <int MODE, int NE, bool DUAL, typename T>
static int launch_bwd_lc_tt(const BwdArgs &A, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> lds_ok{0};   // per-device bits
    if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mix_bwd_lc_kernel<MODE, NE, DUAL, T>, lds_ok, "embed_mix_bwd_lc_kernel")) return rc_lds;
    int64_t blocks = (A.n_tokens + 16 * kLcWaves - 1) / (16 * kLcWaves);   // >= 16 sorted positions per wave
    if (blocks > 256) blocks = 256;   // one workgroup per CU
    hipLaunchKernelGGL((embed_mix_bwd_lc_kernel<MODE, NE, DUAL, T>), dim3((unsigned)blocks), dim3(kLcThreads), lds, stream, A);
    return check_launch("embed_mix_bwd_lc_kernel");
}
template <int MODE, int NE>
static int launch_bwd_lc(const BwdArgs &A, size_t lds, hipStream_t stream) {
    if constexpr (MODE == MOT_MIX_SUM) {
        if (A.ids_b) return launch_bwd_lc_t<MODE, NE, true>(A, lds, stream);
    }
    return launch_bwd_lc_t<MODE, NE, false>(A, lds, stream);
}

template <int MODE>
static int dispatch_ne_lc(const BwdArgs &A, size_t lds, hipStream_t stream) {
    switch (A.D / 64) {
        case 4: return launch_bwd_lc<MODE, 4>(A, lds, stream);
        case 8: return launch_bwd_lc<MODE, 8>(A, lds, stream);
        case 12: return launch_bwd_lc<MODE, 12>(A, lds, stream);
        default: return launch_bwd_lc<MODE, 16>(A, lds, stream);
    }
}

// ------------------------------------------------------------------------------------------
// Round 3: the headline configuration by itself.  norm(E_tok[t] + concat_k E_byte[id_k]) of runs/71_*.py:227-230 has no
// per-embedding norm and no learned scalar, so nothing of the general kernel's token / byte side bookkeeping is needed:
//     y = a + b,   dy = ry g - (ry^3 mean(g y)) y,   d_tok[t] += dy,   d_byte[id_k] += dy[slot k]
// What the counters and timing ablations said about embed_mix_bwd_lc_kernel on this configuration (profiles/r03_c4_backward_*):
// its HBM traffic IS the algorithmic traffic (FETCH 1.81 GB + WRITE 0.18 GB + 0.19 GB of atomic requests against 1.61 GB of
// gradient rows + token rows + ids); the LDS adds cost 0.03-0.05 ms, the token-row flushes 0.10 ms, everything else 0.34 ms
// (1.8 GB of random 3 KB rows at 5.4 TB/s, the rate the CDNA guide measures for such gathers), and the parts do not overlap.
// This kernel changes what could be changed inside hipcc's own s_waitcnt placement:
//   * rows are read as 16-byte chunks lane, lane + 64, ...: every row load is a fully coalesced 1 KiB wave-instruction that
//     touches each 128-byte line once (the lane-contiguous layout reads a row with NV instructions that each touch all of its
//     lines; with more rows in flight than the 32 KB L1 holds, lines are fetched up to three times -- requesting rows further
//     ahead made that kernel SLOWER);
//   * the current run's token row lives in the wave's LDS strip (NV ds_read_b128 per place instead of NE registers), the bpt
//     ids of a place are ONE 8-byte load per lane handed out with ds_bpermute, the running sum and the byte-row terms share one
//     scaled form (dys = 2^k dy feeds both the fixed-point LDS adds and the register sum);
//   * lookahead indices are clamped inside a 128-place segment, so every step issues the same requests, and the step is
//     written twice (same run / new run): hipcc counts its waits per straight-line path and takes the SMALLER count wherever
//     two paths join, so one shared body made every step wait for the gradient row of the NEXT place.
// What is left on the table, measured ("Backward, round 3"): a new run still drains the wave's memory queue
// (the compiler-visible token-row load behind the 12 atomics: two memory round trips per run, 17 runs per wave, 0.10 ms); the
// versions that avoid it -- atomics or a plain read-add-store of the d_tok row at the END of the step, requests from inline asm
// with hand-counted vmcnt -- lost to hipcc copying in-flight registers in front of the tied waits (v_mov_b64 of the id pair, then
// the wait on the copy; with physical-register constraints: copies in and out of the pinned registers and 46 spills).
// Everything else -- the sorted order, the privatised 64-bit byte-table sums with 49-word rows, the exact global-atomic path
// for what has no LDS slot or does not convert -- is embed_mix_bwd_lc_kernel's.
// ------------------------------------------------------------------------------------------
constexpr int kPlThreads = 768, kPlWaves = kPlThreads / 64;   // 12 waves, 3 per SIMD: 168 registers per lane

// T: element type of the tables and of grad_out (float, or __bf16 as the production loop runs them, train_gpt.py:1124-1126); all
// arithmetic and every gradient buffer are fp32.  A lane's chunk is FOUR elements either way (16 bytes of fp32, 8 bytes of bf16).
template <int MODE, int NE, bool NORM_OUT, typename T>
__global__ __launch_bounds__(kPlThreads) void embed_mix_bwd_plain_kernel(const BwdArgs A) {
#pragma clang fp contract(fast)
    constexpr int D = 64 * NE, NV = NE / 4;
    constexpr bool BYTES = MODE != MOT_MIX_NOOP;
    constexpr int kSeg = 128;                                     // places per pipeline segment: their (position, token) pairs sit in 2 x 2 registers
#ifdef MOT_DEV_ABLATION   // timing-only switches (results are wrong): 1 no LDS byte adds, 2 no token-row atomics, 512 no wave sums
#define PL_ABL(bit) (A.abl & (bit))
#else
#define PL_ABL(bit) false
#endif
    extern __shared__ unsigned long long lds_q[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qs = A.Db + 1;
    const int nbyte = (A.priv_rows * qs + 1) & ~1;
    unsigned long long *dbyte_q = lds_q;
    uint32_t *fx_bits = (uint32_t *)(lds_q + nbyte);
    float *xp = (float *)(lds_q + nbyte + 2) + wave * D;
    for (int i = tid; i < nbyte; i += kPlThreads) dbyte_q[i] = 0ull;
    if (tid == 0) *fx_bits = 0u;
    const float inv_d = 1.0f / (float)D;
    // Lane l owns the 16-byte chunks l, l + 64, ... of a row (elements 4c .. 4c + 3 of chunk c).  A chunk lies inside one byte
    // slot (Db a multiple of 4): chunk v of this lane belongs to slot sl[v] and starts at element wo[v] of that slot's byte row.
    const int cps = BYTES ? A.Db / 4 : 1;                         // chunks per slot
    int sl[NV], wo[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { const int c = lane + 64 * v; sl[v] = BYTES ? c / cps : 0; wo[v] = BYTES ? 4 * (c - sl[v] * cps) : 0; }
    constexpr uint32_t kChunk = 4u * (uint32_t)sizeof(T);          // bytes of a lane's chunk in the tables / grad_out
    const uint32_t lane_off = (uint32_t)lane * kChunk;
    const T *tok_table = (const T *)A.tok_table, *byte_table = (const T *)A.byte_table, *grad_out = (const T *)A.grad_out;
    auto load_row = [&](float4v (&dst)[NV], const T *rbase) {     // one row of D elements in chunk order, widened
#pragma unroll
        for (int v = 0; v < NV; ++v) dst[v] = Elem<T>::load4((const T *)((const char *)rbase + (lane_off + 64u * kChunk * v)));
    };
    auto load_byte_rows = [&](float4v (&dst)[NV], const int (&id)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; ++v) dst[v] = Elem<T>::load4(byte_table + (int64_t)id[v] * A.Db + wo[v]);
    };
    auto byte_slot = [&](int id) { return id < A.priv_lo ? id : (id >= A.priv_hi0 ? id - A.priv_hi0 + A.priv_lo : -1); };
    // The bpt ids of a position are one cache line: lane k < bpt loads id k (one 8-byte load per position), range-checks it, and
    // the NV slots of every lane pick theirs with ds_bpermute (the LDS crossbar: no memory, no bank conflicts).
    const int idl = min(lane, max(A.bpt, 1) - 1);
    auto load_ids = [&](int64_t n) { return A.ids_a[n * A.bpt + idl]; };
    auto clamp_id = [&](int64_t raw) {
        const bool bad = (uint64_t)raw >= (uint64_t)A.byte_rows;
        if (bad && A.status) atomicOr(A.status, kStatusByteOor);
        return bad ? 0 : (int)raw;
    };
    auto spread_ids = [&](int idv, int (&id)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; ++v) id[v] = __builtin_amdgcn_ds_bpermute(sl[v] * 4, idv);
    };
    const int64_t per_wg = (A.n_tokens + gridDim.x - 1) / gridDim.x;
    const int64_t wg_lo = min(A.n_tokens, (int64_t)blockIdx.x * per_wg), wg_hi = min(A.n_tokens, wg_lo + per_wg);
    const int64_t per_wave = (wg_hi - wg_lo + kPlWaves - 1) / kPlWaves;
    const int64_t s_begin = min(wg_hi, wg_lo + wave * per_wave), s_end = min(wg_hi, s_begin + per_wave);
    __syncthreads();
    // fixed-point scale 2^fx_k of the privatised sums from one sampled gradient row per wave (sample maximum -> 2^27); with no
    // non-zero finite sample (m == 0) NOTHING takes the fixed-point path: every non-zero term goes out as an exact float atomic
    int fx_k = 0;
    bool fx_on = false;
    {
        float gmax = 0.f;
        if (s_begin < s_end) {
            float4v gs[NV];
            load_row(gs, grad_out + (int64_t)A.pos_sorted[s_begin] * D);
#pragma unroll
            for (int v = 0; v < NV; ++v) gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(gs[v].x), fabsf(gs[v].y))), fmaxf(fabsf(gs[v].z), fabsf(gs[v].w)));
            gmax = wave_max(gmax);
            if (lane == 0 && gmax > 0.f && gmax < INFINITY) atomicMax(fx_bits, __float_as_uint(gmax));
        }
        __syncthreads();
        const float m = __uint_as_float(*fx_bits);
        fx_on = m > 0.f;
        fx_k = fx_on ? max(-100, min(27 - ilogbf(m), 100)) : 0;
    }
    const float fx_s = ldexpf(1.0f, fx_k), fx_back = ldexpf(1.0f, -fx_k);
    float4v acc[NV];
    int cur = -1;
    // The current run's token row lives in the wave's LDS strip between two flushes; the strip is also where a finished row is
    // turned into 256-byte-contiguous atomic instructions.
    auto strip_read = [&](float4v (&dst)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; ++v) dst[v] = *(const float4v *)(xp + 4 * (lane + 64 * v));
    };
    auto strip_write = [&](const float4v (&src)[NV], float f) {
#pragma unroll
        for (int v = 0; v < NV; ++v) *(float4v *)(xp + 4 * (lane + 64 * v)) = src[v] * f;
    };
    auto flush = [&]() {                                          // acc (scaled) -> strip -> NE atomic wave-instructions of 256 contiguous bytes
        if (cur < 0) return;
        seg_sync(); strip_write(acc, fx_back); seg_sync();
        float *drow = A.d_tok + (int64_t)cur * D;
        float tr[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) tr[j] = xp[lane + 64 * j];
        if (!PL_ABL(2)) {
#pragma unroll
            for (int j = 0; j < NE; ++j) atomicAdd(drow + lane + 64 * j, tr[j]);
        }
        seg_sync();
    };
    auto new_run = [&](int tok) {                                 // the finished row leaves, this token's row comes into the strip, an empty sum
        flush();
        cur = tok;
        load_row(acc, tok_table + (int64_t)tok * D);
        __builtin_amdgcn_s_waitcnt(0x0f70);                        // vmcnt(0) (gfx9 encoding: vmcnt [3:0] + [15:14]; expcnt, lgkmcnt left alone)
        strip_write(acc, 1.0f);
        seg_sync();
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = float4v{0.f, 0.f, 0.f, 0.f};
    };
    for (int64_t seg0 = s_begin; seg0 < s_end; seg0 += kSeg) {
        const int len = (int)min((int64_t)kSeg, s_end - seg0);
        int vpos = 0, vtok = 0, vpos_hi = 0, vtok_hi = 0;        // places 0..63 and 64..127 of the segment, a lane each
        if (lane < len) { vpos = A.pos_sorted[seg0 + lane]; vtok = A.tok_sorted[seg0 + lane]; }
        if (64 + lane < len) { vpos_hi = A.pos_sorted[seg0 + 64 + lane]; vtok_hi = A.tok_sorted[seg0 + 64 + lane]; }
        auto pos_at = [&](int i) { return i < 64 ? __builtin_amdgcn_readlane(vpos, i) : __builtin_amdgcn_readlane(vpos_hi, i - 64); };
        auto tok_at = [&](int i) { return i < 64 ? __builtin_amdgcn_readlane(vtok, i) : __builtin_amdgcn_readlane(vtok_hi, i - 64); };
        auto g_row = [&](int i) { return grad_out + (int64_t)pos_at(min(i, len - 1)) * D; };
        float4v b_nx[NV], G0[NV], G1[NV];
        int id_cur[NV], idv_n1 = 0;                               // this place's ids per slot of the lane; the next place's, one per lane
        int64_t idraw_n1 = 0;
        // ---- pipeline fill, requests in the order a step leaves them in (ids, byte rows, gradient row): the first step's waits
        //      are counted for the worse of its two predecessors.  Lookahead indices are clamped to the segment's last place, so
        //      every step issues the same vector-memory instructions (a re-read of a row that is in flight anyway).
        if (BYTES) {
            const int64_t r0 = load_ids(pos_at(0));
            idraw_n1 = load_ids(pos_at(min(1, len - 1)));
            idv_n1 = clamp_id(r0);
            spread_ids(idv_n1, id_cur);
            load_byte_rows(b_nx, id_cur);
        }
        load_row(G0, g_row(0));
        auto step = [&](float4v (&gc)[NV], float4v (&gfree)[NV], const int i) {     // gc: this place's gradient row; gfree: where the next place's goes
            const int tok = tok_at(i);
            if (BYTES) spread_ids(idv_n1, id_cur);
            auto rest = [&]() {
                // (1) the raw ids two places ahead (addresses from a scalar position: depend on nothing in flight)
                int64_t idraw_n2 = 0;
                if (BYTES) idraw_n2 = load_ids(pos_at(min(i + 2, len - 1)));
                // (2) y = a + b_i, then the next place's byte rows go into b's registers (their ids were requested one step ago)
                float4v y[NV];
                strip_read(y);
                if (BYTES) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) y[v] += b_nx[v];
                    int id_n1[NV];
                    idv_n1 = clamp_id(idraw_n1);
                    spread_ids(idv_n1, id_n1);
                    load_byte_rows(b_nx, id_n1);
                    idraw_n1 = idraw_n2;
                }
                // (3) the next place's gradient row
                load_row(gfree, g_row(i + 1));
                // (4) this place
                float ry_s = fx_s, c2_s = 0.f;
                bool finite = true;                                   // wave-uniform: a NaN / infinity anywhere in g or y shows in the two sums
                if (NORM_OUT) {
                    float4v ssv{0.f, 0.f, 0.f, 0.f}, mv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int v = 0; v < NV; ++v) { ssv += y[v] * y[v]; mv += gc[v] * y[v]; }
                    float ss = (ssv.x + ssv.y) + (ssv.z + ssv.w), m = (mv.x + mv.y) + (mv.z + mv.w);
                    if (!PL_ABL(512)) { ss = wave_sum(ss); m = wave_sum(m); }
                    const float ry = rms_scale(ss, D, A.eps);
                    ry_s = ry * fx_s;
                    c2_s = m * inv_d * ry * ry * ry_s;
                    finite = fabsf(c2_s) < INFINITY && fabsf(ry_s) < INFINITY;
                }
                float amax = 0.f;
                uint32_t abits = 0u;                                  // without the norm: max of the |x| bit patterns (orders NaN above infinity; v_max_f32 drops a NaN)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    gc[v] = NORM_OUT ? ry_s * gc[v] - c2_s * y[v] : gc[v] * fx_s;      // dys = 2^k dy
                    acc[v] += gc[v];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (NORM_OUT) amax = fmaxf(amax, fabsf(gc[v][e]));
                        else abits = max(abits, __float_as_uint(gc[v][e]) & 0x7fffffffu);
                    }
                }
                if (!NORM_OUT) amax = abits > 0x7f800000u ? INFINITY : __uint_as_float(abits);
                if (BYTES) {
                    // A lane converts all its NE terms or none: its slots need LDS rows, the workgroup a usable scale, and its largest
                    // |x| must lie in [2^12, 2^31) (or be zero: nothing to add).  The absolute rounding error of a converted term is
                    // <= 2^-(k+1), i.e. 2^-28 of the sampled maximum; a lane whose largest term is 2^-15 of that maximum or less
                    // (rows far below the sample: masked or down-weighted positions) keeps full relative precision through the
                    // float atomics.
                    int sa[NV];
                    bool rows = true;
#pragma unroll
                    for (int v = 0; v < NV; ++v) { sa[v] = byte_slot(id_cur[v]); rows &= sa[v] >= 0; }
                    const bool conv = fx_on && finite && amax < 0x1p31f && amax >= 0x1p12f;
                    const bool ok = (finite && amax == 0.f) || (conv && rows);
                    if (PL_ABL(1)) {
                        if (amax == 123.456f) dbyte_q[lane] = 1;
                    } else if (__all(ok)) {
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            unsigned long long *rq = dbyte_q + (size_t)max(sa[v], 0) * qs + wo[v];
#pragma unroll
                            for (int e = 0; e < 4; ++e) atomicAdd(rq + e, (unsigned long long)(long long)__float2int_rn(gc[v][e]));
                        }
                    } else {
                        // rare (a byte id without an LDS row, a term that does not convert): the lane's terms go through its part of the
                        // wave's LDS strip, one at a time, so that this path costs the common one no registers; the token row the strip
                        // held is fetched again afterwards
                        seg_sync(); strip_write(gc, 1.0f); seg_sync();
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            const bool lds_ok = conv && sa[v] >= 0;
                            unsigned long long *rq = dbyte_q + (size_t)max(sa[v], 0) * qs + wo[v];
                            float *gq = A.d_byte + (int64_t)id_cur[v] * A.Db + wo[v];
#pragma unroll 1
                            for (int e = 0; e < 4; ++e) {
                                const float x = xp[4 * (lane + 64 * v) + e];
                                if (lds_ok) atomicAdd(rq + e, (unsigned long long)(long long)__float2int_rn(x));
                                else if (x != 0.f) atomicAdd(gq + e, x * fx_back);   // (NaN != 0: it is added)
                            }
                        }
                        seg_sync();
                        load_row(y, tok_table + (int64_t)cur * D);
                        __builtin_amdgcn_s_waitcnt(0x0f70);
                        strip_write(y, 1.0f);
                        seg_sync();
                    }
                }
            };
            // A new run is the only conditional vector-memory work of a step, and it ends with everything this wave has in flight
            // drained; each path then runs its OWN copy of the rest of the step (see the header: waits are counted per path).
            if (tok != cur) { new_run(tok); __builtin_amdgcn_sched_barrier(0); rest(); } else { rest(); }
        };
        int i = 0;
        for (; i + 1 < len; i += 2) {   // two row buffers as register names
            step(G0, G1, i);
            step(G1, G0, i + 1);
        }
        if (i < len) step(G0, G1, i);
    }
    flush();
    if (BYTES) {
        __syncthreads();
        for (int i = tid; i < A.priv_rows * A.Db; i += kPlThreads) {
            const int sl_ = i / A.Db, wi = i - sl_ * A.Db;
            const long long q = (long long)dbyte_q[sl_ * qs + wi];
            if (q == 0) continue;
            const int row = sl_ < A.priv_lo ? sl_ : sl_ - A.priv_lo + A.priv_hi0;
            atomicAdd(A.d_byte + row * A.Db + wi, (float)ldexp((double)q, -fx_k));
        }
    }
#undef PL_ABL
}

// the configuration this kernel is for: SUM / NOOP over full rows of 256, 512 or 768 columns (D = 64 NE, a lane's chunks are four
// elements), one id tensor, no per-embedding norm, no learned scalars; fp32 or bf16 tables and gradient rows
template <int MODE>
static bool plain_layout(const BwdArgs &A) {
    if ((A.D & 255) || A.D > 768 || A.Dt != A.D || A.tok_lo != 0) return false;
    if (MODE == MOT_MIX_SUM && (A.byte_lo != 0 || A.nbk != A.D)) return false;
    if (A.norm_tok || A.norm_byte || A.scale_tok || A.scale_byte || A.d_scale_tok || A.d_scale_byte || A.ids_b || A.g_ld || A.no_tok || A.slot0) return false;
    if (MODE == MOT_MIX_SUM && ((A.Db & 3) || A.Db > 128)) return false;   // a chunk of a row lies inside one byte slot; wide byte rows
                                                                            // (the D-wide "slot" of the cross-attention mixin's two-id backward) do
                                                                            // not fit LDS, and this kernel's path for rows without an LDS slot is slow
    return true;
}

template <int MODE, int NE, bool NORM_OUT, typename T>
static int launch_bwd_plain_t(const BwdArgs &A, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> lds_ok{0};   // per-device bits
    if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mix_bwd_plain_kernel<MODE, NE, NORM_OUT, T>, lds_ok, "embed_mix_bwd_plain_kernel")) return rc_lds;
    int64_t blocks = (A.n_tokens + 16 * kPlWaves - 1) / (16 * kPlWaves);   // >= 16 sorted positions per wave
    if (blocks > 256) blocks = 256;   // one workgroup per CU
    hipLaunchKernelGGL((embed_mix_bwd_plain_kernel<MODE, NE, NORM_OUT, T>), dim3((unsigned)blocks), dim3(kPlThreads), lds, stream, A);
    return check_launch("embed_mix_bwd_plain_kernel");
}
template <int MODE, int NE>
static int launch_bwd_plain(const BwdArgs &A, size_t lds, hipStream_t stream) {
    if (A.in_bf16) {
        if (A.norm_out) return launch_bwd_plain_t<MODE, NE, true, __bf16>(A, lds, stream);
        return launch_bwd_plain_t<MODE, NE, false, __bf16>(A, lds, stream);
    }
    if (A.norm_out) return launch_bwd_plain_t<MODE, NE, true, float>(A, lds, stream);
    return launch_bwd_plain_t<MODE, NE, false, float>(A, lds, stream);
}

template <int MODE>
static int dispatch_ne_plain(const BwdArgs &A, size_t lds, hipStream_t stream) {
    switch (A.D / 64) {
        case 4: return launch_bwd_plain<MODE, 4>(A, lds, stream);
        case 8: return launch_bwd_plain<MODE, 8>(A, lds, stream);
        default: return launch_bwd_plain<MODE, 12>(A, lds, stream);
    }
}

// ---- grouping of the token positions by (clamped) token id: a counting sort in three small kernels.
// bwd_rank_kernel: a workgroup sorts (token << 11 | index) for 2048 positions in LDS (bitonic), so equal tokens become
// runs; the head of a run reserves the run's places in the token's group with ONE atomicAdd(counts[token], length)
// (a hot token costs one atomic per workgroup, not one per occurrence) and every position gets its rank in the group.
// bwd_scan_kernel: group starts.  bwd_place_kernel: pos_sorted[start[token] + rank] = position (no atomics).
constexpr int kRankThreads = 512;
__device__ __forceinline__ int lower_bound_u32(const uint32_t *a, int n, uint32_t v) {   // first index with a[i] >= v
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(kThreads) void zero_i32_kernel(int32_t *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) p[i] = 0;
}

int launch_zero_words(void *p, int64_t n_words, hipStream_t stream) {
    if (n_words <= 0) return MOT_OK;
    int64_t blocks = (n_words + kThreads - 1) / kThreads;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, (int32_t *)p, n_words);
    return check_launch("zero_i32_kernel");
}

template <int kRankChunk>   // positions per workgroup: 2048, or 512 when there are too few positions to fill the chip with 2048s
__global__ __launch_bounds__(kRankThreads) void bwd_rank_kernel(const int32_t *__restrict__ tokens, int64_t n, int64_t rows,
                                                                int32_t *__restrict__ counts, int32_t *__restrict__ rank,
                                                                uint32_t *status) {
    __shared__ uint32_t skey[kRankChunk];
    __shared__ int32_t runbase[kRankChunk];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kRankChunk;
    for (int i = tid; i < kRankChunk; i += kRankThreads) {
        uint32_t key = 0xffffffffu;
        if (base + i < n) {
            uint32_t t = (uint32_t)tokens[base + i];
            if ((uint64_t)t >= (uint64_t)rows) { if (status) atomicOr(status, kStatusTokenOor); t = 0; }
            key = (t << 11) | (uint32_t)i;
        }
        skey[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= kRankChunk; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < kRankChunk / 2; p += kRankThreads) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = i | j;
                const uint32_t x = skey[i], y = skey[ixj];
                if ((x > y) == ((i & k) == 0)) { skey[i] = y; skey[ixj] = x; }
            }
            __syncthreads();
        }
    constexpr int kPer = kRankChunk / kRankThreads;
    int head[kPer];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
        const int si = tid + r * kRankThreads;
        const uint32_t key = skey[si];
        head[r] = -1;
        if (key == 0xffffffffu) continue;
        const uint32_t tok = key >> 11;
        const int h = (si == 0 || (skey[si - 1] >> 11) != tok) ? si : lower_bound_u32(skey, si, tok << 11);
        head[r] = h;
        if (h == si) {
            const int e = lower_bound_u32(skey, kRankChunk, (tok + 1) << 11);   // padding keys are larger than any token's
            runbase[si] = atomicAdd(&counts[tok], e - si);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
        const int si = tid + r * kRankThreads;
        if (head[r] < 0) continue;
        rank[base + (skey[si] & 2047)] = runbase[head[r]] + (si - head[r]);
    }
}

__global__ __launch_bounds__(kThreads) void bwd_place_kernel(const int32_t *__restrict__ tokens, int64_t n, int64_t rows,
                                                             const int32_t *__restrict__ starts, const int32_t *__restrict__ rank,
                                                             int32_t *__restrict__ pos_sorted, int32_t *__restrict__ tok_sorted) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        int t = tokens[i];
        if ((uint64_t)(uint32_t)t >= (uint64_t)rows) t = 0;
        const int32_t at = starts[t] + rank[i];
        pos_sorted[at] = (int32_t)i;
        tok_sorted[at] = t;
    }
}

// exclusive scan of counts[0..rows) into starts.  Workgroup b owns the 1024 counts of tile b: it first sums everything in
// front of its tile (coalesced reads of an L2-resident array, at most a few hundred KB), then scans its own tile.
__global__ __launch_bounds__(1024) void bwd_scan_kernel(const int32_t *__restrict__ counts, int64_t rows,
                                                        int32_t *__restrict__ starts) {
    __shared__ int32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * 1024;
    int32_t before = 0;
    for (int64_t i = tid; i < t0; i += 1024) before += counts[i];
    const int64_t i = t0 + tid;
    const int32_t own = i < rows ? counts[i] : 0;
    const int32_t incl = wave_incl_add(own, lane);
    const int32_t bsum = wave_incl_add(before, lane);
    if (lane == 63) wsum[wave] = incl + bsum;     // this wave's share of (everything before the tile + the tile)
    __syncthreads();
    int32_t off = incl - own;
    for (int w = 0; w < 16; ++w) off += w < wave ? wsum[w] : 0;
    // the `before` parts of the later waves belong in front of every element of the tile as well
    __shared__ int32_t bpart[16];
    if (lane == 63) bpart[wave] = bsum;
    __syncthreads();
    for (int w = wave; w < 16; ++w) off += bpart[w];
    if (i < rows) starts[i] = off;
}

template <int MODE, int NE>
static int launch_bwd(const BwdArgs &A, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> lds_ok{0};   // per-device bits
    if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mix_bwd_kernel<MODE, NE>, lds_ok, "embed_mix_bwd_kernel")) return rc_lds;
    int64_t blocks = ((A.n_tokens + kWindow - 1) / kWindow + kBwdWaves - 1) / kBwdWaves;
    if (blocks > 256) blocks = 256;  // one persistent workgroup per CU
    hipLaunchKernelGGL((embed_mix_bwd_kernel<MODE, NE>), dim3((unsigned)blocks), dim3(kBwdThreads), lds, stream, A);
    return check_launch("embed_mix_bwd_kernel");
}

template <int MODE, int NE, bool BF>
static int launch_bwd_full_t(const BwdArgs &A, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> lds_ok{0};   // per-device bits
    if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mix_bwd_full_kernel<MODE, NE, BF>, lds_ok, "embed_mix_bwd_full_kernel")) return rc_lds;
    int64_t blocks = (A.n_tokens + 16 * kBwdWaves - 1) / (16 * kBwdWaves);   // >= 16 sorted positions per wave
    if (blocks > 256) blocks = 256;   // one workgroup per CU
    hipLaunchKernelGGL((embed_mix_bwd_full_kernel<MODE, NE, BF>), dim3((unsigned)blocks), dim3(kBwdThreads), lds, stream, A);
    return check_launch("embed_mix_bwd_full_kernel");
}

template <int MODE, int NE>
static int launch_bwd_full(const BwdArgs &A, size_t lds, hipStream_t stream) {
    if constexpr (MODE != MOT_MIX_CONCAT_LINEAR) {
        if (A.in_bf16) return launch_bwd_full_t<MODE, NE, true>(A, lds, stream);
    } else if (A.in_bf16) {
        return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: the CONCAT_LINEAR scatter takes fp32 operands");
    }
    return launch_bwd_full_t<MODE, NE, false>(A, lds, stream);
}

// rows that are "full" (see embed_mix_bwd_full_kernel): SUM / NOOP, D a multiple of 64 with a built NE
template <int MODE>
static bool full_layout(const BwdArgs &A) {
    if (A.D & 63) return false;
    if (MODE == MOT_MIX_CONCAT_LINEAR) {   // split row: every 64-element chunk wholly token part or wholly byte part
        if ((A.Dt & 63) || (A.tok_lo & 63) || (A.byte_lo & 63) || (A.nbk & 63) || A.Dt + A.nbk != A.D || A.Db > 0xffff || A.in_bf16) return false;
    } else {
        if (A.Dt != A.D || A.tok_lo != 0) return false;
        if (MODE == MOT_MIX_SUM && (A.byte_lo != 0 || A.nbk != A.D || A.Db > 0xffff)) return false;
        if (A.in_bf16 && ((A.D & 127) || (MODE == MOT_MIX_SUM && (A.Db & 1)))) return false;   // bf16 rows are read as pairs
    }
    if (A.abl & 4) return false;   // abl 4: dev switch back to the general kernel
    const int ne = A.D / 64;
    return ne == 1 || ne == 2 || ne == 4 || ne == 8 || ne == 12 || ne == 16 || ne == 24 || ne == 32;
}

template <int MODE>
static int dispatch_ne_full(const BwdArgs &A, size_t lds, hipStream_t stream) {
    switch (A.D / 64) {
        case 1: return launch_bwd_full<MODE, 1>(A, lds, stream);
        case 2: return launch_bwd_full<MODE, 2>(A, lds, stream);
        case 4: return launch_bwd_full<MODE, 4>(A, lds, stream);
        case 8: return launch_bwd_full<MODE, 8>(A, lds, stream);
        case 12: return launch_bwd_full<MODE, 12>(A, lds, stream);
        case 16: return launch_bwd_full<MODE, 16>(A, lds, stream);
        case 24: return launch_bwd_full<MODE, 24>(A, lds, stream);
        default: return launch_bwd_full<MODE, 32>(A, lds, stream);
    }
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

// ------------------------------------------------------------------------------------------
// scatter stage shared by all modes: sort the positions by token id, then embed_mix_bwd_kernel.
// `ws_ints` = [counts: tok_rows][starts: tok_rows][rank: N][pos_sorted: N][tok_sorted: N] (int32).
// ------------------------------------------------------------------------------------------
// The counting sort by itself: positions 0..n-1 grouped by ids[position] (clamped into [0, rows)); `ws_ints` holds
// group_positions_ws_ints(n, rows) int32.  *pos_sorted / *id_sorted point into it.
size_t group_positions_ws_ints(int64_t n, int64_t rows) { return 2 * (size_t)rows + 3 * (size_t)n; }
int launch_group_positions(const int32_t *ids, int64_t n, int64_t rows, int32_t *ws_ints, const int32_t **pos_sorted_out, const int32_t **id_sorted_out,
                           uint32_t *status, hipStream_t stream) {
    if (rows >= (1 << 21) - 1) return set_error(MOT_EUNSUPPORTED, "group_positions: %lld rows (>= 2^21 - 1) are not built", (long long)rows);
    int32_t *counts = ws_ints, *starts = counts + rows, *rank = starts + rows, *pos_sorted = rank + n, *id_sorted = pos_sorted + n;
    int rc;
    if ((rc = launch_zero_words(counts, rows, stream))) return rc;
    const int rank_chunk = n >= 256 * 2048 ? 2048 : 512;
    const int64_t rb = (n + rank_chunk - 1) / rank_chunk;
    int64_t pb = (n + kThreads - 1) / kThreads;
    if (pb > 2048) pb = 2048;
    if (rb > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "group_positions: too many positions");
    if (n > 0) {
        if (rank_chunk == 2048)
            hipLaunchKernelGGL(bwd_rank_kernel<2048>, dim3((unsigned)rb), dim3(kRankThreads), 0, stream, ids, n, rows, counts, rank, status);
        else
            hipLaunchKernelGGL(bwd_rank_kernel<512>, dim3((unsigned)rb), dim3(kRankThreads), 0, stream, ids, n, rows, counts, rank, status);
        hipLaunchKernelGGL(bwd_scan_kernel, dim3((unsigned)((rows + 1023) / 1024)), dim3(1024), 0, stream, counts, rows, starts);
        hipLaunchKernelGGL(bwd_place_kernel, dim3((unsigned)pb), dim3(kThreads), 0, stream, ids, n, rows, starts, rank, pos_sorted, id_sorted);
        if ((rc = check_launch("group_positions kernels"))) return rc;
    }
    *pos_sorted_out = pos_sorted;
    *id_sorted_out = id_sorted;
    return MOT_OK;
}

static size_t scatter_ws_ints(const MotEmbedMixDesc &d) { return 2 * (size_t)d.tok_rows + 3 * (size_t)(d.n_rows * d.tokens_per_row); }

template <int MODE>
static int run_scatter(BwdArgs &A, const MotEmbedMixDesc &d, int32_t *ws_ints, float *rnorm_ws, hipStream_t stream) {
    int rc;
    A.abl = 0;
#ifdef MOT_DEV_ABLATION
    if (getenv("MOT_BWD_ABL")) A.abl = atoi(getenv("MOT_BWD_ABL"));
#endif
    if (d.tok_rows >= (1 << 21) - 1)   // (token << 11 | index) of bwd_rank_kernel must stay below its 0xffffffff padding key
        return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: token tables of %lld rows (>= 2^21 - 1) are not built", (long long)d.tok_rows);
    const bool full = full_layout<MODE>(A) && !A.g_ld;   // (a row stride is known to the lane-contiguous kernel and to the general one)
    bool lc = false, plain = false;
    if constexpr (MODE != MOT_MIX_CONCAT_LINEAR) {
        lc = lc_layout<MODE>(A) && !(A.abl & 8);   // abl 8: dev switch back to the strided kernels
        plain = plain_layout<MODE>(A) && !(A.abl & 24);   // abl 16: dev switch back to the general kernels
        lc = lc || plain;                           // (LDS layout below: one fp32 row per wave, byte-table rows padded by one sum)
    }
    // LDS besides the privatised byte-table sums: per-wave per-slot accumulators (strided kernels) or per-wave transposition rows (lane-contiguous)
    size_t lds = lc ? (size_t)(plain ? kPlWaves : kLcWaves) * A.D * sizeof(float) + 16 : 2 * (size_t)kBwdWaves * kMaxBpt * sizeof(float) + 16;
    // the positions grouped by token id: given by the caller (mot_token_order, same layout as the workspace) or made here
    if (A.pos_sorted == nullptr || A.tok_sorted == nullptr)
        if ((rc = launch_group_positions(A.tokens, A.n_tokens, A.tok_rows, ws_ints, &A.pos_sorted, &A.tok_sorted, A.status, stream))) return rc;
#ifdef MOT_DEV_ABLATION
    if (getenv("MOT_BWD_SORT_ONLY")) return MOT_OK;  // dev: inspect the sort prologue's workspace from the host
#endif
    A.priv_lo = 0; A.priv_hi0 = (int)d.byte_rows; A.priv_rows = 0;
    if (MODE != MOT_MIX_NOOP) {
        // as many byte-table rows as 150 KiB of LDS hold at 8 bytes per element; when not all fit, the last 32 rows
        // (pad / eot and other specials sit at the end of the byte vocabulary) and the first cap-32
        const size_t row_q = (size_t)d.byte_dim + (lc ? 1 : 0);   // the lane-contiguous kernel pads its rows by one sum
        const int64_t cap = (int64_t)((150 * 1024 - lds) / (row_q * 8));
        if (cap >= d.byte_rows) { A.priv_lo = (int)d.byte_rows; A.priv_rows = (int)d.byte_rows; }
        else if (cap > 64) { A.priv_lo = (int)cap - 32; A.priv_hi0 = (int)d.byte_rows - 32; A.priv_rows = (int)cap; }
        lds += (size_t)A.priv_rows * row_q * 8 + 8;
        if (d.norm_byte && !(MODE == MOT_MIX_CONCAT_LINEAR && d.ids_b)) {
            rc = launch_rows_rnorm(A.byte_table, d.byte_rows, d.byte_dim, A.eps, rnorm_ws, A.in_bf16 ? MOT_BF16 : MOT_F32, stream);
            if (rc) return rc;
            A.byte_rnorm = rnorm_ws;
        }
    }
    if constexpr (MODE != MOT_MIX_CONCAT_LINEAR) {
        if (plain) return dispatch_ne_plain<MODE>(A, lds, stream);
        if (lc) return dispatch_ne_lc<MODE>(A, lds, stream);
    }
    if (full) return dispatch_ne_full<MODE>(A, lds, stream);
    return dispatch_ne<MODE>(A, lds, stream);
}

static void fill_bwd_args(BwdArgs &A, const MotEmbedMixDesc &d, const MotEmbedMixGrads &gr) {
    A.tokens = d.tokens; A.n_tokens = d.n_rows * d.tokens_per_row; A.bpt = d.bpt;
    A.ids_a = d.ids_a; A.ids_b = d.ids_b;
    A.tok_table = (const float *)d.tok_table; A.tok_rows = d.tok_rows; A.D = d.tok_dim;
    A.byte_table = (const float *)d.byte_table; A.byte_rows = d.byte_rows; A.Db = d.byte_dim;
    A.norm_tok = d.norm_tok; A.norm_byte = d.norm_byte; A.norm_out = d.norm_out;
    A.in_bf16 = d.dtype == MOT_BF16;
    A.eps = d.eps > 0.f ? d.eps : (A.in_bf16 ? kBf16Eps : FLT_EPSILON);
    A.scale_tok = d.scale_tok; A.scale_byte = d.scale_byte; A.byte_rnorm = nullptr;
    A.grad_out = (const float *)gr.grad_out;
    A.d_tok = (float *)gr.d_tok_table; A.d_byte = (float *)gr.d_byte_table;
    A.d_scale_tok = gr.d_scale_tok; A.d_scale_byte = gr.d_scale_byte;
    A.status = d.status;
    A.pos_sorted = A.tok_sorted = nullptr;
    A.g_ld = 0; A.no_tok = 0; A.slot0 = 0;
    if (gr.token_order) {   // [counts: rows][starts: rows][rank: n][pos_sorted: n][tok_sorted: n], as launch_group_positions lays it out
        const int64_t n = d.n_rows * d.tokens_per_row;
        A.pos_sorted = gr.token_order + 2 * d.tok_rows + n;
        A.tok_sorted = A.pos_sorted + n;
    }
    A.Dt = d.tok_dim; A.tok_lo = 0; A.byte_lo = 0; A.nbk = d.bpt * d.byte_dim;
}

// ==========================================================================================
// CONCAT_LINEAR backward:  x = rms_norm?(y), y = W u + bias, u = cat(a, b_*)
//   dy = r_y (g - x mean(g x))                      dy_kernel (one wave per row)
//   du = dy . W          (N x Dm) @ (Dm x K)        the forward MFMA kernel with dy as dense "token rows";
//                                                   W in nn.Linear layout IS the k-major operand it wants
//   dW += dy^T . u       (Dm x N) @ (N x K)         gemm_tn_kernel: split over tokens, fp32 MFMA, atomic accumulate;
//                                                   u = the seam tensors (gather_rows with the norms/scales applied)
//   dbias += colsum(dy)                             colsum_kernel
//   table gradients: the scatter stage above on du (row layout = the concat layout)
// ==========================================================================================
__global__ __launch_bounds__(kThreads) void dy_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                      const float *__restrict__ rnorm, int64_t n, int Dm, float *__restrict__ dy) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= n) return;
    const float *gr = g + r * Dm, *xr = x + r * Dm;
    float m = 0.f;
    for (int j = lane; j < Dm; j += 64) m += gr[j] * xr[j];
    m = wave_sum(m) / (float)Dm;
    const float ry = rnorm[r];
    for (int j = lane; j < Dm; j += 64) dy[r * Dm + j] = ry * (gr[j] - xr[j] * m);
}

// the same from bf16 g and x, result in bf16 (the bf16 route never needs an fp32 dy): Dm a multiple of 8, <= 4096
__global__ __launch_bounds__(kThreads) void dy16_kernel(const __bf16 *__restrict__ g, const __bf16 *__restrict__ x, const float *__restrict__ rnorm, int64_t n,
                                                        int Dm, __bf16 *__restrict__ dy) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= n) return;
    const __bf16 *gr = g + r * Dm, *xr = x + r * Dm;
    float8v gv[8], xv[8];
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = 8 * (lane + 64 * i);
        gv[i] = (float8v)(0.f); xv[i] = (float8v)(0.f);
        if (c < Dm) {
            gv[i] = Elem<__bf16>::loadv(gr + c);
            xv[i] = Elem<__bf16>::loadv(xr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) m += gv[i][e] * xv[i][e];
        }
    }
    m = wave_sum(m) / (float)Dm;
    const float ry = rnorm[r];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = 8 * (lane + 64 * i);
        if (c < Dm) Elem<__bf16>::storev_nt(dy + r * Dm + c, (gv[i] - xv[i] * m) * ry);
    }
}

__global__ __launch_bounds__(kThreads) void colsum_kernel(const float *__restrict__ a, int64_t n, int cols, float *__restrict__ out) {
    // each workgroup sums a strip of rows for every column, then one atomic per column
    const int64_t rows_per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * rows_per, hi = min(n, lo + rows_per);
    for (int c = threadIdx.x; c < cols; c += kThreads) {
        float s = 0.f;
        for (int64_t r = lo; r < hi; ++r) s += a[r * cols + c];
        if (lo < hi) atomicAdd(out + c, s);
    }
}

__global__ __launch_bounds__(kThreads) void iota_kernel(int32_t *p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) p[i] = (int32_t)i;
}

__global__ __launch_bounds__(kThreads) void pad_copy_kernel(const float *__restrict__ src, int rows, int cols, float *__restrict__ dst,
                                                            int rows_pad, int cols_pad) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < (int64_t)rows_pad * cols_pad; i += (int64_t)gridDim.x * kThreads) {
        const int r = (int)(i / cols_pad), c = (int)(i - (int64_t)r * cols_pad);
        dst[i] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
    }
}

int launch_pad_copy(const float *src, int rows, int cols, float *dst, int rows_pad, int cols_pad, hipStream_t stream) {
    hipLaunchKernelGGL(pad_copy_kernel, dim3(512), dim3(kThreads), 0, stream, src, rows, cols, dst, rows_pad, cols_pad);
    return check_launch("pad_copy_kernel");
}

// C[j][k] += sum_n A[n][j] * B[n][k]   (A: n x M, B: n x Nc, C: M x Nc with leading dimension ldc), fp32 MFMA.
// Workgroup = 128 x 128 output block (4 waves as 2 x 2, each 64 x 64 = 2 x 2 tiles of 32 x 32) over one
// slice of the rows; 16 rows per step, double-buffered LDS, rows ARE the MFMA k index so both operands are
// staged in their natural row-major layout.  Partial blocks are accumulated with float atomics
// (128-byte contiguous segments per instruction).
typedef float f32x16b __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(kThreads) void gemm_tn_kernel(const float *__restrict__ A_, int lda, int M, const float *__restrict__ B_, int ldb,
                                                           int Nc, int64_t n, int64_t rows_per_split, float *__restrict__ C, int ldc) {
    __shared__ __attribute__((aligned(16))) float lA[2][16 * 128], lB[2][16 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
    const int j0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
    const int64_t r0 = (int64_t)blockIdx.z * rows_per_split, r1 = min(n, r0 + rows_per_split);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16b acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // staging role: 16 rows x 32 float4 per operand = 512 float4 -> 2 per thread
    float4v ra[2], rb[2];
    const bool va = (lda & 3) == 0 && ((uintptr_t)A_ & 15) == 0, vb = (ldb & 3) == 0 && ((uintptr_t)B_ & 15) == 0;   // 16-byte loads allowed
    auto load_stage = [&](int64_t r) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int q = p * kThreads + tid, row = q >> 5, c4 = (q & 31) * 4;
            const int64_t rr = r + row;
            ra[p] = (float4v)(0.f); rb[p] = (float4v)(0.f);
            if (rr < r1) {
                const float *pa = A_ + rr * lda + j0 + c4, *pb = B_ + rr * ldb + k0 + c4;
                if (va && j0 + c4 + 3 < M) ra[p] = *(const float4v *)pa;
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e)   // element-wise guards keep ragged right edges correct
                        if (j0 + c4 + e < M) ra[p][e] = pa[e];
                if (vb && k0 + c4 + 3 < Nc) rb[p] = *(const float4v *)pb;
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k0 + c4 + e < Nc) rb[p][e] = pb[e];
            }
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int q = p * kThreads + tid;
            *(float4v *)(&lA[buf][q * 4]) = ra[p];
            *(float4v *)(&lB[buf][q * 4]) = rb[p];
        }
    };
    load_stage(r0);
    store_stage(0);
    __syncthreads();
    int buf = 0;
    for (int64_t r = r0; r < r1; r += 16, buf ^= 1) {
        const bool more = r + 16 < r1;
        if (more) load_stage(r + 16);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float a0 = lA[buf][(kk + h) * 128 + wm + li], a1 = lA[buf][(kk + h) * 128 + wm + 32 + li];
            const float b0 = lB[buf][(kk + h) * 128 + wn + li], b1 = lB[buf][(kk + h) * 128 + wn + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) store_stage(buf ^ 1);
        __syncthreads();
    }
    // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); A is the "row" (j) operand
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, k = k0 + wn + b * 32 + li;
                if (j < M && k < Nc) atomicAdd(C + (int64_t)j * ldc + k, acc[a][b][r]);
            }
}

int launch_gemm_tn(const float *A_, int lda, int M, const float *B_, int ldb, int Nc, int64_t n, float *C, int ldc, hipStream_t stream) {
    if (M <= 0 || Nc <= 0 || n <= 0) return MOT_OK;
    const int gx = (M + 127) / 128, gy = (Nc + 127) / 128;
    int64_t splits = (1024 + gx * gy - 1) / (gx * gy);                  // ~1024 workgroups in total
    int64_t rows_per = ((n + splits - 1) / splits + 15) / 16 * 16;     // whole 16-row steps
    if (rows_per < 256) rows_per = 256;
    splits = (n + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)splits), dim3(kThreads), 0, stream, A_, lda, M, B_, ldb,
                       Nc, n, rows_per, C, ldc);
    return check_launch("gemm_tn_kernel");
}

// C[n][c] = sum_r A[n][r] * (BT ? B[c][r] : B[r][c])   (A: n x R rows, C: n x Nc; leading dimensions lda / ldb / ldc), fp32 MFMA.
// Same block shape and inner loop as gemm_tn_kernel -- 128 x 128 output block, 16 reduction indices per step, both operands in
// LDS as [reduction index][block row / column], one conflict-free ds_read_b32 per MFMA operand -- with the operand whose rows
// are contiguous along r (A always, B when BT) transposed on its way into LDS: a lane takes 4 consecutive r of one row, 4
// lanes one 64-byte row segment, and writes them as four ds_write_b32 down a padded column.  The reduction is whole
// inside the workgroup (plain stores).  The fused gather + norm kernel (mot_linear.hip) runs its dense-row mode at 48 % of
// the fp32 MFMA peak; this loop reaches ~75 %.
template <bool BT>
__device__ __forceinline__ void gemm_rows_body(const float *__restrict__ A_, int lda, int64_t n, const float *__restrict__ B_, int ldb,
                                               int R, int Nc, float *__restrict__ C, int ldc, const float *__restrict__ bias, int accumulate,
                                               bool plain_order = false) {
    // transposed operands sit in LDS with a row stride of 132 floats: the 4 lanes that share a source row (coalesced 64-byte
    // reads) then write to banks 16 apart, two lanes per bank -- the minimum for 64 dword writes
    constexpr int LDA = 132, LDB = BT ? 132 : 128;
    __shared__ __attribute__((aligned(16))) float lA[2][16 * LDA], lB[2][16 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
    // XCD-aware block order (1-D grid, workgroup ids round-robin over the 8 XCDs): an XCD walks all column blocks of a row panel
    // back to back, so the panel of A is fetched into that XCD's L2 once instead of once per column block
    const int64_t gx = (n + 127) / 128;
    const int gy = (Nc + 127) / 128;
    // (plain_order: gx * gy blocks, no padding -- the sliced few-row launches, where the padded panels were most of the workgroups and
    //  their dispatch most of the time: 132 rows = 2 panels padded to 8, 2816 workgroups of which 704 work, 100 us)
    const int64_t bid = blockIdx.x, seq = plain_order ? bid : bid >> 3, panel = plain_order ? bid / gy : (seq / gy) * 8 + (bid & 7);
    if (panel >= gx) return;   // the grid is padded to whole groups of 8 panels
    const int64_t j0 = panel * 128;
    const int k0 = (int)(seq % gy) * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    // acc is the MFMA accumulator of kFold reduction steps at a time; it is then folded into `sum` with vector adds and
    // restarted, so no fp32 summation chain is longer than 8 * kFold MFMA steps (blocked summation, like the reference's
    // CPU sgemm: one chain over K = 768 ends up 4x as far from the float64 result as the reference, the parity bar is 2x)
    constexpr int kFold = 8;
    f32x16b acc[2][2], sum[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; sum[a][b][r] = 0.f; }
    const bool va = (lda & 3) == 0 && ((uintptr_t)A_ & 15) == 0, vb = (ldb & 3) == 0 && ((uintptr_t)B_ & 15) == 0;
    float4v ra[2], rb[2];
    // rows-contiguous-along-r operand: thread -> (row = q >> 2, 4 consecutive r starting at (q & 3) * 4): 4 lanes read one 64-byte row segment
    auto load_t = [&](const float *P, int ld, int64_t row0, int64_t rows, bool vec, int r, float4v (&dst)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int q = p * kThreads + tid, row = q >> 2, c4 = (q & 3) * 4;
            dst[p] = (float4v)(0.f);
            if (row0 + row < rows) {
                const float *src = P + (row0 + row) * ld + r + c4;
                if (vec && r + c4 + 3 < R) dst[p] = *(const float4v *)src;
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (r + c4 + e < R) dst[p][e] = src[e];
            }
        }
    };
    auto store_t = [&](float *L, const float4v (&srcv)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int q = p * kThreads + tid, row = q >> 2, c4 = (q & 3) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) L[(c4 + e) * 132 + row] = srcv[p][e];
        }
    };
    auto load_stage = [&](int r) {
        load_t(A_, lda, j0, n, va, r, ra);
        if (BT) {
            load_t(B_, ldb, k0, Nc, vb, r, rb);
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p) {   // natural layout: 16 reduction rows x 32 float4
                const int q = p * kThreads + tid, row = q >> 5, c4 = (q & 31) * 4;
                rb[p] = (float4v)(0.f);
                if (r + row < R) {
                    const float *src = B_ + (int64_t)(r + row) * ldb + k0 + c4;
                    if (vb && k0 + c4 + 3 < Nc) rb[p] = *(const float4v *)src;
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (k0 + c4 + e < Nc) rb[p][e] = src[e];
                }
            }
        }
    };
    auto store_stage = [&](int buf) {
        store_t(lA[buf], ra);
        if (BT) store_t(lB[buf], rb);
        else
#pragma unroll
            for (int p = 0; p < 2; ++p) *(float4v *)(&lB[buf][(p * kThreads + tid) * 4]) = rb[p];
    };
    load_stage(0);
    store_stage(0);
    __syncthreads();
    int buf = 0;
    for (int r = 0; r < R; r += 16, buf ^= 1) {
        const bool more = r + 16 < R;
        if (more) load_stage(r + 16);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float a0 = lA[buf][(kk + h) * LDA + wm + li], a1 = lA[buf][(kk + h) * LDA + wm + 32 + li];
            const float b0 = lB[buf][(kk + h) * LDB + wn + li], b1 = lB[buf][(kk + h) * LDB + wn + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (((r >> 4) & (kFold - 1)) == kFold - 1) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 16; ++q) { sum[a][b][q] += acc[a][b][q]; acc[a][b][q] = 0.f; }
        }
        if (more) store_stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t j = j0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int k = k0 + wn + b * 32 + li;
                if (j < n && k < Nc) {
                    float v = sum[a][b][r] + acc[a][b][r];
                    if (bias) v += bias[k];
                    if (accumulate) v += C[j * ldc + k];
                    C[j * ldc + k] = v;
                }
            }
}

// The register budget is set per variant: with B transposed the body fits 168 registers (3 waves per SIMD); with B in its
// natural layout that cap spills inside the loop (0.92 ms instead of 0.72), so that variant runs at 2 waves per SIMD.
// (blockIdx.y = slice of the reduction, `slice` indices long, whose block goes to C + y * part_stride: launch_gemm_rows_sliced)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 4))) void gemm_rows_bt_kernel(
    const float *__restrict__ A_, int lda, int64_t n, const float *__restrict__ B_, int ldb, int R, int Nc, float *__restrict__ C, int ldc,
    const float *__restrict__ bias, int accumulate, int slice, int64_t part_stride) {
    const int r0 = blockIdx.y * slice;
    gemm_rows_body<true>(A_ + r0, lda, n, B_ + r0, ldb, slice ? min(slice, R - r0) : R, Nc, C + blockIdx.y * part_stride, ldc, bias, accumulate, slice != 0);
}
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 4))) void gemm_rows_kernel(
    const float *__restrict__ A_, int lda, int64_t n, const float *__restrict__ B_, int ldb, int R, int Nc, float *__restrict__ C, int ldc,
    const float *__restrict__ bias, int accumulate, int slice, int64_t part_stride) {
    const int r0 = blockIdx.y * slice;
    gemm_rows_body<false>(A_ + r0, lda, n, B_ + (int64_t)r0 * ldb, ldb, slice ? min(slice, R - r0) : R, Nc, C + blockIdx.y * part_stride, ldc, bias, accumulate,
                          slice != 0);
}
// C[i] = part[0][i] + part[1][i] + ... in that order (C rows ldc apart, the partial blocks dense [n][Nc])
__global__ __launch_bounds__(kThreads) void gemm_rows_sum_slices_kernel(const float *__restrict__ part, int slices, int64_t n, int Nc, float *__restrict__ C, int ldc) {
    const int64_t total = n * Nc;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        float v = part[i];
        for (int s = 1; s < slices; ++s) v += part[(int64_t)s * total + i];
        C[(i / Nc) * ldc + i % Nc] = v;
    }
}

// (the reduction can be cut into launches that add to C; with the in-kernel blocked summation one launch covers any K)
constexpr int kGemmRowsPass = 1 << 30;
int launch_gemm_rows(const float *A_, int lda, int64_t n, const float *B_, int ldb, int R, int Nc, float *C, int ldc, bool b_transposed,
                     hipStream_t stream, const float *bias, bool accumulate) {
    if (n <= 0 || Nc <= 0) return MOT_OK;
    if (b_transposed && R > 0 && gemm_rows_f32_256_usable(A_, lda, n, B_, ldb, R, Nc))   // 256 x 256 blocks by LDS-DMA (mot_gemm_bf16.hip)
        return launch_gemm_rows_f32_256(A_, lda, n, B_, ldb, R, Nc, C, ldc, bias, accumulate, stream);
    const int64_t gx = (n + 127) / 128;
    const int gy = (Nc + 127) / 128;
    const int64_t blocks = (gx + 7) / 8 * 8 * gy;   // 1-D, see the block order in the kernel
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "gemm_rows: too many rows");
    for (int r0 = 0; r0 < R || r0 == 0; r0 += kGemmRowsPass) {
        const int rn = R - r0 < kGemmRowsPass ? R - r0 : kGemmRowsPass;
        const float *a = A_ + r0, *b = b_transposed ? B_ + r0 : B_ + (int64_t)r0 * ldb;
        if (b_transposed)
            hipLaunchKernelGGL(gemm_rows_bt_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, a, lda, n, b, ldb, rn, Nc, C, ldc,
                               r0 ? nullptr : bias, (r0 || accumulate) ? 1 : 0, 0, (int64_t)0);
        else
            hipLaunchKernelGGL(gemm_rows_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, a, lda, n, b, ldb, rn, Nc, C, ldc,
                               r0 ? nullptr : bias, (r0 || accumulate) ? 1 : 0, 0, (int64_t)0);
        if (r0 + kGemmRowsPass >= R) break;
    }
    return check_launch("gemm_rows_kernel");
}

// how the reduction of a few-row product is cut: slices of a multiple of 16 indices, at least 64, as many as fill the chip
static int gemm_rows_slices(int64_t n, int R, int Nc, int *slice_len) {
    const int64_t blocks = ((n + 127) / 128 + 7) / 8 * 8 * ((Nc + 127) / 128), live = ((n + 127) / 128) * ((Nc + 127) / 128);
    if (n > 1024 || R < 256 || live >= 128 || blocks > 4096) return 1;
    int want = (int)(768 / live);   // (three workgroups a CU: a step of this kernel is one exposed load latency, ~5 us of it per step measured)
    if (want > R / 64) want = R / 64;
    if (want > 32) want = 32;
    if (want < 2) return 1;
    const int len = ((R + want - 1) / want + 15) / 16 * 16;
    *slice_len = len;
    return (R + len - 1) / len;
}
size_t gemm_rows_sliced_floats(int64_t n, int R, int Nc) {
    int len = 0;
    const int s = gemm_rows_slices(n, R, Nc, &len);
    return s > 1 ? (size_t)s * n * Nc : 0;
}
int launch_gemm_rows_sliced(const float *A_, int lda, int64_t n, const float *B_, int ldb, int R, int Nc, float *C, int ldc, bool b_transposed, float *part,
                            size_t part_floats, hipStream_t stream) {
    int len = 0;
    const int slices = n > 0 && Nc > 0 ? gemm_rows_slices(n, R, Nc, &len) : 1;
    if (slices < 2 || !part || part_floats < (size_t)slices * n * Nc) return launch_gemm_rows(A_, lda, n, B_, ldb, R, Nc, C, ldc, b_transposed, stream);
    const int64_t blocks = ((n + 127) / 128) * ((Nc + 127) / 128);   // (plain block order in the sliced launches: no padded panels)
    const dim3 grid((unsigned)blocks, (unsigned)slices);
    if (b_transposed)
        hipLaunchKernelGGL(gemm_rows_bt_kernel, grid, dim3(kThreads), 0, stream, A_, lda, n, B_, ldb, R, Nc, part, Nc, (const float *)nullptr, 0, len, n * Nc);
    else
        hipLaunchKernelGGL(gemm_rows_kernel, grid, dim3(kThreads), 0, stream, A_, lda, n, B_, ldb, R, Nc, part, Nc, (const float *)nullptr, 0, len, n * Nc);
    if (int rc = check_launch("gemm_rows_kernel")) return rc;
    const int64_t total = n * Nc;
    hipLaunchKernelGGL(gemm_rows_sum_slices_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads < 2048 ? (total + kThreads - 1) / kThreads : 2048)), dim3(kThreads), 0,
                       stream, part, slices, n, Nc, C, ldc);
    return check_launch("gemm_rows_sum_slices_kernel");
}

// workspace of the CONCAT backward, in floats unless noted:
//   [rnorm: byte_rows][dy: N*Dm][du: N*K][u_tok: N*Dt][u_byte: N*bpt*Db][Wk: Dm16*K128][byte0: 4][iota: N int32][zero ids: 0]
//   [sort ints: 3*tok_rows + N]
struct LinBwdLayout { size_t rnorm, dy, du, utok, ubyte, wk, byte0, iota, sort, total; int Kp, Dmp; };
static LinBwdLayout lin_bwd_layout(const MotEmbedMixDesc &d) {
    LinBwdLayout L;
    const size_t N = (size_t)(d.n_rows * d.tokens_per_row), K = (size_t)d.tok_dim + (size_t)d.bpt * d.byte_dim;
    L.Kp = (int)((K + 127) / 128 * 128);            // output columns of the du GEMM, padded as the MFMA kernel pads them
    if (L.Kp > 512 && L.Kp <= 768) L.Kp = 768; else if (L.Kp > 768) L.Kp = 1024;
    L.Dmp = (d.model_dim + 15) / 16 * 16;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
    L.rnorm = take(d.byte_rows); L.dy = take(N * d.model_dim); L.du = take(N * K); L.utok = take(N * d.tok_dim);
    L.ubyte = take(N * d.bpt * d.byte_dim); L.wk = take((size_t)L.Dmp * L.Kp); L.byte0 = take(4); L.iota = take(N);
    L.sort = take(scatter_ws_ints(d)); L.total = o;
    return L;
}

// bf16 CONCAT_LINEAR backward: the operands are widened once into fp32 workspace copies and the fp32 pipeline
// above runs on them (fp32 MFMA and fp32 accumulation throughout -- never less precise than bf16 autograd;
// the bf16-MFMA version of the three GEMMs is the open item).  Layout in floats, in front of LinBwdLayout.
struct UpLayout { size_t tok, byte, w, bias, g, x, total; };
static UpLayout up_layout(const MotEmbedMixDesc &d) {
    UpLayout U;
    const size_t N = (size_t)(d.n_rows * d.tokens_per_row), K = (size_t)d.tok_dim + (size_t)d.bpt * d.byte_dim;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
    U.tok = take((size_t)d.tok_rows * d.tok_dim); U.byte = take((size_t)d.byte_rows * d.byte_dim); U.w = take((size_t)d.model_dim * K);
    U.bias = take(d.bias ? d.model_dim : 0); U.g = take(N * d.model_dim); U.x = take(d.norm_out ? N * d.model_dim : 0);
    U.total = o;
    return U;
}

__global__ __launch_bounds__(kThreads) void widen_kernel(const __bf16 *__restrict__ src, int64_t n, float *__restrict__ dst) {
    for (int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * kThreads * 8) {
        if (i + 8 <= n && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
            const float8v v = Elem<__bf16>::loadv(src + i);
            *(float4v *)(dst + i) = __builtin_shufflevector(v, v, 0, 1, 2, 3);
            *(float4v *)(dst + i + 4) = __builtin_shufflevector(v, v, 4, 5, 6, 7);
        } else {
            for (int64_t j = i; j < min(n, i + 8); ++j) dst[j] = (float)src[j];
        }
    }
}
static int launch_widen(const void *src, size_t n, float *dst, hipStream_t stream) {
    if (!n) return MOT_OK;
    size_t blocks = (n / 8 + kThreads) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(widen_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, (const __bf16 *)src, (int64_t)n, dst);
    return check_launch("widen_kernel");
}

// du = dy . W of the bf16 backward on the bf16 MFMA (what autograd does for bf16 parameters): dy rounded to bf16,
// W^T as the [K, Dm] "weight" of the forward bf16 kernel in dense-row mode, du widened back for the scatter stage.
// Scratch behind the fp32 layouts, in bytes: [dy16: N*Dm*2][wt16: K*Dm*2][u16: N*K*2].
struct Du16Layout { size_t dy16, wt16, uT, total; };
static Du16Layout du16_layout(const MotEmbedMixDesc &d) {
    Du16Layout U;
    const size_t N = (size_t)(d.n_rows * d.tokens_per_row), K = (size_t)d.tok_dim + (size_t)d.bpt * d.byte_dim, Dm = (size_t)d.model_dim;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    U.dy16 = take(N * Dm * 2); U.wt16 = take(K * Dm * 2);
    U.uT = take(N * K * 2);   // the concat operand in bf16, row-major (dW)
    U.total = o;
    return U;
}
static bool du16_usable(const MotEmbedMixDesc &d) {
    const int K = d.tok_dim + d.bpt * d.byte_dim;
    return d.dtype == MOT_BF16 && (d.model_dim & 7) == 0 && (K & 7) == 0 && K <= 4096 && ((d.n_rows * d.tokens_per_row) & 7) == 0 &&
           !(d.flags & MOT_FLAG_BWD_DU_FP32);
}

__global__ __launch_bounds__(kThreads) void narrow_kernel(const float *__restrict__ src, int64_t n, __bf16 *__restrict__ dst) {
    for (int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * kThreads * 8) {
        if (i + 8 <= n) {   // both buffers are 256-byte aligned workspace regions
            float8v v;
            const float4v a = *(const float4v *)(src + i), b = *(const float4v *)(src + i + 4);
            v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
            Elem<__bf16>::storev_nt(dst + i, v);
        } else {
            for (int64_t j = i; j < n; ++j) dst[j] = (__bf16)src[j];
        }
    }
}


// C[m][k] += sum_n A[n][m] * B[n][k]   (A: rows x M, B: rows x Kc, both bf16 ROW-major as the forward and dy_kernel leave them;
// C fp32, leading dimension ldc) on v_mfma_f32_32x32x16_bf16: dW = dy^T u with the token index as the contraction index.
// Both MFMA operands want 8 consecutive CONTRACTION elements per lane, i.e. a column of the row-major tiles: the tiles go into
// LDS as they are (64 token rows x 128 columns, rows padded to 320 bytes) and are read back with ds_read_b64_tr_b16, the
// transposing LDS read of gfx950 -- a 16-lane group fetches 4 rows x 16 columns and every lane receives ONE column's 4 rows;
// two reads make a lane's 8 contraction elements.  (Round 1 transposed dy and u in HBM first -- narrow_transpose /
// transpose_bf16, 0.2 ms at 65 536 x 768 -- and contracted the token-minor copies: 0.30 ms more, with 85-fold split-k atomics.)
// With 320-byte rows the four rows of a read sit 80 dwords apart: a 32-lane half (two groups, 32 columns) covers all 64 banks once.
// Workgroup tile 128 x 128, 2 x 4 waves as 2 x 2, each 64 x 64; contraction split over blockIdx.z; partial tiles are added with
// float atomics (128-byte contiguous segments).  128 x 128 keeps the split count -- and with it the atomic volume
// (splits x M x Kc x 4 bytes) -- at 8 for 768 x 768 (two workgroups per CU).
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef short s16x4w __attribute__((ext_vector_type(4)));
constexpr int kTnRows = 64, kTnLd = 160;   // token rows per step; elements per staged row (128 data + 32 pad)
constexpr int kTnThreads = 512;            // 8 waves: two per SIMD.  Waves 0-3 and 4-7 are the same 2 x 2 grid of 64 x 64 sub-tiles and split the
                                           // step's four 16-row contraction slices between them; the two partial tiles meet in LDS at the end
__global__ __launch_bounds__(kTnThreads) void gemm_tn_bf16_kernel(const __bf16 *__restrict__ A_, int lda, int M, const __bf16 *__restrict__ B_, int ldb, int Kc,
                                                                  int64_t rows, int64_t rper, int nz, float *__restrict__ C, int ldc) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds_tn[];   // [2][A | B][kTnRows][kTnLd]
    constexpr int kTile = kTnRows * kTnLd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
    // 1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs, so ids 8 g .. 8 g + 7 take the SAME output tile and the
    // contraction slices z = 0 .. 7: all tiles of one slice then run on one XCD and its rows of A and B come out of that XCD's L2
    // (every row is wanted by gx + gy tiles; without this they were fetched once per XCD and tile: 1.2 GB instead of 0.2 at 65 536 x 768)
    const int gx = (M + 127) / 128, tiles = gx * ((Kc + 127) / 128);
    int tile, z;
    if ((nz & 7) == 0) { const int g = blockIdx.x >> 3; tile = g % tiles; z = (blockIdx.x & 7) + 8 * (g / tiles); }
    else { tile = blockIdx.x % tiles; z = blockIdx.x / tiles; }
    const int m0 = (tile % gx) * 128, k0 = (tile / gx) * 128;
    const int64_t r_lo = (int64_t)z * rper, r_hi = min(rows, r_lo + rper);
    const int wm = ((wave >> 1) & 1) * 64, wk = (wave & 1) * 64, ws = wave >> 2;   // ws: which two of the four contraction slices
    f32x16b acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // staging: 64 rows x 16 pieces of 16 bytes per operand = 1024 pieces -> 2 per thread (16 lanes read one 256-byte row segment)
    bf16x8w ra[2], rb[2];
    auto load_stage = [&](int64_t r0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int q = p * kTnThreads + tid, row = q >> 4, c = (q & 15) * 8;
            ra[p] = (bf16x8w)((__bf16)0.f); rb[p] = (bf16x8w)((__bf16)0.f);
            if (r0 + row < r_hi) {   // M, Kc, lda, ldb are multiples of 8: a piece is wholly inside or outside
                if (m0 + c < M) ra[p] = *(const bf16x8w *)(A_ + (r0 + row) * lda + m0 + c);
                if (k0 + c < Kc) rb[p] = *(const bf16x8w *)(B_ + (r0 + row) * ldb + k0 + c);
            }
        }
    };
    auto store_stage = [&](int buf) {
        __bf16 *sA = lds_tn + buf * 2 * kTile, *sB = sA + kTile;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int q = p * kTnThreads + tid, row = q >> 4, c = (q & 15) * 8;
            *(bf16x8w *)(sA + row * kTnLd + c) = ra[p];
            *(bf16x8w *)(sB + row * kTnLd + c) = rb[p];
        }
    };
    // transposed fragment: lane 4 q + p of a 16-lane group g addresses row (r0 + q), columns c0 + 4 p .. + 3 of the group's 4 x 16 block;
    // lane i of the group receives column c0 + i.  Group g: contraction half h = g >> 1, columns 16 (g & 1) .. + 15 of the 32-wide tile.
    const int grp = lane >> 4, gi = lane & 15;
    const int tr_off = ((gi >> 2) + 8 * (grp >> 1)) * kTnLd + 16 * (grp & 1) + 4 * (gi & 3);   // elements, inside a 16-row x 32-column operand block
    auto frag = [&](const __bf16 *tile, int s16, int col0) {   // rows 16 s16 .. + 15 (contraction), columns col0 .. + 31
        const __bf16 *p = tile + (16 * s16) * kTnLd + col0 + tr_off;
        const s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w *)p);
        const s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w *)(p + 4 * kTnLd));
        typedef short s16x8w __attribute__((ext_vector_type(8)));
        const s16x8w v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8w, v);
    };
    if (r_lo < r_hi) {
        load_stage(r_lo);
        store_stage(0);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t r0 = r_lo; r0 < r_hi; r0 += kTnRows, buf ^= 1) {
        const bool more = r0 + kTnRows < r_hi;
        if (more) load_stage(r0 + kTnRows);
        const __bf16 *sA = lds_tn + buf * 2 * kTile, *sB = sA + kTile;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int s16 = 2 * ws + j;
            bf16x8w af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = frag(sA, s16, wm + 32 * a);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = frag(sB, s16, wk + 32 * b);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_stage(buf ^ 1);
        __syncthreads();
    }
    // the second wave group's partial tile joins the first's through LDS (64 KB: the stage buffers are free now)
    float *red = (float *)lds_tn + (size_t)(wave & 3) * 64 * 64;   // [a][b][r][lane] of one 64 x 64 sub-tile
    if (ws == 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((a * 2 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (ws == 1) return;
    // D[i][j]: lane -> j (B column = output column k), registers -> i (A column = output row m)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, k = k0 + wk + b * 32 + li;
                if (m < M && k < Kc) atomicAdd(C + (int64_t)m * ldc + k, acc[a][b][r] + red[((a * 2 + b) * 16 + r) * 64 + lane]);
            }
}

// A, B: 16-byte aligned, lda / ldb / M / Kc multiples of 8
int launch_gemm_tn_bf16(const __bf16 *A_, int lda, int M, const __bf16 *B_, int ldb, int Kc, int64_t rows, float *C, int ldc, hipStream_t stream) {
    if (M <= 0 || Kc <= 0 || rows <= 0) return MOT_OK;
    const int tiles = ((M + 127) / 128) * ((Kc + 127) / 128);
    // contraction slices: two workgroups per CU (80 KB of LDS each), a multiple of 8 for the XCD mapping, few enough to keep the
    // atomic volume (slices x M x Kc x 4 bytes) small
    int64_t splits = (512 / tiles) & ~7;
    if (splits < 8) splits = 8;
    if (splits > 32) splits = 32;
    int64_t rper = ((rows + splits - 1) / splits + kTnRows - 1) / kTnRows * kTnRows;
    if (rper < 4 * kTnRows) rper = 4 * kTnRows;
    splits = (rows + rper - 1) / rper;
    if ((int64_t)tiles * splits > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "gemm_tn_bf16: too many tiles");
    const size_t lds = (size_t)4 * kTnRows * kTnLd * sizeof(__bf16);
    static std::atomic<uint64_t> lds_ok{0};
    if (int rc = ensure_max_dyn_lds((const void *)gemm_tn_bf16_kernel, lds_ok, "gemm_tn_bf16_kernel")) return rc;
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3((unsigned)(tiles * splits)), dim3(kTnThreads), lds, stream, A_, lda, M, B_, ldb, Kc, rows, rper, (int)splits,
                       C, ldc);
    return check_launch("gemm_tn_bf16_kernel");
}

// dst[c][r] = src[r][c]   (rows x cols -> cols x rows), bf16, 32 x 32 tiles through LDS
__global__ __launch_bounds__(kThreads) void transpose_bf16_kernel(const __bf16 *__restrict__ src, int rows, int cols, __bf16 *__restrict__ dst) {
    __shared__ __bf16 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? src[(int64_t)(r0 + r) * cols + c0 + tx] : (__bf16)0.f;
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < cols && r0 + tx < rows) dst[(int64_t)(c0 + c) * rows + r0 + tx] = tile[tx][c];
}

// dst[c][r] = bf16(src[r][c])   (fp32 rows x cols -> bf16 cols x rows): the k-major copy of a weight for gemm_rows_bf16
__global__ __launch_bounds__(kThreads) void narrow_transpose_kernel(const float *__restrict__ src, int rows, int cols, __bf16 *__restrict__ dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? src[(int64_t)(r0 + r) * cols + c0 + tx] : 0.f;
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < cols && r0 + tx < rows) dst[(int64_t)(c0 + c) * rows + r0 + tx] = (__bf16)tile[tx][c];
}
// dst[c][r] = src[r][c]   (fp32 rows x cols -> cols x rows): a k-major weight as the row-major operand of the LDS-DMA product kernel
__global__ __launch_bounds__(kThreads) void transpose_f32_kernel(const float *__restrict__ src, int rows, int cols, float *__restrict__ dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? src[(int64_t)(r0 + r) * cols + c0 + tx] : 0.f;
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < cols && r0 + tx < rows) dst[(int64_t)(c0 + c) * rows + r0 + tx] = tile[tx][c];
}
int launch_transpose_f32(const float *src, int rows, int cols, float *dst, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return MOT_OK;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(kThreads), 0, stream, src, rows, cols, dst);
    return check_launch("transpose_f32_kernel");
}
int launch_narrow_transpose(const float *src, int rows, int cols, void *dst, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return MOT_OK;
    hipLaunchKernelGGL(narrow_transpose_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(kThreads), 0, stream, src, rows, cols,
                       (__bf16 *)dst);
    return check_launch("narrow_transpose_kernel");
}
// dst[i] = bf16(src[i]); both 16-byte aligned
int launch_narrow(const float *src, int64_t n, void *dst, hipStream_t stream) {
    if (n <= 0) return MOT_OK;
    size_t nb = ((size_t)n / 8 + kThreads) / kThreads;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)nb), dim3(kThreads), 0, stream, src, n, (__bf16 *)dst);
    return check_launch("narrow_kernel");
}

// out[n] = ids[n * bpt + k] as int32 (out-of-range ids flagged and clamped to 0, as the forward does): one byte slot's ids as the
// "tokens" of a plain embedding backward (the slot-wise scatter of wide concat rows, below)
__global__ __launch_bounds__(kThreads) void ids_column_i32_kernel(const int64_t *__restrict__ ids, int64_t n, int bpt, int k, int64_t rows, int32_t *__restrict__ out,
                                                                  uint32_t *status) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        int64_t v = ids[i * bpt + k];
        if ((uint64_t)v >= (uint64_t)rows) { if (status) atomicOr(status, kStatusByteOor); v = 0; }
        out[i] = (int32_t)v;
    }
}

static size_t bwd_rnorm_floats(const MotEmbedMixDesc &d) { return d.mode == MOT_MIX_SUM ? ((size_t)d.byte_rows + 3) & ~(size_t)3 : 0; }
size_t embed_mix_bwd_mean_workspace_bytes(const MotEmbedMixDesc &d);
size_t embed_mix_bwd_workspace_bytes(const MotEmbedMixDesc &d) {
    if (d.mode == MOT_MIX_MEAN) return embed_mix_bwd_mean_workspace_bytes(d);
    if (d.mode == MOT_MIX_CONCAT_LINEAR) {
        return (lin_bwd_layout(d).total + (d.dtype == MOT_BF16 ? up_layout(d).total : 0)) * 4 + 256 + (du16_usable(d) ? du16_layout(d).total : 0);
    }
    return (bwd_rnorm_floats(d) + scatter_ws_ints(d)) * 4;
}

// `w16` / `ws16` (optional): the bf16 weight and the Du16Layout scratch -- then du and dW run on the bf16 MFMA; `g16` / `x16`
// (optional with them): the bf16 upstream gradient and forward output -- then dy is produced in bf16 directly and
// gr.grad_out / d.out (fp32) are never read; `d16`: the caller's descriptor with the bf16 tables (the concat operand of dW is then
// gathered from them directly, as the forward's was)
static int launch_embed_mix_bwd_linear(const MotEmbedMixDesc &d, const MotEmbedMixGrads &gr, hipStream_t stream, const void *w16 = nullptr,
                                       char *ws16 = nullptr, const void *g16 = nullptr, const void *x16 = nullptr, const MotEmbedMixDesc *d16 = nullptr) {
    if (d.id_source != MOT_IDS_GIVEN) return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: pass the byte ids the forward returned (MOT_IDS_GIVEN)");
    if (!gr.d_weight) return set_error(MOT_EINVAL, "embed_mix_bwd concat_linear: d_weight missing");
    if (d.norm_out && (!d.out || !d.out_row_rnorm)) return set_error(MOT_EINVAL, "embed_mix_bwd concat_linear: needs the forward's out and out_row_rnorm");
    const int64_t N = d.n_rows * d.tokens_per_row;
    const int Dm = d.model_dim, Dt = d.tok_dim, nbk = d.bpt * d.byte_dim, K = Dt + nbk;
    // rows wider than 1024 (mathblations' defaults: 768 + 3 x 768, model.py:21-24, 256-268) only where the table gradients can be
    // scattered part by part on the lane-contiguous kernel: token part and every byte slot a multiple of 256 columns, <= 1024 each
    // (rows up to 2048 columns: the strided kernels take them whole; wider ones -- the reference's dimension sweeps reach 1024 + 16 x
    //  128 = 3072, experiments100_000steps.sh, mathblations' defaults 768 + 3 x 768 -- only where the part-wise scatter below applies)
    if (Dm > 2048) return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd concat_linear: model_dim %d > 2048", Dm);
    const LinBwdLayout L = lin_bwd_layout(d);
    if (d.dtype == MOT_BF16) {
        const UpLayout U = up_layout(d);
        if (!d.workspace || d.workspace_bytes < (U.total + L.total) * 4)
            return set_error(MOT_EWORKSPACE, "embed_mix_bwd: needs %zu workspace bytes, got %zu", (U.total + L.total) * 4, d.workspace_bytes);
        float *up = (float *)d.workspace;
        const size_t Nn = (size_t)N * Dm;
        int rc;
        if ((rc = launch_widen(d.tok_table, (size_t)d.tok_rows * Dt, up + U.tok, stream))) return rc;
        if ((rc = launch_widen(d.byte_table, (size_t)d.byte_rows * d.byte_dim, up + U.byte, stream))) return rc;
        if ((rc = launch_widen(d.weight, (size_t)Dm * K, up + U.w, stream))) return rc;
        if (d.bias && (rc = launch_widen(d.bias, Dm, up + U.bias, stream))) return rc;
        const bool route16 = du16_usable(d);
        const bool dy_in_bf16 = route16 && !d.bias && Dm <= 4096;   // the bias gradient is a column sum of an fp32 dy
        if (!dy_in_bf16) {
            if ((rc = launch_widen(gr.grad_out, Nn, up + U.g, stream))) return rc;
            if (d.norm_out && (rc = launch_widen(d.out, Nn, up + U.x, stream))) return rc;
        }
        MotEmbedMixDesc d32 = d;
        MotEmbedMixGrads g32 = gr;
        d32.dtype = MOT_F32;
        d32.tok_table = up + U.tok; d32.byte_table = up + U.byte; d32.weight = up + U.w; d32.bias = d.bias ? up + U.bias : nullptr;
        d32.out = d.norm_out ? (void *)(up + U.x) : nullptr;
        d32.eps = d.eps > 0.f ? d.eps : kBf16Eps;   // the forward normalised with the bf16 epsilon
        d32.workspace = up + U.total; d32.workspace_bytes = d.workspace_bytes - U.total * 4;
        g32.grad_out = up + U.g;
        if (route16) {
            const size_t off = ((U.total + L.total) * 4 + 255) & ~(size_t)255;
            if (d.workspace_bytes < off + du16_layout(d).total)
                return set_error(MOT_EWORKSPACE, "embed_mix_bwd: needs %zu workspace bytes, got %zu", off + du16_layout(d).total, d.workspace_bytes);
            return launch_embed_mix_bwd_linear(d32, g32, stream, d.weight, (char *)d.workspace + off, dy_in_bf16 ? gr.grad_out : nullptr,
                                               dy_in_bf16 ? d.out : nullptr, &d);
        }
        return launch_embed_mix_bwd_linear(d32, g32, stream);
    }
    if (!d.workspace || d.workspace_bytes < L.total * 4)
        return set_error(MOT_EWORKSPACE, "embed_mix_bwd: needs %zu workspace bytes, got %zu", L.total * 4, d.workspace_bytes);
    float *ws = (float *)d.workspace;
    float *rn = ws + L.rnorm, *dy = ws + L.dy, *du = ws + L.du, *utok = ws + L.utok, *ubyte = ws + L.ubyte, *wk = ws + L.wk, *byte0 = ws + L.byte0;
    int32_t *iota = (int32_t *)(ws + L.iota), *sort_ints = (int32_t *)(ws + L.sort);
    (void)wk; (void)byte0; (void)ubyte;   // slots of the layout the fp32 du product no longer uses (u is built in place: utok .. ubyte)
    const float eps = d.eps > 0.f ? d.eps : FLT_EPSILON;
    int rc;
    // 1. dy
    const float *dyp = (const float *)gr.grad_out;
    const __bf16 *dy16p = nullptr;   // bf16 route with bf16 inputs: dy exists in bf16 only
    if (g16) {
        const Du16Layout U = du16_layout(d);
        dy16p = (const __bf16 *)g16;
        if (d.norm_out) {
            __bf16 *dy16 = (__bf16 *)(ws16 + U.dy16);
            hipLaunchKernelGGL(dy16_kernel, dim3((unsigned)((N + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, (const __bf16 *)g16, (const __bf16 *)x16,
                               d.out_row_rnorm, N, Dm, dy16);
            if ((rc = check_launch("dy16_kernel"))) return rc;
            dy16p = dy16;
        }
        dyp = nullptr;
    } else if (d.norm_out) {
        hipLaunchKernelGGL(dy_kernel, dim3((unsigned)((N + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, (const float *)gr.grad_out,
                           (const float *)d.out, d.out_row_rnorm, N, Dm, dy);
        if ((rc = check_launch("dy_kernel"))) return rc;
        dyp = dy;
    }
    if (gr.d_bias) {
        hipLaunchKernelGGL(colsum_kernel, dim3(256), dim3(kThreads), 0, stream, dyp, N, Dm, (float *)gr.d_bias);
        if ((rc = check_launch("colsum_kernel"))) return rc;
    }
    // 2. u = the seam tensors (norms and scalars applied), and dW += dy^T u
    const int64_t blk = 2048;
    (void)blk;
    const int tok_lo = d.bytes_first ? nbk : 0, byte_lo = d.bytes_first ? 0 : Dt;
    float *dW = (float *)gr.d_weight;
    // (w16: the concat operand goes straight to bf16, below)
    if (!w16) {  // fp32: the concat operand u [N, K] itself, in the two (adjacent) scratch regions, so dW is ONE contraction
        if ((rc = launch_gather_rows_placed(d.tokens, nullptr, 4, N, d.tok_table, d.tok_rows, Dt, d.norm_tok, eps, d.scale_tok, utok + tok_lo, 1, K,
                                            d.status, kStatusTokenOor, MOT_F32, stream))) return rc;
        if ((rc = launch_gather_rows_placed(d.ids_a, d.ids_b, 8, N * d.bpt, d.byte_table, d.byte_rows, d.byte_dim, d.norm_byte, eps, d.scale_byte,
                                            utok + byte_lo, d.bpt, K, d.status, kStatusByteOor, MOT_F32, stream))) return rc;
        if ((rc = launch_gemm_tn(dyp, Dm, Dm, utok, K, K, N, dW, K, stream))) return rc;
    } else {
        // 2'. dW on the bf16 MFMA: dy [N, Dm] and u [N, K] in bf16, ROW-major as they are, contracted over the tokens by
        // gemm_tn_bf16_kernel (transposing LDS reads).  u is the forward's operand: gathered from the bf16 tables by the forward's
        // own concat_rows_kernel when that applies (one id tensor, no learned scalars, 16-byte pieces), else gathered in fp32 from
        // the widened tables and narrowed.
        const Du16Layout U = du16_layout(d);
        __bf16 *dy16 = (__bf16 *)(ws16 + U.dy16), *u16 = (__bf16 *)(ws16 + U.uT);
        const __bf16 *dyr = dy16p;
        if (!dyr) {
            size_t nb = ((size_t)N * Dm / 8 + kThreads) / kThreads;
            if (nb > 4096) nb = 4096;
            hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)nb), dim3(kThreads), 0, stream, dyp, (int64_t)N * Dm, dy16);
            if ((rc = check_launch("narrow_kernel"))) return rc;
            dyr = dy16;
        }
        if (d16 && !d.ids_b && !d.scale_tok && !d.scale_byte && (Dt & 7) == 0 && (d.byte_dim & 7) == 0) {
            if (d.norm_byte && (rc = launch_rows_rnorm(d16->byte_table, d.byte_rows, d.byte_dim, eps, rn, MOT_BF16, stream))) return rc;
            if ((rc = launch_concat_rows(d.tokens, d.ids_a, N, d16->tok_table, d.tok_rows, Dt, d16->byte_table, d.byte_rows, d.byte_dim, d.bpt, d.norm_tok,
                                         d.norm_byte ? rn : nullptr, eps, u16, K, tok_lo, byte_lo, d.status, MOT_BF16, stream))) return rc;
        } else {
            if ((rc = launch_gather_rows_placed(d.tokens, nullptr, 4, N, d.tok_table, d.tok_rows, Dt, d.norm_tok, eps, d.scale_tok, utok + tok_lo, 1, K,
                                                d.status, kStatusTokenOor, MOT_F32, stream))) return rc;
            if ((rc = launch_gather_rows_placed(d.ids_a, d.ids_b, 8, N * d.bpt, d.byte_table, d.byte_rows, d.byte_dim, d.norm_byte, eps, d.scale_byte,
                                                utok + byte_lo, d.bpt, K, d.status, kStatusByteOor, MOT_F32, stream))) return rc;
            size_t nb = ((size_t)N * K / 8 + kThreads) / kThreads;
            if (nb > 4096) nb = 4096;
            hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)nb), dim3(kThreads), 0, stream, utok, (int64_t)N * K, u16);
            if ((rc = check_launch("narrow_kernel"))) return rc;
        }
        if ((rc = launch_gemm_tn_bf16(dyr, Dm, Dm, u16, K, K, N, dW, K, stream))) return rc;
    }
    hipLaunchKernelGGL(iota_kernel, dim3(256), dim3(kThreads), 0, stream, iota, N);
    if (w16) {
        // 3'. du on the bf16 MFMA: bf16(dy) rows x W^T on the dense bf16 kernel
        const Du16Layout U = du16_layout(d);
        __bf16 *dy16 = (__bf16 *)(ws16 + U.dy16), *wt16 = (__bf16 *)(ws16 + U.wt16);
        size_t nb = ((size_t)N * Dm / 8 + kThreads) / kThreads;
        if (nb > 4096) nb = 4096;
        if (dy16p) dy16 = const_cast<__bf16 *>(dy16p);   // (else narrowed for dW above)
        (void)nb;
        hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)((K + 31) / 32), (unsigned)((Dm + 31) / 32)), dim3(kThreads), 0, stream,
                           (const __bf16 *)w16, Dm, K, wt16);
        if ((rc = check_launch("narrow/transpose"))) return rc;
        // du[n][k] = sum_m dy16[n][m] * wt16[k][m], accumulated and written in fp32 (no bf16 round trip before the scatter)
        if ((rc = launch_gemm_rows_bf16(dy16, Dm, N, wt16, Dm, Dm, K, du, K, false, nullptr, stream))) return rc;
    } else {
    // 3. du = dy . W   (N x Dm) @ (Dm x K): both row-major as they are (nn.Linear keeps W as [Dm][K])
    //    (W transposed once and the product on the LDS-DMA kernel, as the cross-attention backward does with its k-major products:
    //     measured, 2.625 against 2.63 ms for forward + backward -- not kept here)
    if ((rc = launch_gemm_rows(dyp, Dm, N, (const float *)d.weight, K, Dm, K, du, K, false, stream))) return rc;
    }
    // 4. table gradients from du (its row layout is the concat layout)
    BwdArgs A;
    fill_bwd_args(A, d, gr);
    A.grad_out = du; A.D = K; A.norm_out = 0;
    A.Dt = Dt; A.tok_lo = tok_lo; A.byte_lo = byte_lo; A.nbk = nbk;
    // The two halves of a du row are two embedding backwards: the token part a plain one (NOOP) over Dt columns, the byte part a SUM
    // over byte slots with no token table -- both on the lane-contiguous kernel, reading their columns of du in place (row stride K),
    // sharing one grouping of the positions.  (The strided kernel of round 1 took 242 us of the 870 us the bf16 concat forward +
    // backward takes at 65 536 tokens.)  One id tensor only: norm_byte over two id tensors normalises the SUM of two rows.
    if (!d.ids_b) {
        BwdArgs At = A, Ab = A;
        At.D = At.Dt = Dt; At.tok_lo = At.byte_lo = 0; At.nbk = 0; At.grad_out = du + tok_lo; At.g_ld = K; At.d_byte = nullptr;
        Ab.D = Ab.Dt = nbk; Ab.tok_lo = Ab.byte_lo = 0; Ab.nbk = nbk; Ab.grad_out = du + byte_lo; Ab.g_ld = K; Ab.no_tok = 1; Ab.d_tok = nullptr;
        Ab.norm_tok = 0; Ab.tok_table = nullptr;
        // the byte part in blocks of whole slots, <= 1024 columns each (16 x 128-wide slots are two blocks of 8)
        int per = d.bpt;
        while (per > 1 && per * d.byte_dim > 1024) per = (per + 1) / 2;
        Ab.D = Ab.Dt = Ab.nbk = per * d.byte_dim;
        // (the token part: the lane-contiguous kernel, or -- 896 columns -- the general one, which knows the row stride too)
        bool split = d.bpt % per == 0 && lc_layout<MOT_MIX_SUM>(Ab) && (lc_layout<MOT_MIX_NOOP>(At) || K > 2048);
#ifdef MOT_DEV_ABLATION
        if (getenv("MOT_CONCAT_SCATTER_OLD")) split = false;
#endif
        if (split) {
            if ((rc = run_scatter<MOT_MIX_NOOP>(At, d, sort_ints, rn, stream))) return rc;
            for (int s0 = 0; s0 < d.bpt; s0 += per) {
                BwdArgs Ac = Ab;
                Ac.pos_sorted = At.pos_sorted; Ac.tok_sorted = At.tok_sorted;
                Ac.slot0 = s0; Ac.grad_out = du + byte_lo + s0 * d.byte_dim;
                if ((rc = run_scatter<MOT_MIX_SUM>(Ac, d, sort_ints, rn, stream))) return rc;
            }
            return MOT_OK;
        }
        // Wide byte slots (a slot is a whole embedding row: the digit mixin): every slot is a plain embedding backward of its own, the
        // slot's ids as the "tokens", the byte table as the table, its columns of du as the gradient rows
        if (lc_layout<MOT_MIX_NOOP>(At) && (d.byte_dim & 255) == 0 && d.byte_dim <= 1024 && !d.scale_tok && !d.scale_byte) {
            if ((rc = run_scatter<MOT_MIX_NOOP>(At, d, sort_ints, rn, stream))) return rc;
            for (int k = 0; k < d.bpt; ++k) {
                hipLaunchKernelGGL(ids_column_i32_kernel, dim3(256), dim3(kThreads), 0, stream, d.ids_a, N, d.bpt, k, d.byte_rows, iota, d.status);
                if ((rc = check_launch("ids_column_i32_kernel"))) return rc;
                BwdArgs As = A;
                As.tokens = iota; As.tok_table = A.byte_table; As.tok_rows = d.byte_rows; As.norm_tok = d.norm_byte;
                As.D = As.Dt = d.byte_dim; As.tok_lo = As.byte_lo = 0; As.nbk = 0; As.grad_out = du + byte_lo + k * d.byte_dim; As.g_ld = K;
                As.d_tok = A.d_byte; As.d_byte = nullptr; As.pos_sorted = As.tok_sorted = nullptr;
                if (!lc_layout<MOT_MIX_NOOP>(As)) return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd concat_linear: byte slots of %d columns", d.byte_dim);
                MotEmbedMixDesc ds = d;   // (run_scatter reads the table height from the descriptor)
                ds.tok_rows = d.byte_rows;
                if ((rc = run_scatter<MOT_MIX_NOOP>(As, ds, sort_ints, rn, stream))) return rc;
            }
            return MOT_OK;
        }
    }
    if (K > 2048)
        return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd concat_linear: K %d > 2048 needs the part-wise scatter (one id tensor, no learned scalars, "
                         "byte slots that tile blocks of <= 1024 columns)", K);
    return run_scatter<MOT_MIX_CONCAT_LINEAR>(A, d, sort_ints, rn, stream);
}

// ==========================================================================================
// MEAN backward:  x = s_t a + s_c mean_k v_k,  v = rms_norm?(E_c[id])      (inference/inference.py:266-267 under autograd)
// The character table has a few hundred rows at most (132), so its gradient is a dense product instead of a scatter:
//   cnt[n][r] = number of slots of token n holding character r                  (mean_counts_kernel)
//   M1 = cnt^T G   [rows, D]                                                    (gemm_tn: fp32 MFMA, split over the tokens)
//   S  = G V^T     [N, rows],   w_r = sum_n cnt[n][r] S[n][r]                   (gemm_rows + mean_colsum_kernel)
//   d E_c[r] += (s_c / bpt) rn_r (M1[r] - v_r w_r / D)   (no norm: (s_c / bpt) M1[r]);   d s_c += sum_r w_r / bpt
// S and w are only needed for the norm's backward and for d s_c.  The token side is the tokens-only backward with a scale.
// ==========================================================================================
__global__ __launch_bounds__(kThreads) void mean_counts_kernel(const int64_t *__restrict__ ids, int64_t n, int bpt, int rows, int ld,
                                                               float *__restrict__ cnt, uint32_t *status) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (t >= n) return;
    int64_t v = lane < bpt ? ids[t * bpt + lane] : -1;
    if (lane < bpt && (uint64_t)v >= (uint64_t)rows) { if (status) atomicOr(status, kStatusByteOor); v = 0; }
    const int id = (int)v;
    for (int r0 = 0; r0 < ld; r0 += 64) {
        const int r = r0 + lane;
        float c = 0.f;
        for (int k = 0; k < bpt; ++k) c += __builtin_amdgcn_readlane(id, k) == r ? 1.f : 0.f;
        if (r < ld) cnt[t * ld + r] = c;
    }
}

// w[r] += sum_n cnt[n][r] * S[n][r]: a workgroup takes a stretch of tokens, a thread a column (rows <= 1024)
__global__ __launch_bounds__(kThreads) void mean_colsum_kernel(const float *__restrict__ cnt, const float *__restrict__ S, int64_t n, int rows, int ld,
                                                               int64_t per_block, float *__restrict__ w) {
    const int64_t n0 = (int64_t)blockIdx.x * per_block, n1 = min(n, n0 + per_block);
    for (int r = threadIdx.x; r < rows; r += kThreads) {
        float acc = 0.f;
        for (int64_t t = n0; t < n1; ++t) acc += cnt[t * ld + r] * S[t * ld + r];
        if (acc != 0.f) atomicAdd(w + r, acc);
    }
}

__global__ __launch_bounds__(kThreads) void mean_finalize_kernel(const float *__restrict__ M1, const float *__restrict__ table, const float *__restrict__ rn,
                                                                 const float *__restrict__ w, int rows, int D, int bpt, const float *scale_byte,
                                                                 float *__restrict__ d_table, float *d_scale) {
    const float c = (scale_byte ? *scale_byte : 1.0f) / (float)bpt;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < (int64_t)rows * D; i += (int64_t)gridDim.x * kThreads) {
        const int r = (int)(i / D);
        float g = M1[i];
        if (rn) g = rn[r] * (g - (table[i] * rn[r]) * (w[r] / (float)D));
        d_table[i] += c * g;
    }
    if (d_scale && blockIdx.x == 0 && threadIdx.x < 64) {
        float sacc = 0.f;
        for (int r = threadIdx.x; r < rows; r += 64) sacc += w[r];
        sacc = wave_sum(sacc);
        if (threadIdx.x == 0) atomicAdd(d_scale, sacc / (float)bpt);
    }
}

constexpr int64_t kMeanSlab = 65536;
struct MeanBwdLayout { size_t rn, vn, m1, w, cnt, s, scatter, tab32, g32, total; int ld; };
static MeanBwdLayout mean_bwd_layout(const MotEmbedMixDesc &d) {
    MeanBwdLayout L;
    const int64_t N = d.n_rows * d.tokens_per_row, slab = N < kMeanSlab ? N : kMeanSlab;
    L.ld = (int)((d.byte_rows + 3) & ~3);
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    L.rn = take(d.byte_rows); L.vn = take((size_t)d.byte_rows * d.byte_dim); L.m1 = take((size_t)d.byte_rows * d.byte_dim); L.w = take(d.byte_rows);
    L.cnt = take((size_t)slab * L.ld); L.s = take((size_t)slab * L.ld); L.scatter = take(scatter_ws_ints(d));
    // bf16 tables / gradient rows: the character side runs on fp32 copies -- the 132-row table once, the gradient rows a slab at a time
    L.tab32 = L.g32 = 0;
    if (d.dtype == MOT_BF16) { L.tab32 = take((size_t)d.byte_rows * d.byte_dim); L.g32 = take((size_t)slab * d.byte_dim); }
    L.total = o;
    return L;
}

__global__ __launch_bounds__(kThreads) void scale_rows_kernel(const float *__restrict__ src, const float *__restrict__ rn, int rows, int D, float *__restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < (int64_t)rows * D; i += (int64_t)gridDim.x * kThreads) dst[i] = src[i] * rn[i / D];
}

static int launch_embed_mix_bwd_mean(const MotEmbedMixDesc &d, const MotEmbedMixGrads &gr, hipStream_t stream) {
    // bf16 (round 3): the token side reads bf16 rows natively (the NOOP scatter kernel); the character side -- dense products over
    // the token axis on the fp32 MFMA -- runs on operands widened once (the table) or slab by slab (the gradient rows); sums fp32.
    const bool bf = d.dtype == MOT_BF16;
    if (d.norm_out) return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: MEAN with an output norm has no backward (the reference's residual, inference.py:267, has none)");
    if (d.id_source != MOT_IDS_GIVEN || d.ids_b) return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd MEAN: one given id tensor");
    if (d.byte_rows > 1024) return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd MEAN: %lld character rows (> 1024): the dense formulation is for small tables", (long long)d.byte_rows);
    if (!gr.d_byte_table) return set_error(MOT_EINVAL, "embed_mix_bwd: d_byte_table missing");
    const MeanBwdLayout L = mean_bwd_layout(d);
    if (!d.workspace || d.workspace_bytes < L.total * 4) return set_error(MOT_EWORKSPACE, "embed_mix_bwd: needs %zu workspace bytes, got %zu", L.total * 4, d.workspace_bytes);
    float *ws = (float *)d.workspace;
    const int64_t N = d.n_rows * d.tokens_per_row, slab = N < kMeanSlab ? N : kMeanSlab;
    const int rows = (int)d.byte_rows, D = d.byte_dim;
    const float eps = d.eps > 0.f ? d.eps : (d.dtype == MOT_BF16 ? kBf16Eps : FLT_EPSILON);   // F.rms_norm(eps=None): finfo of the input dtype
    int rc;
    // ---- token side: x = s_t * norm?(E_t[tok]) + (...) is the tokens-only mix as far as the token table and s_t are concerned
    {
        MotEmbedMixDesc t = d;
        t.mode = MOT_MIX_NOOP; t.bpt = 0; t.id_source = MOT_IDS_NONE; t.ids_a = nullptr; t.byte_table = nullptr; t.norm_byte = 0; t.scale_byte = nullptr;
        MotEmbedMixGrads gt = gr;
        gt.d_byte_table = nullptr; gt.d_scale_byte = nullptr;
        BwdArgs A;
        fill_bwd_args(A, t, gt);
        if ((rc = run_scatter<MOT_MIX_NOOP>(A, t, (int32_t *)(ws + L.scatter), nullptr, stream))) return rc;
    }
    // ---- character side
    const float *tab = (const float *)d.byte_table;
    if (bf) {
        if ((rc = launch_widen(d.byte_table, (size_t)rows * D, ws + L.tab32, stream))) return rc;
        tab = ws + L.tab32;
    }
    const float *V = tab, *rn = nullptr;
    const bool need_s = d.norm_byte || gr.d_scale_byte;
    if (d.norm_byte) {
        if ((rc = launch_rows_rnorm(tab, rows, D, eps, ws + L.rn, MOT_F32, stream))) return rc;
        hipLaunchKernelGGL(scale_rows_kernel, dim3(256), dim3(kThreads), 0, stream, tab, ws + L.rn, rows, D, ws + L.vn);
        V = ws + L.vn; rn = ws + L.rn;
    }
    if ((rc = launch_zero_words(ws + L.m1, (int64_t)rows * D, stream))) return rc;
    if ((rc = launch_zero_words(ws + L.w, rows, stream))) return rc;
    for (int64_t n0 = 0; n0 < N; n0 += slab) {
        const int64_t nn = N - n0 < slab ? N - n0 : slab;
        const float *G = (const float *)gr.grad_out + n0 * D;
        if (bf) {
            if ((rc = launch_widen((const __bf16 *)gr.grad_out + n0 * D, (size_t)nn * D, ws + L.g32, stream))) return rc;
            G = ws + L.g32;
        }
        hipLaunchKernelGGL(mean_counts_kernel, dim3((unsigned)((nn + kWaves - 1) / kWaves)), dim3(kThreads), 0, stream, d.ids_a + n0 * d.bpt, nn, d.bpt, rows,
                           L.ld, ws + L.cnt, d.status);
        if ((rc = check_launch("mean_counts_kernel"))) return rc;
        if ((rc = launch_gemm_tn(ws + L.cnt, L.ld, rows, G, D, D, nn, ws + L.m1, D, stream))) return rc;
        if (need_s) {
            if ((rc = launch_gemm_rows(G, D, nn, V, D, D, rows, ws + L.s, L.ld, true, stream))) return rc;
            const int64_t per = 256;
            hipLaunchKernelGGL(mean_colsum_kernel, dim3((unsigned)((nn + per - 1) / per)), dim3(kThreads), 0, stream, ws + L.cnt, ws + L.s, nn, rows, L.ld, per,
                               ws + L.w);
            if ((rc = check_launch("mean_colsum_kernel"))) return rc;
        }
    }
    hipLaunchKernelGGL(mean_finalize_kernel, dim3(256), dim3(kThreads), 0, stream, ws + L.m1, tab, rn, ws + L.w, rows, D, d.bpt,
                       d.scale_byte, (float *)gr.d_byte_table, gr.d_scale_byte);
    return check_launch("mean_finalize_kernel");
}

size_t embed_mix_bwd_mean_workspace_bytes(const MotEmbedMixDesc &d) { return mean_bwd_layout(d).total * 4; }

int launch_embed_mix_bwd(const MotEmbedMixDesc &d, const MotEmbedMixGrads &gr, hipStream_t stream) {
    if (d.n_rows * d.tokens_per_row > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: more than 2^31 tokens");
    if (d.mode == MOT_MIX_CONCAT_LINEAR) return launch_embed_mix_bwd_linear(d, gr, stream);
    if (d.mode == MOT_MIX_MEAN) return launch_embed_mix_bwd_mean(d, gr, stream);
    if (d.mode == MOT_MIX_SUM && d.id_source != MOT_IDS_GIVEN)
        return set_error(MOT_EUNSUPPORTED, "embed_mix_bwd: pass the byte ids the forward returned (MOT_IDS_GIVEN)");
    BwdArgs A;
    fill_bwd_args(A, d, gr);
    const size_t need = embed_mix_bwd_workspace_bytes(d);
    if (!d.workspace || d.workspace_bytes < need)
        return set_error(MOT_EWORKSPACE, "embed_mix_bwd: needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
    float *rn = (float *)d.workspace;
    int32_t *ints = (int32_t *)(rn + bwd_rnorm_floats(d));
    if (d.mode == MOT_MIX_SUM) return run_scatter<MOT_MIX_SUM>(A, d, ints, rn, stream);
    return run_scatter<MOT_MIX_NOOP>(A, d, ints, rn, stream);
}

}  // namespace mot
