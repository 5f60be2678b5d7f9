// mot_mix.hpp -- pieces shared by the fused forward kernels (mot_embed.hip: SUM/MEAN/NOOP,
// mot_linear.hip: CONCAT_LINEAR): the kernel argument block and phase 1 (byte ids of a tile
// into LDS, from the token->byte table + pull or from precomputed int64 ids).
#pragma once
#include <float.h>

#include "mot_internal.hpp"
#include "mot_tile.hpp"

namespace mot {

typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

// Element access for the two table/output formats.  All arithmetic is fp32; bf16 rows are widened on
// load (8 B per lane-chunk) and results rounded once, to nearest-even, on store (v_cvt_pk_bf16_f32).
template <typename T> struct Elem;
template <> struct Elem<float> {
    typedef float4v vec;               // one 16-byte lane load
    typedef float4v raw;               // as it sits in registers while the load is in flight
    static constexpr int kVec = 4;
    static __device__ __forceinline__ raw load_raw(const float *p) { return *(const float4v *)p; }
    static __device__ __forceinline__ vec widen(raw r) { return r; }
    static __device__ __forceinline__ float4v loadv(const float *p) { return *(const float4v *)p; }
    static __device__ __forceinline__ void storev_nt(float *p, float4v v) { __builtin_nontemporal_store(v, (float4v *)p); }
    static __device__ __forceinline__ float4v load4(const float *p) { return *(const float4v *)p; }
    static __device__ __forceinline__ float load1(const float *p) { return *p; }
    static __device__ __forceinline__ void store4_nt(float *p, float4v v) { __builtin_nontemporal_store(v, (float4v *)p); }
    static __device__ __forceinline__ void store1(float *p, float v) { *p = v; }
};
template <> struct Elem<__bf16> {
    typedef float8v vec;
    typedef bf16x8v raw;               // 4 VGPRs in flight instead of 8: twice the tokens per wave
    static constexpr int kVec = 8;
    static __device__ __forceinline__ raw load_raw(const __bf16 *p) { return *(const bf16x8v *)p; }
    static __device__ __forceinline__ vec widen(raw r) { return __builtin_convertvector(r, float8v); }
    static __device__ __forceinline__ float8v loadv(const __bf16 *p) { return __builtin_convertvector(*(const bf16x8v *)p, float8v); }
    static __device__ __forceinline__ void storev_nt(__bf16 *p, float8v v) { __builtin_nontemporal_store(__builtin_convertvector(v, bf16x8v), (bf16x8v *)p); }
    static __device__ __forceinline__ float4v load4(const __bf16 *p) { return __builtin_convertvector(*(const bf16x4v *)p, float4v); }
    static __device__ __forceinline__ float load1(const __bf16 *p) { return (float)*p; }
    static __device__ __forceinline__ void store4_nt(__bf16 *p, float4v v) { __builtin_nontemporal_store(__builtin_convertvector(v, bf16x4v), (bf16x4v *)p); }
    static __device__ __forceinline__ void store1(__bf16 *p, float v) { *p = (__bf16)v; }
};

struct MixArgs {
    // ids
    const int32_t *tokens;
    int64_t T;  // tokens per row
    int bpt;
    int id_source, pull_dir;
    const void *ttb;
    int64_t ttb_rows;
    int ttb_elem;
    int add_padded;
    int32_t pad, eot;
    const int64_t *ids_a, *ids_b;
    // tables
    const float *tok_table;
    int64_t tok_rows;
    int Dt;
    const float *byte_table;
    int64_t byte_rows;
    int Db;
    int norm_tok, norm_byte, norm_out;
    float eps;
    const float *scale_tok, *scale_byte;
    const float *byte_rnorm;  // workspace: 1/rms of every byte-table row (norm_byte)
    float *out;
    __bf16 *add16;            // LDS-table MEAN kernel, fp32 tables: the result is added to the bf16 rows here (fp32 sum, one rounding) instead of stored to out
    int add_out;              // the same kernel: out += result (fp32 rows)
    int64_t *out_ids_padded, *out_ids_pulled, *counters;
    uint32_t *status;
    int tile_tokens, tiles_per_row;   // tile kernels (mot_linear.hip)
    int unit, wave_lds;               // wave kernels (mot_embed.hip): tokens per wave, LDS bytes per wave
    int64_t units_per_row, n_units;
};

constexpr float kBf16Eps = 0.0078125f;  // torch.finfo(torch.bfloat16).eps: what F.rms_norm(eps=None) uses on bf16 input

__device__ __forceinline__ float rms_scale(float sumsq, int dim, float eps) {
    // F.rms_norm: x * rsqrt(mean(x^2) + eps)   (train_gpt.py:172-173).  v_rcp_f32 and v_rsq_f32 (<= 1 ulp each) instead of the
    // IEEE divide and square root sequences: ~40 VALU instructions per row less in kernels that are VALU-bound; the result is
    // within 2.5e-7 relative of the correctly rounded one (the parity bar for these outputs is 1e-6).
    return __builtin_amdgcn_rsqf(sumsq * __builtin_amdgcn_rcpf((float)dim) + eps);
}

__device__ __forceinline__ int clamp_byte_id(int id, int64_t byte_rows, uint32_t *status) {
    if ((uint64_t)(uint32_t)id >= (uint64_t)byte_rows) {
        if (status) atomicOr(status, kStatusByteOor);
        return 0;
    }
    return id;
}

// ------------------------------------------------------------------------------------------ phase 1
// Leaves L.tok (token ids clamped to the ttb), L.ids (idsA, clamped to the byte table) and, when
// `dual`, L.val (idsB, clamped) ready for phase 2.  Writes the optional parity outputs/counters.
template <int DIR>
__device__ __forceinline__ void phase1_from_ttb(const MixArgs &A, const TileLds &L, int64_t row, int64_t t0, int ntok) {
    const int bpt = A.bpt, sv = bpt | 1;
    SrcTable src{A.tokens + row * A.T, A.ttb, A.ttb_rows, A.ttb_elem, bpt, A.pad, A.eot, A.status};
    if (DIR != kPullNone && (threadIdx.x >> 6) == kWaves - 1) halo_walk<DIR == kPullNone ? kPullLeft : DIR>(src, t0, ntok, A.T, bpt, L);
    fill_table_tile(src, t0, ntok, bpt, L);
    if (DIR != kPullNone) tile_scan_and_compact<DIR == kPullNone ? kPullLeft : DIR>(src, t0, ntok, A.T, bpt, L);
    const SlotLayout S(bpt);
    int pads_before = 0, pads_after = 0;
    if (S.kq < bpt) {
        const int64_t obase = (row * A.T + t0) * bpt;
        for (int t = S.tq; t < ntok; t += S.tstride) {
            const int own = L.val[t * sv + S.kq];
            int v = own;
            if (DIR != kPullNone) {
                int kind;
                const int payload = pulled_slot<DIR == kPullNone ? kPullLeft : DIR>(L, t, S.kq, ntok, bpt, &kind);
                v = kind == 1 ? A.pad : (kind == 2 ? own : payload);
            }
            if (A.out_ids_padded) A.out_ids_padded[obase + t * bpt + S.kq] = own;
            if (A.out_ids_pulled) A.out_ids_pulled[obase + t * bpt + S.kq] = v;
            pads_before += own == A.pad;
            pads_after += v == A.pad;
            L.ids[t * sv + S.kq] = clamp_byte_id(v, A.byte_rows, A.status);
        }
    }
    if (A.counters) {  // runs/79_mot-in_toks-valemb.py:484-488
        pads_before = (int)wave_sum((float)pads_before);  // <= 64*64 per wave: exact in fp32
        pads_after = (int)wave_sum((float)pads_after);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd((unsigned long long *)A.counters + 2, (unsigned long long)pads_before);
            atomicAdd((unsigned long long *)A.counters + 3, (unsigned long long)pads_after);
        }
        if (threadIdx.x == 0) {
            atomicAdd((unsigned long long *)A.counters + 0, (unsigned long long)ntok);
            atomicAdd((unsigned long long *)A.counters + 1, (unsigned long long)ntok * bpt);
        }
    }
    __syncthreads();  // every pulled_slot read of L.val/stream is done
    if (A.add_padded && S.kq < bpt)
        for (int t = S.tq; t < ntok; t += S.tstride)
            L.val[t * sv + S.kq] = clamp_byte_id(L.val[t * sv + S.kq], A.byte_rows, A.status);
    __syncthreads();
}

__device__ __forceinline__ void phase1_given(const MixArgs &A, const TileLds &L, int64_t row, int64_t t0, int ntok) {
    const int bpt = A.bpt, sv = bpt | 1;
    if ((int)threadIdx.x < ntok) L.tok[threadIdx.x] = A.tokens[row * A.T + t0 + threadIdx.x];
    const SlotLayout S(bpt);
    if (S.kq < bpt) {
        const int64_t ibase = (row * A.T + t0) * bpt;
        for (int t = S.tq; t < ntok; t += S.tstride) {
            const int64_t a = A.ids_a[ibase + t * bpt + S.kq];
            int ia = (int)a;
            if ((uint64_t)a >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); ia = 0; }
            L.ids[t * sv + S.kq] = ia;
            if (A.ids_b) {
                const int64_t b = A.ids_b[ibase + t * bpt + S.kq];
                int ib = (int)b;
                if ((uint64_t)b >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); ib = 0; }
                L.val[t * sv + S.kq] = ib;
            }
        }
    }
    if (A.counters && threadIdx.x == 0) {
        atomicAdd((unsigned long long *)A.counters + 0, (unsigned long long)ntok);
        atomicAdd((unsigned long long *)A.counters + 1, (unsigned long long)ntok * bpt);
    }
    __syncthreads();
}


// Fills the id-related fields of MixArgs from the public descriptor.
inline void fill_mix_args(MixArgs &A, const MotEmbedMixDesc &d) {
    A.tokens = d.tokens; A.T = d.tokens_per_row; A.bpt = d.mode == MOT_MIX_NOOP ? 0 : d.bpt;
    A.id_source = d.id_source; A.pull_dir = d.pull_dir; A.ttb = d.ttb; A.ttb_rows = d.ttb_rows;
    A.ttb_elem = d.ttb_elem_bytes; A.add_padded = d.add_padded; A.pad = d.pad_byte; A.eot = d.eot_byte;
    A.ids_a = d.ids_a; A.ids_b = d.ids_b;
    A.tok_table = (const float *)d.tok_table; A.tok_rows = d.tok_rows; A.Dt = d.tok_dim;
    A.byte_table = (const float *)d.byte_table; A.byte_rows = d.byte_rows; A.Db = d.byte_dim;
    A.norm_tok = d.norm_tok; A.norm_byte = d.norm_byte; A.norm_out = d.norm_out;
    // F.rms_norm(eps=None) uses torch.finfo(x.dtype).eps: 2^-23 for fp32 inputs, 2^-7 for bf16 inputs
    A.eps = d.eps > 0.f ? d.eps : (d.dtype == MOT_BF16 ? kBf16Eps : FLT_EPSILON);
    A.scale_tok = d.scale_tok; A.scale_byte = d.scale_byte;
    A.byte_rnorm = nullptr;
    A.out = (float *)d.out;
    A.add16 = nullptr;
    A.add_out = 0;
    A.out_ids_padded = d.out_ids_padded; A.out_ids_pulled = d.out_ids_pulled; A.counters = d.counters;
    A.status = d.status;
}

}  // namespace mot
