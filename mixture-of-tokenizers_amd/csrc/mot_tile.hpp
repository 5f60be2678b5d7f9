// mot_tile.hpp -- per-tile byte-index machinery shared by every kernel (gfx950, wave64).
//
// A *tile* is up to 256 consecutive tokens of ONE batch row, owned by one 256-thread
// workgroup.  For a tile the code below reproduces, entirely in LDS, what the reference's
// pull_from_left / pull_from_right compute with cumsum / nonzero / searchsorted over a whole
// row (scaled-pre-train/data_creation.py:71-176, 179-305):
//
//   stream  = the tile's non-pad slots, compacted in order            (flat_valid_bytes)
//   cum[t]  = number of valid slots in tile tokens [0, t)              (cum_valid_bytes)
//   halo    = the <= bpt valid slots just outside the tile that a window can still reach:
//             before the tile (pull-left) or after it (pull-right).  A window never crosses
//             an all-EOT token and never needs more than bpt slots, so one wave walking
//             outwards 64 tokens at a time stops after the first step in practice; this is
//             what makes tiles independent (no cross-workgroup scan, no row-wide pass).
//   use[t]  = how many stream slots token t's window holds (bytes_to_use / bytes_to_pull)
//
// Slots are described by a 32-bit *payload*: for the token->byte-table source it is the byte id
// itself; for a raw int64 byte tensor it is the slot's position in its row, so values of any
// width are moved bit-exactly by re-reading the input at that position.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mot {

constexpr int kThreads = 256;          // workgroup size of every tile kernel
constexpr int kWaves = kThreads / 64;  // wave64
constexpr int kMaxTileTokens = 256;    // one token per thread in the scan passes
constexpr int kMaxBpt = 64;

enum : int { kPullNone = 0, kPullLeft = 1, kPullRight = 2 };
enum : uint32_t { kStatusTokenOor = 1u, kStatusByteOor = 2u };

// flags of one slot
constexpr int kValid = 1;  // slot != pad_byte   (non_pad_mask, data_creation.py:93/199)
constexpr int kEotB = 2;   // slot == eot_byte   (is_eot_token needs all slots, :94/200)

// ---------------------------------------------------------------- wave / block scans
__device__ __forceinline__ int wave_incl_add(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ int wave_incl_max(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o, 64);
        if (lane >= o) v = max(v, u);
    }
    return v;
}
// inclusive min-scan running from lane 63 down to lane 0
__device__ __forceinline__ int wave_rincl_min(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_down(v, o, 64);
        if (lane + o < 64) v = min(v, u);
    }
    return v;
}
// Wave-wide reductions on the DPP cross-lane path (no LDS crossbar round trips): two quad permutes and two row rotates
// leave every lane of a 16-lane row with the row's total; row_bcast:15 / row_bcast:31 then chain the four rows, so lane 63
// holds the wave's total, which comes back as a wave-uniform (scalar) value.  6 VALU instructions + 1 readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float masked, float v) {   // lanes outside ROW_MASK keep `masked`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(masked), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E, kDppRowRor4 = 0x124, kDppRowRor8 = 0x128, kDppRowBcast15 = 0x142,
              kDppRowBcast31 = 0x143;
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<kDppQuadXor1, 0xf>(0.f, v);
    v += dpp_move<kDppQuadXor2, 0xf>(0.f, v);
    v += dpp_move<kDppRowRor4, 0xf>(0.f, v);
    v += dpp_move<kDppRowRor8, 0xf>(0.f, v);
    v += dpp_move<kDppRowBcast15, 0xa>(0.f, v);
    v += dpp_move<kDppRowBcast31, 0xc>(0.f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<kDppQuadXor1, 0xf>(v, v));
    v = fmaxf(v, dpp_move<kDppQuadXor2, 0xf>(v, v));
    v = fmaxf(v, dpp_move<kDppRowRor4, 0xf>(v, v));
    v = fmaxf(v, dpp_move<kDppRowRor8, 0xf>(v, v));
    v = fmaxf(v, dpp_move<kDppRowBcast15, 0xa>(v, v));
    v = fmaxf(v, dpp_move<kDppRowBcast31, 0xc>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// LDS carve-up of one tile.  SV = bpt|1 is the odd row stride of the per-slot arrays, which
// keeps the one-thread-per-token passes (stride-SV reads) free of bank conflicts.
struct TileLds {
    int32_t *val;     // [tile_tokens*SV] slot payloads (table source) or slot flags (raw source)
    int32_t *ids;     // [tile_tokens*SV] pulled byte ids (fused kernels only)
    int32_t *stream;  // [tile_tokens*bpt]
    int32_t *cum;     // [tile_tokens+1]
    int32_t *use;     // [tile_tokens]
    int32_t *iseot;   // [tile_tokens]
    int32_t *tok;     // [tile_tokens]
    int32_t *halo;    // [kMaxBpt]
    int32_t *misc;    // [16]: 0-3 wave sums, 4-7 wave max/min, 8 halo count
};
constexpr int kMiscHalo = 8;

__host__ __device__ inline size_t tile_lds_bytes(int tile_tokens, int bpt, bool with_ids) {
    size_t sv = (size_t)(bpt | 1);
    size_t w = (size_t)tile_tokens * sv * (with_ids ? 2 : 1) + (size_t)tile_tokens * bpt +
               (size_t)(tile_tokens + 1) + 3 * (size_t)tile_tokens + kMaxBpt + 16;
    return w * sizeof(int32_t);
}

__device__ __forceinline__ TileLds tile_lds_carve(int32_t *base, int tile_tokens, int bpt, bool with_ids) {
    TileLds L;
    const int sv = bpt | 1;
    L.val = base;            base += tile_tokens * sv;
    L.ids = base;            if (with_ids) base += tile_tokens * sv;
    L.stream = base;         base += tile_tokens * bpt;
    L.cum = base;            base += tile_tokens + 1;
    L.use = base;            base += tile_tokens;
    L.iseot = base;          base += tile_tokens;
    L.tok = base;            base += tile_tokens;
    L.halo = base;           base += kMaxBpt;
    L.misc = base;
    return L;
}

// ---------------------------------------------------------------- slot sources
// Token->byte table source: slot (t,k) = ttb[tokens[t]][k]  (tokens_to_bytes, data_creation.py:61-67)
// The table is int16 (458 byte ids, 14 digits) or int32 (a second BPE vocabulary).
struct SrcTable {
    static constexpr bool kRaw = false;
    const int32_t *tokens;  // this row's tokens
    const void *ttb;
    int64_t ttb_rows;
    int elem;  // 2 | 4
    int bpt;
    int32_t pad, eot;
    uint32_t *status;
    __device__ __forceinline__ int token(int64_t t) const {
        int id = tokens[t];
        if ((uint64_t)(uint32_t)id >= (uint64_t)ttb_rows) {  // nn.Embedding would raise IndexError
            if (status) atomicOr(status, kStatusTokenOor);
            id = 0;
        }
        return id;
    }
    __device__ __forceinline__ int32_t value(int tokid, int k) const {
        const int64_t i = (int64_t)tokid * bpt + k;
        return elem == 2 ? (int32_t)((const int16_t *)ttb)[i] : ((const int32_t *)ttb)[i];
    }
    __device__ __forceinline__ int flags_of(int32_t v) const { return (v != pad ? kValid : 0) | (v == eot ? kEotB : 0); }
};

// Raw int64 byte tensor source (pull_from_* called on an existing tensor).
struct SrcRaw {
    static constexpr bool kRaw = true;
    const int64_t *row;  // this row's T*bpt slots
    int bpt;
    int64_t pad, eot;
    __device__ __forceinline__ int flags_at(int64_t slot) const {
        int64_t v = row[slot];
        return (v != pad ? kValid : 0) | (v == eot ? kEotB : 0);
    }
};

// ---------------------------------------------------------------- halo walk (one wave)
// Collects, nearest first, the valid slots outside the tile that a window may reach:
// DIR left: tokens t0-1, t0-2, ... ; DIR right: tokens t0+ntok, ...   It stops at the first
// all-EOT token (excluded), at the row boundary, or once bpt slots are found.
template <int DIR, class Src>
__device__ __forceinline__ void halo_walk(const Src &src, int64_t t0, int ntok, int64_t T, int bpt,
                                          const TileLds &L) {
    const int lane = threadIdx.x & 63;
    int h = 0;
    int64_t base = DIR == kPullLeft ? t0 : t0 + ntok;
    while (h < bpt && (DIR == kPullLeft ? base > 0 : base < T)) {
        const int64_t tt = DIR == kPullLeft ? base - 1 - lane : base + lane;
        const bool active = DIR == kPullLeft ? tt >= 0 : tt < T;
        int cnt = 0, e = 0, tokid = 0;
        if (active) {
            e = 1;
            if constexpr (!Src::kRaw) tokid = src.token(tt);
            for (int k = 0; k < bpt; ++k) {
                int f;
                if constexpr (Src::kRaw) f = src.flags_at(tt * bpt + k);
                else f = src.flags_of(src.value(tokid, k));
                cnt += f & kValid;
                e &= (f >> 1);
            }
        }
        const unsigned long long em = __ballot(active && e);
        const int first = em ? __builtin_ctzll(em) : 64;  // nearest EOT token in this step
        const int c = (active && lane < first) ? cnt : 0;
        const int incl = wave_incl_add(c, lane);
        int p = h + incl - c;  // halo position of this token's first contributed slot
        if (c > 0 && p < bpt) {
            for (int kk = 0; kk < bpt; ++kk) {
                const int k = DIR == kPullLeft ? bpt - 1 - kk : kk;  // nearest slot first
                int f, payload;
                if constexpr (Src::kRaw) {
                    f = src.flags_at(tt * bpt + k);
                    payload = (int)(tt * bpt + k);
                } else {
                    payload = src.value(tokid, k);
                    f = src.flags_of(payload);
                }
                if (f & kValid) {
                    if (p < bpt) L.halo[p] = payload;
                    ++p;
                }
            }
        }
        h = min(bpt, h + __shfl(incl, 63, 64));
        if (first < 64) break;
        base += DIR == kPullLeft ? -64 : 64;
    }
    if (lane == 0) L.misc[kMiscHalo] = h;
}

// ---------------------------------------------------------------- the tile pass
// Precondition for the table source: L.tok[0..ntok) holds clamped token ids and L.val holds the
// slot values (fill_table_tile).  For the raw source L.val holds slot flags (fill_raw_tile).
// Postcondition (after the trailing barrier): cum, use, iseot, stream, halo are valid.
template <int DIR, class Src>
__device__ __forceinline__ void tile_scan_and_compact(const Src &src, int64_t t0, int ntok, int64_t T, int bpt,
                                                      const TileLds &L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sv = bpt | 1;
    // per-token valid count and all-EOT flag (one token per thread)
    int cnt = 0, e = 0;
    if (tid < ntok) {
        e = 1;
        for (int k = 0; k < bpt; ++k) {
            int f;
            if constexpr (Src::kRaw) f = L.val[tid * sv + k];
            else f = src.flags_of(L.val[tid * sv + k]);
            cnt += f & kValid;
            e &= (f >> 1);
        }
    }
    // block scans: prefix sum of counts; nearest EOT token at or before (left) / at or after (right)
    const int winc = wave_incl_add(cnt, lane);
    int bnd;
    if (DIR == kPullLeft) bnd = wave_incl_max(e ? tid : -1, lane);
    else bnd = wave_rincl_min(e ? tid : kMaxTileTokens, lane);
    // (a kernel may run these tile passes in a workgroup of more than kThreads threads: the extra waves
    // hold no token and only meet the barriers)
    if (lane == 63 && wave < kWaves) L.misc[wave] = winc;
    if (DIR == kPullLeft) { if (lane == 63 && wave < kWaves) L.misc[4 + wave] = bnd; }
    else { if (lane == 0 && wave < kWaves) L.misc[4 + wave] = bnd; }
    if (tid == 0) L.cum[0] = 0;
    __syncthreads();
    int incl = winc;
    for (int w = 0; w < min(wave, kWaves); ++w) incl += L.misc[w];
    if (DIR == kPullLeft) { for (int w = 0; w < min(wave, kWaves); ++w) bnd = max(bnd, L.misc[4 + w]); }
    else { for (int w = wave + 1; w < kWaves; ++w) bnd = min(bnd, L.misc[4 + w]); }
    if (tid < kMaxTileTokens && tid < ntok) L.cum[tid + 1] = incl;
    __syncthreads();
    if (tid < ntok) {
        // compaction (flat_valid_bytes, data_creation.py:131-132 / 248-249)
        int r = incl - cnt;
        for (int k = 0; k < bpt; ++k) {
            const int v = L.val[tid * sv + k];
            int f;
            if constexpr (Src::kRaw) f = v; else f = src.flags_of(v);
            if (f & kValid) L.stream[r++] = Src::kRaw ? (int)((t0 + tid) * bpt + k) : v;
        }
        const int h = L.misc[kMiscHalo];
        int use;
        if (DIR == kPullLeft) {
            // pull_range_start = cum[prev_eot+1] (or what the halo holds), data_creation.py:228-242
            const int seg_start = bnd >= 0 ? L.cum[bnd + 1] : -h;
            use = min(bpt, incl - seg_start);
        } else {
            // next_eot_valid_byte_start - start, data_creation.py:123-128
            const int seg_end = bnd < ntok ? L.cum[bnd] : L.cum[ntok] + h;
            use = max(0, min(bpt, seg_end - (incl - cnt)));
        }
        L.use[tid] = use;
        L.iseot[tid] = e;
    }
    __syncthreads();
}

// Result of the pull for slot (t,k) of the tile.  Returns the payload; *kind: 0 payload from the
// stream/halo, 1 pad, 2 the slot's own original content (EOT tokens keep their bytes,
// data_creation.py:169-173 / 298-302).
template <int DIR>
__device__ __forceinline__ int pulled_slot(const TileLds &L, int t, int k, int ntok, int bpt, int *kind) {
    if (L.iseot[t]) { *kind = 2; return 0; }
    const int use = L.use[t];
    if (DIR == kPullLeft) {
        if (k < bpt - use) { *kind = 1; return 0; }
        const int r = L.cum[t + 1] - bpt + k;  // gather_start + k_relative, right-aligned (:245,280)
        *kind = 0;
        return r >= 0 ? L.stream[r] : L.halo[-r - 1];
    } else {
        if (k >= use) { *kind = 1; return 0; }
        const int r = L.cum[t] + k;  // start_valid_byte_idx + k (:142)
        const int n = L.cum[ntok];
        *kind = 0;
        return r < n ? L.stream[r] : L.halo[r - n];
    }
}

// (tq, kq) thread layout for per-slot passes: kq = slot within the token (power-of-two padded so
// no integer division is needed), tq strides over tokens.  Lanes with kq >= bpt idle.
struct SlotLayout {
    int kq, tq, tstride;
    __device__ __forceinline__ explicit SlotLayout(int bpt) {
        const int bp2 = bpt <= 1 ? 1 : 1 << (32 - __builtin_clz(bpt - 1));
        kq = threadIdx.x < kThreads ? (int)(threadIdx.x & (bp2 - 1)) : 1 << 20;   // threads beyond kThreads idle
        tq = threadIdx.x / bp2;
        tstride = kThreads / bp2;
    }
};

// Loads the tile's token ids and table rows into LDS (tokens_to_bytes for the tile).
__device__ __forceinline__ void fill_table_tile(const SrcTable &src, int64_t t0, int ntok, int bpt,
                                                const TileLds &L) {
    const int sv = bpt | 1;
    if ((int)threadIdx.x < ntok) L.tok[threadIdx.x] = src.token(t0 + threadIdx.x);
    __syncthreads();
    const SlotLayout S(bpt);
    if (S.kq < bpt)
        for (int t = S.tq; t < ntok; t += S.tstride) L.val[t * sv + S.kq] = src.value(L.tok[t], S.kq);
    __syncthreads();
}

__device__ __forceinline__ void fill_raw_tile(const SrcRaw &src, int64_t t0, int ntok, int bpt, const TileLds &L) {
    const int sv = bpt | 1;
    const SlotLayout S(bpt);
    if (S.kq < bpt)
        for (int t = S.tq; t < ntok; t += S.tstride) L.val[t * sv + S.kq] = src.flags_at((t0 + t) * bpt + S.kq);
    __syncthreads();
}

}  // namespace mot
