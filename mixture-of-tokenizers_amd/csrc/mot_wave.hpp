// mot_wave.hpp -- the byte-index work of the fused forward done by ONE WAVE, with no workgroup barrier (gfx950, wave64).
//
// The tile machinery of mot_tile.hpp gives a 256-thread workgroup a tile of up to 256 tokens and meets at seven barriers per
// tile; at the 65 536-token shard a GPU sees under 8-way batch sharding every workgroup of the launch is resident at once, so
// those barriers and the three dependent fetches behind them (token ids, token->byte rows, halo) were fully exposed: 66 us
// with the ids pulled in-kernel against 57 us with the ids given.  Here a wave owns a *unit* of <= 64 consecutive tokens of
// one row and looks at a *window* of 64 tokens, one per lane, that holds the unit at its end (pull-left: the tokens in front
// of the unit are the halo a window may reach back into) or at its start (pull-right).  Valid counts, the EOT-bounded
// segments and the compaction of the window's non-pad bytes are wave scans and wave-private LDS; a window whose built-in
// halo is too short (fewer than bpt valid bytes in front of the unit and no EOT token among them) walks further out, 64
// tokens per step, exactly like halo_walk of mot_tile.hpp.  Waves never wait for each other.
//
// Restates, for the unit's tokens, scaled-pre-train/data_creation.py:61-67 (tokens_to_bytes), 179-305 (pull_from_left) and
// 71-176 (pull_from_right); the results are the same integers the tile kernels produce (tests/test_gpu_index.py compares both
// with the reference's goldens).
#pragma once
#include "mot_mix.hpp"

namespace mot {

// compiler-level ordering of one wave's LDS traffic: DS instructions of a wave execute in issue order, so all that is needed is
// that the compiler keeps the stores of one pass in front of the loads of the next (lanes read what OTHER lanes wrote)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-wide inclusive scans on the DPP path (row_shr 1, 2, 4, 8 inside each 16-lane row, then row_bcast:15 / row_bcast:31 chain
// the rows): six dependent VALU instructions instead of six LDS-crossbar round trips -- the scans sit on the latency chain of
// every wave's first tokens.  Lanes whose source falls outside the row keep the identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_int(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_scan_add(int v) {
    v += dpp_int<0x111, 0xf>(0, v);
    v += dpp_int<0x112, 0xf>(0, v);
    v += dpp_int<0x114, 0xf>(0, v);
    v += dpp_int<0x118, 0xf>(0, v);
    v += dpp_int<kDppRowBcast15, 0xa>(0, v);
    v += dpp_int<kDppRowBcast31, 0xc>(0, v);
    return v;
}
__device__ __forceinline__ int wave_scan_max(int v, int identity) {
    v = max(v, dpp_int<0x111, 0xf>(identity, v));
    v = max(v, dpp_int<0x112, 0xf>(identity, v));
    v = max(v, dpp_int<0x114, 0xf>(identity, v));
    v = max(v, dpp_int<0x118, 0xf>(identity, v));
    v = max(v, dpp_int<kDppRowBcast15, 0xa>(identity, v));
    v = max(v, dpp_int<kDppRowBcast31, 0xc>(identity, v));
    return v;
}
// inclusive min-scan running from lane 63 down to lane 0: a forward max-scan of the negated values over the mirrored lanes
__device__ __forceinline__ int wave_rscan_min(int v, int lane) {
    const int mirror = (63 - lane) << 2;
    int m = __builtin_amdgcn_ds_bpermute(mirror, -v);
    m = wave_scan_max(m, (int)0x80000000);
    return -__builtin_amdgcn_ds_bpermute(mirror, m);
}

// LDS of one wave: ids (and ids2 with two id tensors) of the unit's tokens for phase 2; for the index pass one array `ext` of
// table elements = [bpt halo slots | the window's compacted byte stream, 64 * bpt | bpt halo slots | 1 dump slot].  A pull-left
// window reads byte number r of the stream at ext[bpt + r], r >= -bpt: the bytes in front of the window sit, nearest last, in the
// slots before the stream, so one LDS read serves both; a pull-right window finds the bytes behind the window right after the
// stream's last byte.  The dump slot takes the stores of pad slots (the compaction is branch-free).
struct WaveLds {
    int32_t *ids;     // [unit * sv]  idsA of the unit's tokens, clamped to the byte table
    int32_t *ids2;    // [unit * sv]  idsB (two id tensors), else unused
    void *ext;        // [bpt + 64 * bpt + bpt + 1] table element type
};
__host__ __device__ inline size_t wave_lds_bytes(int unit, int bpt, bool dual, int stream_elem_bytes) {
    const size_t sv = (size_t)(bpt | 1);
    size_t b = (size_t)unit * sv * 4 * (dual ? 2 : 1) + ((size_t)66 * bpt + 1) * stream_elem_bytes;
    return (b + 15) & ~(size_t)15;
}
__device__ __forceinline__ WaveLds wave_lds_carve(unsigned char *base, int unit, int bpt, bool dual, int stream_elem_bytes) {
    WaveLds W;
    const int sv = bpt | 1;
    W.ids = (int32_t *)base;               base += (size_t)unit * sv * 4;
    W.ids2 = (int32_t *)base;              if (dual) base += (size_t)unit * sv * 4;
    W.ext = base;
    return W;
}

// One row of the token->byte table, visited slot by slot.  Rows whose byte length is a multiple of 16 (16 x int16, 8 x int16,
// 32 x int16, 8 x int32 ...) are fetched with 16-byte loads and the first two vectors stay in registers, so the three passes of
// the index work (count, compact, emit) cost ONE fetch; anything else is read element by element from L1.
template <typename E>
struct TtbRow {
    static constexpr int EPV = 16 / (int)sizeof(E);
    typedef E vec_t __attribute__((ext_vector_type(EPV)));
    const E *rowp;
    int bpt, ncached;   // vectors held in w0, w1 (wave-uniform)
    vec_t w0, w1;
    __device__ __forceinline__ void load(const E *p, int bpt_, bool vec_ok) {
        rowp = p; bpt = bpt_;
        ncached = vec_ok ? min(2, bpt_ / EPV) : 0;
        if (ncached > 0) w0 = *(const vec_t *)p;
        if (ncached > 1) w1 = *(const vec_t *)(p + EPV);
    }
    template <class F>
    __device__ __forceinline__ void each(F &&f) const {   // f(k, value) for k = 0 .. bpt-1, in order
        int k = 0;
        if (ncached > 0) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) f(e, (int32_t)w0[e]);
            k = EPV;
        }
        if (ncached > 1) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) f(EPV + e, (int32_t)w1[e]);
            k = 2 * EPV;
        }
        for (; k < bpt; ++k) f(k, (int32_t)rowp[k]);
    }
};

// Walks outwards from the window, 64 tokens per step, collecting nearest-first the valid bytes a window of the unit may still
// reach: DIR left: tokens base-1, base-2, ...; DIR right: tokens base, base+1, ...  Stops at the first all-EOT token
// (excluded), at the row boundary, or once bpt bytes are found.  Returns their number (wave-uniform); halo byte number q
// (0 = nearest) is stored at halo[q * hstep] (hstep -1: the slots in front of the stream, +1: behind its end).
template <int DIR, typename E>
__device__ __forceinline__ int wave_halo_walk(const SrcTable &src, int64_t base, int64_t T, int bpt, bool vec_ok, E *halo, int hstep) {
    const int lane = threadIdx.x & 63;
    int h = 0;
    while (h < bpt && (DIR == kPullLeft ? base > 0 : base < T)) {
        const int64_t tt = DIR == kPullLeft ? base - 1 - lane : base + lane;
        const bool active = DIR == kPullLeft ? tt >= 0 : tt < T;
        int cnt = 0, e = 0;
        TtbRow<E> r;
        r.load((const E *)src.ttb + (int64_t)(active ? src.token(tt) : 0) * bpt, bpt, vec_ok);
        if (active) {
            e = 1;
            r.each([&](int, int32_t v) {
                cnt += v != src.pad;
                e &= v == src.eot;
            });
        }
        const unsigned long long em = __ballot(active && e);
        const int first = em ? __builtin_ctzll(em) : 64;      // nearest EOT token of this step
        const int c = (active && lane < first) ? cnt : 0;
        const int incl = wave_scan_add(c);
        int p = h + incl - c;                                   // halo number of this token's nearest valid byte
        if (c > 0 && p < bpt) {
            // nearest byte first: a left walk numbers a token's bytes from its last slot to its first
            if (DIR == kPullLeft) p += c - 1;
            r.each([&](int, int32_t v) {
                if (v != src.pad) {
                    if (p < bpt) halo[p * hstep] = (E)v;
                    p += DIR == kPullLeft ? -1 : 1;
                }
            });
        }
        h = min(bpt, h + __shfl(incl, 63, 64));
        if (first < 64) break;
        base += DIR == kPullLeft ? -64 : 64;
    }
    return h;
}

// The byte ids of a unit from the token->byte table (+ pull), in three steps so that the caller can put its own memory requests
// between them: tokens() (the window's token ids), load_rows() (their table rows), finish() (everything else: LDS only, unless
// the window has to walk outwards for its halo).  finish() fills W.ids (and W.ids2 = the unpulled rows when `dual`) and
// writes the optional parity outputs and statistics.
template <int DIR, typename E>
struct WaveIndexer {
    const MixArgs &A;
    const WaveLds &W;
    int64_t row, u0, w0;
    int ntok, lane, unit_lane0, j, tokid;
    bool dual, in_unit, act, vec_ok;
    TtbRow<E> r;

    __device__ __forceinline__ WaveIndexer(const MixArgs &A_, const WaveLds &W_, int64_t row_, int64_t u0_, int ntok_, bool dual_)
        : A(A_), W(W_), row(row_), u0(u0_), ntok(ntok_), dual(dual_) {
        lane = threadIdx.x & 63;
        // the window: 64 tokens, one per lane, the unit at its end (pull-left) or at its start
        w0 = DIR == kPullLeft ? u0 + ntok - 64 : u0;
        unit_lane0 = DIR == kPullLeft ? 64 - ntok : 0;
        j = lane - unit_lane0;                                    // index of the lane's token in the unit
        in_unit = j >= 0 && j < ntok;
        const int64_t tt = w0 + lane;
        act = DIR == kPullNone ? in_unit : (tt >= 0 && tt < A.T);
        vec_ok = ((A.bpt * (int)sizeof(E)) & 15) == 0 && (((uintptr_t)A.ttb) & 15) == 0;
        tokid = 0;
    }
    __device__ __forceinline__ SrcTable src() const {
        return SrcTable{A.tokens + row * A.T, A.ttb, A.ttb_rows, A.ttb_elem, A.bpt, A.pad, A.eot, A.status};
    }
    // the lane's token id, clamped to the token->byte table (0 for lanes outside the row)
    __device__ __forceinline__ int tokens() {
        if (act) tokid = src().token(w0 + lane);
        return tokid;
    }
    __device__ __forceinline__ void load_rows() { r.load((const E *)A.ttb + (int64_t)tokid * A.bpt, A.bpt, vec_ok); }

    __device__ __forceinline__ void finish() {
        const int bpt = A.bpt, sv = bpt | 1;
        const int64_t T = A.T;
        E *ext = (E *)W.ext, *stream = ext + bpt, *dump = ext + 66 * bpt;
        int cnt = 0, e = 0;
        if (DIR != kPullNone) {
            e = act;
            r.each([&](int, int32_t v) {
                cnt += v != A.pad;                                // non_pad_mask, data_creation.py:93 / 199
                e &= v == A.eot;                                  // is_eot_token: every slot, :94 / 200
            });
            if (!act) cnt = 0;
        }
        int use = 0, incl = 0, total = 0;
        if (DIR != kPullNone) {
            incl = wave_scan_add(cnt);                            // cum_valid_bytes over the window
            total = __shfl(incl, 63, 64);
            const int excl = incl - cnt;
            // compaction of the window's valid bytes (flat_valid_bytes, :131-132 / 248-249), branch-free: a pad slot's store
            // goes to the dump slot
            {
                E *p = stream + excl;
                r.each([&](int, int32_t v) {
                    const bool valid = act && v != A.pad;
                    *(valid ? p : dump) = (E)v;
                    p += valid;
                });
            }
            int h = 0;
            if (DIR == kPullLeft) {
                // nearest all-EOT token at or before the lane; what the window reaches back to (:228-242)
                const int bnd = wave_scan_max(e ? lane : -1, -1);
                const int at_bnd = __shfl(incl, max(bnd, 0), 64); // cum[bnd + 1]
                const bool wants_halo = in_unit && !e && bnd < 0 && incl < bpt && w0 > 0;
                if (__any(wants_halo)) h = wave_halo_walk<kPullLeft, E>(src(), w0, T, bpt, vec_ok, stream - 1, -1);
                const int seg_start = bnd >= 0 ? at_bnd : -h;
                use = min(bpt, incl - seg_start);
            } else {
                // nearest all-EOT token at or after the lane (:123-128)
                const int bnd = wave_rscan_min(e ? lane : 64, lane);
                const int at_bnd = __shfl(excl, min(bnd, 63), 64); // cum[bnd]
                const bool wants_halo = in_unit && !e && bnd >= 64 && total - excl < bpt && w0 + 64 < T;
                if (__any(wants_halo)) h = wave_halo_walk<kPullRight, E>(src(), w0 + 64, T, bpt, vec_ok, stream + total, 1);
                const int seg_end = bnd < 64 ? at_bnd : total + h;
                use = max(0, min(bpt, seg_end - excl));
            }
            wave_lds_sync();                                      // stream / halo bytes written by other lanes are read below
        }
        // the unit's ids, slot by slot
        int pads_before = 0, pads_after = 0;
        bool oor = false;
        if (in_unit) {
            int32_t *ia = W.ids + j * sv, *ib = W.ids2 + j * sv;
            int64_t *op = A.out_ids_padded ? A.out_ids_padded + ((row * T + u0 + j) * bpt) : nullptr;
            int64_t *ol = A.out_ids_pulled ? A.out_ids_pulled + ((row * T + u0 + j) * bpt) : nullptr;
            // slot k of a pulled row is byte number first + k of the stream (halo slots included) when lo <= k < hi, else pad;
            // all-EOT tokens keep their row (:169-173 / 298-302)
            const int first = DIR == kPullLeft ? incl - bpt : incl - cnt;   // gather_start (:245, 280) / start_valid_byte_idx (:142)
            const int lo = DIR == kPullLeft ? bpt - use : 0, hi = DIR == kPullLeft ? bpt : use;
            r.each([&](int k, int32_t own) {
                int32_t v = own;
                if (DIR != kPullNone) {
                    const bool take = !e && k >= lo && k < hi;
                    const int32_t s = (int32_t)stream[take ? first + k : 0];
                    v = e ? own : (take ? s : A.pad);
                }
                if (op) op[k] = own;
                if (ol) ol[k] = v;
                pads_before += own == A.pad;
                pads_after += v == A.pad;
                const bool bad = (uint64_t)(uint32_t)v >= (uint64_t)A.byte_rows;
                oor |= bad;
                ia[k] = bad ? 0 : v;
                if (dual) {
                    const bool bad2 = (uint64_t)(uint32_t)own >= (uint64_t)A.byte_rows;
                    oor |= bad2;
                    ib[k] = bad2 ? 0 : own;
                }
            });
        }
        if (A.status && oor) atomicOr(A.status, kStatusByteOor);
        if (A.counters) {  // runs/79_mot-in_toks-valemb.py:484-488
            pads_before = (int)wave_sum((float)pads_before);       // <= 64 * 64 per wave: exact in fp32
            pads_after = (int)wave_sum((float)pads_after);
            if (lane == 0) {
                atomicAdd((unsigned long long *)A.counters + 0, (unsigned long long)ntok);
                atomicAdd((unsigned long long *)A.counters + 1, (unsigned long long)ntok * bpt);
                atomicAdd((unsigned long long *)A.counters + 2, (unsigned long long)pads_before);
                atomicAdd((unsigned long long *)A.counters + 3, (unsigned long long)pads_after);
            }
        }
        wave_lds_sync();
    }
};

// ids given (the module seam: int64 tensors as the reference's loader emits them): the unit's ids, coalesced, into W.ids / W.ids2
__device__ __forceinline__ void wave_ids_given(const MixArgs &A, const WaveLds &W, int64_t row, int64_t u0, int ntok) {
    const int lane = threadIdx.x & 63;
    const int bpt = A.bpt, sv = bpt | 1;
    const int64_t base = (row * A.T + u0) * bpt;
    const int n = ntok * bpt;
    const float inv = 1.0f / (float)bpt;
    for (int i = lane; i < n; i += 64) {
        const int t = __float2int_rd(((float)i + 0.5f) * inv);     // i / bpt, exact for i < 2^22
        const int k = i - t * bpt;
        const int64_t a = A.ids_a[base + i];
        int ia = (int)a;
        if ((uint64_t)a >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); ia = 0; }
        W.ids[t * sv + k] = ia;
        if (A.ids_b) {
            const int64_t b = A.ids_b[base + i];
            int ib = (int)b;
            if ((uint64_t)b >= (uint64_t)A.byte_rows) { if (A.status) atomicOr(A.status, kStatusByteOor); ib = 0; }
            W.ids2[t * sv + k] = ib;
        }
    }
    if (A.counters && lane == 0) {
        atomicAdd((unsigned long long *)A.counters + 0, (unsigned long long)ntok);
        atomicAdd((unsigned long long *)A.counters + 1, (unsigned long long)ntok * bpt);
    }
    wave_lds_sync();
}

}  // namespace mot
