// mot_concat16.hip -- the concat + linear mixin in the production dtype as ONE dense-MFMA kernel (gfx950):
//     x = rms_norm?( bf16( W . cat(norm?(E_t[tok]), norm?(E_b[id_0]), ..., norm?(E_b[id_{bpt-1}])) + bias ) )
// (ByteMixinConcat on FlexibleEmbedding's outputs, scaled-pre-train/train_gpt.py:342-379, 430-443, CastedLinear 185-186;
//  DigitMixinConcat, mathblations/model.py:256-268), bf16 tables / weight / output, fp32 accumulation.
//
// The composed path (index kernels -> concat_rows_kernel -> gemm_rows_bf16_kernel -> rows_rms_inplace_kernel) spends 40 % of
// its time outside the contraction and runs the contraction itself at 23 % of the bf16 MFMA peak: 128 x 128 blocks read one
// 16-byte LDS fragment per MFMA, and the concat operand makes two trips through HBM.  Here
//   * a workgroup owns WHOLE output rows (64 MT tokens x all Dm = 128 NT columns; 8 waves as 2 x 4, a wave 32 MT x 32 NT), so
//     the row norm is an epilogue, and a fragment feeds MT or NT MFMAs;
//   * the A operand is GATHERED: per 32-deep step a thread fetches one 16-byte piece of a token row or byte row, scales it by
//     the row's rms factor, rounds to bf16 (the reference's rounding point: norm() returns a bf16 tensor) and writes it into the
//     step's LDS tile -- the concat tensor never exists;
//   * W is staged by LDS-DMA (global_load_lds, 16 bytes per lane, no staging registers) into NS stages, XOR-swizzled on the
//     SOURCE side so that the lane-linear LDS image reads back without bank conflicts: piece p of row n sits at
//     n * 64 + ((p ^ (n >> 2)) & 3) * 16;
//   * the output tile leaves through LDS (the W stages, free by then) as whole 16-byte pieces: a lane holds NT consecutive outputs
//     of a row (the stage rows of W are permuted for that), 32 lanes then take a row, sum its squares, scale and store.
// Two barriers per step (the gathered tile is single-buffered: LDS is full); the DMA of steps s + 1 and s + 2 and the gathered
// pieces of steps s + 1 and s + 2 are in flight while step s multiplies; every wait in the loop is counted (see the step).
#include "mot_wave.hpp"
#include <type_traits>
// (clang wants the explicit captures below for operands of inline asm inside generic lambdas, and then calls them unused)
#pragma clang diagnostic ignored "-Wunused-lambda-capture"

namespace mot {

typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef float f32x16c __attribute__((ext_vector_type(16)));

struct C16Args {
    const int32_t *tokens;     // [n]
    const int64_t *ids;        // [n, bpt] as the reference's loader emits them, or
    const uint16_t *ids16;     // [n, bpt] compact and range-checked, from wave_ids16_kernel (ids pulled from the token->byte table)
    int64_t n;
    const __bf16 *tok_table; int64_t tok_rows; int Dt;
    const __bf16 *byte_table; int64_t byte_rows; int Db; int bpt;
    int norm_tok;              // token rows are rms-normalised (factor computed per tile, below)
    const float *byte_rnorm;   // [byte_rows] rms factors of the byte table, or null: no byte norm, or
    int norm_byte_here;        // the workgroups compute the factors themselves (tables of up to kC16NormHere rows)
    const __bf16 *W;           // [Dm, K]
    const __bf16 *bias;        // [Dm] or null
    int K, Dm, tok_lo, byte_lo;
    int norm_out;
    float eps;
    __bf16 *out;               // [n, Dm]
    float *row_rnorm;          // optional [n]
    uint32_t *status;
};

constexpr int kC16Threads = 512;
constexpr int kC16NormHere = 1024;
#ifdef C16_STAMPS   // dev: wall-clock stamps (10 ns units) of workgroup phases, printed once by the launcher
__device__ unsigned long long c16_stamps[4096 * 8];
#define C16_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) c16_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define C16_STAMP(i) do {} while (0)
#endif
typedef int i32x4c __attribute__((ext_vector_type(4)));

// LDS reads the COMPILER must not see.  hipcc orders every LDS read it emits behind every LDS-DMA still in flight (it cannot
// tell which bytes the DMA writes), i.e. s_waitcnt vmcnt(0) in front of the first ds_read after a global_load_lds -- which
// would serialise the DMA of step s + 2 with the multiplies of step s.  The reads of the loop are therefore inline asm, ordered
// against the DMA by hand: a stage is read only after the barrier behind the wait that retired it (see the loop).
#define C16_FRAG(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
__device__ __forceinline__ uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }
__device__ __forceinline__ uint32_t lds_u16_now(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ float lds_f32_now(uint32_t addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// workgroup barrier that waits for this wave's LDS traffic only (no vmcnt drain: the DMA and the gathered pieces stay in flight)
__device__ __forceinline__ void c16_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// NS stages of W in LDS (NS - 1 steps of DMA in flight), ONE stage of the gathered operand (its next step waits in registers)
#define C16_PASS_BEGIN for (int hh = 0; hh < NH; ++hh) {
#define C16_PASS_END }
// NH: column passes.  model_dim 1024 does not fit a 128-token tile's accumulators (256 registers); as 64-token tiles it streamed all
// of W per 64 tokens and paid 8 DMA instructions per 16 MFMAs.  NH = 2 keeps the 128-token tile and computes its columns in two
// passes of 512 (the tile's ids, rms factors and token-row norms are made once; the gathered operand is walked twice, out of L2):
// pass 0 stores y un-normalised and keeps each row's sum of squares, pass 1 completes the sum, stores its half normalised and
// rescales the first half in place (every lane re-reads exactly the pieces it wrote).
template <int MT, int NT, int NS, int NH = 1>
__global__ __launch_bounds__(kC16Threads) void concat16_gemm_kernel(const C16Args P) {
    constexpr int BM = 64 * MT, BN = 128 * NT, WMR = 32 * MT, WNR = 32 * NT, PD = NS - 1;
    constexpr int kStageB = BN * 64, kStageA = BM * 64, kDma = BN * 4 / kC16Threads;
    static_assert(PD == 1 || PD == 2, "one or two steps ahead");
    extern __shared__ __attribute__((aligned(16))) char lds_c[];
    char *sA = lds_c + NS * kStageB;
    uint16_t *sIds = (uint16_t *)(sA + kStageA);                                  // [BM * bpt]
    float *sRn = (float *)(lds_c + NS * kStageB + kStageA + ((BM * P.bpt * 2 + 15) & ~15));   // [byte_rows] rms factors of the byte rows (1 when that part is not normalised)
    float *sSS = sRn + P.byte_rows;                                               // [BM] (NH = 2) a row's sum of squares over the first column pass
    const uint32_t oB = lds_off(lds_c), oA = lds_off(sA), oIds = lds_off(sIds), oRn = lds_off(sRn);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 2) * WMR, wn = (wave & 3) * WNR;
    const int64_t j0 = (int64_t)blockIdx.x * BM;
    const int nrows = (int)min((int64_t)BM, P.n - j0);
    const int K = P.K, bpt = P.bpt;
    // ---- W by LDS-DMA: wave-instruction j = i * 8 + wave writes 1 KiB = stage rows 16 j .. 16 j + 15, lane -> (row, physical piece).
    // Stage row q of a wave's strip holds W row strip + (q % 32) * NT + q / 32: MFMA column li of tile b is output column
    // li * NT + b, so that a lane ends up with NT CONSECUTIVE outputs of a row (packed stores in the epilogue).
    uint32_t goff[kDma];   // byte offset into W of this lane's piece of step 0, per DMA instruction
#pragma unroll
    for (int i = 0; i < kDma; ++i) {
        const int q = (i * 8 + wave) * 16 + (lane >> 2), strip = q / WNR, within = q - strip * WNR;
        const int nrow = strip * WNR + (within & 31) * NT + (within >> 5);
        goff[i] = (uint32_t)(nrow * K + 8 * (((lane & 3) ^ (q >> 2)) & 3)) * 2u;
    }
    C16_STAMP(0);
    const int nsteps = K / 32;
    const char *Wb = (const char *)P.W;   // the W rows of the current column pass
    auto b_request = [&](int s) {   // (a step past the end re-reads the last one into a stage nobody reads: the loop stays branch-free)
        char *sB = lds_c + (s % NS) * kStageB;
        const uint32_t ko = 64u * (uint32_t)min(s, nsteps - 1);
#pragma unroll
        for (int i = 0; i < kDma; ++i) {
            const char *g = Wb + (goff[i] + ko);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(sB + (i * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    b_request(0);
    if (PD == 2) b_request(1);
    // ---- the tile's own inputs.  Everything that does not depend on another load is requested first (the token id, the byte ids,
    // the byte rows' rms factors), so that the prologue costs two memory round trips -- ids, then the token rows for their norm --
    // instead of one per item.
    const bool a_thread = tid < BM * 4;   // (whole waves: BM * 4 is a multiple of 64)
    const int arow = tid >> 2, apiece = tid & 3;
    int tok = a_thread ? P.tokens[j0 + min(arow, nrows - 1)] : 0;
    const int n_ids = nrows * bpt;        // (rows past the batch repeat the last valid token: computed, never stored)
    if (P.ids16) {   // compact ids from the wave-local index pass: 8 bytes per thread and trip
        for (int i0 = tid * 4; i0 < BM * bpt; i0 += 4 * kC16Threads) {
            uint16_t v[4];
            if (i0 + 3 < n_ids && (bpt & 3) == 0) {
                const uint2 w = *(const uint2 *)(P.ids16 + j0 * bpt + i0);
                v[0] = (uint16_t)w.x; v[1] = (uint16_t)(w.x >> 16); v[2] = (uint16_t)w.y; v[3] = (uint16_t)(w.y >> 16);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u, r = min(i / bpt, nrows - 1);
                    v[u] = P.ids16[(j0 + r) * bpt + i % bpt];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u < BM * bpt) sIds[i0 + u] = v[u];
        }
    } else {
        for (int i0 = tid; i0 < BM * bpt; i0 += 4 * kC16Threads) {   // four loads in flight per thread
            int64_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * kC16Threads, BM * bpt - 1);
                v[u] = P.ids[(j0 + min(i / bpt, nrows - 1)) * bpt + i % bpt];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kC16Threads;
                if ((uint64_t)v[u] >= (uint64_t)P.byte_rows) { if (P.status) atomicOr(P.status, kStatusByteOor); v[u] = 0; }
                if (i < BM * bpt) sIds[i] = (uint16_t)v[u];
            }
        }
    }
    // rms factors of the byte rows: from the caller's table, or (small tables: norm_byte_here) computed here, a thread per row --
    // the table is L2-resident and a separate launch for 458 rows costs more than the 29 KB every workgroup re-reads
    for (int i = tid; i < (int)P.byte_rows; i += kC16Threads) {
        float r = 1.f;
        if (P.byte_rnorm) r = P.byte_rnorm[i];
        else if (P.norm_byte_here) {
            float ss = 0.f;
            const __bf16 *brow = P.byte_table + (int64_t)i * P.Db;
            for (int p0 = 0; p0 < P.Db / 8; p0 += 4) {   // four loads in flight
                bf16x8c v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8c *)(brow + 8 * min(p0 + u, P.Db / 8 - 1));
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (p0 + u < P.Db / 8) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ss += (float)v[u][e] * (float)v[u][e];
                    }
            }
            r = rms_scale(ss, P.Db, P.eps);
        }
        sRn[i] = r;
    }
    // ---- this thread's piece of the gathered operand: row arow, logical 16-byte piece apiece of every 64-byte step row
    float rn_tok = 1.f;
    if ((uint64_t)(uint32_t)tok >= (uint64_t)P.tok_rows) { if (P.status) atomicOr(P.status, kStatusTokenOor); tok = 0; }
    const __bf16 *trow = P.tok_table + (int64_t)tok * P.Dt;
    if (P.norm_tok) {   // the four threads of a row share its sum of squares (the row comes back out of L2 in the steps below)
        float ss = 0.f;
        if (a_thread)
            for (int p0 = apiece; p0 < P.Dt / 8; p0 += 32) {   // eight loads in flight
                bf16x8c v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const bf16x8c *)(trow + 8 * min(p0 + 4 * u, P.Dt / 8 - 1));
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (p0 + 4 * u < P.Dt / 8) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ss += (float)v[u][e] * (float)v[u][e];
                    }
            }
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        rn_tok = rms_scale(ss, P.Dt, P.eps);
    }
    const uint32_t a_dst = oA + arow * 64 + ((apiece ^ (arow >> 2)) & 3) * 16;
    // the walk over this thread's pieces: k = 8 apiece, + 32 per step; inside the byte part (slot, within) advance with it.
    // Branch-free (and one global load per request whatever the part) so that hipcc can COUNT the loads in flight at the commit.
    int ak = 8 * apiece, slot = 0, within = 0;
    {   // the first of them inside the byte part
        const int d = P.Dt - 8 * apiece;
        const int first = P.byte_lo == 0 ? 8 * apiece : (d > 0 ? (d + 31) / 32 : 0) * 32 - d;
        slot = first / P.Db; within = first - slot * P.Db;
    }
    const int dslot = 32 / P.Db, dwithin = 32 - dslot * P.Db;
    C16_STAMP(1);
    __syncthreads();   // sIds, sRn  (hipcc drains the DMA of the first stages here: once per tile)
    C16_STAMP(2);
    auto a_request = [&](i32x4c &raw, float &scale) {
        const int kt = ak - P.tok_lo;
        const bool in_tok = (unsigned)kt < (unsigned)P.Dt;
        const int sl = min(slot, bpt - 1);   // (requests past the last step read a valid row and are never used)
        const int id = (int)lds_u16_now(oIds + (arow * bpt + sl) * 2);
        const float rb = lds_f32_now(oRn + id * 4);
        const __bf16 *src = in_tok ? trow + kt : P.byte_table + (int64_t)id * P.Db + within;
        // (asm: hipcc answers a plain load among LDS-DMA with vmcnt(0) at its use; the commit below counts instead.  The registers
        //  stay pending until that wait: nothing else may touch them -- they are outputs here and operands of the wait only)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw) : "v"(src) : "memory");
        scale = in_tok ? rn_tok : rb;
        int w2 = within + dwithin, s2 = slot + dslot;
        if (w2 >= P.Db) { w2 -= P.Db; ++s2; }
        within = in_tok ? within : w2;
        slot = in_tok ? slot : s2;
        ak += 32;
    };
    // scale in fp32, round once to bf16 (the reference's norm() output), into the step's tile (the caller has waited for the piece)
    auto a_commit = [&](const i32x4c &raw_bits, float scale) {
        const bf16x8c raw = __builtin_bit_cast(bf16x8c, raw_bits);
        bf16x8c v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)raw[e] * scale);
        *(__attribute__((address_space(3))) bf16x8c *)(uintptr_t)a_dst = v;
    };
    f32x16c acc[MT][NT];
    // fragment addresses: row = (wm | wn) + 32 t + li, piece (2 kk + h) ^ (row >> 2): the tile index t only adds t * 2048, kk flips bit 5
    const uint32_t fa0 = oA + (wm + li) * 64 + ((h ^ (li >> 2)) & 3) * 16;
    const uint32_t fb0 = (wn + li) * 64 + ((h ^ (li >> 2)) & 3) * 16;
    i32x4c r0_raw = {0, 0, 0, 0}, r1_raw = {0, 0, 0, 0};
    float r0_scale = 1.f, r1_scale = 1.f;
    constexpr bool kAllGather = BM * 4 == kC16Threads;   // every thread carries a piece: no branch around the requests
    constexpr int kInflight = PD == 2 ? kDma + 1 : 0;
    const int ak_first = ak, slot_first = slot, within_first = within;
    C16_PASS_BEGIN
    if (NH > 1 && hh > 0) {   // the next column pass: its rows of W, the walk over the gathered operand from the start
        __syncthreads();      // every wave is done with the staging area of the previous pass (it lies in the stages of W)
        Wb = (const char *)P.W + (size_t)hh * BN * K * 2;
        b_request(0);
        if (PD == 2) b_request(1);
        ak = ak_first; slot = slot_first; within = within_first;
    }
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    if (kAllGather || a_thread) {
        a_request(r0_raw, r0_scale);
        if (PD == 2) a_request(r1_raw, r1_scale);
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r0_raw) : "n"(PD - 1) : "memory");
        a_commit(r0_raw, r0_scale);
    }
    if (NH > 1 && hh > 0) {   // (pass 0: the barrier of the prologue) this wave's DMA of the first stages is older than the piece it just
        if (!(kAllGather || a_thread)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // waited for; every wave's must have landed
        c16_barrier();
    }
    // ---- the step.  One wave's share of step s is NP = 2 NT "positions" (kk, b) of MT MFMAs each; all eight waves run in lockstep
    // between the two barriers of a step, so whatever is not an MFMA has to be issued BETWEEN MFMAs or the matrix pipes idle:
    //   * the fragments of W roll through four register slots, read three positions ahead (LDS latency under the MFMAs of the
    //     positions in between), the first three of a step right behind the previous step's second barrier;
    //   * the NT DMA instructions of W step s + PD go out one per position, the gathered piece s + PD at position NP - 3 (its
    //     LDS id read drains the LDS queue: every fragment of the step is requested by then);
    //   * waits are COUNTED: lgkmcnt(n) leaves the younger fragment reads in flight, vmcnt(NT + 1) the requests of this step.
    // Order: [barrier 1: the gathered tile s is committed] fragments of the gathered tile; positions; wait for piece s + 1 and the
    // DMA of W step s + 1 (loads retire in order: one count covers both); [barrier 2: everyone has read tile s, W step s + 1 is
    // in LDS for everyone] first fragments of W step s + 1; commit piece s + 1.
    constexpr int NP = 2 * NT;
    i32x4c af[2 * MT], bq[4];
    auto frag_b = [&bq](auto idx, uint32_t fb) {   // fragment idx = (kk, b) of the stage at fb -> slot idx % 4
        constexpr int i = decltype(idx)::value;
        C16_FRAG(bq[i % 4], fb ^ ((i / NT) * 32), (i % NT) * 2048);
    };
    auto dma_piece = [&](int s, auto ic) {
        constexpr int i = decltype(ic)::value;
        char *sB = lds_c + (s % NS) * kStageB;
        const char *g = Wb + (goff[i] + 64u * (uint32_t)min(s, nsteps - 1));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(sB + (i * 8 + wave) * 1024), 16, 0, 0);
    };
    auto step = [&](int s, i32x4c &ld_raw, float &ld_scale, i32x4c &cm_raw, const float &cm_scale) {
        c16_barrier();
        const uint32_t fb = oB + (s % NS) * kStageB + fb0;
        static_for<0, 2 * MT>([&af, fa0](auto jc) {
            constexpr int j = decltype(jc)::value;
            C16_FRAG(af[j], fa0 ^ ((j / MT) * 32), (j % MT) * 2048);
        });
        static_for<0, NP>([&, &af = af, &bq = bq](auto pc) {
            constexpr int p = decltype(pc)::value, kk = p / NT, b = p % NT;
            if constexpr (p == (NP >= 3 ? NP - 3 : 0))
                if (kAllGather || a_thread) a_request(ld_raw, ld_scale);
            // LDS reads behind barrier 1, in order: the 2 MT fragments of the gathered tile, then W fragments 3, 4, ... (one per position)
            constexpr int issued = 2 * MT + (p < NP - 3 ? p : NP - 3);
            constexpr int need_b = p >= 3 ? 2 * MT + p - 3 : -1;
            constexpr int need_a = p == 0 ? MT - 1 : (p == NT ? 2 * MT - 1 : -1);
            constexpr int need = need_b > need_a ? need_b : need_a;
            if constexpr (need >= 0) {
                if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(bq[p % 4]), "+v"(af[kk * MT]), "+v"(af[kk * MT + 1]) : "n"(issued - need - 1));
                else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bq[p % 4]), "+v"(af[kk * MT]) : "n"(issued - need - 1));
            } else {
                asm volatile("" : "+v"(bq[p % 4]));
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8c, af[kk * MT + a]), __builtin_bit_cast(bf16x8c, bq[p % 4]),
                                                                    acc[a][b], 0, 0, 0);
            if constexpr (p + 3 < NP) frag_b(std::integral_constant<int, p + 3>{}, fb);
            if constexpr (p < kDma) dma_piece(s + PD, pc);
        });
        if (kAllGather || a_thread) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(cm_raw) : "n"(kInflight) : "memory");
        else if (PD == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");   // a wave without pieces: its DMA of step s + 1
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        c16_barrier();
        const uint32_t fbn = oB + ((s + 1) % NS) * kStageB + fb0;
        static_for<0, 3>([&](auto ic) { frag_b(ic, fbn); });
        if (kAllGather || a_thread) a_commit(cm_raw, cm_scale);
    };
    static_for<0, 3>([&](auto ic) { frag_b(ic, oB + fb0); });   // (the stage landed in front of the barrier above)
    C16_STAMP(3);
    if (PD == 2) {
        for (int s = 0; s < nsteps; s += 2) {
            step(s, r0_raw, r0_scale, r1_raw, r1_scale);
            if (s + 1 < nsteps) step(s + 1, r1_raw, r1_scale, r0_raw, r0_scale);
        }
    } else {
        for (int s = 0; s < nsteps; ++s) step(s, r0_raw, r0_scale, r0_raw, r0_scale);
    }
    // What was requested past the last step is still on its way INTO registers the compiler considers free from here on: the
    // fragment reads behind the last barrier, the gathered pieces.  Hold the registers until everything has landed.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(r0_raw), "+v"(r1_raw), "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) : : "memory");
    C16_STAMP(4);
    // ---- epilogue.  C/D layout: MFMA column li of tile b = output column wn + li NT + b, row = wm + 32 a + (reg & 3) + 8 (reg >> 2) + 4 h.
    // y = bf16(acc + bias): F.linear on bf16 operands returns a bf16 tensor (train_gpt.py:185-186); norm() upcasts it (172-173, 443).
    // The tile leaves through LDS in halves of 32 MT rows (they fit in the stages of W): a lane packs its NT consecutive outputs of a
    // row; then 32 lanes take a row, sum its squares, scale and store whole 16-byte pieces.
    const int cb = NH > 1 ? hh * BN : 0;   // first output column of this pass
    float bv[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) bv[b] = P.bias ? (float)P.bias[cb + wn + li * NT + b] : 0.f;
    __syncthreads();   // every wave is done with the last step's tiles
    C16_STAMP(5);
    __bf16 *stage = (__bf16 *)lds_c;   // [WMR][BN]
    static_assert(WMR * BN * 2 <= NS * kStageB, "a half tile fits in the stages");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((wave >> 2) == half) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;   // row inside the half
                    uint32_t *dst = (uint32_t *)(stage + lr * BN + wn + li * NT);
#pragma unroll
                    for (int b = 0; b < NT; b += 2) {
                        typedef __bf16 bf16x2c __attribute__((ext_vector_type(2)));
                        bf16x2c pr;
                        pr[0] = (__bf16)(acc[a][b][r] + bv[b]);
                        pr[1] = (__bf16)(acc[a][b + 1][r] + bv[b + 1]);
                        dst[b / 2] = __builtin_bit_cast(uint32_t, pr);
                    }
                }
        }
        c16_barrier();   // (LDS only: the first half's stores to HBM stay in flight under the second half's staging)
        constexpr int PP = NT / 2;   // 16-byte pieces of a row per lane: BN / 8 pieces over 32 lanes
        for (int lr = wave * 2 + h; lr < WMR; lr += 16) {
            const int row = half * WMR + lr;
            bf16x8c v[PP];
            float ss = 0.f;
#pragma unroll
            for (int p = 0; p < PP; ++p) {
                v[p] = *(const bf16x8c *)(stage + lr * BN + 8 * (li + 32 * p));
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += (float)v[p][e] * (float)v[p][e];
            }
            float rs = 1.f;
            const bool first_of_two = NH > 1 && hh + 1 < NH;   // the row's other columns are still to come: no factor yet
            if (P.norm_out) {
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) ss += __shfl_xor(ss, o, 64);
                if (first_of_two) {
                    if (li == 0) sSS[row] = ss;
                } else {
                    if (NH > 1) ss += sSS[row];
                    rs = rms_scale(ss, P.Dm, P.eps);
                    if (P.row_rnorm && li == 0 && row < nrows) P.row_rnorm[j0 + row] = rs;
#pragma unroll
                    for (int p = 0; p < PP; ++p)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[p][e] = (__bf16)((float)v[p][e] * rs);
                }
            }
            if (row < nrows) {
                __bf16 *orow = P.out + (j0 + row) * (int64_t)P.Dm + cb;
#pragma unroll
                for (int p = 0; p < PP; ++p) {
                    if (first_of_two && P.norm_out) *(bf16x8c *)(orow + 8 * (li + 32 * p)) = v[p];   // (comes back in the last pass: keep it in L2)
                    else __builtin_nontemporal_store(v[p], (bf16x8c *)(orow + 8 * (li + 32 * p)));
                }
                if (NH > 1 && !first_of_two && P.norm_out) {
                    // the first pass's columns of this row, un-normalised so far: this lane re-reads the pieces IT stored (device-scope
                    // loads: not through this CU's L1) and rescales them
                    for (int c0 = 0; c0 < cb; c0 += BN)
#pragma unroll
                        for (int p = 0; p < PP; ++p) {
                            uint32_t *q = (uint32_t *)(orow - cb + c0 + 8 * (li + 32 * p));
                            i32x4c w;
#pragma unroll
                            for (int e = 0; e < 4; ++e) w[e] = (int)__hip_atomic_load(q + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            bf16x8c u = __builtin_bit_cast(bf16x8c, w);
#pragma unroll
                            for (int e = 0; e < 8; ++e) u[e] = (__bf16)((float)u[e] * rs);
                            __builtin_nontemporal_store(u, (bf16x8c *)q);
                        }
                }
            }
        }
        if (half == 0) c16_barrier();   // the staging area is rewritten
        C16_STAMP(6 + half);
    }
    C16_PASS_END
}

static size_t c16_lds_base(int MT, int NT, int NS, int bpt);
template <int MT, int NT, int NS, int NH = 1>
static int launch_c16(const C16Args &P0, hipStream_t stream) {
    constexpr int BM = 64 * MT;
    const C16Args &P = P0;
    const size_t lds = c16_lds_base(MT, NT, NS, P.bpt) + (size_t)P.byte_rows * 4 + (NH > 1 ? BM * 4 : 0);
    if (lds > 160 * 1024) return set_error(MOT_EUNSUPPORTED, "concat16: needs %zu B of LDS", lds);
    static std::atomic<uint64_t> ok{0};
    if (int rc = ensure_max_dyn_lds((const void *)concat16_gemm_kernel<MT, NT, NS, NH>, ok, "concat16_gemm_kernel")) return rc;
    const int64_t blocks = (P.n + BM - 1) / BM;
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "concat16: too many rows");
    hipLaunchKernelGGL((concat16_gemm_kernel<MT, NT, NS, NH>), dim3((unsigned)blocks), dim3(kC16Threads), lds, stream, P);
#ifdef C16_STAMPS
    static int calls = 0;
    if (++calls == 60) {
        static unsigned long long h[4096 * 8];
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h, HIP_SYMBOL(c16_stamps), sizeof(h));
        const int nb = (int)(blocks < 4096 ? blocks : 4096);
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nb; ++b) t0 = h[b * 8] < t0 ? h[b * 8] : t0;
        double sum[8] = {0}, mx[8] = {0};
        for (int b = 0; b < nb; ++b)
            for (int i = 0; i < 8; ++i) { const double v = (double)(h[b * 8 + i] - t0) * 0.01; sum[i] += v; mx[i] = v > mx[i] ? v : mx[i]; }
        fprintf(stderr, "c16 stamps (us since first start; mean / max over %d workgroups):", nb);
        for (int i = 0; i < 8; ++i) fprintf(stderr, "  [%d] %.1f/%.1f", i, sum[i] / nb, mx[i]);
        fprintf(stderr, "\n  first-round groups only (start < 2 us):");
        double s2[8] = {0}; int n2 = 0;
        for (int b = 0; b < nb; ++b) if ((h[b * 8] - t0) < 200) { ++n2; for (int i = 0; i < 8; ++i) s2[i] += (double)(h[b * 8 + i] - t0) * 0.01; }
        for (int i = 0; i < 8; ++i) fprintf(stderr, "  [%d] %.1f", i, s2[i] / (n2 ? n2 : 1));
        fprintf(stderr, "  (%d groups)\n", n2);
    }
#endif
    return check_launch("concat16_gemm_kernel");
}

// ------------------------------------------------------------------------------------------ ids pulled from the token->byte table
// The byte-index work of the loader (tokens_to_bytes + pull, data_creation.py:60-76, 79-176, 179-305) for the gather-GEMM above:
// the wave-local indexer of the fused SUM kernel (mot_wave.hpp: a unit of 16 or 32 tokens per wave, its 64-token window, halo walk
// across the window's edge) writes the pulled ids ONCE, as 16-bit values (2 bytes per slot instead of the two int64 tensors the
// separate index kernels write and read back: 2 MB instead of 2 x 8 + 8 MB at 65 536 tokens), range-checked, plus the int64 parity
// outputs and the pad statistics when the caller asked for them.
// tokens per wave: 32 (7.2 us at 65 536 tokens; 9.0 with 16, 13.4 with 8: fewer windows to build); 16 for small batches
static int ids16_unit(int64_t n) { return n >= 16384 ? 32 : 16; }
template <int DIR, typename E>
__global__ __launch_bounds__(kThreads) void wave_ids16_kernel(const MixArgs A, uint16_t *__restrict__ ids16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_wave[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t unit_id = (int64_t)blockIdx.x * kWaves + wave;
    if (unit_id >= A.n_units) return;             // no workgroup barrier below: a wave may leave on its own
    const int64_t row = unit_id / A.units_per_row;
    const int64_t u0 = (unit_id - row * A.units_per_row) * A.unit;
    const int ntok = (int)min((int64_t)A.unit, A.T - u0);
    const WaveLds W = wave_lds_carve(lds_wave + (size_t)wave * A.wave_lds, A.unit, A.bpt, false, DIR != kPullNone ? (int)sizeof(E) : 0);
    WaveIndexer<DIR, E> ix(A, W, row, u0, ntok, false);
    ix.tokens();
    ix.load_rows();
    ix.finish();
    const int bpt = A.bpt, sv = bpt | 1, n = ntok * bpt;
    uint16_t *dst = ids16 + (row * A.T + u0) * bpt;
    const float inv = 1.0f / (float)bpt;
    for (int i = lane; i < n; i += 64) {
        const int t = __float2int_rd(((float)i + 0.5f) * inv);     // i / bpt, exact for i < 2^22
        dst[i] = (uint16_t)W.ids[t * sv + (i - t * bpt)];
    }
}

int launch_wave_ids16(const MotEmbedMixDesc &d, uint16_t *ids16, hipStream_t stream) {
    MixArgs A;
    fill_mix_args(A, d);
    A.unit = ids16_unit(d.n_rows * d.tokens_per_row);
    A.units_per_row = (d.tokens_per_row + A.unit - 1) / A.unit;
    A.n_units = d.n_rows * A.units_per_row;
    const int64_t blocks = (A.n_units + kWaves - 1) / kWaves;
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "concat16: too many units");
    if (blocks == 0) return MOT_OK;
    A.wave_lds = (int)wave_lds_bytes(A.unit, d.bpt, false, d.pull_dir != MOT_PULL_NONE ? d.ttb_elem_bytes : 0);
    const size_t lds = (size_t)A.wave_lds * kWaves;
    if (lds > 64 * 1024) return set_error(MOT_EUNSUPPORTED, "concat16: the index pass needs %zu B of LDS", lds);
#define MOT_IDS16_LAUNCH(DIR, E) hipLaunchKernelGGL((wave_ids16_kernel<DIR, E>), dim3((unsigned)blocks), dim3(kThreads), lds, stream, A, ids16)
    if (d.ttb_elem_bytes == 2) {
        if (d.pull_dir == MOT_PULL_LEFT) MOT_IDS16_LAUNCH(kPullLeft, int16_t);
        else if (d.pull_dir == MOT_PULL_RIGHT) MOT_IDS16_LAUNCH(kPullRight, int16_t);
        else MOT_IDS16_LAUNCH(kPullNone, int16_t);
    } else {
        if (d.pull_dir == MOT_PULL_LEFT) MOT_IDS16_LAUNCH(kPullLeft, int32_t);
        else if (d.pull_dir == MOT_PULL_RIGHT) MOT_IDS16_LAUNCH(kPullRight, int32_t);
        else MOT_IDS16_LAUNCH(kPullNone, int32_t);
    }
#undef MOT_IDS16_LAUNCH
    return check_launch("wave_ids16_kernel");
}

static size_t c16_lds_base(int MT, int NT, int NS, int bpt) {
    return (size_t)NS * 128 * NT * 64 + (size_t)64 * MT * 64 + (((size_t)64 * MT * bpt * 2 + 15) & ~(size_t)15);
}
struct C16Shape { int MT, NT, NS, NH; };
static bool c16_two_pass() {   // model_dim 1024 as two column passes of a 128-token tile (dev builds can switch back to 64-token tiles)
#ifdef MOT_DEV_ABLATION
    if (getenv("MOT_C16_1024_OLD")) return false;
#endif
    return true;
}
static C16Shape c16_shape(int Dm) {
    switch (Dm) {
        case 256: return {2, 2, 3, 1};
        case 512: return {2, 4, 3, 1};
        case 768: return {2, 6, 3, 1};
        default: return c16_two_pass() ? C16Shape{2, 4, 3, 2} : C16Shape{1, 8, 2, 1};   // 1024
    }
}

// the shapes this kernel takes: bf16, one id tensor, pieces of 8 elements that never straddle a part (Dt, Db multiples of 8),
// whole 32-deep steps, model_dim = 128 NT with an accumulator tile that fits (NT <= 6 at 128-token tiles, 8 at 64-token tiles),
// 16-bit byte ids, and the tile's ids and the byte rows' rms factors beside the stages in LDS
bool concat16_norm_in_kernel(const MotEmbedMixDesc &d) { return d.byte_rows <= kC16NormHere; }
bool concat16_usable(const MotEmbedMixDesc &d) {
    if (d.dtype != MOT_BF16 || d.ids_b || d.scale_tok || d.scale_byte || d.bpt < 1) return false;
    const int K = d.tok_dim + d.bpt * d.byte_dim;
    if ((d.tok_dim & 7) || (d.byte_dim & 7) || (K & 31) || d.byte_rows > 65536) return false;
    const int Dm = d.model_dim;
    if (Dm != 256 && Dm != 512 && Dm != 768 && Dm != 1024) return false;
    if (((uintptr_t)d.weight | (uintptr_t)d.tok_table | (uintptr_t)d.byte_table | (uintptr_t)d.out) & 15) return false;
    const C16Shape sh = c16_shape(Dm);
    return c16_lds_base(sh.MT, sh.NT, sh.NS, d.bpt) + (size_t)d.byte_rows * 4 + (sh.NH > 1 ? 64 * sh.MT * 4 : 0) <= 160 * 1024;
}

// rn_byte: per-row rms factors of the byte table (launch_rows_rnorm); null when the byte part is not normalised or the table is
// small enough for the workgroups to compute them (concat16_norm_in_kernel)
int launch_concat16(const MotEmbedMixDesc &d, const int32_t *tokens, const int64_t *ids, const uint16_t *ids16, int64_t n, const float *rn_byte,
                    void *out, float *row_rnorm, hipStream_t stream) {
    C16Args P;
    P.tokens = tokens; P.ids = ids; P.ids16 = ids16; P.n = n;
    P.tok_table = (const __bf16 *)d.tok_table; P.tok_rows = d.tok_rows; P.Dt = d.tok_dim;
    P.byte_table = (const __bf16 *)d.byte_table; P.byte_rows = d.byte_rows; P.Db = d.byte_dim; P.bpt = d.bpt;
    P.norm_tok = d.norm_tok; P.byte_rnorm = rn_byte; P.norm_byte_here = d.norm_byte && !rn_byte;
    P.W = (const __bf16 *)d.weight; P.bias = (const __bf16 *)d.bias;
    P.K = d.tok_dim + d.bpt * d.byte_dim; P.Dm = d.model_dim;
    P.tok_lo = d.bytes_first ? d.bpt * d.byte_dim : 0; P.byte_lo = d.bytes_first ? 0 : d.tok_dim;
    P.norm_out = d.norm_out; P.eps = d.eps > 0.f ? d.eps : kBf16Eps;
    P.out = (__bf16 *)out; P.row_rnorm = d.norm_out ? row_rnorm : nullptr; P.status = d.status;
    switch (d.model_dim) {
        case 256: return launch_c16<2, 2, 3>(P, stream);
        case 512: return launch_c16<2, 4, 3>(P, stream);
        case 768: return launch_c16<2, 6, 3>(P, stream);
        default: return c16_two_pass() ? launch_c16<2, 4, 3, 2>(P, stream) : launch_c16<1, 8, 2>(P, stream);
    }
}

}  // namespace mot
