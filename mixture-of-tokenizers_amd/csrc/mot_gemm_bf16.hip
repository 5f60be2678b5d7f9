// mot_gemm_bf16.hip -- plain dense bf16 contraction on v_mfma_f32_32x32x16_bf16 (gfx950), fp32 accumulation:
//     C[n][c] = sum_r A[n][r] * B[c][r] (+ bias[c])        A: n x R, B: Nc x R, both row-major bf16 (r contiguous)
// That is F.linear(A, B) for bf16 operands (CastedLinear, train_gpt.py:185-186): both operands are consumed in their natural
// layout -- a lane's MFMA fragment is 8 consecutive r of one row, i.e. one 16-byte ds_read_b128.
// 128 x 128 output block per 256-thread workgroup (4 waves, each 64 x 64 = 2 x 2 tiles of 32 x 32), 32 reduction indices per
// step, double-buffered LDS with 80-byte rows (64 data + 16 pad: the 32 rows a half-wave reads tile the 32 banks four times,
// the minimum for 512 bytes), the next step's global loads in flight while the current one is multiplied.
// Used by the composed bf16 concat + linear forward (mot_linear.hip) -- the fused tile kernel of mot_linear_bf16.hip spends
// its time gathering and staging, not multiplying.
#include <type_traits>
#include <atomic>
#include <stdlib.h>

#include "mot_internal.hpp"
#include "mot_tile.hpp"

namespace mot {

typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));
typedef float f32x16g __attribute__((ext_vector_type(16)));
constexpr int kGK = 32;             // reduction indices per step
constexpr int kGRow = 2 * kGK + 16; // bytes per staged row

template <bool OUT_BF16, int WM>   // WM waves along the rows x 2 along the columns: a (64 WM) x 128 output block per workgroup
__global__ __launch_bounds__(128 * WM) __attribute__((amdgpu_waves_per_eu(3, 4))) void gemm_rows_bf16_kernel(const __bf16 *__restrict__ A_, int lda, int64_t n, const __bf16 *__restrict__ B_, int ldb,
                                                                  int R, int Nc, void *__restrict__ C_, int ldc, const __bf16 *__restrict__ bias, const float *__restrict__ addend) {
    constexpr int TM = 64 * WM, NTHR = 128 * WM, PA = TM * 4 / NTHR, PB = (128 * 4 + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) char lA[2][TM * kGRow], lB[2][128 * kGRow];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
    // XCD-aware block order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so XCD x takes the row panels
    // x, x + 8, ... and walks all column blocks of a panel back to back: the panel of A comes out of that XCD's L2 for every
    // column block after the first (with the plain 2-D order the six column blocks of a panel run far apart on different
    // XCDs and A was re-read from the fabric six times: 0.6 GB per 133 us at config 2, the bound).
    const int64_t gx = (n + TM - 1) / TM;
    const int gy = (Nc + 127) / 128;
    const int64_t id = blockIdx.x, seq = id >> 3;
    const int64_t panel = (seq / gy) * 8 + (id & 7);
    if (panel >= gx) return;   // the grid is padded to whole groups of 8 panels
    const int64_t j0 = panel * TM;
    const int k0 = (int)(seq % gy) * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16g acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // staging: 128 rows x 4 pieces of 16 bytes per operand = 512 pieces -> 2 per thread; 4 lanes read one 64-byte row segment.
    // Two register sets: the loads of step s + 2 are issued while step s is multiplied, so a load has two steps (16 MFMAs per
    // wave, x 3 waves per SIMD) to arrive -- one step ahead left the waves waiting on HBM / L2 latency.
    bf16x8g ra[2][PA], rb[2][PB];
    const int nsteps = (R + kGK - 1) / kGK;
    auto load_stage = [&](int st, auto setc) {
        constexpr int SET = decltype(setc)::value;
        const int r = st * kGK;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int q = p * NTHR + tid, row = q >> 2, k = r + (q & 3) * 8;
            ra[SET][p] = (bf16x8g)((__bf16)0.f);
            if (st < nsteps && k < R && j0 + row < n) ra[SET][p] = *(const bf16x8g *)(A_ + (j0 + row) * lda + k);   // R % 8 == 0: whole pieces
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int q = p * NTHR + tid, row = q >> 2, k = r + (q & 3) * 8;
            rb[SET][p] = (bf16x8g)((__bf16)0.f);
            if (q < 512 && st < nsteps && k < R && k0 + row < Nc) rb[SET][p] = *(const bf16x8g *)(B_ + (int64_t)(k0 + row) * ldb + k);
        }
    };
    auto store_stage = [&](int buf, auto setc) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int q = p * NTHR + tid, row = q >> 2, piece = q & 3;
            *(bf16x8g *)(lA[buf] + row * kGRow + piece * 16) = ra[SET][p];
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int q = p * NTHR + tid, row = q >> 2, piece = q & 3;
            if (q < 512) *(bf16x8g *)(lB[buf] + row * kGRow + piece * 16) = rb[SET][p];
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto step = [&](int st, auto setc, auto setn) {   // setc: free set, takes step st + 2; setn: holds step st + 1
        load_stage(st + 2, setc);
        const int buf = st & 1;
#pragma unroll
        for (int kk = 0; kk < kGK / 16; ++kk) {
            const bf16x8g a0 = *(const bf16x8g *)(lA[buf] + (wm + li) * kGRow + 32 * kk + 16 * h);
            const bf16x8g a1 = *(const bf16x8g *)(lA[buf] + (wm + 32 + li) * kGRow + 32 * kk + 16 * h);
            const bf16x8g b0 = *(const bf16x8g *)(lB[buf] + (wn + li) * kGRow + 32 * kk + 16 * h);
            const bf16x8g b1 = *(const bf16x8g *)(lB[buf] + (wn + 32 + li) * kGRow + 32 * kk + 16 * h);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (st + 1 < nsteps) store_stage(buf ^ 1, setn);
        __syncthreads();
    };
    load_stage(0, S0{});
    store_stage(0, S0{});
    load_stage(1, S1{});
    __syncthreads();
    int st = 0;
    for (; st + 2 <= nsteps; st += 2) {
        step(st, S0{}, S1{});
        step(st + 1, S1{}, S0{});
    }
    if (st < nsteps) step(st, S0{}, S1{});
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); A is the row operand
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int k = k0 + wn + b * 32 + li;
            const float bv = (bias && k < Nc) ? (float)bias[k] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t j = j0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (j < n && k < Nc) {
                    float v = acc[a][b][r] + bv;
                    if (addend) v += addend[j * ldc + k];   // (C += : addend = C; a bf16 result on top of an fp32 residual: another buffer, same rows)
                    if constexpr (OUT_BF16) ((__bf16 *)C_)[j * ldc + k] = (__bf16)v;
                    else ((float *)C_)[j * ldc + k] = v;
                }
            }
        }
}

// ----------------------------------------------------------------------------------------------------------------------------------
// Round 3: 256 x 256 output blocks, both operands by LDS-DMA, one barrier per step.  The 128 x 128 kernel above stages through
// registers, re-reads every fragment per 2 MFMAs and reaches 23-25 % of the bf16 MFMA peak; since round 3 it also carries the
// cross-attention mixin's and the character mixer's products.  Here
//   * 8 waves as 2 x 4, a wave 128 x 64 = 4 x 2 tiles of 32 x 32 (128 accumulator registers): a fragment of A feeds 2 MFMAs, one of
//     B feeds 4 -- 12 fragment reads per 16 MFMAs and step;
//   * both operands arrive by global_load_lds (16 bytes per lane, no staging registers) into FOUR stages of 32 reduction indices
//     (2 x 16 KB each), XOR-swizzled on the source side exactly as the W stages of mot_concat16.hip (piece p of stage row q at
//     q * 64 + ((p ^ (q >> 2)) & 3) * 16: conflict-free ds_read_b128 fragments), requested THREE steps ahead;
//   * the barrier in front of step s guarantees stage s + 1 as well, so the fragments of a step's second k-block are read during
//     its first and those of the next step's first k-block during its second: when the barrier opens, the operands of the next
//     eight MFMAs are in registers, and every LDS wait sits behind eight MFMAs.  One barrier per step; nothing serial behind it.
// LDS reads and waits are inline asm for the reason given in mot_concat16.hip (hipcc drains the DMA in front of its own LDS reads).
// Shapes: Nc a multiple of 256, R a multiple of 32, rows 16-byte aligned; the launcher falls back to the kernel above otherwise.
#pragma clang diagnostic ignored "-Wunused-lambda-capture"
#define G256_FRAG(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
typedef int i32x4g __attribute__((ext_vector_type(4)));
constexpr int kG2Threads = 512, kG2NS = 4, kG2PD = 3;
template <int I, int N, class F>
__device__ __forceinline__ void static_for_g(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_g<I + 1, N>(f);
    }
}
// E = float: the same kernel on v_mfma_f32_32x32x2f32 -- a stage row is again 64 bytes, now 16 reduction indices, and a lane's
// 16-byte fragment holds the k of FOUR MFMAs (lanes 0-31 take piece 2 kk, lanes 32-63 piece 2 kk + 1, so the MFMA that uses
// register e of the fragment contracts k = 8 kk + e and 8 kk + 4 + e: any pairing serves as long as both operands use it).
// A 64-cycle MFMA leaves the LDS pipe idle whatever the tile, so the fp32 form takes 64 x 64 per wave (8 waves as 4 x 2: 256 x 128
// per workgroup) and spends the registers on a second accumulator set: `acc` collects 8 steps (128 products), then is folded into
// `sum` with vector adds -- blocked summation like the 128 x 128 kernel of mot_backward.hip and the reference's BLAS, which the
// concat parity bar (twice the reference's own fp32 error) needs at K = 768.
template <typename E, bool OUT_BF16, int MT, int NT, int WC>
__global__ __launch_bounds__(kG2Threads) void gemm_rows_256_kernel(const E *__restrict__ A_, int lda, int64_t n, const E *__restrict__ B_, int ldb,
                                                                   int R, int Nc, void *__restrict__ C_, int ldc, const E *__restrict__ bias, const float *__restrict__ addend) {
    constexpr int EPR = 64 / (int)sizeof(E), EPP = 16 / (int)sizeof(E);   // elements per stage row / per 16-byte piece
    constexpr int BM = (8 / WC) * 32 * MT, BN = WC * 32 * NT, kStage = (BM + BN) * 64, NA = BM / 128, NB = BN / 128, ND = NA + NB;
    constexpr bool kFoldSums = std::is_same_v<E, float>;
    extern __shared__ __attribute__((aligned(16))) char lds_g2[];   // [4 stages][A: BM rows x 64 B | B: BN rows x 64 B]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WC) * 32 * MT, wn = (wave % WC) * 32 * NT;
    const int64_t gx = (n + BM - 1) / BM;
    const int gy = Nc / BN;
    const int64_t id = blockIdx.x, seq = id >> 3;
    const int64_t panel = (seq / gy) * 8 + (id & 7);   // (the XCD-aware order of the kernel above)
    if (panel >= gx) return;
    const int64_t j0 = panel * BM;
    const int k0 = (int)(seq % gy) * BN;
    const uint32_t oS = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)lds_g2;
    // DMA: wave-instruction j writes stage rows 16 j .. 16 j + 15 (1 KiB); this wave: j = NA wave + i of A, NB wave + i of B
    uint32_t goffA[NA], goffB[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = (NA * wave + i) * 16 + (lane >> 2), pc = ((lane & 3) ^ (q >> 2)) & 3;
        const int64_t ra = min(j0 + q, n - 1) - j0;   // rows past the end repeat the last one (computed, never stored)
        goffA[i] = (uint32_t)((ra * lda + EPP * pc) * sizeof(E));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int q = (NB * wave + i) * 16 + (lane >> 2), pc = ((lane & 3) ^ (q >> 2)) & 3;
        goffB[i] = (uint32_t)(((int64_t)q * ldb + EPP * pc) * sizeof(E));
    }
    const char *Ab = (const char *)(A_ + j0 * lda), *Bb = (const char *)(B_ + (int64_t)k0 * ldb);
    const int nsteps = R / EPR;
    auto dma_all = [&](int s) {   // (a step past the end re-reads the last one into a stage nobody reads)
        char *st = lds_g2 + (s % kG2NS) * kStage;
        const uint32_t ko = 64u * (uint32_t)min(s, nsteps - 1);
        static_for_g<0, NA>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Ab + (goffA[i] + ko)),
                                             (__attribute__((address_space(3))) void *)(st + (NA * wave + i) * 1024), 16, 0, 0);
        });
        static_for_g<0, NB>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Bb + (goffB[i] + ko)),
                                             (__attribute__((address_space(3))) void *)(st + BM * 64 + (NB * wave + i) * 1024), 16, 0, 0);
        });
    };
    dma_all(0); dma_all(1); dma_all(2);
    f32x16g acc[MT][NT], sum[kFoldSums ? MT : 1][kFoldSums ? NT : 1];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[a][b][r] = 0.f;
                if constexpr (kFoldSums) sum[a][b][r] = 0.f;
            }
    // fragment addresses inside a stage: row = (wm | wn) + 32 t + li, piece (2 kk + h) ^ (row >> 2): t adds t * 2048, kk flips bit 5
    const uint32_t fa0 = oS + (wm + li) * 64 + ((h ^ (li >> 2)) & 3) * 16;
    const uint32_t fb0 = oS + BM * 64 + (wn + li) * 64 + ((h ^ (li >> 2)) & 3) * 16;
    i32x4g fA[2][MT], fB[2][NT];   // [k-block][tile]
    auto frags = [&, &fA = fA, &fB = fB](auto kkc, int s) {   // (explicit captures: clang wants them for asm operands in generic lambdas)
        constexpr int kk = decltype(kkc)::value;
        const uint32_t so = (uint32_t)(s % kG2NS) * kStage;
        const uint32_t pa = (fa0 + so) ^ (kk * 32), pb = (fb0 + so) ^ (kk * 32);
        static_for_g<0, NT>([&fB = fB, pb](auto tc) { G256_FRAG(fB[kk][decltype(tc)::value], pb, decltype(tc)::value * 2048); });
        static_for_g<0, MT>([&fA = fA, pa](auto tc) { G256_FRAG(fA[kk][decltype(tc)::value], pa, decltype(tc)::value * 2048); });
    };
    auto landed = [&, &fA = fA, &fB = fB](auto kkc) {   // the fragments of a k-block have landed (the wait is in front): tie them to it
        constexpr int kk = decltype(kkc)::value;
        static_for_g<0, NT>([&fB = fB](auto tc) { asm volatile("" : "+v"(fB[kk][decltype(tc)::value])); });
        static_for_g<0, MT>([&fA = fA](auto tc) { asm volatile("" : "+v"(fA[kk][decltype(tc)::value])); });
    };
    auto mfmas = [&, &fA = fA, &fB = fB](auto kkc) {
        constexpr int kk = decltype(kkc)::value;
        if constexpr (std::is_same_v<E, float>) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) {
                        // (element copies first: __builtin_bit_cast applied to the element expression itself compiled to element 0
                        //  for every e with this hipcc -- found from the ISA, four identical MFMAs per tile)
                        const int ai = fA[kk][a][e], bi = fB[kk][b][e];
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, ai), __builtin_bit_cast(float, bi), acc[a][b], 0, 0, 0);
                    }
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8g, fA[kk][a]), __builtin_bit_cast(bf16x8g, fB[kk][b]), acc[a][b], 0, 0, 0);
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    // stage 0 for everyone, then the first k-block of step 0 into registers
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ND) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    frags(K0{}, 0);
    for (int s = 0; s < nsteps; ++s) {
        // this wave's share of stage s + 1 has landed (younger: stage s + 2); the fragments requested in the previous step too
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ND) : "memory");
        landed(K0{});
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        frags(K1{}, s);              // second k-block of this step: stage s is complete since the previous barrier
        dma_all(s + kG2PD);          // into the stage read in step s - 1: everyone is past it
        mfmas(K0{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        landed(K1{});
        frags(K0{}, s + 1);          // first k-block of the next step: stage s + 1 is complete since this step's barrier
        mfmas(K1{});
        if constexpr (kFoldSums) {
            if ((s & 7) == 7) {
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) {
                        sum[a][b] += acc[a][b];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
                    }
            }
        }
    }
    // (what was requested past the last step is still on its way: hold the registers and the stages until it has landed)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    landed(K0{});
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); A is the row operand
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int k = k0 + wn + b * 32 + li;
            const float bv = bias ? (float)bias[k] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t j = j0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (j < n) {
                    float v = acc[a][b][r] + bv;
                    if constexpr (kFoldSums) v = (sum[a][b][r] + acc[a][b][r]) + bv;
                    if (addend) v += addend[j * ldc + k];
                    if constexpr (OUT_BF16) ((__bf16 *)C_)[j * ldc + k] = (__bf16)v;
                    else ((float *)C_)[j * ldc + k] = v;
                }
            }
        }
}

// fp32 operands: C[n][c] (+)= sum_r A[n][r] * B[c][r] (+ bias[c]) on the LDS-DMA kernel (256 x 128 blocks); false when the shape is
// not its own (Nc % 128, R % 16, rows not 16-byte aligned, fewer than 512 rows) and the caller keeps its 128 x 128 kernel
bool gemm_rows_f32_256_usable(const float *A_, int lda, int64_t n, const float *B_, int ldb, int R, int Nc) {
    bool ok = (Nc % 128) == 0 && (R % 16) == 0 && n >= 512 && !(lda & 3) && !(ldb & 3) && !(((uintptr_t)A_ | (uintptr_t)B_) & 15) &&
              (int64_t)256 * (lda > ldb ? lda : ldb) * 4 < 0x7fffffffLL;
#ifdef MOT_DEV_ABLATION
    if (getenv("MOT_GEMM32_OLD")) ok = false;
#endif
    return ok;
}
int launch_gemm_rows_f32_256(const float *A_, int lda, int64_t n, const float *B_, int ldb, int R, int Nc, float *C, int ldc, const float *bias,
                             bool accumulate, hipStream_t stream) {
    const int64_t gx2 = (n + 255) / 256, blocks2 = (gx2 + 7) / 8 * 8 * (Nc / 128);
    if (blocks2 > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "gemm_rows: too many rows");
    static std::atomic<uint64_t> ok{0};
    if (int rc = ensure_max_dyn_lds((const void *)gemm_rows_256_kernel<float, false, 2, 2, 2>, ok, "gemm_rows_256_kernel")) return rc;
    hipLaunchKernelGGL((gemm_rows_256_kernel<float, false, 2, 2, 2>), dim3((unsigned)blocks2), dim3(kG2Threads), (size_t)kG2NS * (256 + 128) * 64, stream, A_, lda, n,
                       B_, ldb, R, Nc, (void *)C, ldc, bias, accumulate ? (const float *)C : (const float *)nullptr);
    return check_launch("gemm_rows_256_kernel");
}

int launch_gemm_rows_bf16(const void *A_, int lda, int64_t n, const void *B_, int ldb, int R, int Nc, void *C, int ldc, bool out_bf16,
                          const void *bias, hipStream_t stream, bool accumulate, const float *addend) {
    if (n <= 0 || Nc <= 0) return MOT_OK;
    if (accumulate && out_bf16) return set_error(MOT_EUNSUPPORTED, "gemm_rows_bf16: C += needs an fp32 result");
    if (accumulate) addend = (const float *)C;
    if ((R & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A_ & 15) || ((uintptr_t)B_ & 15))
        return set_error(MOT_EUNSUPPORTED, "gemm_rows_bf16: rows must be 16-byte aligned multiples of 8 elements (R %d, lda %d, ldb %d)", R, lda, ldb);
    bool big = (Nc % 256) == 0 && (R % 32) == 0 && n >= 512 && (int64_t)256 * (lda > ldb ? lda : ldb) * 2 < 0x7fffffffLL;
#ifdef MOT_DEV_ABLATION
    if (getenv("MOT_GEMM16_OLD")) big = false;
#endif
    if (big) {   // 256 x 256 blocks by LDS-DMA
        const int64_t gx2 = (n + 255) / 256, blocks2 = (gx2 + 7) / 8 * 8 * (Nc / 256);
        if (blocks2 > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "gemm_rows_bf16: too many rows");
        const size_t lds = (size_t)kG2NS * (256 + 256) * 64;
        static std::atomic<uint64_t> ok0{0}, ok1{0};
        if (out_bf16) {
            if (int rc = ensure_max_dyn_lds((const void *)gemm_rows_256_kernel<__bf16, true, 4, 2, 4>, ok1, "gemm_rows_256_kernel")) return rc;
            hipLaunchKernelGGL((gemm_rows_256_kernel<__bf16, true, 4, 2, 4>), dim3((unsigned)blocks2), dim3(kG2Threads), lds, stream, (const __bf16 *)A_, lda, n,
                               (const __bf16 *)B_, ldb, R, Nc, C, ldc, (const __bf16 *)bias, addend);
        } else {
            if (int rc = ensure_max_dyn_lds((const void *)gemm_rows_256_kernel<__bf16, false, 4, 2, 4>, ok0, "gemm_rows_256_kernel")) return rc;
            hipLaunchKernelGGL((gemm_rows_256_kernel<__bf16, false, 4, 2, 4>), dim3((unsigned)blocks2), dim3(kG2Threads), lds, stream, (const __bf16 *)A_, lda, n,
                               (const __bf16 *)B_, ldb, R, Nc, C, ldc, (const __bf16 *)bias, addend);
        }
        return check_launch("gemm_rows_256_kernel");
    }
    const int gy = (Nc + 127) / 128;
    // (WM = 4, 256-row blocks on 8 waves, halves the reads of W per output but measured 3-7 % slower at 65 536 x 768 x 768)
    constexpr int WM = 2, TM = 64 * WM;
    const int64_t gx = (n + TM - 1) / TM;
    const int64_t blocks = (gx + 7) / 8 * 8 * gy;   // 1-D, see the block order in the kernel
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "gemm_rows_bf16: too many rows");
    if (out_bf16)
        hipLaunchKernelGGL((gemm_rows_bf16_kernel<true, WM>), dim3((unsigned)blocks), dim3(128 * WM), 0, stream, (const __bf16 *)A_, lda, n,
                           (const __bf16 *)B_, ldb, R, Nc, C, ldc, (const __bf16 *)bias, addend);
    else
        hipLaunchKernelGGL((gemm_rows_bf16_kernel<false, WM>), dim3((unsigned)blocks), dim3(128 * WM), 0, stream, (const __bf16 *)A_, lda, n,
                           (const __bf16 *)B_, ldb, R, Nc, C, ldc, (const __bf16 *)bias, addend);
    return check_launch("gemm_rows_bf16_kernel");
}

}  // namespace mot
