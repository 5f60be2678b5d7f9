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

#include "mot_internal.hpp"
#include "mot_tile.hpp"

namespace mot {

typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));
typedef float f32x16g __attribute__((ext_vector_type(16)));
constexpr int kGK = 32;             // reduction indices per step
constexpr int kGRow = 2 * kGK + 16; // bytes per staged row

template <bool OUT_BF16, int WM>   // WM waves along the rows x 2 along the columns: a (64 WM) x 128 output block per workgroup
__global__ __launch_bounds__(128 * WM) __attribute__((amdgpu_waves_per_eu(3, 4))) void gemm_rows_bf16_kernel(const __bf16 *__restrict__ A_, int lda, int64_t n, const __bf16 *__restrict__ B_, int ldb,
                                                                  int R, int Nc, void *__restrict__ C_, int ldc, const __bf16 *__restrict__ bias, int accumulate) {
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
                    const float v = acc[a][b][r] + bv;
                    if constexpr (OUT_BF16) ((__bf16 *)C_)[j * ldc + k] = (__bf16)v;
                    else ((float *)C_)[j * ldc + k] = accumulate ? ((const float *)C_)[j * ldc + k] + v : v;   // (C += : fp32 results only)
                }
            }
        }
}

int launch_gemm_rows_bf16(const void *A_, int lda, int64_t n, const void *B_, int ldb, int R, int Nc, void *C, int ldc, bool out_bf16,
                          const void *bias, hipStream_t stream, bool accumulate) {
    if (n <= 0 || Nc <= 0) return MOT_OK;
    if (accumulate && out_bf16) return set_error(MOT_EUNSUPPORTED, "gemm_rows_bf16: C += needs an fp32 result");
    if ((R & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A_ & 15) || ((uintptr_t)B_ & 15))
        return set_error(MOT_EUNSUPPORTED, "gemm_rows_bf16: rows must be 16-byte aligned multiples of 8 elements (R %d, lda %d, ldb %d)", R, lda, ldb);
    const int gy = (Nc + 127) / 128;
    // (WM = 4, 256-row blocks on 8 waves, halves the reads of W per output but measured 3-7 % slower at 65 536 x 768 x 768)
    constexpr int WM = 2, TM = 64 * WM;
    const int64_t gx = (n + TM - 1) / TM;
    const int64_t blocks = (gx + 7) / 8 * 8 * gy;   // 1-D, see the block order in the kernel
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "gemm_rows_bf16: too many rows");
    if (out_bf16)
        hipLaunchKernelGGL((gemm_rows_bf16_kernel<true, WM>), dim3((unsigned)blocks), dim3(128 * WM), 0, stream, (const __bf16 *)A_, lda, n,
                           (const __bf16 *)B_, ldb, R, Nc, C, ldc, (const __bf16 *)bias, accumulate ? 1 : 0);
    else
        hipLaunchKernelGGL((gemm_rows_bf16_kernel<false, WM>), dim3((unsigned)blocks), dim3(128 * WM), 0, stream, (const __bf16 *)A_, lda, n,
                           (const __bf16 *)B_, ldb, R, Nc, C, ldc, (const __bf16 *)bias, accumulate ? 1 : 0);
    return check_launch("gemm_rows_bf16_kernel");
}

}  // namespace mot
