// mot_linear.hip -- CONCAT_LINEAR mode of the fused front-end (gfx950):
//     x = rms_norm?( W . cat(a, b_0 .. b_{bpt-1}) + bias )
// i.e. ByteMixinConcat after FlexibleEmbedding (scaled-pre-train/train_gpt.py:327-379, 430-443:
// tokens first, no bias, norms everywhere) and DigitMixinConcat after wte/dte
// (mathblations/model.py:256-268, 323-327: digits first, bias, no norms).
//
// Two implementations live here.  The DEFAULT is composed from plain kernels (see "composed path" below: index kernels, seam
// gather writing the concat operand, dense MFMA kernel, row norm; fp32 and bf16) and runs 20-30 % faster; the one-launch fused
// tile kernel described next was the first path and is kept for learned embedding scalars and behind MOT_LIN_FUSED=1.
//
// This mode is a dense contraction over K = Dt + bpt*Db per token (~285 FLOP/B at
// K = Dm = 768), so it is MFMA-bound, not HBM-bound, and uses the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain: no precision is traded away).  What makes it a
// *fused* kernel is the A operand: the (tokens x K) concat matrix is never materialised -- each
// K-step's A tile is gathered straight from the embedding tables (rows chosen by the token ids and
// by the byte ids that phase 1 left in LDS), scaled by the per-segment rms factor and written to
// LDS k-major, where it is consumed as MFMA fragments.
//
// Geometry: one 256-thread workgroup (4 waves, one per SIMD, so each wave may use the whole
// 512-entry register file) per tile of 64 tokens x all Dm output columns -- the post-norm needs
// complete rows.  Wave w owns columns [w*NT*32, (w+1)*NT*32): 2 x NT accumulator tiles of 32x32.
// K is walked in steps of 16 with double-buffered LDS: the next step's W chunk (from the k-major
// transposed copy the prologue kernel writes to the workspace) and A chunk are loaded to registers
// before the current step's MFMAs and written to the other buffer after them; one barrier per step.
#include <stdlib.h>

#include "mot_mix.hpp"

namespace mot {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBKmax = 16;  // K-step is 16 (8 for the widest accumulator tile, to stay inside 512 registers)

struct LinArgs {
    MixArgs M;
    const float *Wt;    // [Kpad, DmPad] k-major copy of the weight, zero padded
    const float *bias;  // [Dm] or null
    float *row_rnorm;   // optional [n_rows*T]: the post-norm factor of every row (for the backward)
    const float *tok_rnorm;  // optional [tok_rows]: 1/rms of every token-table row (norm_tok), computed once per call
    int K, Kpad, Dm, DmPad, bytes_first, dual;
};

struct LinLds {
    float *W0, *A0, *scale, *rowss;  // W0/A0: two consecutive buffers each
    int32_t *tokc;
    TileLds tile;
};

__host__ __device__ inline size_t lin_lds_floats_before_tile(int DmPad, int bpt, int bk, int tm) {
    return 2 * (size_t)bk * DmPad + 2 * (size_t)bk * tm + (size_t)tm * (1 + bpt) + 4 * tm + tm;
}
__host__ __device__ inline size_t lin_lds_bytes(int DmPad, int bpt, int bk, int tm) {
    size_t f = lin_lds_floats_before_tile(DmPad, bpt, bk, tm);
    f = (f + 3) & ~(size_t)3;
    return f * 4 + tile_lds_bytes(tm, bpt, true);
}

// W [Dm, K] (nn.Linear layout) -> Wt [Kpad, DmPad], zero padded; 32x32 LDS-tiled transpose.
__global__ __launch_bounds__(kThreads) void transpose_pad_kernel(const float *__restrict__ W, int Dm, int K,
                                                                 float *__restrict__ Wt, int Kpad, int DmPad) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int k0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, k = k0 + tx;
        tile[r][tx] = (j < Dm && k < K) ? W[(int64_t)j * K + k] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, j = j0 + tx;
        if (k < Kpad && j < DmPad) Wt[(int64_t)k * DmPad + j] = tile[tx][r];
    }
}

// MT x NT accumulator tiles of 32x32 per wave: the workgroup tile is kTM = 32*MT tokens x 128*NT columns.
// ABL: timing-only ablation bits (results are wrong when set): 1 = chain straight into the running
// accumulator (no blocked-sum adds), 2 = no per-step global loads, 4 = no per-step LDS stores, 8 = no barrier.
template <int MT, int NT, int kBK, int OCC, int ABL = 0>
__global__ __launch_bounds__(kThreads, OCC) void embed_mix_linear_kernel(const LinArgs P) {
    constexpr int kTM = 32 * MT;
    constexpr int WP = kBK * NT / 8;  // float4 of the W chunk per thread
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    const MixArgs &A = P.M;
    const int bpt = A.bpt, sv = bpt | 1, SS = 1 + bpt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    LinLds S;
    {
        float *f = (float *)lds;
        S.W0 = f; f += 2 * kBK * P.DmPad;
        S.A0 = f; f += 2 * kBK * kTM;
        S.scale = f; f += kTM * SS;
        S.rowss = f; f += 4 * kTM;
        S.tokc = (int32_t *)f; f += kTM;
        size_t used = (size_t)(f - (float *)lds);
        used = (used + 3) & ~(size_t)3;
        S.tile = tile_lds_carve(lds + used, kTM, bpt, true);
    }
    const TileLds &L = S.tile;
    const int64_t row = blockIdx.x / A.tiles_per_row;
    const int64_t t0 = (int64_t)(blockIdx.x % A.tiles_per_row) * kTM;
    const int ntok = (int)min((int64_t)kTM, A.T - t0);

    // ---- phase 1: byte ids of the tile
    if (A.id_source == MOT_IDS_FROM_TTB) {
        if (A.pull_dir == kPullLeft) phase1_from_ttb<kPullLeft>(A, L, row, t0, ntok);
        else if (A.pull_dir == kPullRight) phase1_from_ttb<kPullRight>(A, L, row, t0, ntok);
        else phase1_from_ttb<kPullNone>(A, L, row, t0, ntok);
    } else {
        phase1_given(A, L, row, t0, ntok);
    }

    // ---- per-segment rms factors (rms_norm of a segment = raw * r: folded into the A operand)
    for (int i = tid; i < kTM * SS; i += kThreads) S.scale[i] = 1.0f;
    if (tid < kTM) {
        int tok = tid < ntok ? L.tok[tid] : 0;
        if ((uint64_t)(uint32_t)tok >= (uint64_t)A.tok_rows) {
            if (A.status) atomicOr(A.status, kStatusTokenOor);
            tok = 0;
        }
        S.tokc[tid] = tok;
    }
    __syncthreads();
    if (A.norm_tok && P.tok_rnorm) {   // table computed once per call: no dependent row fetch per token per tile
        for (int t = tid; t < ntok; t += kThreads) S.scale[t * SS] = P.tok_rnorm[S.tokc[t]];
    } else if (A.norm_tok) {
        for (int t = wave; t < ntok; t += kWaves) {
            const float *trow = A.tok_table + (int64_t)S.tokc[t] * A.Dt;
            float ss = 0.f;
            for (int c = lane; c < (A.Dt >> 2); c += 64) {
                const float4v v = *(const float4v *)(trow + 4 * c);
                ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
            ss = wave_sum(ss);
            if (lane == 0) S.scale[t * SS] = rms_scale(ss, A.Dt, A.eps);
        }
    }
    if (A.norm_byte) {
        for (int p = tid; p < ntok * bpt; p += kThreads) {
            const int t = p / bpt, k = p - t * bpt;
            const int id = L.ids[t * sv + k];
            float r;
            if (!P.dual) {
                r = A.byte_rnorm[id];
            } else {  // norm(emb(padded) + emb(pulled)), train_gpt.py:378
                const int id2 = L.val[t * sv + k];
                const float *pa = A.byte_table + (int64_t)id * A.Db, *pb = A.byte_table + (int64_t)id2 * A.Db;
                float ss = 0.f;
                for (int j = 0; j < A.Db; j += 4) {
                    const float4v v = *(const float4v *)(pa + j) + *(const float4v *)(pb + j);
                    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                }
                r = rms_scale(ss, A.Db, A.eps);
            }
            S.scale[t * SS + 1 + k] = r;
        }
    }
    __syncthreads();

    // ---- K loop
    const float s_tok = A.scale_tok ? *A.scale_tok : 1.0f;
    const float s_byte = A.scale_byte ? *A.scale_byte : 1.0f;
    const bool scale_t = A.scale_tok != nullptr, scale_b = A.scale_byte != nullptr;
    const int am = tid % kTM, akq = tid / kTM;  // A staging role: token am, float4 (akq mod kBK/4) of the K-step
    const int nbytes_k = bpt * A.Db;
    float4v wreg[WP], areg;
    float afac;   // factor of the staged A piece, applied when it is written to LDS: scaling at load time would wait for the load
    const float inv_db = 1.0f / (float)A.Db;

    // Rows past the tile's end read token/byte id 0 (valid memory; never stored); K-padding columns
    // meet zero rows of Wt.  Branch-free on purpose: the K loop body must stay one basic block so
    // that the scheduler can overlap the partial-sum adds with the next tile's MFMAs.
    const int amr = min(am, ntok - 1);  // row used for every lookup: always a real token of the tile
    const int tok_m = S.tokc[amr];
    auto load_stage = [&](int s) {
        const int k0 = s * kBK;
        const float4v *src = (const float4v *)(P.Wt + (int64_t)k0 * P.DmPad);
#pragma unroll
        for (int p = 0; p < WP; ++p) wreg[p] = src[p * kThreads + tid];
        const int k = min(k0 + 4 * (akq & (kBK / 4 - 1)), P.K - 4);
        const bool is_tok = P.bytes_first ? k >= nbytes_k : k < A.Dt;
        const int kb = P.bytes_first ? k : k - A.Dt;             // offset inside the byte part
        const int slot = is_tok ? 0 : __float2int_rd(((float)kb + 0.5f) * inv_db);   // kb / Db without the integer division
        const int off = is_tok ? (P.bytes_first ? k - nbytes_k : k) : kb - slot * A.Db;
        const int id1 = L.ids[amr * sv + slot];
        const float *p1 = is_tok ? A.tok_table + (int64_t)tok_m * A.Dt + off : A.byte_table + (int64_t)id1 * A.Db + off;
        float4v v = *(const float4v *)p1;
        if (P.dual) {
            const float4v v2 = *(const float4v *)(A.byte_table + (int64_t)L.val[amr * sv + slot] * A.Db + off);
            v += is_tok ? (float4v)(0.f) : v2;
        }
        float f = S.scale[amr * SS + (is_tok ? 0 : 1 + slot)];
        const float sc = is_tok ? s_tok : s_byte;
        if (is_tok ? scale_t : scale_b) f *= sc;
        areg = v;
        afac = f;
    };
    auto store_stage = [&](int buf) {
        float4v *dst = (float4v *)(S.W0 + buf * (kBK * P.DmPad));
#pragma unroll
        for (int p = 0; p < WP; ++p) dst[p * kThreads + tid] = wreg[p];
        // k-major A[k][m]; with kBK == 8 waves 2-3 duplicate the writes of waves 0-1 (same values)
        float *a = S.A0 + buf * (kBK * kTM) + (4 * (akq & (kBK / 4 - 1))) * kTM + am;
        a[0] = areg.x * afac; a[kTM] = areg.y * afac; a[2 * kTM] = areg.z * afac; a[3 * kTM] = areg.w * afac;
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int h = lane >> 5, li = lane & 31;
    const int n0 = wave * (NT * 32);
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    const int nsteps = P.Kpad / kBK;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        if (!(ABL & 2)) load_stage(min(s + 1, nsteps - 1));  // the last step re-loads itself: keeps the body branch-free
        const float *Ab = S.A0 + (s & 1) * (kBK * kTM), *Wb = S.W0 + (s & 1) * (kBK * P.DmPad);
        // v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].
        // Blocked summation: the kBK products of this step are chained in a fresh accumulator and
        // that partial sum is added to the running one, so the long chain has K/kBK terms instead
        // of K (a single K-long fp32 chain is ~2x less accurate than the reference's blocked sgemm).
        float af[MT][kBK / 2];
#pragma unroll
        for (int kp = 0; kp < kBK / 2; ++kp)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt][kp] = Ab[(2 * kp + h) * kTM + mt * 32 + li];
        // Software pipeline over the MT*NT accumulator tiles of this step.  While tile j's kBK/2 chained
        // MFMAs issue (64 cycles each on the SIMD's matrix pipe), the other pipes work in their shadow:
        //   - the VALU adds tile j-1's finished partial sum into the running accumulator,
        //   - LDS reads fetch the B fragments of the NEXT column block (so no MFMA waits on a read that
        //     was issued one MFMA earlier: ds_read latency is about one MFMA long),
        // and the sched_group_barrier sequence pins that interleave per MFMA.
        float bf[2][kBK / 2];
#pragma unroll
        for (int kp = 0; kp < kBK / 2; ++kp) bf[0][kp] = Wb[(2 * kp + h) * P.DmPad + n0 + li];
        f32x16 prev0 = zero16, prev1 = zero16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt + 1 < NT) {
#pragma unroll
                for (int kp = 0; kp < kBK / 2; ++kp) bf[(nt + 1) & 1][kp] = Wb[(2 * kp + h) * P.DmPad + n0 + (nt + 1) * 32 + li];
            }
            if (ABL & 1) {   // timing only: straight chains into the running accumulators, no adds
#pragma unroll
                for (int kp = 0; kp < kBK / 2; ++kp)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][kp], bf[nt & 1][kp], acc[mt][nt], 0, 0, 0);
            } else if (MT == 2) {
                // the two row tiles of this column block share the B fragments; their chains are
                // interleaved so that consecutive MFMAs never depend on each other
                f32x16 p0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][0], bf[nt & 1][0], zero16, 0, 0, 0);
                f32x16 p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[MT - 1][0], bf[nt & 1][0], zero16, 0, 0, 0);
#pragma unroll
                for (int kp = 1; kp < kBK / 2; ++kp) {
                    p0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][kp], bf[nt & 1][kp], p0, 0, 0, 0);
                    p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[MT - 1][kp], bf[nt & 1][kp], p1, 0, 0, 0);
                }
                if (nt > 0) { acc[0][nt - 1] += prev0; acc[MT - 1][nt - 1] += prev1; }
                prev0 = p0; prev1 = p1;
#pragma unroll
                for (int kp = 0; kp < kBK; ++kp) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
                    if ((kp & 1) == 0 && nt + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one B-fragment read
                    __builtin_amdgcn_sched_group_barrier(0x002, 48 / (kBK / 2), 0);        // a slice of the adds
                }
            } else {
                f32x16 part = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][0], bf[nt & 1][0], zero16, 0, 0, 0);
#pragma unroll
                for (int kp = 1; kp < kBK / 2; ++kp) part = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][kp], bf[nt & 1][kp], part, 0, 0, 0);
                if (nt > 0) acc[0][nt - 1] += prev0;
                prev0 = part;
#pragma unroll
                for (int kp = 0; kp < kBK / 2; ++kp) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (nt + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 48 / (kBK / 2), 0);
                }
            }
        }
        acc[0][NT - 1] += prev0;
        if (MT == 2) acc[MT - 1][NT - 1] += prev1;
        if (!(ABL & 4)) store_stage((s + 1) & 1);
        if (!(ABL & 8)) __syncthreads();
    }

    // ---- epilogue.  C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (P.bias) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + nt * 32 + li;
            const float bv = col < P.Dm ? P.bias[col] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] += bv;
        }
    }
    if (A.norm_out) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ss = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) ss += acc[mt][nt][r] * acc[mt][nt][r];  // padded columns hold 0
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);  // stays inside the 32-lane half
                if (li == 0) S.rowss[wave * kTM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = ss;
            }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float tot = ((S.rowss[m] + S.rowss[kTM + m]) + S.rowss[2 * kTM + m]) + S.rowss[3 * kTM + m];
                const float rs = rms_scale(tot, P.Dm, A.eps);
                if (P.row_rnorm && wave == 0 && li == 0 && m < ntok) P.row_rnorm[row * A.T + t0 + m] = rs;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] *= rs;
            }
    }
    float *orow = A.out + (row * A.T + t0) * (int64_t)P.Dm;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < ntok) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = n0 + nt * 32 + li;
                    if (col < P.Dm) __builtin_nontemporal_store(acc[mt][nt][r], orow + (int64_t)m * P.Dm + col);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------ launchers
static int nt_of(int Dm) {
    const int nt = (Dm + 127) / 128;
    if (nt <= 4) return nt;
    if (nt <= 6) return 6;
    if (nt <= 8) return 8;
    return -1;
}

// ------------------------------------------------------------------------------------------ composed path (ids given)
// With the byte ids already in HBM (the module seam: FlexibleEmbedding hands them over as the reference's loader made
// them) the forward is composed from three plain kernels: the seam gather writes the normalised concat operand u
// [tokens, K] (token part and byte part placed side by side), gemm_rows_kernel contracts it with W at ~68 % of the fp32
// MFMA peak (the fused tile kernel below reaches 48 %), and a row pass applies the output norm.  u costs two extra passes
// over tokens x K floats, which the faster contraction more than pays for; it is built in slabs of kSlabRows rows so the
// scratch stays bounded.  The fused kernel remains the path for ids pulled in-kernel from the token->byte table.
constexpr int64_t kSlabRows = 65536;
bool embed_mix_linear_is_composed(const MotEmbedMixDesc &d) {
    if (d.flags & MOT_FLAG_LINEAR_ONE_LAUNCH) return false;   // A-B / test switch: the one-launch tile kernels for everything
    return d.bpt > 0 && !d.scale_tok && !d.scale_byte;
}
static bool composed_path(const MotEmbedMixDesc &d) { return embed_mix_linear_is_composed(d); }
// u in units of 4 bytes (fp32: one element; bf16: two), rounded up to a multiple of 4 so that what follows stays 16-byte aligned
static size_t composed_floats(const MotEmbedMixDesc &d) {
    const int64_t n = d.n_rows * d.tokens_per_row;
    const size_t elems = (size_t)(n < kSlabRows ? n : kSlabRows) * (size_t)(d.tok_dim + d.bpt * d.byte_dim);
    return ((d.dtype == MOT_BF16 ? (elems + 1) / 2 : elems) + 3) & ~(size_t)3;
}
static size_t composed_rnorm_floats(const MotEmbedMixDesc &d) { return ((size_t)d.byte_rows + 3) & ~(size_t)3; }   // rms factors of the byte-table rows
// ids pulled from the token->byte table: the two index kernels of the loader path run first, into the caller's out_ids_*
// buffers when it asked for them, else into scratch behind u (2 x tokens x bpt int64)
static size_t composed_id_words(const MotEmbedMixDesc &d) {
    return d.id_source == MOT_IDS_FROM_TTB ? 2 * (size_t)(d.n_rows * d.tokens_per_row) * (size_t)d.bpt : 0;
}

// x[r] *= rsqrt(mean(x[r]^2) + eps) in place, one wave per row; the factor is kept for the backward.  For bf16 the row IS
// the bf16 tensor the reference's CastedLinear returns (train_gpt.py:185-186); norm() upcasts it (172-173), one rounding on store.
template <typename T>
__global__ __launch_bounds__(kThreads) void rows_rms_inplace_kernel(T *__restrict__ x, int64_t n, int dim, float eps, float *__restrict__ row_rnorm) {
    using vec_t = typename Elem<T>::vec;            // 16 bytes: 4 floats / 8 bf16, widened to floats
    constexpr int VEC = Elem<T>::kVec, kKeep = 4;    // a lane keeps up to kKeep vectors of the row between the two passes
    auto sumsq = [](const vec_t &v) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) t += v[e] * v[e];
        return t;
    };
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= n) return;
    T *p = x + r * dim;
    if ((dim % VEC) == 0 && (((uintptr_t)x | ((size_t)dim * sizeof(T))) & 15) == 0) {
        const int nv = dim / VEC;
        vec_t keep[kKeep];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < kKeep; ++i) {
            const int j = lane + 64 * i;
            keep[i] = j < nv ? Elem<T>::loadv(p + VEC * j) : (vec_t)(0.f);
            ss += sumsq(keep[i]);
        }
        for (int j = lane + 64 * kKeep; j < nv; j += 64) ss += sumsq(Elem<T>::loadv(p + VEC * j));
        const float rs = rms_scale(wave_sum(ss), dim, eps);
#pragma unroll
        for (int i = 0; i < kKeep; ++i) {
            const int j = lane + 64 * i;
            if (j < nv) Elem<T>::storev_nt(p + VEC * j, keep[i] * rs);
        }
        for (int j = lane + 64 * kKeep; j < nv; j += 64) Elem<T>::storev_nt(p + VEC * j, Elem<T>::loadv(p + VEC * j) * rs);
        if (row_rnorm && lane == 0) row_rnorm[r] = rs;
        return;
    }
    float ss = 0.f;
    for (int j = lane; j < dim; j += 64) { const float v = (float)p[j]; ss += v * v; }
    const float rs = rms_scale(wave_sum(ss), dim, eps);
    for (int j = lane; j < dim; j += 64) p[j] = (T)((float)p[j] * rs);
    if (row_rnorm && lane == 0) row_rnorm[r] = rs;
}

// the statistics of runs/79_*.py:484-488 from the two id tensors: tokens, byte slots, pads before the pull, pads after
__global__ __launch_bounds__(kThreads) void count_pads_kernel(const int64_t *__restrict__ padded, const int64_t *__restrict__ after, int64_t n_slots,
                                                              int64_t pad, int64_t n_tokens, int64_t *counters) {
    int before = 0, aft = 0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * kThreads) {
        before += padded[i] == pad;
        aft += after[i] == pad;
    }
    before = (int)wave_sum((float)before);   // <= 64 * a few thousand per wave: exact in fp32
    aft = (int)wave_sum((float)aft);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long *)counters + 2, (unsigned long long)before);
        atomicAdd((unsigned long long *)counters + 3, (unsigned long long)aft);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd((unsigned long long *)counters + 0, (unsigned long long)n_tokens);
        atomicAdd((unsigned long long *)counters + 1, (unsigned long long)n_slots);
    }
}

static int launch_composed(const MotEmbedMixDesc &d, hipStream_t stream);

static int launch_composed_from_ttb(const MotEmbedMixDesc &d, hipStream_t stream) {
    const int64_t N = d.n_rows * d.tokens_per_row, slots = N * d.bpt;
    const size_t need = (composed_floats(d) + composed_rnorm_floats(d)) * sizeof(float) + composed_id_words(d) * sizeof(int64_t);
    if (!d.workspace || d.workspace_bytes < need)
        return set_error(MOT_EWORKSPACE, "embed_mix concat_linear: needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
    int64_t *ws_ids = (int64_t *)((float *)d.workspace + composed_floats(d) + composed_rnorm_floats(d));
    int rc;
    // bf16 at the shapes of mot_concat16.hip: one wave-local index pass (16-bit ids, parity outputs and statistics included) and
    // the gather-GEMM
    if (d.dtype == MOT_BF16 && !(d.flags & MOT_FLAG_LINEAR_COMPOSED) && !d.add_padded && concat16_usable(d)) {
        if (N == 0) return MOT_OK;
        uint16_t *ids16 = (uint16_t *)ws_ids;
        float *rn = (float *)d.workspace + composed_floats(d);
        if ((rc = launch_wave_ids16(d, ids16, stream))) return rc;
        const bool rn_table = d.norm_byte && !concat16_norm_in_kernel(d);
        if (rn_table && (rc = launch_rows_rnorm(d.byte_table, d.byte_rows, d.byte_dim, d.eps > 0.f ? d.eps : kBf16Eps, rn, d.dtype, stream))) return rc;
        return launch_concat16(d, d.tokens, nullptr, ids16, N, rn_table ? rn : nullptr, d.out, d.out_row_rnorm, stream);
    }
    int64_t *padded = d.out_ids_padded ? d.out_ids_padded : ws_ids;
    int64_t *pulled = d.out_ids_pulled ? d.out_ids_pulled : ws_ids + slots;
    if ((rc = launch_tokens_to_bytes(d.tokens, N, d.ttb, d.ttb_elem_bytes, d.ttb_rows, d.bpt, padded, d.status, stream))) return rc;
    const int64_t *after = padded;
    if (d.pull_dir != MOT_PULL_NONE) {
        if ((rc = launch_pull_bytes(padded, pulled, d.n_rows, d.tokens_per_row, d.bpt, d.pad_byte, d.eot_byte,
                                    d.pull_dir == MOT_PULL_LEFT ? kPullLeft : kPullRight, stream))) return rc;
        after = pulled;
    } else if (d.out_ids_pulled) {   // nothing is pulled: that output is the padded tensor again
        if ((rc = launch_tokens_to_bytes(d.tokens, N, d.ttb, d.ttb_elem_bytes, d.ttb_rows, d.bpt, d.out_ids_pulled, nullptr, stream))) return rc;
    }
    if (d.counters) {
        int64_t blocks = (slots + kThreads * 16 - 1) / (kThreads * 16);
        if (blocks > 1024) blocks = 1024;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(count_pads_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, padded, after, slots, (int64_t)d.pad_byte, N, d.counters);
        if ((rc = check_launch("count_pads_kernel"))) return rc;
    }
    MotEmbedMixDesc g = d;
    g.id_source = MOT_IDS_GIVEN;
    g.ids_a = after;
    g.ids_b = d.add_padded ? padded : nullptr;
    return launch_composed(g, stream);
}

static int launch_composed(const MotEmbedMixDesc &d, hipStream_t stream) {
    const int64_t N = d.n_rows * d.tokens_per_row;
    const int Dt = d.tok_dim, Db = d.byte_dim, bpt = d.bpt, K = Dt + bpt * Db, Dm = d.model_dim;
    const int tok_lo = d.bytes_first ? bpt * Db : 0, byte_lo = d.bytes_first ? 0 : Dt;
    const bool bf = d.dtype == MOT_BF16;
    const size_t esz = bf ? 2 : 4;
    const float eps = d.eps > 0.f ? d.eps : (bf ? kBf16Eps : FLT_EPSILON);
    const size_t need = (composed_floats(d) + composed_rnorm_floats(d)) * sizeof(float);
    if (!d.workspace || d.workspace_bytes < need)
        return set_error(MOT_EWORKSPACE, "embed_mix concat_linear: needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
    char *u = (char *)d.workspace;
    float *rn = (float *)d.workspace + composed_floats(d);
    if (N == 0) return MOT_OK;
    if (bf && ((Dt & 7) || (Db & 7)))
        return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear bf16: tok_dim/byte_dim must be multiples of 8 (got %d, %d)", Dt, Db);
    int rc;
    // one id tensor and 16-byte-vector rows: the concat operand comes from ONE kernel (concat_rows_kernel, byte-row rms factors from a
    // table built once per call); otherwise from the two seam gathers
    const int vec = bf ? 8 : 4;
    const bool one_kernel = !d.ids_b && (Dt % vec) == 0 && (Db % vec) == 0;
    // bf16 at the shapes mot_concat16.hip takes: gather, contraction, bias and output norm are ONE kernel and u is never built
    const bool fused16 = bf && !(d.flags & MOT_FLAG_LINEAR_COMPOSED) && concat16_usable(d);
    if (fused16) {
        const bool rn_table = d.norm_byte && !concat16_norm_in_kernel(d);
        if (rn_table && (rc = launch_rows_rnorm(d.byte_table, d.byte_rows, Db, eps, rn, d.dtype, stream))) return rc;
        return launch_concat16(d, d.tokens, d.ids_a, nullptr, N, rn_table ? rn : nullptr, d.out, d.out_row_rnorm, stream);
    }
    if (one_kernel && d.norm_byte && (rc = launch_rows_rnorm(d.byte_table, d.byte_rows, Db, eps, rn, d.dtype, stream))) return rc;
    for (int64_t r0 = 0; r0 < N; r0 += kSlabRows) {
        const int64_t n = N - r0 < kSlabRows ? N - r0 : kSlabRows;
        if (one_kernel) {
            if ((rc = launch_concat_rows(d.tokens + r0, d.ids_a + r0 * bpt, n, d.tok_table, d.tok_rows, Dt, d.byte_table, d.byte_rows, Db, bpt, d.norm_tok,
                                         d.norm_byte ? rn : nullptr, eps, u, K, tok_lo, byte_lo, d.status, d.dtype, stream))) return rc;
        } else {
            if ((rc = launch_gather_rows_placed(d.tokens + r0, nullptr, 4, n, d.tok_table, d.tok_rows, Dt, d.norm_tok, eps, nullptr, u + tok_lo * esz, 1, K,
                                                d.status, kStatusTokenOor, d.dtype, stream))) return rc;
            if ((rc = launch_gather_rows_placed(d.ids_a + r0 * bpt, d.ids_b ? d.ids_b + r0 * bpt : nullptr, 8, n * bpt, d.byte_table, d.byte_rows, Db,
                                                d.norm_byte, eps, nullptr, u + byte_lo * esz, bpt, K, d.status, kStatusByteOor, d.dtype, stream))) return rc;
        }
        char *out = (char *)d.out + r0 * Dm * esz;
        float *rr = d.out_row_rnorm ? d.out_row_rnorm + r0 : nullptr;
        const unsigned nb = (unsigned)((n + kWaves - 1) / kWaves);
        if (bf) {
            if ((rc = launch_gemm_rows_bf16(u, K, n, d.weight, K, K, Dm, out, Dm, true, d.bias, stream))) return rc;
            if (d.norm_out) hipLaunchKernelGGL(rows_rms_inplace_kernel<__bf16>, dim3(nb), dim3(kThreads), 0, stream, (__bf16 *)out, n, Dm, eps, rr);
        } else {
            if ((rc = launch_gemm_rows((const float *)u, K, n, (const float *)d.weight, K, K, Dm, (float *)out, Dm, true, stream, (const float *)d.bias))) return rc;
            if (d.norm_out) hipLaunchKernelGGL(rows_rms_inplace_kernel<float>, dim3(nb), dim3(kThreads), 0, stream, (float *)out, n, Dm, eps, rr);
        }
        if (d.norm_out && (rc = check_launch("rows_rms_inplace_kernel"))) return rc;
    }
    return MOT_OK;
}

size_t embed_mix_linear_composed_workspace_bytes(const MotEmbedMixDesc &d) {
    return (composed_floats(d) + composed_rnorm_floats(d)) * sizeof(float) + composed_id_words(d) * sizeof(int64_t);
}
int launch_embed_mix_linear_composed(const MotEmbedMixDesc &d, hipStream_t stream) {
    return d.id_source == MOT_IDS_FROM_TTB ? launch_composed_from_ttb(d, stream) : launch_composed(d, stream);
}

size_t embed_mix_linear_workspace_bytes(const MotEmbedMixDesc &d) {
    if (composed_path(d)) return embed_mix_linear_composed_workspace_bytes(d);
    const int nt = nt_of(d.model_dim);
    if (nt < 0) return 0;
    const int K = d.tok_dim + d.bpt * d.byte_dim;
    const size_t Kpad = (size_t)(K + kBKmax - 1) / kBKmax * kBKmax, DmPad = (size_t)nt * 128;
    return (Kpad * DmPad + (((size_t)d.byte_rows + 3) & ~(size_t)3) + (d.norm_tok ? (size_t)d.tok_rows : 0)) * sizeof(float);
}

template <int MT, int NT, int BK, int OCC = 1, int ABL = 0>
static int launch_lin(LinArgs &P, const MotEmbedMixDesc &d, hipStream_t stream) {
    constexpr int TM = 32 * MT;
    P.M.tile_tokens = TM;
    const int64_t tiles_per_row = (d.tokens_per_row + TM - 1) / TM;
    P.M.tiles_per_row = (int)tiles_per_row;
    const int64_t blocks = d.n_rows * tiles_per_row;
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear: too many tiles");
    const size_t lds = lin_lds_bytes(P.DmPad, P.M.bpt, BK, TM);
    if (lds > 160 * 1024)
        return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear: needs %zu B of LDS (model_dim %d, bpt %d) > 160 KiB", lds, P.Dm, P.M.bpt);
    static std::atomic<uint64_t> lds_ok{0};   // per-device bits
    if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mix_linear_kernel<MT, NT, BK, OCC, ABL>, lds_ok, "embed_mix_linear_kernel")) return rc_lds;
    hipLaunchKernelGGL((embed_mix_linear_kernel<MT, NT, BK, OCC, ABL>), dim3((unsigned)blocks), dim3(kThreads), lds, stream, P);
    return check_launch("embed_mix_linear_kernel");
}

int launch_embed_mix_linear(const MotEmbedMixDesc &d, hipStream_t stream) { return launch_embed_mix_linear_ex(d, nullptr, 0, stream); }

// `wt_prebuilt` (optional): the k-major, zero-padded weight operand [Kpad rows][wt_cols columns] supplied by
// the caller (the backward passes W itself: for du = dy.W the nn.Linear layout already is k-major).
int launch_embed_mix_linear_ex(const MotEmbedMixDesc &d, const float *wt_prebuilt, int wt_cols, hipStream_t stream) {
    if (!wt_prebuilt && composed_path(d)) return launch_embed_mix_linear_composed(d, stream);
    const int nt = nt_of(d.model_dim);
    if (nt < 0) return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear: model_dim %d > 1024 is not built", d.model_dim);
    if ((d.tok_dim & 3) || (d.byte_dim & 3))
        return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear: tok_dim/byte_dim must be multiples of 4 (got %d, %d)", d.tok_dim, d.byte_dim);
    LinArgs P;
    fill_mix_args(P.M, d);
    P.K = d.tok_dim + d.bpt * d.byte_dim;
    P.Kpad = (P.K + kBKmax - 1) / kBKmax * kBKmax;
    P.Dm = d.model_dim;
    P.DmPad = nt * 128;
    P.bytes_first = d.bytes_first;
    P.dual = d.id_source == MOT_IDS_FROM_TTB ? d.add_padded != 0 : d.ids_b != nullptr;
    P.bias = (const float *)d.bias;
    P.row_rnorm = d.norm_out ? d.out_row_rnorm : nullptr;
    P.tok_rnorm = nullptr;
    int rc;
    float *rn = nullptr;
    if (wt_prebuilt) {
        if (wt_cols != P.DmPad) return set_error(MOT_EINVAL, "embed_mix concat_linear: prebuilt weight has %d columns, kernel wants %d", wt_cols, P.DmPad);
        P.Wt = wt_prebuilt;
    } else {
        const size_t need = embed_mix_linear_workspace_bytes(d);
        if (!d.workspace || d.workspace_bytes < need)
            return set_error(MOT_EWORKSPACE, "embed_mix concat_linear: needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
        float *Wt = (float *)d.workspace;
        rn = Wt + (size_t)P.Kpad * P.DmPad;
        P.Wt = Wt;
        hipLaunchKernelGGL(transpose_pad_kernel, dim3((unsigned)((P.Kpad + 31) / 32), (unsigned)((P.DmPad + 31) / 32)),
                           dim3(kThreads), 0, stream, (const float *)d.weight, P.Dm, P.K, Wt, P.Kpad, P.DmPad);
        if ((rc = check_launch("transpose_pad_kernel"))) return rc;
    }
    if (d.norm_byte && !P.dual && rn) {
        rc = launch_rows_rnorm(P.M.byte_table, d.byte_rows, d.byte_dim, P.M.eps, rn, MOT_F32, stream);
        if (rc) return rc;
        P.M.byte_rnorm = rn;
    }
    if (d.norm_tok && rn) {   // one streaming pass over the token table instead of a dependent row fetch per token per tile
        float *tr = rn + (((size_t)d.byte_rows + 3) & ~(size_t)3);
        rc = launch_rows_rnorm(P.M.tok_table, d.tok_rows, d.tok_dim, P.M.eps, tr, MOT_F32, stream);
        if (rc) return rc;
        P.tok_rnorm = tr;
    }
    // 64-token tiles while the accumulators (2*NT*16 registers) leave room for the partial sums in the
    // 256 architectural VGPRs; the 1024-column variant runs 32-token tiles instead.
    switch (nt) {
        case 1: return launch_lin<2, 1, 16>(P, d, stream);
        case 2: return launch_lin<2, 2, 16>(P, d, stream);
        case 3: return launch_lin<2, 3, 16>(P, d, stream);
        case 4: return launch_lin<2, 4, 16>(P, d, stream);
        case 6: {
#ifdef MOT_DEV_ABLATION  // timing-only variants (wrong results), see the ABL template parameter
            const char *abl = getenv("MOT_LIN_ABL");
            if (abl) switch (atoi(abl)) {
                case 1: return launch_lin<2, 6, 16, 1, 1>(P, d, stream);
                case 6: return launch_lin<2, 6, 16, 1, 6>(P, d, stream);
                case 14: return launch_lin<2, 6, 16, 1, 14>(P, d, stream);
                case 15: return launch_lin<2, 6, 16, 1, 15>(P, d, stream);
            }
#endif
            return launch_lin<2, 6, 16>(P, d, stream);
        }
        default: return launch_lin<1, 8, 16>(P, d, stream);
    }
}

}  // namespace mot
