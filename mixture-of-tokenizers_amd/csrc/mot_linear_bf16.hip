// mot_linear_bf16.hip -- CONCAT_LINEAR with bf16 tables / weight / output (gfx950): what the
// reference's training loop executes -- nn.Embedding cast to bf16 (scaled-pre-train/train_gpt.py:1124-1126),
// CastedLinear casting its weight to the activations' dtype (:185-186), rms_norm in fp32 internally with
// eps = finfo(bfloat16).eps (:172-173).
//
//   x = bf16( rms_norm?( W . cat(seg_0 .. seg_bpt) + bias ) ),   seg = bf16( raw * r [* scale] )
//
// v_mfma_f32_32x32x16_bf16, fp32 accumulation.  Both operands are consumed in their NATURAL layouts:
// lane (r = l&31, h = l>>5) of the A fragment wants A[row r][k = 8h..8h+7] -- 16 contiguous bytes of a
// gathered row -- and of the B fragment B[k = 8h..8h+7][col r] = W[r][8h..8h+7] -- 16 contiguous bytes of
// a row of the nn.Linear weight.  So there is no transposed copy and no workspace: the weight is staged
// row by row exactly as it lies in memory.
//
// Geometry: RG row groups x 4 column groups of waves (256*RG threads), a tile of TM = 32*MT*RG tokens x all Dm
// columns (complete rows for the post-norm); wave (rg, cg) owns rows [rg*32*MT, ..) and columns
// [cg*NT*32, ..).  K is walked in steps of 16 with double-buffered LDS (rows padded to 48 bytes:
// conflict-free ds_read_b128).  Measured on config 2 (timing ablations, 64-token tiles, 4 waves, one workgroup per
// CU): 42 % of the time was outside the K loop (tile index phase, a serial per-token rms pass, 2-byte stores) and
// the loop waited on L2 -> LDS staging with nothing to overlap; MFMA time was invisible.  Hence: two waves per
// SIMD (RG = 2) on a 128-token tile -- W is staged half as often per token and a second wave per SIMD covers
// the staging latency -- and the token rows' rms factors come from a table computed once per call.
#include <stdlib.h>
#include <type_traits>

#include "mot_mix.hpp"

namespace mot {

typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBK16 = 16;               // K-step in elements = one MFMA k-slice
constexpr int kPPR = kBK16 / 8;         // 16-byte pieces per staged row
constexpr int kRowB = 2 * kBK16 + 16;   // bytes per staged row: 32 data + 16 pad (lane slots 3r+h mod 16 are distinct)

struct LinArgs16 {
    MixArgs M;
    const __bf16 *W;     // step-major slabs [nsteps][DmPad][16] (pack_w_bf16_kernel)
    const __bf16 *bias;  // [Dm] or null
    int K, Dm, bytes_first, dual;
    float *row_rnorm;    // optional [n_rows*T]: the post-norm factor of every row (for the backward)
    const float *tok_rnorm;  // optional [tok_rows]: 1/rms of every token-table row (norm_tok), computed once per call
    int abl;             // dev-only timing ablations (MOT_DEV_ABLATION builds): 1 no global loads after step 0, 2 no LDS stores, 4 no MFMA, 8 a single K step
};

__host__ __device__ inline size_t lin16_lds_bytes(int DmPad, int bpt, int tm) {
    size_t b = 2 * (size_t)(DmPad + tm) * kRowB + (size_t)tm * (1 + bpt) * 4 + 4 * (size_t)tm * 4 + (size_t)tm * 4;
    b = (b + 15) & ~(size_t)15;
    return b + tile_lds_bytes(tm, bpt, true);
}

// W [Dm, K] (nn.Linear layout) -> step-major slabs Wp[step][n < DmPad][16]: one K step of the main kernel then reads
// one contiguous DmPad*32-byte slab (whole 128-byte lines).  Read straight from W, a step touches 32 bytes of each of
// the Dm rows -- a quarter of every line it pulls through L2 -- and the kernel ran at the L2 line rate (measured:
// 0.18 of 0.40 ms stalled on these loads at config 2).  Rows n >= Dm and columns k >= K are zero.
__global__ __launch_bounds__(kThreads) void pack_w_bf16_kernel(const __bf16 *__restrict__ W, int Dm, int K, int DmPad, int nsteps,
                                                               __bf16 *__restrict__ Wp) {
    const int64_t pieces = (int64_t)nsteps * DmPad * kPPR;   // 8-element pieces
    for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < pieces; q += (int64_t)gridDim.x * kThreads) {
        const int c = (int)(q % kPPR);
        const int64_t r = q / kPPR;
        const int n = (int)(r % DmPad), s = (int)(r / DmPad);
        const int k = s * kBK16 + 8 * c;
        bf16x8 v = (bf16x8)((__bf16)0.f);
        if (n < Dm && k < K) v = *(const bf16x8 *)(W + (int64_t)n * K + k);   // K is a multiple of 8
        *(bf16x8 *)(Wp + q * 8) = v;
    }
}

template <int MT, int NT, int RG>
__global__ __launch_bounds__(kThreads * RG, 1) void embed_mix_linear_bf16_kernel(const LinArgs16 P) {
    constexpr int kTM = 32 * MT * RG;
    constexpr int NTHR = kThreads * RG;
    constexpr int DmPad = NT * 128;
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    const MixArgs &A = P.M;
    const int bpt = A.bpt, sv = bpt | 1, SS = 1 + bpt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __bf16 *tok_table = (const __bf16 *)A.tok_table, *byte_table = (const __bf16 *)A.byte_table;
    // LDS: [W buf0][W buf1][A buf0][A buf1][scale][rowss][tokc][tile arrays]
    char *base = (char *)lds;
    char *W0 = base;                       base += 2 * (size_t)DmPad * kRowB;
    char *A0 = base;                       base += 2 * (size_t)kTM * kRowB;
    float *scale = (float *)base;          base += (size_t)kTM * SS * 4;
    float *rowss = (float *)base;          base += 4 * (size_t)kTM * 4;
    int32_t *tokc = (int32_t *)base;       base += (size_t)kTM * 4;
    base = (char *)(((uintptr_t)base + 15) & ~(uintptr_t)15);
    const TileLds L = tile_lds_carve((int32_t *)base, kTM, bpt, true);

    const int64_t row = blockIdx.x / A.tiles_per_row;
    const int64_t t0 = (int64_t)(blockIdx.x % A.tiles_per_row) * kTM;
    const int ntok = (int)min((int64_t)kTM, A.T - t0);

    // ---- phase 1: byte ids of the tile
    if (P.abl & 16) {
        for (int i = tid; i < kTM * sv; i += NTHR) { L.ids[i] = 0; L.val[i] = 0; }
        for (int i = tid; i < kTM; i += NTHR) L.tok[i] = 0;
        __syncthreads();
    } else if (A.id_source == MOT_IDS_FROM_TTB) {
        if (A.pull_dir == kPullLeft) phase1_from_ttb<kPullLeft>(A, L, row, t0, ntok);
        else if (A.pull_dir == kPullRight) phase1_from_ttb<kPullRight>(A, L, row, t0, ntok);
        else phase1_from_ttb<kPullNone>(A, L, row, t0, ntok);
    } else {
        phase1_given(A, L, row, t0, ntok);
    }

    // ---- per-segment rms factors (folded into the A operand before it is rounded to bf16)
    for (int i = tid; i < kTM * SS; i += NTHR) scale[i] = 1.0f;
    if (tid < kTM) {
        int tok = tid < ntok ? L.tok[tid] : 0;
        if ((uint64_t)(uint32_t)tok >= (uint64_t)A.tok_rows) {
            if (A.status) atomicOr(A.status, kStatusTokenOor);
            tok = 0;
        }
        tokc[tid] = tok;
    }
    __syncthreads();
    if (A.norm_tok && P.tok_rnorm) {
        for (int t = tid; t < ntok; t += NTHR) scale[t * SS] = P.tok_rnorm[tokc[t]];
    } else if (A.norm_tok && !(P.abl & 32)) {
        for (int t = wave; t < ntok; t += kWaves * RG) {
            const __bf16 *trow = tok_table + (int64_t)tokc[t] * A.Dt;
            float ss = 0.f;
            for (int c = lane; c < (A.Dt >> 3); c += 64) {
                const float8v v = Elem<__bf16>::loadv(trow + 8 * c);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
            }
            ss = wave_sum(ss);
            if (lane == 0) scale[t * SS] = rms_scale(ss, A.Dt, A.eps);
        }
    }
    if (A.norm_byte) {
        for (int p = tid; p < ntok * bpt; p += NTHR) {
            const int t = p / bpt, k = p - t * bpt;
            const int id = L.ids[t * sv + k];
            float r;
            if (!P.dual) {
                r = A.byte_rnorm[id];
            } else {  // norm(emb(padded) + emb(pulled)), train_gpt.py:378
                const int id2 = L.val[t * sv + k];
                const __bf16 *pa = byte_table + (int64_t)id * A.Db, *pb = byte_table + (int64_t)id2 * A.Db;
                float ss = 0.f;
                for (int j = 0; j < A.Db; j += 8) {
                    const float8v v = Elem<__bf16>::loadv(pa + j) + Elem<__bf16>::loadv(pb + j);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
                }
                r = rms_scale(ss, A.Db, A.eps);
            }
            scale[t * SS + 1 + k] = r;
        }
    }
    __syncthreads();

    // ---- K loop
    const float s_tok = A.scale_tok ? *A.scale_tok : 1.0f;
    const float s_byte = A.scale_byte ? *A.scale_byte : 1.0f;
    const bool scale_t = A.scale_tok != nullptr, scale_b = A.scale_byte != nullptr;
    const int nbytes_k = bpt * A.Db;
    constexpr int WP = (DmPad * kPPR + NTHR - 1) / NTHR;  // 16-byte pieces of the W chunk per thread
    constexpr int AP = (kTM * kPPR + NTHR - 1) / NTHR;
    // three register sets: the loads of K steps s+1, s+2, s+3 are in flight while step s is multiplied (one step of
    // 12 MFMAs is ~0.2 us, an L2 round trip under load 1-2 us)
    constexpr int kSets = NT >= 8 ? 2 : 3;   // the widest variant has no registers for a third set
    bf16x8 wreg[kSets][WP], areg[kSets][AP], areg2[kSets][AP];
    float afac[kSets][AP];
    bool atok[kSets][AP];
    const float inv_db = 1.0f / (float)A.Db;

    auto load_stage = [&](int s, auto setc) {
        constexpr int SET = decltype(setc)::value;
        const int k0 = s * kBK16;
        const __bf16 *slab = P.W + (int64_t)s * (DmPad * kBK16);
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const int q = min(p * NTHR + tid, DmPad * kPPR - 1);   // piece q of the slab: 16 contiguous bytes (row q / 2, half q % 2)
            wreg[SET][p] = *(const bf16x8 *)(slab + q * 8);
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            if (p * NTHR + (tid & ~63) >= kTM * kPPR) continue;   // wave-uniform: only the waves that own A pieces gather
            const int q = p * NTHR + tid, m = min(q / kPPR, kTM - 1), c = q % kPPR;
            const int amr = min(m, ntok - 1);
            const int kreal = k0 + 8 * c;
            const int k = min(kreal, P.K - 8);
            const bool is_tok = P.bytes_first ? k >= nbytes_k : k < A.Dt;
            const int kb = P.bytes_first ? k : k - A.Dt;
            const int slot = is_tok ? 0 : __float2int_rd(((float)kb + 0.5f) * inv_db);
            const int off = is_tok ? (P.bytes_first ? k - nbytes_k : k) : kb - slot * A.Db;
            const int id1 = L.ids[amr * sv + slot];
            const __bf16 *p1 = is_tok ? tok_table + (int64_t)tokc[amr] * A.Dt + off : byte_table + (int64_t)id1 * A.Db + off;
            // raw rows stay packed in registers until store_stage: scaling here would wait for the load at once
            areg[SET][p] = *(const bf16x8 *)p1;
            if (P.dual) areg2[SET][p] = *(const bf16x8 *)(byte_table + (int64_t)L.val[amr * sv + slot] * A.Db + off);
            float f = scale[amr * SS + (is_tok ? 0 : 1 + slot)];
            if (is_tok ? scale_t : scale_b) f *= is_tok ? s_tok : s_byte;
            if (kreal >= P.K) f = 0.f;
            afac[SET][p] = f;
            atok[SET][p] = is_tok;
        }
    };
    auto store_stage = [&](int buf, auto setc) {
        constexpr int SET = decltype(setc)::value;
        char *wb = W0 + (size_t)buf * DmPad * kRowB;
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const int q = p * NTHR + tid, n = q / kPPR, c = q % kPPR;
            if (n < DmPad) *(bf16x8 *)(wb + n * kRowB + 16 * c) = wreg[SET][p];
        }
        char *ab = A0 + (size_t)buf * kTM * kRowB;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int q = p * NTHR + tid, m = q / kPPR, c = q % kPPR;
            if (m < kTM) {
                float8v v = __builtin_convertvector(areg[SET][p], float8v);
                if (P.dual && !atok[SET][p]) v += __builtin_convertvector(areg2[SET][p], float8v);
                v *= afac[SET][p];
                if (afac[SET][p] == 0.f) v = (float8v)(0.f);   // the zero padding of the last step, whatever the clamped load read
                *(bf16x8 *)(ab + m * kRowB + 16 * c) = __builtin_convertvector(v, bf16x8);   // the segment as the reference holds it: bf16
            }
        }
    };

    f32x16c acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int h = lane >> 5, li = lane & 31;
    const int rg = wave >> 2, cg = wave & 3;       // row group, column group
    const int m0 = rg * (32 * MT), n0 = cg * (NT * 32);
    const int nsteps = (P.abl & 8) ? 1 : (P.K + kBK16 - 1) / kBK16;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    // step s: its operands sit in LDS buffer s & 1; register set s % 3 is free (stored during step s-1) and takes the
    // loads of step s+3; set (s+1) % 3 -- issued two steps ago -- is written to the other buffer after the MFMAs
    auto step = [&](int s, auto setc, auto setn) {
        if (!(P.abl & 1)) load_stage(min(s + kSets, nsteps - 1), setc);
        const char *ab = A0 + (size_t)(s & 1) * kTM * kRowB, *wb = W0 + (size_t)(s & 1) * DmPad * kRowB;
#pragma unroll
        for (int kk = 0; kk < kBK16 / 16; ++kk) {
            bf16x8 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *(const bf16x8 *)(ab + (m0 + mt * 32 + li) * kRowB + 32 * kk + 16 * h);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 bf = *(const bf16x8 *)(wb + (n0 + nt * 32 + li) * kRowB + 32 * kk + 16 * h);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (!(P.abl & 4)) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf, acc[mt][nt], 0, 0, 0);
            }
        }
        if (!(P.abl & 2)) store_stage((s + 1) & 1, setn);
        __syncthreads();
    };
    load_stage(0, S0{});
    load_stage(min(1, nsteps - 1), S1{});
    if constexpr (kSets == 3) load_stage(min(2, nsteps - 1), S2{});
    store_stage(0, S0{});
    __syncthreads();
    int s = 0;
    if constexpr (kSets == 3) {
        for (; s + 3 <= nsteps; s += 3) {
            step(s, S0{}, S1{});
            step(s + 1, S1{}, S2{});
            step(s + 2, S2{}, S0{});
        }
        if (s < nsteps) step(s, S0{}, S1{});
        if (s + 1 < nsteps) step(s + 1, S1{}, S2{});
    } else {
        for (; s + 2 <= nsteps; s += 2) {
            step(s, S0{}, S1{});
            step(s + 1, S1{}, S0{});
        }
        if (s < nsteps) step(s, S0{}, S1{});
    }

    // ---- epilogue.  C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (P.bias) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + nt * 32 + li;
            const float bv = col < P.Dm ? (float)P.bias[col] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] += bv;
        }
    }
    if (A.norm_out) {
        // y is a bf16 tensor in the reference (CastedLinear output, train_gpt.py:185-186) before norm() upcasts it (172-173)
        // (rounded where it is used, twice, rather than in a pass of its own: that pass made the widest variant spill)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ss = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { const float yv = (float)(__bf16)acc[mt][nt][r]; ss += yv * yv; }  // padded columns hold 0
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
                if (li == 0) rowss[cg * kTM + m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = ss;
            }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float tot = ((rowss[m] + rowss[kTM + m]) + rowss[2 * kTM + m]) + rowss[3 * kTM + m];
                const float rs = rms_scale(tot, P.Dm, A.eps);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] = (float)(__bf16)acc[mt][nt][r] * rs;
                if (P.row_rnorm && cg == 0 && li == 0 && m < ntok) P.row_rnorm[row * A.T + t0 + m] = rs;
            }
    }
    __bf16 *orow = (__bf16 *)A.out + (row * A.T + t0) * (int64_t)P.Dm;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < ntok && !(P.abl & 64)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = n0 + nt * 32 + li;
                    if (col < P.Dm) orow[(int64_t)m * P.Dm + col] = (__bf16)acc[mt][nt][r];
                }
            }
        }
}

template <int MT, int NT, int RG>
static int launch_lin16(LinArgs16 &P, const MotEmbedMixDesc &d, hipStream_t stream) {
    constexpr int TM = 32 * MT * RG;
    P.M.tile_tokens = TM;
    const int64_t tiles_per_row = (d.tokens_per_row + TM - 1) / TM;
    P.M.tiles_per_row = (int)tiles_per_row;
    const int64_t blocks = d.n_rows * tiles_per_row;
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear: too many tiles");
    const size_t lds = lin16_lds_bytes(NT * 128, P.M.bpt, TM);
    if (lds > 160 * 1024)
        return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear bf16: needs %zu B of LDS (model_dim %d, bpt %d) > 160 KiB", lds, P.Dm, P.M.bpt);
    static std::atomic<uint64_t> lds_ok{0};   // per-device bits
    if (int rc_lds = ensure_max_dyn_lds((const void *)embed_mix_linear_bf16_kernel<MT, NT, RG>, lds_ok, "embed_mix_linear_bf16_kernel")) return rc_lds;
    hipLaunchKernelGGL((embed_mix_linear_bf16_kernel<MT, NT, RG>), dim3((unsigned)blocks), dim3(kThreads * RG), lds, stream, P);
    return check_launch("embed_mix_linear_bf16_kernel");
}

// workspace: [byte-row rms factors: byte_rows (norm_byte)][token-row rms factors: tok_rows (norm_tok)][packed W]
static size_t ws_byte_floats(const MotEmbedMixDesc &d) { return d.norm_byte ? ((size_t)d.byte_rows + 3) & ~(size_t)3 : 0; }
static size_t ws_tok_floats(const MotEmbedMixDesc &d) { return d.norm_tok ? ((size_t)d.tok_rows + 3) & ~(size_t)3 : 0; }
static int nt_of16(int Dm) {   // column tiles of 128 per workgroup, as dispatched below
    const int nt = (Dm + 127) / 128;
    return nt == 5 ? 6 : (nt == 7 ? 8 : nt);
}
static size_t ws_packed_w_bytes(const MotEmbedMixDesc &d) {
    const int K = d.tok_dim + d.bpt * d.byte_dim, nsteps = (K + kBK16 - 1) / kBK16, DmPad = nt_of16(d.model_dim) * 128;
    return (size_t)nsteps * DmPad * kBK16 * 2;
}
size_t embed_mix_linear_bf16_workspace_bytes(const MotEmbedMixDesc &d) {
    if (embed_mix_linear_is_composed(d)) return embed_mix_linear_composed_workspace_bytes(d);
    return (ws_byte_floats(d) + ws_tok_floats(d)) * sizeof(float) + ws_packed_w_bytes(d);
}

int launch_embed_mix_linear_bf16(const MotEmbedMixDesc &d, hipStream_t stream) {
    if (embed_mix_linear_is_composed(d)) return launch_embed_mix_linear_composed(d, stream);   // bpt > 0: not the dense-row mode of the backward
    if (d.model_dim > 1024) return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear bf16: model_dim %d > 1024 is not built", d.model_dim);
    if ((d.tok_dim & 7) || (d.byte_dim & 7))
        return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear bf16: tok_dim/byte_dim must be multiples of 8 (got %d, %d)", d.tok_dim, d.byte_dim);
    LinArgs16 P;
    fill_mix_args(P.M, d);
    P.K = d.tok_dim + d.bpt * d.byte_dim;
    P.Dm = d.model_dim;
    P.bytes_first = d.bytes_first;
    P.dual = d.id_source == MOT_IDS_FROM_TTB ? d.add_padded != 0 : d.ids_b != nullptr;
    P.bias = (const __bf16 *)d.bias;
    P.row_rnorm = d.norm_out ? d.out_row_rnorm : nullptr;
    P.abl = 0;
#ifdef MOT_DEV_ABLATION
    if (getenv("MOT_LIN16_ABL")) P.abl = atoi(getenv("MOT_LIN16_ABL"));
#endif
    const size_t need = embed_mix_linear_bf16_workspace_bytes(d);
    if (need && (!d.workspace || d.workspace_bytes < need))
        return set_error(MOT_EWORKSPACE, "embed_mix concat_linear bf16: needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
    if (d.norm_byte && !P.dual) {
        int rc = launch_rows_rnorm(d.byte_table, d.byte_rows, d.byte_dim, P.M.eps, (float *)d.workspace, MOT_BF16, stream);
        if (rc) return rc;
        P.M.byte_rnorm = (const float *)d.workspace;
    }
    P.tok_rnorm = nullptr;
    if (d.norm_tok) {   // one pass over the token table (L2 / HBM streaming) instead of a dependent row fetch per token per tile
        float *tr = (float *)d.workspace + ws_byte_floats(d);
        int rc = launch_rows_rnorm(d.tok_table, d.tok_rows, d.tok_dim, P.M.eps, tr, MOT_BF16, stream);
        if (rc) return rc;
        P.tok_rnorm = tr;
    }
    {
        const int nsteps = (P.K + kBK16 - 1) / kBK16, DmPad = nt_of16(d.model_dim) * 128;
        __bf16 *wp = (__bf16 *)((float *)d.workspace + ws_byte_floats(d) + ws_tok_floats(d));
        hipLaunchKernelGGL(pack_w_bf16_kernel, dim3(256), dim3(kThreads), 0, stream, (const __bf16 *)d.weight, d.model_dim, P.K, DmPad, nsteps, wp);
        int rc = check_launch("pack_w_bf16_kernel");
        if (rc) return rc;
        P.W = wp;
    }
    const int nt = (d.model_dim + 127) / 128;
    // the tallest tile whose accumulators (MT*NT*16 registers) and LDS image still fit
    // 8 waves (two per SIMD, 256 registers each): 128-token tiles while 2*NT*16 accumulators fit, 64 tokens at NT = 8
    switch (nt) {
        case 1: return launch_lin16<2, 1, 2>(P, d, stream);
        case 2: return launch_lin16<2, 2, 2>(P, d, stream);
        case 3: return launch_lin16<2, 3, 2>(P, d, stream);
        case 4: return launch_lin16<1, 4, 2>(P, d, stream);
        case 6: return launch_lin16<1, 6, 2>(P, d, stream);
        default: return launch_lin16<1, 8, 2>(P, d, stream);
    }
}

}  // namespace mot
