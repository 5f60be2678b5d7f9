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
// Geometry: 256 threads (4 waves, one per SIMD), a tile of TM = 32*MT tokens x all Dm columns (complete
// rows for the post-norm), wave w owning columns [w*NT*32, (w+1)*NT*32).  K is walked in steps of 16
// with double-buffered LDS (rows padded to 48 bytes: conflict-free ds_read_b128).  At 16x the fp32 MFMA
// rate this kernel is bound by streaming W from L2 into LDS once per tile, so the tile is as tall as
// the accumulators allow without spilling (128 tokens up to Dm = 384, 64 tokens above).
#include "mot_mix.hpp"

namespace mot {

typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBK16 = 16;               // K-step in elements = one MFMA k-slice
constexpr int kPPR = kBK16 / 8;         // 16-byte pieces per staged row
constexpr int kRowB = 2 * kBK16 + 16;   // bytes per staged row: 32 data + 16 pad (lane slots 3r+h mod 16 are distinct)

struct LinArgs16 {
    MixArgs M;
    const __bf16 *W;     // [Dm, K] nn.Linear layout
    const __bf16 *bias;  // [Dm] or null
    int K, Dm, bytes_first, dual;
    float *row_rnorm;    // optional [n_rows*T]: the post-norm factor of every row (for the backward)
};

__host__ __device__ inline size_t lin16_lds_bytes(int DmPad, int bpt, int tm) {
    size_t b = 2 * (size_t)(DmPad + tm) * kRowB + (size_t)tm * (1 + bpt) * 4 + 4 * (size_t)tm * 4 + (size_t)tm * 4;
    b = (b + 15) & ~(size_t)15;
    return b + tile_lds_bytes(tm, bpt, true);
}

template <int MT, int NT>
__global__ __launch_bounds__(kThreads, 1) void embed_mix_linear_bf16_kernel(const LinArgs16 P) {
    constexpr int kTM = 32 * MT;
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
    if (A.id_source == MOT_IDS_FROM_TTB) {
        if (A.pull_dir == kPullLeft) phase1_from_ttb<kPullLeft>(A, L, row, t0, ntok);
        else if (A.pull_dir == kPullRight) phase1_from_ttb<kPullRight>(A, L, row, t0, ntok);
        else phase1_from_ttb<kPullNone>(A, L, row, t0, ntok);
    } else {
        phase1_given(A, L, row, t0, ntok);
    }

    // ---- per-segment rms factors (folded into the A operand before it is rounded to bf16)
    for (int i = tid; i < kTM * SS; i += kThreads) scale[i] = 1.0f;
    if (tid < kTM) {
        int tok = tid < ntok ? L.tok[tid] : 0;
        if ((uint64_t)(uint32_t)tok >= (uint64_t)A.tok_rows) {
            if (A.status) atomicOr(A.status, kStatusTokenOor);
            tok = 0;
        }
        tokc[tid] = tok;
    }
    __syncthreads();
    if (A.norm_tok) {
        for (int t = wave; t < ntok; t += kWaves) {
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
        for (int p = tid; p < ntok * bpt; p += kThreads) {
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
    constexpr int WP = DmPad * kPPR / kThreads;  // 16-byte pieces of the W chunk per thread
    constexpr int AP = (kTM * kPPR + kThreads - 1) / kThreads;
    bf16x8 wreg[WP], areg[AP];

    auto load_stage = [&](int s) {
        const int k0 = s * kBK16;
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const int q = p * kThreads + tid, n = q / kPPR, c = q % kPPR;   // row n of W, 8-element piece c of the step
            const int k = min(k0 + 8 * c, P.K - 8);
            const int nn = min(n, P.Dm - 1);
            bf16x8 v = *(const bf16x8 *)(P.W + (int64_t)nn * P.K + k);
            if (n >= P.Dm || k0 + 8 * c >= P.K) v = (bf16x8)((__bf16)0.f);
            wreg[p] = v;
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int q = p * kThreads + tid, m = min(q / kPPR, kTM - 1), c = q % kPPR;
            const int amr = min(m, ntok - 1);
            const int kreal = k0 + 8 * c;
            const int k = min(kreal, P.K - 8);
            const bool is_tok = P.bytes_first ? k >= nbytes_k : k < A.Dt;
            const int kb = P.bytes_first ? k : k - A.Dt;
            const int slot = is_tok ? 0 : kb / A.Db;
            const int off = is_tok ? (P.bytes_first ? k - nbytes_k : k) : kb - slot * A.Db;
            const int id1 = L.ids[amr * sv + slot];
            const __bf16 *p1 = is_tok ? tok_table + (int64_t)tokc[amr] * A.Dt + off : byte_table + (int64_t)id1 * A.Db + off;
            float8v v = Elem<__bf16>::loadv(p1);
            if (P.dual) {
                const float8v v2 = Elem<__bf16>::loadv(byte_table + (int64_t)L.val[amr * sv + slot] * A.Db + off);
                if (!is_tok) v += v2;
            }
            v *= scale[amr * SS + (is_tok ? 0 : 1 + slot)];
            if (is_tok ? scale_t : scale_b) v *= is_tok ? s_tok : s_byte;
            if (kreal >= P.K) v = (float8v)(0.f);
            areg[p] = __builtin_convertvector(v, bf16x8);   // the segment as the reference holds it: bf16
        }
    };
    auto store_stage = [&](int buf) {
        char *wb = W0 + (size_t)buf * DmPad * kRowB;
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const int q = p * kThreads + tid, n = q / kPPR, c = q % kPPR;
            *(bf16x8 *)(wb + n * kRowB + 16 * c) = wreg[p];
        }
        char *ab = A0 + (size_t)buf * kTM * kRowB;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int q = p * kThreads + tid, m = q / kPPR, c = q % kPPR;
            if (m < kTM) *(bf16x8 *)(ab + m * kRowB + 16 * c) = areg[p];
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
    const int n0 = wave * (NT * 32);
    const int nsteps = (P.K + kBK16 - 1) / kBK16;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        load_stage(min(s + 1, nsteps - 1));
        const char *ab = A0 + (size_t)(s & 1) * kTM * kRowB, *wb = W0 + (size_t)(s & 1) * DmPad * kRowB;
#pragma unroll
        for (int kk = 0; kk < kBK16 / 16; ++kk) {
            bf16x8 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *(const bf16x8 *)(ab + (mt * 32 + li) * kRowB + 32 * kk + 16 * h);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 bf = *(const bf16x8 *)(wb + (n0 + nt * 32 + li) * kRowB + 32 * kk + 16 * h);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf, acc[mt][nt], 0, 0, 0);
            }
        }
        store_stage((s + 1) & 1);
        __syncthreads();
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
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ss = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) ss += acc[mt][nt][r] * acc[mt][nt][r];  // padded columns hold 0
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
                if (li == 0) rowss[wave * kTM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = ss;
            }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float tot = ((rowss[m] + rowss[kTM + m]) + rowss[2 * kTM + m]) + rowss[3 * kTM + m];
                const float rs = rms_scale(tot, P.Dm, A.eps);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] *= rs;
                if (P.row_rnorm && wave == 0 && li == 0 && m < ntok) P.row_rnorm[row * A.T + t0 + m] = rs;
            }
    }
    __bf16 *orow = (__bf16 *)A.out + (row * A.T + t0) * (int64_t)P.Dm;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < ntok) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = n0 + nt * 32 + li;
                    if (col < P.Dm) orow[(int64_t)m * P.Dm + col] = (__bf16)acc[mt][nt][r];
                }
            }
        }
}

template <int MT, int NT>
static int launch_lin16(LinArgs16 &P, const MotEmbedMixDesc &d, hipStream_t stream) {
    constexpr int TM = 32 * MT;
    P.M.tile_tokens = TM;
    const int64_t tiles_per_row = (d.tokens_per_row + TM - 1) / TM;
    P.M.tiles_per_row = (int)tiles_per_row;
    const int64_t blocks = d.n_rows * tiles_per_row;
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear: too many tiles");
    const size_t lds = lin16_lds_bytes(NT * 128, P.M.bpt, TM);
    if (lds > 160 * 1024)
        return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear bf16: needs %zu B of LDS (model_dim %d, bpt %d) > 160 KiB", lds, P.Dm, P.M.bpt);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)embed_mix_linear_bf16_kernel<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return set_error(MOT_EHIP, "hipFuncSetAttribute(embed_mix_linear_bf16_kernel): %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((embed_mix_linear_bf16_kernel<MT, NT>), dim3((unsigned)blocks), dim3(kThreads), lds, stream, P);
    return check_launch("embed_mix_linear_bf16_kernel");
}

size_t embed_mix_linear_bf16_workspace_bytes(const MotEmbedMixDesc &d) { return d.norm_byte ? (size_t)d.byte_rows * sizeof(float) : 0; }

int launch_embed_mix_linear_bf16(const MotEmbedMixDesc &d, hipStream_t stream) {
    if (d.model_dim > 1024) return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear bf16: model_dim %d > 1024 is not built", d.model_dim);
    if ((d.tok_dim & 7) || (d.byte_dim & 7))
        return set_error(MOT_EUNSUPPORTED, "embed_mix concat_linear bf16: tok_dim/byte_dim must be multiples of 8 (got %d, %d)", d.tok_dim, d.byte_dim);
    LinArgs16 P;
    fill_mix_args(P.M, d);
    P.K = d.tok_dim + d.bpt * d.byte_dim;
    P.Dm = d.model_dim;
    P.bytes_first = d.bytes_first;
    P.dual = d.id_source == MOT_IDS_FROM_TTB ? d.add_padded != 0 : d.ids_b != nullptr;
    P.W = (const __bf16 *)d.weight;
    P.bias = (const __bf16 *)d.bias;
    P.row_rnorm = d.norm_out ? d.out_row_rnorm : nullptr;
    if (d.norm_byte && !P.dual) {
        const size_t need = (size_t)d.byte_rows * sizeof(float);
        if (!d.workspace || d.workspace_bytes < need)
            return set_error(MOT_EWORKSPACE, "embed_mix concat_linear bf16: needs %zu workspace bytes, got %zu", need, d.workspace_bytes);
        int rc = launch_rows_rnorm(d.byte_table, d.byte_rows, d.byte_dim, P.M.eps, (float *)d.workspace, MOT_BF16, stream);
        if (rc) return rc;
        P.M.byte_rnorm = (const float *)d.workspace;
    }
    const int nt = (d.model_dim + 127) / 128;
    // the tallest tile whose accumulators (MT*NT*16 registers) and LDS image still fit
    switch (nt) {
        case 1: return launch_lin16<4, 1>(P, d, stream);
        case 2: return launch_lin16<4, 2>(P, d, stream);
        case 3: return launch_lin16<4, 3>(P, d, stream);
        case 4: return launch_lin16<2, 4>(P, d, stream);
        case 5:
        case 6: return launch_lin16<2, 6>(P, d, stream);
        default: return launch_lin16<2, 8>(P, d, stream);
    }
}

}  // namespace mot
