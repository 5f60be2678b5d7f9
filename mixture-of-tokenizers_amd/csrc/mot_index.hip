// mot_index.hip -- integer path kernels (gfx950): tokens_to_bytes, pull_from_left/right,
// create_batch.  All outputs are bit-exact with the reference.
#include "mot_internal.hpp"
#include "mot_tile.hpp"

namespace mot {

// ------------------------------------------------------------------------------------------
// tokens_to_bytes (scaled-pre-train/data_creation.py:61-67): out[n, k] = (int64) ttb[tokens[n], k]
// HBM-bound on the int64 output (8*bpt B/token); the table (<= 3.2 MB) is L2-resident.
// ------------------------------------------------------------------------------------------
template <typename TabT>
__global__ __launch_bounds__(kThreads) void tokens_to_bytes_kernel(const int32_t *__restrict__ tokens,
                                                                   int64_t n_tokens, const TabT *__restrict__ ttb,
                                                                   int64_t ttb_rows, int bpt,
                                                                   int64_t *__restrict__ out, uint32_t *status) {
    const SlotLayout S(bpt);
    if (S.kq >= bpt) return;
    const int64_t stride = (int64_t)gridDim.x * S.tstride;
    for (int64_t n = (int64_t)blockIdx.x * S.tstride + S.tq; n < n_tokens; n += stride) {
        int id = tokens[n];
        if ((uint32_t)id >= (uint64_t)ttb_rows) {
            if (status) atomicOr(status, kStatusTokenOor);
            id = 0;
        }
        out[n * bpt + S.kq] = (int64_t)ttb[(int64_t)id * bpt + S.kq];
    }
}

// ------------------------------------------------------------------------------------------
// pull_from_left / pull_from_right on an int64 byte tensor (data_creation.py:179-305 / 71-176).
// One workgroup per tile of `tile_tokens` tokens of one row; see mot_tile.hpp.
// ------------------------------------------------------------------------------------------
template <int DIR>
__global__ __launch_bounds__(kThreads) void pull_bytes_kernel(const int64_t *__restrict__ in, int64_t *__restrict__ out,
                                                              int64_t tokens_per_row, int bpt, int64_t pad, int64_t eot,
                                                              int tile_tokens, int tiles_per_row) {
    extern __shared__ int32_t lds[];
    const TileLds L = tile_lds_carve(lds, tile_tokens, bpt, false);
    const int64_t row = blockIdx.x / tiles_per_row;
    const int64_t t0 = (int64_t)(blockIdx.x % tiles_per_row) * tile_tokens;
    const int ntok = (int)min((int64_t)tile_tokens, tokens_per_row - t0);
    const int64_t row_off = row * tokens_per_row * bpt;
    SrcRaw src{in + row_off, bpt, pad, eot};

    if ((threadIdx.x >> 6) == kWaves - 1) halo_walk<DIR>(src, t0, ntok, tokens_per_row, bpt, L);
    fill_raw_tile(src, t0, ntok, bpt, L);
    tile_scan_and_compact<DIR>(src, t0, ntok, tokens_per_row, bpt, L);

    const SlotLayout S(bpt);
    if (S.kq < bpt) {
        for (int t = S.tq; t < ntok; t += S.tstride) {
            int kind;
            const int payload = pulled_slot<DIR>(L, t, S.kq, ntok, bpt, &kind);
            const int64_t self = (t0 + t) * bpt + S.kq;
            int64_t v;
            if (kind == 1) v = pad;
            else v = src.row[kind == 2 ? self : (int64_t)payload];
            out[row_off + self] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// create_batch (data_creation.py:308-330): per token
//   [token | left-padded | pulled-from-left | right-padded | pulled-from-right]  int64
// Both pulls run on the same tile back to back, reusing the LDS arrays.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void create_batch_kernel(const int32_t *__restrict__ tokens, int64_t tokens_per_row,
                                                                const void *__restrict__ ttb_left,
                                                                const void *__restrict__ ttb_right, int elem,
                                                                int64_t ttb_rows, int bpt, int32_t pad, int32_t eot,
                                                                int64_t *__restrict__ out, uint32_t *status,
                                                                int tile_tokens, int tiles_per_row) {
    extern __shared__ int32_t lds[];
    const TileLds L = tile_lds_carve(lds, tile_tokens, bpt, false);
    const int sv = bpt | 1;
    const int64_t row = blockIdx.x / tiles_per_row;
    const int64_t t0 = (int64_t)(blockIdx.x % tiles_per_row) * tile_tokens;
    const int ntok = (int)min((int64_t)tile_tokens, tokens_per_row - t0);
    const int64_t width = 1 + 4 * (int64_t)bpt;
    int64_t *orow = out + (row * tokens_per_row + t0) * width;
    const SlotLayout S(bpt);

    {   // left-padded table, pull from left
        SrcTable src{tokens + row * tokens_per_row, ttb_left, ttb_rows, elem, bpt, pad, eot, status};
        if ((threadIdx.x >> 6) == kWaves - 1) halo_walk<kPullLeft>(src, t0, ntok, tokens_per_row, bpt, L);
        fill_table_tile(src, t0, ntok, bpt, L);
        tile_scan_and_compact<kPullLeft>(src, t0, ntok, tokens_per_row, bpt, L);
        if ((int)threadIdx.x < ntok) orow[threadIdx.x * width] = tokens[row * tokens_per_row + t0 + threadIdx.x];
        if (S.kq < bpt)
            for (int t = S.tq; t < ntok; t += S.tstride) {
                int kind;
                const int payload = pulled_slot<kPullLeft>(L, t, S.kq, ntok, bpt, &kind);
                const int own = L.val[t * sv + S.kq];
                orow[t * width + 1 + S.kq] = own;
                orow[t * width + 1 + bpt + S.kq] = kind == 1 ? pad : (kind == 2 ? own : payload);
            }
        __syncthreads();
    }
    {   // right-padded table, pull from right
        SrcTable src{tokens + row * tokens_per_row, ttb_right, ttb_rows, elem, bpt, pad, eot, status};
        if ((threadIdx.x >> 6) == kWaves - 1) halo_walk<kPullRight>(src, t0, ntok, tokens_per_row, bpt, L);
        fill_table_tile(src, t0, ntok, bpt, L);
        tile_scan_and_compact<kPullRight>(src, t0, ntok, tokens_per_row, bpt, L);
        if (S.kq < bpt)
            for (int t = S.tq; t < ntok; t += S.tstride) {
                int kind;
                const int payload = pulled_slot<kPullRight>(L, t, S.kq, ntok, bpt, &kind);
                const int own = L.val[t * sv + S.kq];
                orow[t * width + 1 + 2 * bpt + S.kq] = own;
                orow[t * width + 1 + 3 * bpt + S.kq] = kind == 1 ? pad : (kind == 2 ? own : payload);
            }
    }
}

// ------------------------------------------------------------------------------------------
// Character matrix of the Llama front-end (inference/inference.py): chr_tokenize (56-67) + create_char_matrix (79-96).
// Sequence s owns the entries [seq_off[s], seq_off[s+1]) (one per BPE token, in order); entry e owns the code points
// codes[tok_off[e] .. tok_off[e+1]).  Row r of the matrix = entry seq_off[s] + r: its first max_char characters mapped to
// ids (ASCII as is, the tokenizer's leading-space marker -> 128, a code point equal to the BOS / EOS token id -> 129 /
// 130, anything else -> 131; a NEGATIVE code c is a literal id -c - 1, e.g. the [129] row get_tokens prepends, line 73),
// then ONE end-of-word id 130 if the row is not full, the rest (and every row past the entries) 2.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void char_matrix_kernel(const int32_t *__restrict__ codes, const int64_t *__restrict__ tok_off,
                                                               const int64_t *__restrict__ seq_off, int64_t n_seqs, int64_t seq_len, int max_char,
                                                               int32_t leading_space, int32_t bos_id, int32_t eos_id, int64_t *__restrict__ out) {
    const int64_t total = n_seqs * seq_len * max_char;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int k = (int)(i % max_char);
        const int64_t rr = i / max_char, r = rr % seq_len, s = rr / seq_len;
        const int64_t e = seq_off[s] + r;
        int64_t v = 2;                                            // "Initialize character matrices with 2: EOS/PAD", line 82
        if (e < seq_off[s + 1]) {
            const int64_t c0 = tok_off[e], len = tok_off[e + 1] - c0;
            if (k < len) {                                        // characters beyond max_char are dropped, lines 89-91
                const int32_t c = codes[c0 + k];
                if (c < 0) v = -(int64_t)c - 1;
                else if (c <= 127) v = c;
                else if (c == leading_space) v = 128;
                else if (c == bos_id) v = 129;
                else if (c == eos_id) v = 130;
                else v = 131;
            } else if (k == len) {
                v = 130;                                          // "ONE EOW TOKEN IS 130 THEN 2", lines 80, 94-95
            }
        }
        out[i] = v;
    }
}

int launch_char_matrix(const int32_t *codes, const int64_t *tok_off, const int64_t *seq_off, int64_t n_seqs, int64_t seq_len, int max_char,
                       int32_t leading_space, int32_t bos_id, int32_t eos_id, int64_t *out, hipStream_t stream) {
    const int64_t total = n_seqs * seq_len * max_char;
    if (total == 0) return MOT_OK;
    int64_t blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(char_matrix_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, codes, tok_off, seq_off, n_seqs, seq_len, max_char,
                       leading_space, bos_id, eos_id, out);
    return check_launch("char_matrix_kernel");
}

// ------------------------------------------------------------------------------------------ launchers
int pick_tile_tokens(int64_t n_rows, int64_t tokens_per_row, int bpt, bool with_ids) {
    // Largest tile that still gives >= ~2048 workgroups (8 per CU) and <= 64 KB of LDS.
    int tt = 256;
    while (tt > 64 && n_rows * ((tokens_per_row + tt - 1) / tt) < 2048) tt >>= 1;
    while (tt > 64 && tile_lds_bytes(tt, bpt, with_ids) > 64 * 1024) tt >>= 1;
    while (tt > 8 && tile_lds_bytes(tt, bpt, with_ids) > 64 * 1024) tt >>= 1;
    return tt;
}

int launch_tokens_to_bytes(const int32_t *tokens, int64_t n_tokens, const void *ttb, int elem, int64_t ttb_rows,
                           int bpt, int64_t *out, uint32_t *status, hipStream_t stream) {
    if (n_tokens == 0) return MOT_OK;
    const int bp2 = bpt <= 1 ? 1 : 1 << (32 - __builtin_clz(bpt - 1));
    const int64_t per_block = kThreads / bp2;
    int64_t blocks = (n_tokens + per_block - 1) / per_block;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (elem == 2)
        hipLaunchKernelGGL(tokens_to_bytes_kernel<int16_t>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, tokens,
                           n_tokens, (const int16_t *)ttb, ttb_rows, bpt, out, status);
    else
        hipLaunchKernelGGL(tokens_to_bytes_kernel<int32_t>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, tokens,
                           n_tokens, (const int32_t *)ttb, ttb_rows, bpt, out, status);
    return check_launch("tokens_to_bytes_kernel");
}

int launch_pull_bytes(const int64_t *in, int64_t *out, int64_t B, int64_t tokens_per_row, int bpt, int64_t pad,
                      int64_t eot, int dir, hipStream_t stream) {
    const int tt = pick_tile_tokens(B, tokens_per_row, bpt, false);
    const int64_t tiles_per_row = (tokens_per_row + tt - 1) / tt;
    const int64_t blocks = B * tiles_per_row;
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "pull_bytes: too many tiles (%lld)", (long long)blocks);
    const size_t lds = tile_lds_bytes(tt, bpt, false);
    if (dir == kPullLeft)
        hipLaunchKernelGGL(pull_bytes_kernel<kPullLeft>, dim3((unsigned)blocks), dim3(kThreads), lds, stream, in, out,
                           tokens_per_row, bpt, pad, eot, tt, (int)tiles_per_row);
    else
        hipLaunchKernelGGL(pull_bytes_kernel<kPullRight>, dim3((unsigned)blocks), dim3(kThreads), lds, stream, in, out,
                           tokens_per_row, bpt, pad, eot, tt, (int)tiles_per_row);
    return check_launch("pull_bytes_kernel");
}

int launch_create_batch(const int32_t *tokens, int64_t B, int64_t T, const void *ttb_left, const void *ttb_right,
                        int elem, int64_t ttb_rows, int bpt, int32_t pad, int32_t eot, int64_t *out, uint32_t *status,
                        hipStream_t stream) {
    const int tt = pick_tile_tokens(B, T, bpt, false);
    const int64_t tiles_per_row = (T + tt - 1) / tt;
    const int64_t blocks = B * tiles_per_row;
    if (blocks > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "create_batch: too many tiles");
    const size_t lds = tile_lds_bytes(tt, bpt, false);
    hipLaunchKernelGGL(create_batch_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, stream, tokens, T, ttb_left,
                       ttb_right, elem, ttb_rows, bpt, pad, eot, out, status, tt, (int)tiles_per_row);
    return check_launch("create_batch_kernel");
}

}  // namespace mot
