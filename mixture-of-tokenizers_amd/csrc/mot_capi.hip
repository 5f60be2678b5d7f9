// mot_capi.hip -- the extern "C" surface of libmot_hip.so (see include/mot.h): argument
// validation, error strings, dispatch to the kernel launchers.  No allocation, no host sync.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "mot_internal.hpp"

namespace mot {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(MOT_EHIP, "%s: %s", what, hipGetErrorString(e));
    return MOT_OK;
}

int ensure_max_dyn_lds(const void *kernel, std::atomic<uint64_t> &done, const char *name) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(MOT_EHIP, "hipGetDevice: %s", hipGetErrorString(e));
    const uint64_t bit = 1ull << (dev & 63);
    if (dev < 64 && (done.load(std::memory_order_acquire) & bit)) return MOT_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return set_error(MOT_EHIP, "hipFuncSetAttribute(%s): %s", name, hipGetErrorString(e));
    if (dev < 64) done.fetch_or(bit, std::memory_order_release);
    return MOT_OK;
}

static int check_bpt(const char *fn, int bpt) {
    if (bpt < 1 || bpt > MOT_MAX_BPT) return set_error(MOT_EUNSUPPORTED, "%s: bytes_per_token %d outside [1, %d]", fn, bpt, MOT_MAX_BPT);
    return MOT_OK;
}

static int validate_embed_mix(const MotEmbedMixDesc *d) {
    if (!d) return set_error(MOT_EINVAL, "embed_mix: null descriptor");
    if (d->struct_size != sizeof(MotEmbedMixDesc))
        return set_error(MOT_EINVAL, "embed_mix: struct_size %u != %zu (ABI mismatch)", d->struct_size, sizeof(MotEmbedMixDesc));
    if (d->dtype != MOT_F32 && d->dtype != MOT_BF16) return set_error(MOT_EINVAL, "embed_mix: bad dtype %d", d->dtype);
    if ((d->flags & ~(MOT_FLAG_LINEAR_ONE_LAUNCH | MOT_FLAG_MEAN_GENERIC | MOT_FLAG_BWD_DU_FP32 | MOT_FLAG_LINEAR_COMPOSED)) || d->reserved0)
        return set_error(MOT_EINVAL, "embed_mix: unknown flags 0x%x / reserved0 %u", d->flags, d->reserved0);
    if (d->n_rows < 0 || d->tokens_per_row < 0) return set_error(MOT_ESHAPE, "embed_mix: negative shape");
    if (d->mode < MOT_MIX_NOOP || d->mode > MOT_MIX_CONCAT_LINEAR) return set_error(MOT_EINVAL, "embed_mix: bad mode %d", d->mode);
    if (!d->tokens || !d->tok_table || !d->out) return set_error(MOT_EINVAL, "embed_mix: tokens/tok_table/out must be non-null");
    if (d->tok_rows <= 0 || d->tok_dim <= 0 || d->model_dim <= 0) return set_error(MOT_ESHAPE, "embed_mix: empty token table");
    if (d->tokens_per_row * (int64_t)(d->bpt > 0 ? d->bpt : 1) > 0x7fffffffLL)
        return set_error(MOT_EUNSUPPORTED, "embed_mix: T*bpt exceeds 2^31");
    if (d->mode != MOT_MIX_NOOP) {
        int rc = check_bpt("embed_mix", d->bpt);
        if (rc) return rc;
        if (!d->byte_table || d->byte_rows <= 0 || d->byte_dim <= 0) return set_error(MOT_EINVAL, "embed_mix: byte table missing");
        if (d->id_source == MOT_IDS_FROM_TTB) {
            if (!d->ttb || d->ttb_rows <= 0) return set_error(MOT_EINVAL, "embed_mix: ttb missing");
            if (d->ttb_elem_bytes != 2 && d->ttb_elem_bytes != 4) return set_error(MOT_EINVAL, "embed_mix: ttb_elem_bytes must be 2 or 4");
            if (d->pull_dir < MOT_PULL_NONE || d->pull_dir > MOT_PULL_RIGHT) return set_error(MOT_EINVAL, "embed_mix: bad pull_dir %d", d->pull_dir);
        } else if (d->id_source == MOT_IDS_GIVEN) {
            if (!d->ids_a) return set_error(MOT_EINVAL, "embed_mix: ids_a missing");
            if (d->out_ids_padded || d->out_ids_pulled) return set_error(MOT_EINVAL, "embed_mix: out_ids_* need MOT_IDS_FROM_TTB");
        } else {
            return set_error(MOT_EINVAL, "embed_mix: bad id_source %d", d->id_source);
        }
    }
    const bool dual = d->mode != MOT_MIX_NOOP && (d->id_source == MOT_IDS_FROM_TTB ? d->add_padded != 0 : d->ids_b != nullptr);
    switch (d->mode) {
        case MOT_MIX_NOOP:
            if (d->model_dim != d->tok_dim) return set_error(MOT_ESHAPE, "embed_mix noop: model_dim %d != tok_dim %d", d->model_dim, d->tok_dim);
            break;
        case MOT_MIX_SUM:
            if (d->bpt * d->byte_dim != d->tok_dim || d->model_dim != d->tok_dim)
                return set_error(MOT_ESHAPE, "embed_mix sum: need bpt*byte_dim == tok_dim == model_dim (got %d*%d, %d, %d)", d->bpt,
                                 d->byte_dim, d->tok_dim, d->model_dim);
            break;
        case MOT_MIX_MEAN:
            if (d->byte_dim != d->tok_dim || d->model_dim != d->tok_dim)
                return set_error(MOT_ESHAPE, "embed_mix mean: need byte_dim == tok_dim == model_dim");
            break;
        case MOT_MIX_CONCAT_LINEAR:
            if (!d->weight) return set_error(MOT_EINVAL, "embed_mix concat_linear: weight missing");
            break;
    }
    if (d->mode != MOT_MIX_CONCAT_LINEAR) {
        const int vec = d->dtype == MOT_BF16 ? 8 : 4;  // elements per 16-byte lane load
        if ((d->tok_dim % vec) || (d->mode == MOT_MIX_SUM && (d->byte_dim % vec)))
            return set_error(MOT_EUNSUPPORTED, "embed_mix: tok_dim %d / byte_dim %d must be multiples of %d elements (16 bytes)", d->tok_dim,
                             d->byte_dim, vec);
        if (d->tok_dim > 2048) return set_error(MOT_EUNSUPPORTED, "embed_mix: model_dim %d > 2048 is not built", d->tok_dim);
        if (dual && d->norm_byte)
            return set_error(MOT_EUNSUPPORTED, "embed_mix: norm_byte over two id tensors is only built for CONCAT_LINEAR");
    }
    return MOT_OK;
}

}  // namespace mot

using namespace mot;

extern "C" {

int mot_version(void) { return MOT_ABI_VERSION; }

const char *mot_last_error(void) { return g_err; }

const char *mot_build_info(void) {
    return "libmot_hip gfx950 wave64 fp32 | hipcc " __VERSION__ " | built " __DATE__;
}

int mot_tokens_to_bytes(const int32_t *tokens, int64_t n_tokens, const void *ttb, int ttb_elem_bytes, int64_t ttb_rows,
                        int bpt, int64_t *out, uint32_t *status, mot_stream_t stream) {
    if (n_tokens < 0) return set_error(MOT_ESHAPE, "tokens_to_bytes: n_tokens < 0");
    if (n_tokens == 0) return MOT_OK;
    if (!tokens || !ttb || !out) return set_error(MOT_EINVAL, "tokens_to_bytes: null pointer");
    if (ttb_elem_bytes != 2 && ttb_elem_bytes != 4) return set_error(MOT_EINVAL, "tokens_to_bytes: ttb_elem_bytes must be 2 or 4");
    if (ttb_rows <= 0) return set_error(MOT_ESHAPE, "tokens_to_bytes: empty table");
    int rc = check_bpt("tokens_to_bytes", bpt);
    if (rc) return rc;
    return launch_tokens_to_bytes(tokens, n_tokens, ttb, ttb_elem_bytes, ttb_rows, bpt, out, status, (hipStream_t)stream);
}

int mot_pull_bytes(const int64_t *in, int64_t *out, int64_t B, int64_t T, int bpt, int64_t pad_byte, int64_t eot_byte,
                   int dir, mot_stream_t stream) {
    if (B < 0 || T < 0) return set_error(MOT_ESHAPE, "pull_bytes: negative shape");
    if (T == 0 || B == 0) return MOT_OK;  // data_creation.py:82-83, 190
    int rc = check_bpt("pull_bytes", bpt);
    if (rc) return rc;
    if (T % bpt != 0) return set_error(MOT_ESHAPE, "pull_bytes: T must be divisible by bytes_per_token");  // :85, 192
    if (!in || !out) return set_error(MOT_EINVAL, "pull_bytes: null pointer");
    if (in == out) return set_error(MOT_EINVAL, "pull_bytes: in-place operation is not supported");
    if (dir != MOT_PULL_LEFT && dir != MOT_PULL_RIGHT) return set_error(MOT_EINVAL, "pull_bytes: dir must be MOT_PULL_LEFT or MOT_PULL_RIGHT");
    if (T > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "pull_bytes: row longer than 2^31 slots");
    return launch_pull_bytes(in, out, B, T / bpt, bpt, pad_byte, eot_byte, dir, (hipStream_t)stream);
}

int mot_create_batch(const int32_t *tokens, int64_t B, int64_t T, const void *ttb_left, const void *ttb_right,
                     int ttb_elem_bytes, int64_t ttb_rows, int bpt, int64_t pad_byte, int64_t eot_byte, int64_t *out,
                     uint32_t *status, mot_stream_t stream) {
    if (B < 0 || T < 0) return set_error(MOT_ESHAPE, "create_batch: negative shape");
    if (B == 0 || T == 0) return MOT_OK;
    if (!tokens || !ttb_left || !ttb_right || !out) return set_error(MOT_EINVAL, "create_batch: null pointer");
    if (ttb_elem_bytes != 2 && ttb_elem_bytes != 4) return set_error(MOT_EINVAL, "create_batch: ttb_elem_bytes must be 2 or 4");
    int rc = check_bpt("create_batch", bpt);
    if (rc) return rc;
    if (T * (int64_t)bpt > 0x7fffffffLL) return set_error(MOT_EUNSUPPORTED, "create_batch: row longer than 2^31 slots");
    return launch_create_batch(tokens, B, T, ttb_left, ttb_right, ttb_elem_bytes, ttb_rows, bpt, (int32_t)pad_byte,
                               (int32_t)eot_byte, out, status, (hipStream_t)stream);
}

int mot_char_matrix(const int32_t *codes, const int64_t *tok_offsets, const int64_t *seq_offsets, int64_t n_seqs, int64_t seq_len, int max_char,
                    int32_t leading_space, int32_t bos_token_id, int32_t eos_token_id, int64_t *out, mot_stream_t stream) {
    if (n_seqs < 0 || seq_len < 0) return set_error(MOT_ESHAPE, "char_matrix: negative shape");
    if (max_char < 1 || max_char > MOT_MAX_BPT) return set_error(MOT_EUNSUPPORTED, "char_matrix: max_char %d outside [1, %d]", max_char, MOT_MAX_BPT);
    if (n_seqs == 0 || seq_len == 0) return MOT_OK;
    if (!tok_offsets || !seq_offsets || !out) return set_error(MOT_EINVAL, "char_matrix: null pointer");   // codes may be NULL when every entry is empty
    return launch_char_matrix(codes, tok_offsets, seq_offsets, n_seqs, seq_len, max_char, leading_space, bos_token_id, eos_token_id, out,
                              (hipStream_t)stream);
}

int mot_gather_rows(const void *ids_a, const void *ids_b, int ids_elem_bytes, int64_t n, const void *table, int64_t rows,
                    int dim, int rms_norm, float eps, const float *scale, void *out, uint32_t *status, int dtype,
                    mot_stream_t stream) {
    if (dtype != MOT_F32 && dtype != MOT_BF16) return set_error(MOT_EINVAL, "gather_rows: bad dtype %d", dtype);
    if (n < 0) return set_error(MOT_ESHAPE, "gather_rows: n < 0");
    if (n == 0) return MOT_OK;
    if (!ids_a || !table || !out) return set_error(MOT_EINVAL, "gather_rows: null pointer");
    if (ids_elem_bytes != 4 && ids_elem_bytes != 8) return set_error(MOT_EINVAL, "gather_rows: ids_elem_bytes must be 4 or 8");
    if (rows <= 0 || dim <= 0) return set_error(MOT_ESHAPE, "gather_rows: empty table");
    return launch_gather_rows(ids_a, ids_b, ids_elem_bytes, n, table, rows, dim, rms_norm, eps, scale, out, status, dtype,
                              (hipStream_t)stream);
}

size_t mot_embed_mix_desc_size(void) { return sizeof(MotEmbedMixDesc); }

size_t mot_embed_mix_workspace_bytes(const MotEmbedMixDesc *desc) {
    if (!desc || desc->struct_size != sizeof(MotEmbedMixDesc)) return 0;
    return embed_mix_workspace_bytes(*desc);
}

size_t mot_token_order_ints(int64_t n_tokens, int64_t tok_rows) {
    if (n_tokens < 0 || tok_rows < 1) return 0;
    return group_positions_ws_ints(n_tokens, tok_rows);
}

int mot_token_order(const int32_t *tokens, int64_t n_tokens, int64_t tok_rows, int32_t *order, uint32_t *status, mot_stream_t stream) {
    if (n_tokens < 0 || tok_rows < 1 || n_tokens > 0x7fffffffLL) return set_error(MOT_EINVAL, "token_order: bad sizes");
    if (n_tokens == 0) return MOT_OK;
    if (!tokens || !order) return set_error(MOT_EINVAL, "token_order: null pointer");
    const int32_t *pos, *ids;
    return launch_group_positions(tokens, n_tokens, tok_rows, order, &pos, &ids, status, (hipStream_t)stream);
}

size_t mot_embed_mix_bwd_workspace_bytes(const MotEmbedMixDesc *desc) {
    if (!desc || desc->struct_size != sizeof(MotEmbedMixDesc)) return 0;
    return embed_mix_bwd_workspace_bytes(*desc);
}

int mot_embed_mix_bwd(const MotEmbedMixDesc *desc, const MotEmbedMixGrads *grads, mot_stream_t stream) {
    if (!grads || grads->struct_size != sizeof(MotEmbedMixGrads))
        return set_error(MOT_EINVAL, "embed_mix_bwd: grads struct missing or struct_size mismatch");
    MotEmbedMixDesc d;
    if (!desc) return set_error(MOT_EINVAL, "embed_mix_bwd: null descriptor");
    d = *desc;
    if (!d.out) d.out = (void *)grads->grad_out;  // the forward validator wants a non-null `out` (only CONCAT_LINEAR reads it)
    int rc = validate_embed_mix(&d);
    if (rc) return rc;
    if (!grads->grad_out) return set_error(MOT_EINVAL, "embed_mix_bwd: grad_out missing");
    if (!grads->d_tok_table) return set_error(MOT_EINVAL, "embed_mix_bwd: d_tok_table missing");
    if (d.mode == MOT_MIX_SUM && !grads->d_byte_table) return set_error(MOT_EINVAL, "embed_mix_bwd: d_byte_table missing");
    if (d.n_rows == 0 || d.tokens_per_row == 0) return MOT_OK;
    return launch_embed_mix_bwd(d, *grads, (hipStream_t)stream);
}

int mot_embed_mix_fwd(const MotEmbedMixDesc *desc, mot_stream_t stream) {
    int rc = validate_embed_mix(desc);
    if (rc) return rc;
    if (desc->n_rows == 0 || desc->tokens_per_row == 0) return MOT_OK;
    if (desc->mode == MOT_MIX_CONCAT_LINEAR)
        return desc->dtype == MOT_BF16 ? launch_embed_mix_linear_bf16(*desc, (hipStream_t)stream) : launch_embed_mix_linear(*desc, (hipStream_t)stream);
    return launch_embed_mix(*desc, (hipStream_t)stream);
}

size_t mot_char_swa_desc_size(void) { return sizeof(MotCharSwaDesc); }

static int validate_char_swa(const MotCharSwaDesc *d) {
    if (!d) return set_error(MOT_EINVAL, "char_swa: null descriptor");
    if (d->struct_size != sizeof(MotCharSwaDesc))
        return set_error(MOT_EINVAL, "char_swa: struct_size %u != %zu (ABI mismatch)", d->struct_size, sizeof(MotCharSwaDesc));
    if (d->dtype != MOT_F32) return set_error(MOT_EUNSUPPORTED, "char_swa: only MOT_F32 is built");
    if (d->n_rows < 0 || d->tokens_per_row < 0) return set_error(MOT_ESHAPE, "char_swa: negative shape");
    if (d->c_v < 1 || d->window < 1 || d->c_v * d->window > 64)
        return set_error(MOT_EUNSUPPORTED, "char_swa: window %d x %d characters must give 1..64 keys per query", d->window, d->c_v);
    if (d->head_dim != 64 && d->head_dim != 128) return set_error(MOT_EUNSUPPORTED, "char_swa: head_dim %d (64 and 128 are built)", d->head_dim);
    if (d->n_heads < 1 || d->dim < 4 || (d->dim & 3)) return set_error(MOT_ESHAPE, "char_swa: n_heads %d / dim %d (dim must be a multiple of 4)", d->n_heads, d->dim);
    if (d->version < MOT_SWA_NO_RESIDUAL || d->version > MOT_SWA_TWO_RESIDUAL) return set_error(MOT_EINVAL, "char_swa: bad version %d", d->version);
    if (d->matmul_dtype != MOT_F32 && d->matmul_dtype != MOT_BF16) return set_error(MOT_EINVAL, "char_swa: bad matmul_dtype %d", d->matmul_dtype);
    if (d->matmul_dtype == MOT_BF16 && ((d->dim & 7) || ((d->n_heads * d->head_dim) & 7)))
        return set_error(MOT_EUNSUPPORTED, "char_swa: matmul_dtype bf16 needs dim and heads * head_dim to be multiples of 8");
    if ((d->io_dtype != MOT_F32 && !(d->io_dtype == MOT_BF16 && d->matmul_dtype == MOT_BF16)) || d->reserved1)
        return set_error(MOT_EINVAL, "char_swa: io_dtype %d (a bf16 result goes with matmul_dtype bf16) / reserved1 %d", d->io_dtype, d->reserved1);
    if (!d->tokens || !d->char_ids || !d->tok_table || !d->char_table || !d->attn_norm_w || !d->char_norm_w || !d->wq || !d->wk || !d->wv || !d->wo || !d->out)
        return set_error(MOT_EINVAL, "char_swa: tokens/char_ids/tables/norm weights/projections/out must be non-null");
    if (d->tok_rows <= 0 || d->char_rows <= 0) return set_error(MOT_ESHAPE, "char_swa: empty table");
    if (d->version == MOT_SWA_TWO_RESIDUAL && d->dim > 2048) return set_error(MOT_EUNSUPPORTED, "char_swa: the two_residual mean needs dim <= 2048");
    return MOT_OK;
}

size_t mot_char_swa_workspace_bytes(const MotCharSwaDesc *desc) {
    if (!desc || desc->struct_size != sizeof(MotCharSwaDesc) || desc->n_heads < 1 || desc->head_dim < 1 || desc->dim < 1 || desc->char_rows < 1 ||
        desc->n_rows < 0 || desc->tokens_per_row < 0)
        return 0;
    return char_swa_workspace_bytes(*desc);
}

int mot_char_swa_fwd(const MotCharSwaDesc *desc, mot_stream_t stream) {
    int rc = validate_char_swa(desc);
    if (rc) return rc;
    if (desc->n_rows == 0 || desc->tokens_per_row == 0) return MOT_OK;
    return launch_char_swa(*desc, (hipStream_t)stream);
}

size_t mot_cross_attn_desc_size(void) { return sizeof(MotCrossAttnDesc); }

static int validate_cross_attn(const MotCrossAttnDesc *d) {
    if (!d) return set_error(MOT_EINVAL, "cross_attn: null descriptor");
    if (d->struct_size != sizeof(MotCrossAttnDesc))
        return set_error(MOT_EINVAL, "cross_attn: struct_size %u != %zu (ABI mismatch)", d->struct_size, sizeof(MotCrossAttnDesc));
    if (d->dtype != MOT_F32) return set_error(MOT_EUNSUPPORTED, "cross_attn: only MOT_F32 is built");
    if (d->n_tokens < 0) return set_error(MOT_ESHAPE, "cross_attn: negative shape");
    int rc = check_bpt("cross_attn", d->bpt);
    if (rc) return rc;
    if (d->n_tokens * (int64_t)d->bpt * (d->n_heads > 0 ? d->n_heads : 1) > 0x7fffffffLL)
        return set_error(MOT_EUNSUPPORTED, "cross_attn: T*bpt*heads exceeds 2^31");
    if (d->n_heads < 1 || d->dim < 128 || d->n_heads * 128 > d->dim)
        return set_error(MOT_ESHAPE, "cross_attn: n_heads %d x 128 does not fit dim %d (train_gpt.py:457-459: heads = dim // 128)", d->n_heads, d->dim);
    if ((d->dim & 3) || d->dim > 1024) return set_error(MOT_EUNSUPPORTED, "cross_attn: dim %d must be a multiple of 4 and <= 1024", d->dim);
    if (d->matmul_dtype != MOT_F32 && d->matmul_dtype != MOT_BF16) return set_error(MOT_EINVAL, "cross_attn: bad matmul_dtype %d", d->matmul_dtype);
    if (d->matmul_dtype == MOT_BF16 && (d->dim & 7)) return set_error(MOT_EUNSUPPORTED, "cross_attn: matmul_dtype bf16 needs dim %% 8 == 0 (got %d)", d->dim);
    if (d->io_dtype != MOT_F32 && !(d->io_dtype == MOT_BF16 && d->matmul_dtype == MOT_BF16))
        return set_error(MOT_EINVAL, "cross_attn: io_dtype %d (bf16 out / grad_out go with matmul_dtype bf16)", d->io_dtype);
    if (d->head_layout != MOT_HEADS_AS_VIEWED && d->head_layout != MOT_HEADS_PER_TOKEN)
        return set_error(MOT_EINVAL, "cross_attn: bad head_layout %d", d->head_layout);
    if (!d->tokens || !d->ids_a || !d->tok_table || !d->byte_table || !d->q_w || !d->kv_w || !d->proj_w || !d->lambda_factor || !d->out)
        return set_error(MOT_EINVAL, "cross_attn: tokens/ids_a/tables/weights/lambda_factor/out must be non-null");
    if (d->tok_rows <= 0 || d->byte_rows <= 0) return set_error(MOT_ESHAPE, "cross_attn: empty table");
    if (!d->cos_q || !d->sin_q || !d->cos_k || !d->sin_k) return set_error(MOT_EINVAL, "cross_attn: rotary buffers missing");
    if (d->rot_q_len < d->n_tokens || d->rot_k_len < d->n_tokens * (int64_t)d->bpt)   /* Rotary.forward's assert, line 200 */
        return set_error(MOT_ESHAPE, "cross_attn: rotary buffers hold %lld / %lld positions, need %lld / %lld", (long long)d->rot_q_len,
                         (long long)d->rot_k_len, (long long)d->n_tokens, (long long)(d->n_tokens * (int64_t)d->bpt));
    return MOT_OK;
}

size_t mot_cross_attn_workspace_bytes(const MotCrossAttnDesc *desc) {
    if (!desc || desc->struct_size != sizeof(MotCrossAttnDesc) || desc->n_heads < 1 || desc->bpt < 1 || desc->n_tokens < 0) return 0;
    return cross_attn_workspace_bytes(*desc);
}

int mot_cross_attn_fwd(const MotCrossAttnDesc *desc, mot_stream_t stream) {
    int rc = validate_cross_attn(desc);
    if (rc) return rc;
    if (desc->n_tokens == 0) return MOT_OK;
    return launch_cross_attn(*desc, (hipStream_t)stream);
}

size_t mot_cross_attn_bwd_workspace_bytes(const MotCrossAttnDesc *desc) {
    if (!desc || desc->struct_size != sizeof(MotCrossAttnDesc) || desc->n_heads < 1 || desc->bpt < 1 || desc->n_tokens < 0) return 0;
    return cross_attn_bwd_workspace_bytes(*desc);
}

int mot_cross_attn_bwd(const MotCrossAttnDesc *desc, const MotCrossAttnGrads *grads, mot_stream_t stream) {
    if (!grads || grads->struct_size != sizeof(MotCrossAttnGrads))
        return set_error(MOT_EINVAL, "cross_attn_bwd: grads struct missing or struct_size mismatch");
    if (!desc) return set_error(MOT_EINVAL, "cross_attn_bwd: null descriptor");
    MotCrossAttnDesc d = *desc;
    if (!d.out) d.out = (void *)grads->grad_out;   // the forward validator wants a non-null `out`; the backward never writes it
    int rc = validate_cross_attn(&d);
    if (rc) return rc;
    if (!grads->grad_out) return set_error(MOT_EINVAL, "cross_attn_bwd: grad_out missing");
    if (d.n_tokens == 0) return MOT_OK;
    return launch_cross_attn_bwd(d, *grads, (hipStream_t)stream);
}

}  // extern "C"
