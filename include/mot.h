/*
 * mot.h -- C ABI of libmot_hip.so: the MI355X (gfx950) implementation of the
 * mixture-of-tokenizers embedding front-end.
 *
 * The reference (snimu/mixture-of-tokenizers) exposes this path as Python functions and
 * nn.Module.forward() calls, not as an FFI; each entry point below names the reference
 * interface it replaces (paths relative to the reference checkout).  Host bindings live in
 * mixture-of-tokenizers_amd/_capi.py (ctypes); INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless a comment says "host"
 *   - the caller owns all buffers (inputs, outputs, workspace); the library allocates nothing and
 *     reads no environment variable; every choice between kernels is a field of the descriptor
 *     (MotEmbedMixDesc.flags).  The only thing it remembers between calls is, per device, which
 *     kernels already had their dynamic-LDS limit raised (hipFuncSetAttribute, idempotent), so
 *     calls are re-entrant and hipGraph-capturable
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and never synchronises the host
 *   - return value: MOT_OK or a negative MotStatus; mot_last_error() gives a thread-local
 *     message for the last failing call on this thread
 *   - byte tensors use the reference's layout: (B, T*bpt) row-major, token-major / slot-minor
 */
#ifndef MOT_H_
#define MOT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOT_ABI_VERSION 12
#define MOT_MAX_BPT 64 /* bytes (characters) per token; the reference uses 3, 8, 16, 18, 20, 32 */

typedef void *mot_stream_t; /* hipStream_t */

typedef enum MotStatus {
    MOT_OK = 0,
    MOT_EINVAL = -1,       /* null pointer / bad enum / bad struct_size                     */
    MOT_ESHAPE = -2,       /* shape the reference asserts on (e.g. T % bytes_per_token)     */
    MOT_EUNSUPPORTED = -3, /* valid request this build has no kernel for                    */
    MOT_EHIP = -4,         /* HIP runtime error (launch failure, no device)                 */
    MOT_EWORKSPACE = -5    /* workspace missing or smaller than mot_embed_mix_workspace_bytes */
} MotStatus;

/* Bits of the optional device status word (`status` arguments / MotEmbedMixDesc.status).
 * Kernels OR them in with atomics; out-of-range ids are clamped to row 0 so that a bad id can
 * never fault -- the host shim turns a non-zero word into the reference's IndexError. */
#define MOT_STATUS_TOKEN_OOR 1u /* token id outside [0, tok_rows) / [0, ttb_rows) */
#define MOT_STATUS_BYTE_OOR 2u  /* byte id outside [0, byte_rows)                 */

typedef enum MotPullDir {
    MOT_PULL_NONE = 0,
    MOT_PULL_LEFT = 1, /* pull_from_left : window ends at the token, right-aligned  */
    MOT_PULL_RIGHT = 2 /* pull_from_right: window starts at the token, left-aligned */
} MotPullDir;

typedef enum MotMixMode {
    MOT_MIX_NOOP = 0,         /* x = tok part            (train_gpt.py:342-348, 421-427)            */
    MOT_MIX_SUM = 1,          /* x = a + concat_k b_k    (modded-nanogpt/runs/71_*.py:227-230)      */
    MOT_MIX_MEAN = 2,         /* x = a + mean_k b_k      (inference/inference.py:266-267)           */
    MOT_MIX_CONCAT_LINEAR = 3 /* x = W.cat(a, b_*) + bias (train_gpt.py:430-443; model.py:256-268)  */
} MotMixMode;

typedef enum MotIdSource {
    MOT_IDS_NONE = 0,     /* MOT_MIX_NOOP                                                       */
    MOT_IDS_FROM_TTB = 1, /* fully fused: tokens -> ttb gather -> pull, byte ids never leave LDS */
    MOT_IDS_GIVEN = 2     /* ids_a (and ids_b) precomputed int64, as the reference loader emits  */
} MotIdSource;

/* Element type of the tables, the weight and `out` (scalars, workspace and gradients stay fp32).
 * MOT_BF16 is what the production loop runs (nn.Embedding -> bf16, train_gpt.py:1124-1126): rows are
 * widened to fp32 on load, all arithmetic is fp32, results are rounded once (nearest-even) on store;
 * eps defaults to torch.finfo(bfloat16).eps = 2^-7, as F.rms_norm(eps=None) does on bf16 inputs. */
typedef enum MotDType { MOT_F32 = 0, MOT_BF16 = 1 } MotDType;

int mot_version(void);                /* MOT_ABI_VERSION the library was built with */
const char *mot_last_error(void);     /* host string, thread-local, never NULL      */
const char *mot_build_info(void);     /* host string: arch, compiler, build flags   */

/*
 * Replaces tokens_to_bytes(tokens, emb)           scaled-pre-train/data_creation.py:61-67
 * (and the table built by make_embedding, :51-58, which the host keeps as an integer table).
 *   tokens  int32 [n_tokens]         ttb  int16|int32 [ttb_rows, bpt] (ttb_elem_bytes = 2|4)
 *   out     int64 [n_tokens*bpt]     status optional
 */
int mot_tokens_to_bytes(const int32_t *tokens, int64_t n_tokens, const void *ttb,
                        int ttb_elem_bytes, int64_t ttb_rows, int bpt, int64_t *out,
                        uint32_t *status, mot_stream_t stream);

/*
 * Replaces pull_from_left / pull_from_right(byte_tensor, bytes_per_token, pad_byte, eot_byte)
 *                                                 scaled-pre-train/data_creation.py:179-305 / 71-176
 *   in, out  int64 [B, T] with T = tokens_per_row*bpt; any int64 values are legal
 *   T == 0 is a no-op (data_creation.py:82-83,190); T % bpt != 0 -> MOT_ESHAPE (:85,192)
 *   in == out is NOT allowed.
 */
int mot_pull_bytes(const int64_t *in, int64_t *out, int64_t B, int64_t T, int bpt,
                   int64_t pad_byte, int64_t eot_byte, int dir /* MotPullDir */,
                   mot_stream_t stream);

/*
 * Replaces create_batch(tokens, bpt, pad, eot, ttb_right, ttb_left)
 *                                                 scaled-pre-train/data_creation.py:308-330
 *   out int64 [B, T, 1 + 4*bpt] = [token | left-padded | pulled-from-left | right-padded |
 *   pulled-from-right]; one fused launch, no intermediate tensors.
 */
int mot_create_batch(const int32_t *tokens, int64_t B, int64_t T, const void *ttb_left,
                     const void *ttb_right, int ttb_elem_bytes, int64_t ttb_rows, int bpt,
                     int64_t pad_byte, int64_t eot_byte, int64_t *out, uint32_t *status,
                     mot_stream_t stream);

/*
 * Replaces TokenMixByCharStreamingDataset.chr_tokenize + create_char_matrix     inference/inference.py:56-67, 79-96
 * (the producer of the character ids of BASELINE config 5; a host-side Python loop in the reference), for a batch of
 * sequences in one launch.
 *   seq_offsets int64 [n_seqs + 1]: entries (one per BPE token, in order) of sequence s are [seq_offsets[s], seq_offsets[s+1])
 *   tok_offsets int64 [n_entries + 1]: code points of entry e are codes[tok_offsets[e] .. tok_offsets[e+1])
 *   codes       int32: Unicode code points of the token strings (chr_tokenize's `ord(x)`); a NEGATIVE value c is a literal
 *               character id -c - 1 (the [129] row get_tokens prepends for BOS, line 73, is passed as -130)
 *   out         int64 [n_seqs, seq_len, max_char]: ids 0-127 ASCII, 128 the tokenizer's leading-space marker, 129 / 130 a
 *               code point equal to the BOS / EOS token id (the reference compares ord(x) with the TOKEN ids, lines 62-65),
 *               131 any other character; one end-of-word 130 after the last character of a row that is not full; 2 elsewhere
 *               (line 82) and on rows past the sequence's entries; characters beyond max_char are dropped (lines 89-91).
 */
int mot_char_matrix(const int32_t *codes, const int64_t *tok_offsets, const int64_t *seq_offsets, int64_t n_seqs,
                    int64_t seq_len, int max_char, int32_t leading_space, int32_t bos_token_id, int32_t eos_token_id,
                    int64_t *out, mot_stream_t stream);

/*
 * Replaces emb(ids) / norm(emb(ids)) / norm(emb(ids_a) + emb(ids_b)) when the caller wants the
 * tensors at the FlexibleEmbedding seam materialised
 *                                                 scaled-pre-train/train_gpt.py:342-379, 172-173
 *   ids int64|int32 [n] (ids_elem_bytes 8|4), ids_b optional; table f32|bf16 [rows, dim];
 *   out (same type) [n, dim]; rms_norm: x * rsqrt(mean(x^2) + eps), eps <= 0 -> FLT_EPSILON
 *   (F.rms_norm eps=None); scale optional device scalar.
 */
int mot_gather_rows(const void *ids_a, const void *ids_b, int ids_elem_bytes, int64_t n,
                    const void *table, int64_t rows, int dim, int rms_norm, float eps,
                    const float *scale, void *out, uint32_t *status, int dtype /* MotDType */,
                    mot_stream_t stream);

/*
 * The fused front-end.  Replaces, in ONE launch (plus a 458-row prologue when norm_byte is set):
 *   FlexibleEmbedding.forward + ByteMixin.forward  scaled-pre-train/train_gpt.py:327-379, 421-480
 *   (call site train_gpt.py:605-606), optionally with the loader's tokens_to_bytes + pull
 *   (train_gpt.py:686-728) folded in;
 *   GPT.wte / GPT.dte / digit_mixin                mathblations/model.py:256-268, 304-306, 323-327;
 *   embed_tokens / embed_bytes / mixin_bytes       modded-nanogpt/runs/71_*.py:227-230, 312-314
 *   (and the per-embedding-norm / lambda variants runs/71041_*.py:311-313, runs/71081_*.py:302-315).
 *
 * Per token n of row-major (B, T):
 *   a   = tok_table[tokens[n]];            if norm_tok:  a = rms_norm(a);    a *= *scale_tok
 *   b_k = byte_table[idsA[n,k]] (+ byte_table[idsB[n,k]]);
 *                                          if norm_byte: b_k = rms_norm(b_k); b_k *= *scale_byte
 *   x   = mix(a, b_0..b_{bpt-1}) per `mode`;  if norm_out: x = rms_norm(x)
 * where idsA/idsB come from `id_source`:
 *   MOT_IDS_FROM_TTB: padded = ttb[tokens[n]], pulled = pull(padded) along each row;
 *       idsA = pulled (or padded when pull_dir == NONE); idsB = padded iff add_padded
 *   MOT_IDS_GIVEN:    idsA = ids_a, idsB = ids_b (NULL = none)
 */
/* MotEmbedMixDesc.flags: kernel selection, resolved by the caller once (the Python shim reads its own switches at
 * import); results agree to the parity bar either way, the workspace size may differ -- size it with the same flags. */
#define MOT_FLAG_LINEAR_ONE_LAUNCH 1u /* CONCAT_LINEAR: the one-launch tile kernel instead of the composed kernels       */
#define MOT_FLAG_MEAN_GENERIC 2u      /* MEAN: the whole-row kernel even where the LDS column-slice kernel qualifies     */
#define MOT_FLAG_BWD_DU_FP32 4u       /* CONCAT_LINEAR backward, bf16: du = dy.W on the fp32 MFMA instead of the bf16 one */
#define MOT_FLAG_LINEAR_COMPOSED 8u   /* CONCAT_LINEAR, bf16: the multi-kernel path even where the one gather-GEMM qualifies */

typedef struct MotEmbedMixDesc {
    uint32_t struct_size; /* sizeof(MotEmbedMixDesc), checked */
    int32_t dtype;        /* MotDType of tables / weight / out */
    uint32_t flags;       /* MOT_FLAG_* */
    uint32_t reserved0;   /* must be 0 */

    /* problem */
    int64_t n_rows;         /* B */
    int64_t tokens_per_row; /* T (tokens, not byte slots) */
    int32_t bpt;            /* byte slots per token; 0 for MOT_MIX_NOOP */
    int32_t mode;           /* MotMixMode */

    /* ids */
    const int32_t *tokens; /* [B, T] */
    int32_t id_source;     /* MotIdSource */
    int32_t pull_dir;      /* MotPullDir          (FROM_TTB) */
    const void *ttb;       /* [ttb_rows, bpt]     (FROM_TTB) */
    int64_t ttb_rows;
    int32_t ttb_elem_bytes; /* 2 | 4 */
    int32_t add_padded;     /* FROM_TTB: idsB = unpulled row (train_gpt.py:371-379) */
    int32_t pad_byte, eot_byte;
    const int64_t *ids_a; /* [B, T*bpt]          (GIVEN) */
    const int64_t *ids_b; /* optional            (GIVEN) */

    /* tables */
    const void *tok_table; /* [tok_rows, tok_dim], contiguous */
    int64_t tok_rows;
    int32_t tok_dim;
    int32_t byte_dim;
    const void *byte_table; /* [byte_rows, byte_dim] */
    int64_t byte_rows;

    /* mixing */
    int32_t model_dim;   /* output columns; SUM/MEAN/NOOP require == tok_dim */
    int32_t bytes_first; /* CONCAT_LINEAR: 0 = [a, b_*] (train_gpt.py:443), 1 = [b_*, a] (model.py:267) */
    const void *weight;  /* CONCAT_LINEAR: [model_dim, tok_dim + bpt*byte_dim] row-major (nn.Linear) */
    const void *bias;    /* optional [model_dim] */
    int32_t norm_tok, norm_byte, norm_out;
    float eps;                /* <= 0 -> FLT_EPSILON (torch.finfo(float32).eps) */
    const float *scale_tok;   /* optional device scalar */
    const float *scale_byte;  /* optional device scalar */

    /* outputs */
    void *out;               /* [B, T, model_dim] */
    int64_t *out_ids_padded; /* optional [B, T*bpt]: what tokens_to_bytes would return (FROM_TTB) */
    int64_t *out_ids_pulled; /* optional [B, T*bpt]: what pull_from_* would return    (FROM_TTB) */
    int64_t *counters;       /* optional int64[4], atomically incremented: tokens, byte slots,
                                pads before the pull, pads after (runs/79_*.py:484-488)        */
    uint32_t *status;        /* optional device word, see MOT_STATUS_* */
    float *out_row_rnorm;    /* optional [B, T] fp32: rsqrt(mean(y^2)+eps) of every output row when norm_out
                                (CONCAT_LINEAR); saved by autograd so the backward need not redo the GEMM */

    /* scratch */
    void *workspace; /* >= mot_embed_mix_workspace_bytes(desc); may be NULL when that is 0 */
    size_t workspace_bytes;
} MotEmbedMixDesc;

/*
 * Backward of mot_embed_mix_fwd: replaces what autograd does for the modules above when the
 * training loop calls loss.backward() (scaled-pre-train/train_gpt.py:1319; mathblations/main.py:304).
 * `fwd` is the forward's descriptor with id_source == MOT_IDS_GIVEN (pass the byte ids the forward
 * returned through out_ids_*); `out`, `out_ids_*`, `counters` are ignored.  Gradients are ACCUMULATED
 * (+=) into the given buffers, so a parameter's .grad can be passed directly; NULL = not wanted.
 * Built: MOT_MIX_SUM, MOT_MIX_NOOP, MOT_MIX_CONCAT_LINEAR, and MOT_MIX_MEAN without an output norm (the residual of
 * inference.py:267 has none; its small character table makes the table gradient a dense product, mot_backward.hip; with
 * bf16 tables that product runs on operands widened slab by slab into the workspace).
 * With dtype == MOT_BF16 the tables, weight/bias, `out` and grad_out are bf16 as in the forward, while
 * every gradient buffer below stays FP32 (sums of thousands of terms are accumulated in fp32; the
 * caller rounds once when it needs a bf16 .grad, train_gpt.py:1124-1126).
 * CONCAT_LINEAR additionally needs `out` (the forward's x) and, when norm_out, `out_row_rnorm` of the
 * forward in `fwd`, and a workspace of mot_embed_mix_bwd_workspace_bytes(fwd).
 * Sums use float atomics: results are order-dependent in the last bits, like the reference's own
 * GPU embedding backward.
 */
typedef struct MotEmbedMixGrads {
    uint32_t struct_size;  /* sizeof(MotEmbedMixGrads) */
    uint32_t reserved;
    const void *grad_out;  /* [B, T, model_dim] upstream gradient dL/dx */
    void *d_tok_table;     /* [tok_rows, tok_dim]   fp32 */
    void *d_byte_table;    /* [byte_rows, byte_dim] fp32 */
    void *d_weight;        /* [model_dim, K]   fp32 (CONCAT_LINEAR) */
    void *d_bias;          /* [model_dim]      fp32 (CONCAT_LINEAR with bias; optional) */
    float *d_scale_tok;    /* scalar */
    float *d_scale_byte;   /* scalar */
    const int32_t *token_order; /* optional: what mot_token_order wrote for fwd->tokens (same n_tokens, tok_rows); NULL = the
                                   backward groups the positions itself, inside its workspace */
} MotEmbedMixGrads;

/*
 * The grouping of a batch's positions by token id that the table-gradient scatter walks (a counting sort: three small kernels,
 * 0.08 ms at 256 x 2048 tokens).  It depends on `tokens` only -- not on the tables, not on grad_out -- so a caller can produce it
 * once per batch, e.g. beside the forward (the Python autograd node runs it on a side stream while the forward streams), and hand
 * it to every backward over the same tokens through MotEmbedMixGrads.token_order (several embedding tables indexed by one
 * token tensor: modded-nanogpt/runs/71_*.py value embeddings; gradient accumulation does not re-sort either).
 *   tokens int32 [n_tokens]; ids outside [0, tok_rows) are flagged in `status` and grouped under row 0 (as the backward does);
 *   order  int32 [mot_token_order_ints(n_tokens, tok_rows)], opaque.  tok_rows < 2^21 - 1, n_tokens < 2^31.
 */
size_t mot_token_order_ints(int64_t n_tokens, int64_t tok_rows);
int mot_token_order(const int32_t *tokens, int64_t n_tokens, int64_t tok_rows, int32_t *order, uint32_t *status,
                    mot_stream_t stream);

size_t mot_embed_mix_bwd_workspace_bytes(const MotEmbedMixDesc *fwd /* host */);
int mot_embed_mix_bwd(const MotEmbedMixDesc *fwd /* host */, const MotEmbedMixGrads *grads /* host */,
                      mot_stream_t stream);

/*
 * CONCAT_LINEAR runs as several kernels inside one call (index kernels when the ids come from the ttb, a gather that
 * writes the concat operand into the workspace, a dense MFMA kernel, a row-norm pass); in bf16, with one id tensor, embedding
 * dims that are multiples of 8, a concat width that is a multiple of 32 and model_dim 256/512/768/1024, everything behind the
 * index kernels is ONE gather-GEMM kernel (MOT_FLAG_LINEAR_COMPOSED keeps the separate kernels).  MOT_FLAG_LINEAR_ONE_LAUNCH in
 * desc->flags selects the older one-launch tile kernel instead.  Same results to the parity bar; the workspace size differs,
 * so mot_embed_mix_workspace_bytes must see the same flags as the call.
 */
size_t mot_embed_mix_desc_size(void); /* sizeof(MotEmbedMixDesc) in this build, for bindings */
size_t mot_embed_mix_workspace_bytes(const MotEmbedMixDesc *desc /* host */);
int mot_embed_mix_fwd(const MotEmbedMixDesc *desc /* host */, mot_stream_t stream);

/*
 * Cross-attention byte mixin, forward: replaces ByteMixinCrossAttn.forward on FlexibleEmbedding's outputs
 * (scaled-pre-train/train_gpt.py:446-464 -> CrossAttention.forward 271-300; embeddings 342-379).
 * Each token attends to its own `bpt` byte embeddings: q = q_w xq, (k, v) = kv_w xkv, per-head rms-norm of q and
 * k, RoPE with the position in each one's own sequence, v *= lambda, softmax(q.k / sqrt(hd)) over the bpt
 * keys, out = proj_w y.  The reference asserts batch 1 (line 275): tokens is one row of n_tokens.
 * head_dim is 128 (line 459); dim = token_dim = byte_dim = model_dim (line 449).  fp32.
 * head_layout MOT_HEADS_AS_VIEWED reproduces lines 283-284 (k, v are reshaped, not transposed, into
 * (H, T, bpt, hd)); MOT_HEADS_PER_TOKEN is the einops expression in the comment of those lines.
 * cos/sin are the Rotary buffers of the module (lines 190-197), fp32 [len, 64], built by the caller.
 */
typedef enum MotHeadLayout { MOT_HEADS_AS_VIEWED = 0, MOT_HEADS_PER_TOKEN = 1 } MotHeadLayout;

typedef struct MotCrossAttnDesc {
    uint32_t struct_size;     /* sizeof(MotCrossAttnDesc) */
    int32_t dtype;            /* MOT_F32 */
    int64_t n_tokens;         /* Tq; Tkv = n_tokens * bpt */
    int32_t bpt;              /* chars_per_token, line 282 */
    int32_t n_heads;          /* hdim = n_heads * 128 */
    int32_t head_layout;      /* MotHeadLayout */
    int32_t dim;              /* columns of both tables */
    const int32_t *tokens;    /* [n_tokens] */
    const int64_t *ids_a;     /* [n_tokens * bpt] byte ids */
    const int64_t *ids_b;     /* optional second id tensor: xkv = norm?(E[a] + E[b]), line 378 */
    const void *tok_table;    /* [tok_rows, dim] */
    int64_t tok_rows;
    const void *byte_table;   /* [byte_rows, dim] */
    int64_t byte_rows;
    int32_t norm_tok;         /* FlexibleEmbedding applies norm() to both (lines 367-378) */
    int32_t norm_byte;
    const void *q_w;          /* [hdim, dim]    CrossAttention.q_w */
    const void *kv_w;         /* [2, hdim, dim] CrossAttention.kv_w */
    const void *proj_w;       /* [dim, hdim]    CrossAttention.c_proj.weight */
    const float *lambda_factor; /* device scalar */
    const float *cos_q, *sin_q; /* [>= n_tokens, 64] */
    const float *cos_k, *sin_k; /* [>= n_tokens * bpt, 64] */
    int64_t rot_q_len, rot_k_len; /* rows of the two pairs of buffers (line 200 asserts they suffice) */
    float eps;                /* 0 = finfo(float32).eps */
    int32_t kv_tables_ready;  /* forward, one id tensor: 1 = `kv_tables` already holds this call's tables (skip building them) */
    void *out;                /* [n_tokens, dim] */
    uint32_t *status;         /* optional, as in MotEmbedMixDesc */
    void *workspace;          /* mot_cross_attn_workspace_bytes(desc) */
    size_t workspace_bytes;
    /* optional, forward with one id tensor: caller-kept buffer of 2 * byte_rows * n_heads * 128 floats for the per-byte-row
     * key / value tables.  They depend on byte_table, kv_w and lambda_factor only, so an inference loop builds them once:
     * pass the buffer with kv_tables_ready = 0 after those change (the call fills it), = 1 otherwise. */
    void *kv_tables;
    /* optional: buffer of 2 * n_tokens * n_heads * 128 floats.  The forward leaves the projected queries and the attention
     * output there; the backward, given the same buffer, reads them instead of recomputing that part of the forward. */
    void *saved_qy;
    /* MOT_F32 (0) or MOT_BF16: where the products over the tokens run (q = W_q xq, out = c_proj y; backward: dW_p, dy, dW_q, dxq).
     * MOT_BF16 = the bf16 MFMA with fp32 accumulation: their row operands (xq, y, grad_out, dq) are rounded to bf16 first and the
     * weights are taken as bf16 -- the values those operands have in the reference's production cast (CastedLinear and
     * `self.q_w.type_as(x)`, train_gpt.py:185-186, 277-278; x bf16 since 1124-1126).  With one id tensor and bpt <= 16 the attention
     * kernels also read norm(k) and lambda * v from bf16 copies of the two per-row tables (bf16 tensors in that cast, lines 278, 280);
     * the attention arithmetic, the tables themselves and every gradient stay fp32.  Needs dim % 8 == 0.  Workspace sizes depend on it. */
    int32_t matmul_dtype;
    /* MOT_F32 (0) or, with matmul_dtype == MOT_BF16, MOT_BF16: the element type of `out` (forward) and of MotCrossAttnGrads.grad_out
     * (backward).  bf16 is what the reference's module returns and receives in the production cast; the last product then writes
     * bf16 itself (one rounding of the fp32 sums, as a caller's cast of the fp32 result would do) and the backward's bf16 products
     * take grad_out as it is -- a widening and a narrowing pass over [T, dim] less on each side of the boundary. */
    int32_t io_dtype;
    /* optional, with matmul_dtype == MOT_BF16: the caller's bf16 token table [tok_rows, dim] whose widened copy `tok_table` is.
     * The normalised token rows -- the row operand of W_q and of dW_q -- are then gathered in bf16 directly (the same values: norm
     * in fp32, one rounding), without the fp32 detour. */
    const void *tok_table_bf16;
} MotCrossAttnDesc;

/*
 * Backward of mot_cross_attn_fwd (loss.backward() through the modules above, train_gpt.py:1319).  `fwd` is the
 * forward's descriptor (`out` is ignored); everything is recomputed from the inputs.  Gradients are ACCUMULATED (+=)
 * in fp32 into the given buffers.  With two id tensors (ids_b, the add_padded_and_pulled embedding of train_gpt.py:364-372)
 * the key / value rows are per kv position: the workspace then grows with n_tokens * bpt rows of 2 * heads * 128 floats.
 */
typedef struct MotCrossAttnGrads {
    uint32_t struct_size;  /* sizeof(MotCrossAttnGrads) */
    uint32_t reserved;
    const void *grad_out;  /* [n_tokens, dim] */
    void *d_tok_table;     /* [tok_rows, dim]  */
    void *d_byte_table;    /* [byte_rows, dim] */
    void *d_q_w;           /* [hdim, dim]      */
    void *d_kv_w;          /* [2, hdim, dim]   */
    void *d_proj_w;        /* [dim, hdim]      */
    float *d_lambda;       /* scalar           */
} MotCrossAttnGrads;

size_t mot_cross_attn_bwd_workspace_bytes(const MotCrossAttnDesc *fwd /* host */);
int mot_cross_attn_bwd(const MotCrossAttnDesc *fwd /* host */, const MotCrossAttnGrads *grads /* host */, mot_stream_t stream);

size_t mot_cross_attn_desc_size(void);
size_t mot_cross_attn_workspace_bytes(const MotCrossAttnDesc *desc /* host */);
int mot_cross_attn_fwd(const MotCrossAttnDesc *desc /* host */, mot_stream_t stream);

/*
 * Sliding-window token <- character attention of the Llama character mixer, forward: replaces
 *   TokenMixByCharBMM.forward                      inference/inference.py:146-224  (swa_transform 174-179)
 *   the residuals of TokenMixByCharBMMBlock.forward                       :260-267 (the SwiGLU feed-forward after them, 269, is
 *   a plain MLP and stays with the caller) on top of the gathers of CustomLlamaModel.forward, :323-327.
 * Per token t of row-major (B, T):  xn = RMSNorm_a(E_tok[t]), cn = RMSNorm_c(E_char[c]) (x * rsqrt(mean(x^2) + norm_eps) * weight,
 * lines 126-132), q = wq xn, keys / values = wk cn / wv cn of the c_v characters of each of the tokens t-window+1 .. t of the
 * token's batch row (zero vectors in front of the row: they keep their place in the softmax with score 0), rotary embedding of
 * q and of every key at the query's position (see mot_swa.hip: it cancels), p = softmax(q . k / sqrt(head_dim)), y = sum p v,
 * out = wo y  (+ toks  |  + lambda_tok toks + lambda_char mean_c chars).   fp32; head_dim 64 or 128; window * c_v <= 64.
 * The reference cannot be imported offline (hub login at import): PARITY UNPINNED, checked against a hand-written float64 restatement.
 */
typedef enum MotSwaVersion { MOT_SWA_NO_RESIDUAL = 0, MOT_SWA_ONE_RESIDUAL = 1, MOT_SWA_TWO_RESIDUAL = 2 } MotSwaVersion;

typedef struct MotCharSwaDesc {
    uint32_t struct_size;       /* sizeof(MotCharSwaDesc) */
    int32_t dtype;              /* MOT_F32 */
    int64_t n_rows;             /* B */
    int64_t tokens_per_row;     /* T */
    int32_t c_v;                /* characters per token (max_char, 8) */
    int32_t window;             /* TokenMixByCharBMM.window_size (8) */
    int32_t n_heads, head_dim;  /* ModelArgs.n_heads, head_dim (32 x 64 for Llama-3.2-1B) */
    int32_t dim;                /* ModelArgs.dim: columns of both tables and of out */
    int32_t version;            /* MotSwaVersion */
    const int32_t *tokens;      /* [B, T] */
    const int64_t *char_ids;    /* [B, T, c_v] */
    const void *tok_table;      /* [tok_rows, dim]  model.embed_tokens.weight */
    int64_t tok_rows;
    const void *char_table;     /* [char_rows, dim] char_embeddings.weight */
    int32_t char_rows;          /* 132 */
    float norm_eps;             /* <= 0 -> 1e-5 (ModelArgs.norm_eps) */
    const void *attn_norm_w;    /* [dim] attention_norm.weight */
    const void *char_norm_w;    /* [dim] char_norm.weight */
    const void *wq, *wk, *wv;   /* [n_heads * head_dim, dim] nn.Linear weights, no bias */
    const void *wo;             /* [dim, n_heads * head_dim] */
    const float *lambda_tok;    /* device scalars (two_residual); NULL = 1 */
    const float *lambda_char;
    void *out;                  /* [B, T, dim] */
    uint32_t *status;           /* optional, as in MotEmbedMixDesc */
    void *workspace;            /* mot_char_swa_workspace_bytes(desc) */
    size_t workspace_bytes;
    /* MOT_F32 (0) or MOT_BF16: where the two products over the tokens run (xq = wq xn, h = wo y).  MOT_BF16 = the bf16 MFMA with
     * fp32 accumulation, xn and y rounded to bf16 first and the weights taken as bf16 (for callers whose tables and weights
     * hold bf16 values; the reference script runs in float32), and the projected queries, keys and values kept as the bf16 tensors
     * they are in a bf16 cast of the module (fp32 softmax and sums); needs dim % 8 == 0 and (heads * head_dim) % 8 == 0. */
    int32_t matmul_dtype;
    /* 1 = `kv_tables` already holds this call's per-character key / value tables (skip building them) */
    int32_t kv_tables_ready;
    /* optional: caller-kept buffer of 2 * char_rows * n_heads * head_dim floats for the projected key / value rows of the character
     * table.  They depend on char_table, char_norm_w, wk and wv only, so an inference loop (what the reference file is) builds them
     * once: pass the buffer with kv_tables_ready = 0 after those change (the call fills it), = 1 otherwise. */
    void *kv_tables;
    /* MOT_F32 (0) or, with matmul_dtype == MOT_BF16, MOT_BF16: the element type of `out`.  bf16 is what the module returns in a bf16
     * cast: the last product then adds the fp32 residuals and writes bf16 itself (one rounding, as a caller's cast of the fp32 result
     * would do; a pass over [B, T, dim] less on each side of the boundary). */
    int32_t io_dtype;
    int32_t reserved1;   /* must be 0 */
} MotCharSwaDesc;

size_t mot_char_swa_desc_size(void);
size_t mot_char_swa_workspace_bytes(const MotCharSwaDesc *desc /* host */);
int mot_char_swa_fwd(const MotCharSwaDesc *desc /* host */, mot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOT_H_ */
