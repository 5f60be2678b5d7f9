/*
 * mot_oracle.c -- CPU restatement of the mixture-of-tokenizers embedding front-end.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path in
 * mixture-of-tokenizers_amd/csrc.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product never links, imports or
 * falls back to anything in oracle/.
 *
 * Every function restates, in plain C loops, what the reference computes with
 * PyTorch ops; citations are to files under /root/reference (read-only, never
 * copied).  Pinning: tests/test_oracle_golden.py checks this file against
 * the .npz files under tests/golden, which oracle/gen_golden.py produced by importing and
 * running the reference itself (torch 2.10 CPU, eager).
 *
 * Integer work is bit-exact.  Float work exists in two instantiations:
 *   *_f32 : float storage/elementwise arithmetic, reductions accumulated in
 *           double then rounded to float (an order-independent statement of the
 *           fp32 reference; within 1 ulp-ish of torch's vectorised sums)
 *   *_f64 : everything in double (the "exact" oracle of SURVEY 8c)
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MOT_O_OK 0
#define MOT_O_EBADSHAPE (-1)
#define MOT_O_ENOMEM (-2)
#define MOT_O_ERANGE (-3)

/* ------------------------------------------------------------------------ */
/* Integer path                                                             */
/* ------------------------------------------------------------------------ */

/*
 * tokens_to_bytes -- scaled-pre-train/data_creation.py:61-67.
 * The reference keeps the token->byte table as an fp32 nn.Embedding
 * (make_embedding, data_creation.py:51-58), gathers rows and casts with
 * .to(torch.int64), i.e. truncation toward zero.  `table` is that fp32 weight,
 * row-major (vocab, bpt).  Output is token-major, slot-minor: (n_tokens*bpt).
 */
int oracle_tokens_to_bytes(const int32_t *tokens, const float *table,
                           int64_t vocab, int64_t n_tokens, int bpt,
                           int64_t *out)
{
    for (int64_t n = 0; n < n_tokens; ++n) {
        int64_t t = tokens[n];
        if (t < 0 || t >= vocab) return MOT_O_ERANGE; /* nn.Embedding raises IndexError */
        for (int k = 0; k < bpt; ++k)
            out[n * bpt + k] = (int64_t)table[t * bpt + k]; /* C cast == trunc, as torch */
    }
    return MOT_O_OK;
}

/* Shared preprocessing of one batch row -- data_creation.py:90-105 / 196-210.
 * flat   : the row's non-pad bytes in order           (flat_valid_bytes, 131-132 / 248-249)
 * cum    : cum[t] = #valid bytes in tokens [0,t)       (cum_valid_bytes, 99-102 / 204-207)
 * is_eot : every slot of the token == eot_byte         (94 / 200)
 */
static void row_prepare(const int64_t *row, int64_t Tr, int bpt, int64_t pad,
                        int64_t eot, int64_t *flat, int64_t *cum, uint8_t *is_eot)
{
    int64_t nflat = 0;
    cum[0] = 0;
    for (int64_t t = 0; t < Tr; ++t) {
        int all_eot = 1;
        for (int k = 0; k < bpt; ++k) {
            int64_t v = row[t * bpt + k];
            if (v != pad) flat[nflat++] = v;
            if (v != eot) all_eot = 0;
        }
        is_eot[t] = (uint8_t)all_eot;
        cum[t + 1] = nflat;
    }
}

/*
 * pull_from_left -- data_creation.py:179-305.
 * For token t: prev = last EOT token index <= t (or -1)            (212-223)
 *   start = prev==-1 ? 0 : cum[prev+1]                              (228-232)
 *   avail = max(cum[t+1] - start, 0); use = min(avail, bpt)         (235-242)
 *   the `use` stream bytes ending at cum[t+1] go to slots
 *   [bpt-use, bpt); the rest is pad                                 (245-295)
 *   EOT tokens keep their original slots                            (298-302)
 * T == 0 returns the input unchanged (190); T % bpt != 0 is an assert (192).
 */
int oracle_pull_from_left(const int64_t *in, int64_t B, int64_t T, int bpt,
                          int64_t pad, int64_t eot, int64_t *out)
{
    if (T == 0) return MOT_O_OK;
    if (bpt <= 0 || T % bpt != 0) return MOT_O_EBADSHAPE;
    int64_t Tr = T / bpt;
    int rc = MOT_O_OK;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const int64_t *row = in + b * T;
        int64_t *orow = out + b * T;
        int64_t *flat = (int64_t *)malloc(sizeof(int64_t) * (size_t)(T + 1));
        int64_t *cum = (int64_t *)malloc(sizeof(int64_t) * (size_t)(Tr + 1));
        uint8_t *is_eot = (uint8_t *)malloc((size_t)Tr + 1);
        if (!flat || !cum || !is_eot) { rc = MOT_O_ENOMEM; free(flat); free(cum); free(is_eot); continue; }
        row_prepare(row, Tr, bpt, pad, eot, flat, cum, is_eot);
        int64_t prev = -1;
        for (int64_t t = 0; t < Tr; ++t) {
            if (is_eot[t]) prev = t;
            if (is_eot[t]) {
                memcpy(orow + t * bpt, row + t * bpt, sizeof(int64_t) * (size_t)bpt);
                continue;
            }
            int64_t start = prev < 0 ? 0 : cum[prev + 1];
            int64_t end = cum[t + 1];
            int64_t avail = end - start; if (avail < 0) avail = 0;
            int64_t use = avail < bpt ? avail : bpt;
            int64_t g0 = end - use;
            for (int k = 0; k < bpt; ++k) orow[t * bpt + k] = pad;
            for (int64_t k = 0; k < use; ++k) orow[t * bpt + (bpt - use + k)] = flat[g0 + k];
        }
        free(flat); free(cum); free(is_eot);
    }
    return rc;
}

/*
 * pull_from_right -- data_creation.py:71-176.
 * For token t: next = first EOT token index >= t (or Tr)            (107-118)
 *   avail = cum[next] - cum[t]; use = clamp(min(avail, bpt), 0)     (123-128)
 *   the `use` stream bytes starting at cum[t] go to slots [0, use);
 *   the rest is pad                                                 (131-165)
 *   EOT tokens keep their original slots                            (169-173)
 */
int oracle_pull_from_right(const int64_t *in, int64_t B, int64_t T, int bpt,
                           int64_t pad, int64_t eot, int64_t *out)
{
    if (T == 0) return MOT_O_OK;
    if (bpt <= 0 || T % bpt != 0) return MOT_O_EBADSHAPE;
    int64_t Tr = T / bpt;
    int rc = MOT_O_OK;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const int64_t *row = in + b * T;
        int64_t *orow = out + b * T;
        int64_t *flat = (int64_t *)malloc(sizeof(int64_t) * (size_t)(T + 1));
        int64_t *cum = (int64_t *)malloc(sizeof(int64_t) * (size_t)(Tr + 1));
        uint8_t *is_eot = (uint8_t *)malloc((size_t)Tr + 1);
        if (!flat || !cum || !is_eot) { rc = MOT_O_ENOMEM; free(flat); free(cum); free(is_eot); continue; }
        row_prepare(row, Tr, bpt, pad, eot, flat, cum, is_eot);
        int64_t next = Tr;
        for (int64_t t = Tr - 1; t >= 0; --t) {
            if (is_eot[t]) next = t;
            if (is_eot[t]) {
                memcpy(orow + t * bpt, row + t * bpt, sizeof(int64_t) * (size_t)bpt);
                continue;
            }
            int64_t start = cum[t];
            int64_t avail = cum[next] - start;
            int64_t use = avail < bpt ? avail : bpt; if (use < 0) use = 0;
            for (int64_t k = 0; k < use; ++k) orow[t * bpt + k] = flat[start + k];
            for (int64_t k = use; k < bpt; ++k) orow[t * bpt + k] = pad;
        }
        free(flat); free(cum); free(is_eot);
    }
    return rc;
}

/*
 * create_batch -- data_creation.py:308-330: cat([tokens, left_pad, pulled_left,
 * right_pad, pulled_right], -1) -> (B, T, 1 + 4*bpt) int64.
 */
int oracle_create_batch(const int32_t *tokens, const float *table_left,
                        const float *table_right, int64_t vocab, int64_t B,
                        int64_t T, int bpt, int64_t pad, int64_t eot, int64_t *out)
{
    int64_t n = B * T, Tb = T * bpt;
    int64_t *lp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(4 * n * bpt + 4));
    if (!lp) return MOT_O_ENOMEM;
    int64_t *lpull = lp + n * bpt, *rp = lpull + n * bpt, *rpull = rp + n * bpt;
    int rc = oracle_tokens_to_bytes(tokens, table_left, vocab, n, bpt, lp);
    if (!rc) rc = oracle_tokens_to_bytes(tokens, table_right, vocab, n, bpt, rp);
    if (!rc) rc = oracle_pull_from_left(lp, B, Tb, bpt, pad, eot, lpull);
    if (!rc) rc = oracle_pull_from_right(rp, B, Tb, bpt, pad, eot, rpull);
    if (!rc) {
        int64_t w = 1 + 4 * (int64_t)bpt;
        for (int64_t i = 0; i < n; ++i) {
            int64_t *o = out + i * w;
            o[0] = tokens[i];
            memcpy(o + 1, lp + i * bpt, sizeof(int64_t) * (size_t)bpt);
            memcpy(o + 1 + bpt, lpull + i * bpt, sizeof(int64_t) * (size_t)bpt);
            memcpy(o + 1 + 2 * bpt, rp + i * bpt, sizeof(int64_t) * (size_t)bpt);
            memcpy(o + 1 + 3 * bpt, rpull + i * bpt, sizeof(int64_t) * (size_t)bpt);
        }
    }
    free(lp);
    return rc;
}

/*
 * Byte-slot statistics of modded-nanogpt/runs/79_mot-in_toks-valemb.py:484-488:
 * stats = {total slots, pads before the pull, pads after the pull}
 * (pulled = before - after, blocked = after).
 */
void oracle_byte_stats(const int64_t *padded, const int64_t *pulled, int64_t n,
                       int64_t pad, int64_t *stats)
{
    int64_t before = 0, after = 0;
    for (int64_t i = 0; i < n; ++i) { before += padded[i] == pad; after += pulled[i] == pad; }
    stats[0] = n; stats[1] = before; stats[2] = after;
}

/*
 * mathblations GenerateEquations.tokens_to_digits -- mathblations/data.py:92-109.
 * lf = max_digits_per_token slots, fill 13, right-aligned decimal digits;
 * op token -> 10, eq token -> 11, pad token -> 12 in the LAST slot.
 * op = 10^lf, eq = op + 1, padtok = op + 2                       (data.py:57-62)
 */
int oracle_tokens_to_digits(const int64_t *tokens, int64_t n, int lf, int64_t *out)
{
    int64_t op = 1; for (int i = 0; i < lf; ++i) op *= 10;
    for (int64_t i = 0; i < n; ++i) {
        int64_t t = tokens[i];
        int64_t *o = out + i * lf;
        for (int k = 0; k < lf; ++k) o[k] = 13;
        if (t == op) o[lf - 1] = 10;
        else if (t == op + 1) o[lf - 1] = 11;
        else if (t == op + 2) o[lf - 1] = 12;
        else {
            if (t < 0 || t > op + 2) return MOT_O_ERANGE;
            int k = lf - 1;
            do { o[k--] = t % 10; t /= 10; } while (t > 0 && k >= 0);
        }
    }
    return MOT_O_OK;
}

/*
 * Loader slice + shift -- scaled-pre-train/train_gpt.py:795-805 (rank slice) and
 * 686-764 (the _create_data_from_toks_* shift): rank r of W takes
 * data[pos + r*L : pos + (r+1)*L] with L = batch*(seq+1)/W, viewed (-1, seq+1);
 * inputs drop the last token, targets drop the first.
 */
int oracle_rank_slice_shift(const int32_t *data, int64_t pos, int64_t batch,
                            int64_t seq, int rank, int world,
                            int32_t *toks_in, int32_t *targets)
{
    if (world <= 0 || batch % world != 0) return MOT_O_EBADSHAPE;
    int64_t L = batch * (seq + 1) / world, rows = L / (seq + 1);
    const int32_t *src = data + pos + (int64_t)rank * L;
    for (int64_t b = 0; b < rows; ++b)
        for (int64_t t = 0; t < seq; ++t) {
            toks_in[b * seq + t] = src[b * (seq + 1) + t];
            targets[b * seq + t] = src[b * (seq + 1) + t + 1];
        }
    return MOT_O_OK;
}

/* ------------------------------------------------------------------------ */
/* Float path, instantiated twice from mot_oracle_float.inc                 */
/* ------------------------------------------------------------------------ */

/* F.rms_norm(eps=None) takes eps from the INPUT dtype; to restate the bf16 path (tables held as
 * bf16-representable floats) the tests set eps = torch.finfo(bfloat16).eps = 2^-7 here; 0 = per-type default. */
static double g_eps_override = 0.0;
void oracle_set_eps(double eps) { g_eps_override = eps; }

/* bf16 matmul path (train_gpt.py:185-186 + 1124-1126): the normalised/scaled segments are bf16 TENSORS
 * when they enter F.linear, so the concat operand is rounded to bf16 (nearest-even) element by element. */
static int g_round_seg_bf16 = 0;
void oracle_set_round_segments_bf16(int on) { g_round_seg_bf16 = on; }
/* Rotary: scaled-pre-train casts the head to float32 first (train_gpt.py:202); mathblations' apply_rotary_emb
 * (model.py:51-58) keeps the head's dtype.  Only the float64 instantiation can tell the two apart. */
static int g_rotary_f32_cast = 1;
void oracle_set_rotary_f32_cast(int on) { g_rotary_f32_cast = on; }
/* Cross-attention with bf16 tables: norm(k) and lambda * v are bf16 TENSORS in the reference's bf16 cast (train_gpt.py:278, 280);
 * on = round them to bf16 where they are formed (the keys before their rotation), as the HIP path keeps them with bf16 products. */
static int g_round_kv_bf16 = 0;
void oracle_set_round_kv_bf16(int on) { g_round_kv_bf16 = on; }
static double bf16_rne(double x) {
    float f = (float)x; uint32_t u; memcpy(&u, &f, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    memcpy(&f, &u, 4); return (double)f;
}

/* mixing modes (values shared with nothing in the product on purpose) */
#define O_MODE_NOOP 0          /* x = tok part only                                  */
#define O_MODE_SUM 1           /* x = a + concat_k b_k   (runs/71*.py:227-230)       */
#define O_MODE_MEAN 2          /* x = a + mean_k b_k     (inference.py:266-267)      */
#define O_MODE_CONCAT_LINEAR 3 /* x = W.cat(..) + bias   (train_gpt.py:439-443; model.py:263-268) */

#define REAL float
#define ACC double
#define SUFFIX(n) n##_f32
#define REAL_EPS 1.1920928955078125e-07 /* torch.finfo(float32).eps: F.rms_norm eps=None */
#define REAL_RSQRT(v) ((REAL)(1.0f / sqrtf((float)(v))))
#include "mot_oracle_float.inc"
#include "mot_oracle_attn.inc"
#undef REAL
#undef ACC
#undef SUFFIX
#undef REAL_EPS
#undef REAL_RSQRT

#define REAL double
#define ACC double
#define SUFFIX(n) n##_f64
#define REAL_EPS 2.220446049250313e-16 /* torch.finfo(float64).eps */
#define REAL_RSQRT(v) (1.0 / sqrt((double)(v)))
#include "mot_oracle_float.inc"
#include "mot_oracle_attn.inc"
