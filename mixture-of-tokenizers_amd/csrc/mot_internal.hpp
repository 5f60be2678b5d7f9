// mot_internal.hpp -- host-side declarations shared by the translation units of libmot_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/mot.h"

namespace mot {

// Records a thread-local message for mot_last_error() and returns `code`.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// hipGetLastError() after a launch -> MOT_OK / MOT_EHIP (with the kernel name in the message).
int check_launch(const char *what);

// Raises a kernel's dynamic-LDS limit to 160 KiB once per (kernel, device): `done` is that kernel's per-device bit set
// (a function-local static std::atomic at the call site).  hipFuncSetAttribute is idempotent, so a race costs a second call.
int ensure_max_dyn_lds(const void *kernel, std::atomic<uint64_t> &done, const char *name);

int pick_tile_tokens(int64_t n_rows, int64_t tokens_per_row, int bpt, bool with_ids);

int launch_tokens_to_bytes(const int32_t *tokens, int64_t n_tokens, const void *ttb, int elem, int64_t ttb_rows,
                           int bpt, int64_t *out, uint32_t *status, hipStream_t stream);
int launch_pull_bytes(const int64_t *in, int64_t *out, int64_t B, int64_t tokens_per_row, int bpt, int64_t pad,
                      int64_t eot, int dir, hipStream_t stream);
int launch_create_batch(const int32_t *tokens, int64_t B, int64_t T, const void *ttb_left, const void *ttb_right,
                        int elem, int64_t ttb_rows, int bpt, int32_t pad, int32_t eot, int64_t *out, uint32_t *status,
                        hipStream_t stream);
int launch_char_matrix(const int32_t *codes, const int64_t *tok_off, const int64_t *seq_off, int64_t n_seqs, int64_t seq_len, int max_char,
                       int32_t leading_space, int32_t bos_id, int32_t eos_id, int64_t *out, hipStream_t stream);
int launch_gather_rows(const void *ids_a, const void *ids_b, int ids_elem, int64_t n, const void *table, int64_t rows,
                       int dim, int rms_norm, float eps, const float *scale, void *out, uint32_t *status, int dtype,
                       hipStream_t stream);
// the same gather with row r written at out + (r / group) * out_ld + (r % group) * dim, and the status bit to raise
int launch_gather_rows_placed(const void *ids_a, const void *ids_b, int ids_elem, int64_t n, const void *table, int64_t rows,
                              int dim, int rms_norm, float eps, const float *scale, void *out, int group, int64_t out_ld,
                              uint32_t *status, uint32_t oor_flag, int dtype, hipStream_t stream);
int launch_rows_rnorm(const void *table, int64_t rows, int dim, float eps, float *out, int dtype, hipStream_t stream);
// the concat operand [n, K] of the concat + linear mixin in one kernel (one id tensor, Dt and Db multiples of the 16-byte vector)
int launch_concat_rows(const int32_t *tokens, const int64_t *ids, int64_t n, const void *tok_table, int64_t tok_rows, int Dt, const void *byte_table,
                       int64_t byte_rows, int Db, int bpt, int norm_tok, const float *byte_rnorm, float eps, void *u, int K, int tok_lo, int byte_lo,
                       uint32_t *status, int dtype, hipStream_t stream);
size_t embed_mix_workspace_bytes(const MotEmbedMixDesc &d);
int launch_embed_mix(const MotEmbedMixDesc &d, hipStream_t stream, __bf16 *add16 = nullptr, bool add_out = false);   // SUM / MEAN / NOOP (MixArgs.add16 / add_out)
bool embed_mix_mean_takes_add16(const MotEmbedMixDesc &d);
int launch_embed_mix_linear(const MotEmbedMixDesc &d, hipStream_t stream);  // CONCAT_LINEAR
int launch_embed_mix_linear_ex(const MotEmbedMixDesc &d, const float *wt_prebuilt, int wt_cols, hipStream_t stream);
size_t embed_mix_linear_workspace_bytes(const MotEmbedMixDesc &d);
size_t embed_mix_linear_bf16_workspace_bytes(const MotEmbedMixDesc &d);
int launch_embed_mix_linear_bf16(const MotEmbedMixDesc &d, hipStream_t stream);
size_t embed_mix_bwd_workspace_bytes(const MotEmbedMixDesc &d);
int launch_embed_mix_bwd(const MotEmbedMixDesc &d, const MotEmbedMixGrads &g, hipStream_t stream);
// C[j][k] += sum_n A[n][j] * B[n][k]  (A: n x M, B: n x Nc; fp32 MFMA, atomic accumulate)
int launch_gemm_tn(const float *A, int lda, int M, const float *B, int ldb, int Nc, int64_t n, float *C, int ldc, hipStream_t stream);
// C[n][c] = sum_r A[n][r] * (b_transposed ? B[c][r] : B[r][c]) (+ bias[c]), fp32 MFMA, plain stores
int launch_gemm_rows(const float *A, int lda, int64_t n, const float *B, int ldb, int R, int Nc, float *C, int ldc, bool b_transposed, hipStream_t stream,
                     const float *bias = nullptr, bool accumulate = false);
// the same product for FEW rows (a 458-row byte table, 132 character rows) over a long reduction: the reduction is cut into slices,
// one workgroup per (output block, slice), whose partial blocks go to `part` (gemm_rows_sliced_floats(n, R, Nc) floats, 0 = the shape
// gains nothing) and are summed in slice order by a second kernel -- the same bits on every run.  Falls back to launch_gemm_rows.
size_t gemm_rows_sliced_floats(int64_t n, int R, int Nc);
int launch_gemm_rows_sliced(const float *A, int lda, int64_t n, const float *B, int ldb, int R, int Nc, float *C, int ldc, bool b_transposed, float *part,
                            size_t part_floats, hipStream_t stream);
// the composed concat + linear forward (index kernels, seam gather, dense MFMA kernel, row norm), fp32 and bf16 (mot_linear.hip)
bool embed_mix_linear_is_composed(const MotEmbedMixDesc &d);
size_t embed_mix_linear_composed_workspace_bytes(const MotEmbedMixDesc &d);
int launch_embed_mix_linear_composed(const MotEmbedMixDesc &d, hipStream_t stream);
// C[n][c] = sum_r A[n][r] * B[c][r] (+ bias[c]) for bf16 A, B (and bias), fp32 accumulation, C bf16 or fp32
// mot_concat16.hip: gather + contraction + bias + output norm of the bf16 concat + linear mixin in one kernel
bool concat16_usable(const MotEmbedMixDesc &d);
bool concat16_norm_in_kernel(const MotEmbedMixDesc &d);   // the byte rows' rms factors need no table from the caller
int launch_concat16(const MotEmbedMixDesc &d, const int32_t *tokens, const int64_t *ids, const uint16_t *ids16, int64_t n, const float *rn_byte,
                    void *out, float *row_rnorm, hipStream_t stream);
int launch_wave_ids16(const MotEmbedMixDesc &d, uint16_t *ids16, hipStream_t stream);   // ids from the token->byte table, 16-bit, + parity outputs
int launch_gemm_rows_bf16(const void *A, int lda, int64_t n, const void *B, int ldb, int R, int Nc, void *C, int ldc, bool out_bf16,
                          const void *bias, hipStream_t stream, bool accumulate = false, const float *addend = nullptr);   // addend: fp32 [n][ldc] added before the store
// the 256 x 256 LDS-DMA kernel of mot_gemm_bf16.hip with fp32 operands (B transposed: B[c][r])
bool gemm_rows_f32_256_usable(const float *A, int lda, int64_t n, const float *B, int ldb, int R, int Nc);
int launch_gemm_rows_f32_256(const float *A, int lda, int64_t n, const float *B, int ldb, int R, int Nc, float *C, int ldc, const float *bias, bool accumulate,
                             hipStream_t stream);
// C[m][k] += sum_n A[n][m] * B[n][k]  (bf16 row-major operands, fp32 atomics into C; mot_backward.hip); lda / ldb / M / Kc multiples of 8
int launch_gemm_tn_bf16(const __bf16 *A, int lda, int M, const __bf16 *B, int ldb, int Kc, int64_t rows, float *C, int ldc, hipStream_t stream);
int launch_narrow(const float *src, int64_t n, void *dst_bf16, hipStream_t stream);                                  // dst[i] = bf16(src[i])
int launch_transpose_f32(const float *src, int rows, int cols, float *dst, hipStream_t stream);                        // dst[c][r] = src[r][c]
int launch_narrow_transpose(const float *src, int rows, int cols, void *dst_bf16, hipStream_t stream);               // dst[c][r] = bf16(src[r][c])
int launch_pad_copy(const float *src, int rows, int cols, float *dst, int rows_pad, int cols_pad, hipStream_t stream);
// counting sort of positions 0..n-1 by ids[position] (mot_backward.hip); ws_ints: group_positions_ws_ints(n, rows) int32
size_t group_positions_ws_ints(int64_t n, int64_t rows);
int launch_group_positions(const int32_t *ids, int64_t n, int64_t rows, int32_t *ws_ints, const int32_t **pos_sorted, const int32_t **id_sorted,
                           uint32_t *status, hipStream_t stream);
// zero n 32-bit words with a kernel (not hipMemsetAsync: a memset node aborts on graph replay with this runtime)
int launch_zero_words(void *p, int64_t n_words, hipStream_t stream);
size_t cross_attn_bwd_workspace_bytes(const MotCrossAttnDesc &d);
int launch_cross_attn_bwd(const MotCrossAttnDesc &d, const MotCrossAttnGrads &g, hipStream_t stream);
size_t char_swa_workspace_bytes(const MotCharSwaDesc &d);
int launch_char_swa(const MotCharSwaDesc &d, hipStream_t stream);
size_t cross_attn_workspace_bytes(const MotCrossAttnDesc &d);
int launch_cross_attn(const MotCrossAttnDesc &d, hipStream_t stream);

}  // namespace mot
