/*
 * lrp_hip.h -- C ABI of liblrp_hip.so: MI355X (gfx950) kernels for the AttnLRP hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces an ATen op
 * sequence that the reference reaches through PyTorch autograd; the reference file:line each
 * one stands in for is cited per function ("ref:" = /root/reference/...).
 *
 * Contract
 *   - plain pointers + sizes, no torch types.  All buffers are DEVICE pointers owned by the
 *     caller (borrowed for the launch); the library never allocates or frees.
 *   - asynchronous on the hipStream_t passed last (void* here so the header needs no HIP).
 *   - return 0 on success, a negative LRP_E* code on a rejected call (nothing launched).
 *   - re-entrant and thread-safe: no global mutable state (called from the Python main
 *     thread in forward and from the autograd worker thread in backward).
 *   - "gradient form": the propagated quantity is G = R / activation;
 *     relevance of any activation a is a (*) G_a.  eps == 0 selects lxt.efficient semantics
 *     (no stabiliser), eps > 0 the lxt.explicit stabiliser z/(c z + eps), UNSIGNED as in
 *     ref: lxt/explicit/functional.py:266-273.
 *   - dtype codes: LRP_F32 = 0 (fp32 storage, exact-fp32 MFMA), LRP_BF16 = 1 (bf16 storage,
 *     fp32 accumulate).  Row statistics (rstd, lse, D) are always fp32.
 */
#ifndef LRP_HIP_H
#define LRP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRP_F32 0
#define LRP_BF16 1

#define LRP_OK 0
#define LRP_EINVAL -1   /* bad argument (null pointer, negative size, unknown dtype)   */
#define LRP_EALIGN -2   /* pointer / leading dimension not aligned as the kernel needs */
#define LRP_ESHAPE -3   /* shape outside what the kernel supports                      */
#define LRP_ELAUNCH -4  /* hipLaunch failed; lrp_last_hip_error() has the HIP code     */

#define LRP_ACT_SILU 0
#define LRP_ACT_GELU_TANH 1
#define LRP_ACT_GELU 2
#define LRP_ACT_TANH 3       /* BERT pooler (identity rule on nn.Tanh, ref: lxt/explicit/models/bert.py:60-65) */

/* library identity / sanity */
int lrp_version(void);                 /* ABI version, currently 8 (round 6, second step: + the Gemma-3 site kernels lrp_sandwich_norm_fwd / _bwd / _ok, lrp_qk_norm_rope_fwd, lrp_qkv_bwd_pack; - the pair entries lrp_gemm_gated_fwd / _bwd / _bwd_ws, which the host never called).  Version 7 (round 6: the fused gated-MLP GEMMs stash the backward's COEFFICIENTS -- lrp_gemm_gated_coef_ok / _fwd_coef / _bwd_coef, and RoPE rides in the QKV forward's epilogue -- lrp_gemm_nt_rs_rope[_ok]; lrp_gemm_gated_fwd / _bwd are the GEMM + element-wise pair only; lrp_gemm_gated_fwd_rs (now the rs argument of _fwd_coef) and the de-phased tile walk -- lrp_set_gemm_scratch / lrp_gemm_scratch_bytes, measured negative in round 5 -- are GONE: the library holds no mutable state again).  Version 6 (round 5) added the K1n family -- lrp_gemm_norm_fused_ok, lrp_gemm_res_ssq, lrp_rms_rstd, lrp_gemm_nt_rs, lrp_gemm_nn_rs, lrp_gemm_gated_fwd_rs, lrp_gemm_nn_rs_res -- lrp_set_gemm_scratch / lrp_gemm_scratch_bytes, lrp_attn_bwd_dq_d[_ok], lrp_gqa_reduce_rope and lrp_linear_stream_fwd_tk / _splits / _ws / _tickets; version 5 added lrp_linear_stream_dgrad_tk / _tickets; version 4: (round 4 added lrp_linear_stream_fwd / _ok, lrp_linear_stream_dgrad / _ok / _ws, lrp_act_grad, lrp_layernorm_bwd_plain; nothing else changed).  Version 3: the one-pass lrp_linear_eps_smallm[_ws] of version 2 is gone (superseded by
                                          lrp_linear_smallm_fwd / _dgrad and lrp_gemm_skinny), lrp_gemm_skinny accepts any row count;
                                          added lrp_head_rmsnorm_fwd / _bwd, lrp_gemm_nn, lrp_gemm_skinny[_ws|_splits], lrp_gemm_gated_fwd / _bwd[_ws], lrp_gated_act_*_il.
                                          Every other version-2 signature is unchanged. */
const char* lrp_build_arch(void);      /* "gfx950" */
int lrp_last_hip_error(void);          /* last HIP error code seen by this thread */

/* ---------------------------------------------------------------------------------------
 * K1  Linear.   ref: forward  lxt/explicit/functional.py:345-351 (F.linear),
 *               backward lxt/explicit/functional.py:355-364, lxt/explicit/rules.py:206-222
 * lrp_gemm_nt: C[b][M,N] = A[b][M,K] . B[b][N,K]^T (+ bias[N]), fp32 accumulate on MFMA.
 *   Both operands are K-contiguous ("NT"): the forward z = x W^T with W in its stored [out,in] layout.  The backward
 *   c = s W uses lrp_gemm_nn on the SAME stored weight (bf16; no W^T copy exists); only the fp32 parity path and shapes
 *   lrp_gemm_nn refuses transpose a weight once (host side) and come back here.
 *   lda/ldb/ldc in elements; K, lda, ldb multiples of 16 bytes' worth of elements.
 *   batch >= 1 with element strides sA/sB/sC (sB may be 0 to share B).
 *   out_dtype may differ from dtype only as LRP_F32 (fp32 output from bf16 operands).
 * --------------------------------------------------------------------------------------- */
int lrp_gemm_nt(const void* A, const void* B, void* C, const void* bias,
                int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                int batch, int64_t sA, int64_t sB, int64_t sC,
                int dtype, int out_dtype, void* stream);

/* lrp_gemm_nn: C[M,N] = A[M,K] . Bt[K,N] (+ bias[N]) -- the eps-rule's redistribution  c = s W  with W in its STORED forward layout
 * [out, in] as Bt (contraction over W's rows): no W^T copy of a weight exists anywhere.  bf16 operands, out bf16 / fp32, K a multiple
 * of 64 (>= 128), Bt below 2^30 elements and 256 rows of A below 2^30 elements (more rows of A are issued as several launches over
 * row chunks); other shapes return LRP_ESHAPE (the caller then transposes once and uses lrp_gemm_nt).
 * ref: lxt/explicit/functional.py:355-364 (backward of linear_epsilon_fn: `relevance_norm @ weight`). */
int lrp_gemm_nn(const void* A, const void* Bt, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                int64_t ldc, int dtype, int out_dtype, void* stream);

/* lrp_linear_stream_fwd: z[M,N] = x[M,K] . W[N,K]^T (+ bias[N]) for 1 <= M <= 128 rows in ONE launch -- the Linear forward in its
 * HBM-bound regime as a narrow-N, full-K weight-streaming MFMA kernel (csrc/linear_stream.hip): a workgroup owns 64 rows of W and the
 * whole K range, W is read from HBM exactly once (no split-K, no fp32 slabs, no second launch), x is staged through LDS, 16-row
 * blocks past M are not multiplied.  bf16 operands, out bf16 / fp32, K a multiple of 512, ldx / ldw multiples of 8 elements, operands
 * below 2^30 elements; anything else returns LRP_ESHAPE / LRP_EALIGN.  lrp_linear_stream_ok(M, N, K, ldx, ldw) = 1 when the kernel
 * applies AND its ceil(N / 64) workgroups fill the chip (>= 192); otherwise the caller uses lrp_gemm_skinny.
 * ref: lxt/explicit/functional.py:345-351 (forward of linear_epsilon_fn), lxt/explicit/rules.py:188-205. */
int lrp_linear_stream_ok(int M, int N, int K, int64_t ldx, int64_t ldw);
/* Round 5 (ABI 6): narrow weights (N = 4096: 64 workgroups of full K) run the same kernel with 2 ... 8 K SPLITS -- fp32 partial slabs in `ws`
 * (lrp_linear_stream_fwd_ws bytes), summed inside the kernel by the last workgroup to arrive on a 64-row block (`tickets`:
 * lrp_linear_stream_fwd_tickets zeroed 32-bit words, left zero again; slab order: deterministic) -- still ONE launch.  With ws / tickets NULL, or
 * through lrp_linear_stream_fwd, the full-K form runs whatever the workgroup count. */
int lrp_linear_stream_fwd_splits(int M, int N, int K);
int64_t lrp_linear_stream_fwd_ws(int M, int N, int K);
int lrp_linear_stream_fwd_tickets(int M, int N, int K);
int lrp_linear_stream_fwd_tk(const void* x, const void* W, const void* bias, void* z, int M, int N, int K, int64_t ldx, int64_t ldw,
                             int64_t ldz, int dtype, int out_dtype, void* ws, void* tickets, void* stream);
int lrp_linear_stream_fwd(const void* x, const void* W, const void* bias, void* z, int M, int N, int K, int64_t ldx, int64_t ldw,
                          int64_t ldz, int dtype, int out_dtype, void* stream);

/* lrp_linear_stream_dgrad: c[M,Kout] = s[M,N] . W[N,Kout] (the eps-rule's redistribution from the STORED weight: contraction over W's rows) for
 * 1 <= M <= 64 rows -- the dgrad counterpart of lrp_linear_stream_fwd: a workgroup owns 64 output columns (128-byte segments of every W row)
 * and one of `splits` ranges of contraction rows, operands go through wave-private LDS rings (no barrier in the contraction loop), the
 * MFMA's W operand is gathered by ds_read_b64_tr_b16.  splits > 1 (few column blocks, e.g. Kout = 4096) writes fp32 slabs into `ws`
 * (lrp_linear_stream_dgrad_ws BYTES, caller-allocated) that a second small kernel sums in slab order; otherwise the result is written directly.
 * z (optional, [M,N] with pitch ldz; NULL = none): the Linear's forward output -- the eps-rule's stabiliser is then formed on the fly,
 * s' = s z / (z + eps) (s = incoming gradient) or, relevance_in = 1, s / (z + eps) (s = incoming relevance), rounded to bf16 (v_rcp_f32 quotient:
 * may differ from lrp_eps_scale's exact division in the last bf16 bit): one launch for  R/(z+eps) -> (.) W  (ref
 * lxt/explicit/functional.py:355-358).  z with eps == 0 -> LRP_EINVAL (no stabiliser: pass z = NULL).
 * bf16 operands, out bf16 / fp32, N a multiple of 128, Kout of 64; lrp_linear_stream_dgrad_ok = 1 when the kernel applies and fills the chip.
 * ref: lxt/explicit/functional.py:355-364 (`relevance_norm @ weight`), lxt/explicit/rules.py:206-222. */
int lrp_linear_stream_dgrad_ok(int M, int N, int Kout, int64_t lds, int64_t ldw);
int64_t lrp_linear_stream_dgrad_ws(int M, int N, int Kout);
int lrp_linear_stream_dgrad(const void* s, const void* z, const void* W, void* c, int M, int N, int Kout, int64_t lds, int64_t ldz,
                            int64_t ldw, int64_t ldc, float eps, int relevance_in, int dtype, int out_dtype, void* ws, void* stream);
/* lrp_linear_stream_dgrad_tk (ABI 5): the same with the contraction splits reduced INSIDE the launch -- every split's workgroup publishes its fp32
 * partial tile into `ws` write-through and takes a ticket of its 64-column block; the last arriver sums the slabs in slab order (deterministic)
 * and writes c.  `tickets`: lrp_linear_stream_dgrad_tickets(M, N, Kout) 32-bit words, ZERO-INITIALISED ONCE by the caller, private to one stream
 * and never written by anything else (the last arriver of a launch re-arms its word to 0: no host-side reset, hipGraph-replayable); NULL = the
 * two-launch form. */
int lrp_linear_stream_dgrad_tickets(int M, int N, int Kout);
int lrp_linear_stream_dgrad_tk(const void* s, const void* z, const void* W, void* c, int M, int N, int Kout, int64_t lds, int64_t ldz,
                               int64_t ldw, int64_t ldc, float eps, int relevance_in, int dtype, int out_dtype, void* ws, void* tickets,
                               void* stream);

/* lrp_gemm_skinny: the same two products with SPLIT-K, for problems whose 256 x 256 tile count alone leaves CUs idle: 1 <= M <= 256 rows
 * (the HBM-bound regime of the Linear eps-rule, SURVEY.md 8d: arithmetic
 * intensity ~2 M FLOP/B in bf16): split-K over the K tiles so that every CU streams its share of the weight exactly once, fp32 partial
 * slabs in `ws` (lrp_gemm_skinny_ws(M,N,K) BYTES, caller-allocated), reduced (+ bias, cast) by a second small kernel; when the tile
 * count alone fills the chip (the LM head) there is one split and the kernel writes C directly.  Larger M is accepted too: with <= 128
 * tiles (M = 2048 rows against a 4096-row weight) the K range is split 2 ... 8 ways so that every CU has a workgroup, and with 257 ... 384
 * tiles and K >= 8192 (8192 x 2560: 1.25 rounds of the chip) two ways.
 *   nn = 0: C = A[M,K] . B[N,K]^T (forward z = x W^T);   nn = 1: C = A[M,K] . B[K,N] (redistribution c = s W, W as stored).
 * Same operand restrictions as lrp_gemm_nn.  ref: lxt/explicit/functional.py:351 (forward), :355-364 (backward). */
int64_t lrp_gemm_skinny_ws(int M, int N, int K);
int lrp_gemm_skinny_splits(int M, int N, int K);      /* K splits lrp_gemm_skinny would use (1: the plain GEMM serves the problem as well) */
int lrp_gemm_skinny(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                    int64_t ldc, int nn, int dtype, int out_dtype, void* ws, void* stream);

/* out = g (*) z/(c z + eps)      (mode 0: gradient form; eps==0 -> g/c)
 * out = r / (c z + eps)          (mode 1: relevance form, the reference's R_out/(z+eps))
 * ref: lxt/explicit/functional.py:360 (linear), :403 (matmul, c=2), :445 (add2) */
int lrp_eps_scale(const void* g, const void* z, void* out, int64_t n, float c, float eps,
                  int mode, int dtype, void* stream);

/* same, on 2-D strided views [rows, cols] (slices of fused GEMM outputs); row strides in elements */
int lrp_eps_scale2d(const void* g, const void* z, void* out, int rows, int cols, int64_t ldg,
                    int64_t ldz, int64_t ldo, float c, float eps, int mode, int dtype, void* stream);

/* out = a (*) b  (R = x (*) G read-out of a rule, ref: functional.py:362 ".mul_(inputs)") */
int lrp_mul(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);

/* Small-M Linear (M <= 16) as W-streaming kernels -- what the engine runs for its one-row-per-prompt top layer, the
 * last-token LM head and dense logit seeds, and lxt_amd.explicit.functional.linear_epsilon for M <= 16.  W [N,K] row-major is
 * read ONCE per call and no W^T copy is needed in either direction (HBM-bound: roofline 8 TB/s).
 *   fwd  : z[M,N] = x[M,K] W^T (+ bias)                 z dtype = out_dtype (dtype or LRP_F32), row strides ldx / ldz
 *   dgrad: s = g                       (z == NULL)
 *          s = g (*) z/(z+eps)          (z given, relevance_in == 0; eps == 0 -> g)
 *          s = g / (z+eps)              (z given, relevance_in == 1: g is the relevance R_out)
 *          out[M,K] = s W  [ (*) x if relevance_out ]   out dtype = out_dtype, contiguous [M,K]; g/z row strides ldg / ldz
 *   workspace: fp32 scratch of lrp_linear_smallm_ws(M,N,K,dtype) floats (k-slab / row-range partials, no atomics).
 * ref: lxt/explicit/functional.py:345-364 (forward :351, backward :355-364). */
int64_t lrp_linear_smallm_ws(int M, int N, int K, int dtype);
int lrp_linear_smallm_fwd(const void* x, const void* W, const void* bias, void* z, float* workspace,
                          int M, int N, int K, int64_t ldx, int64_t ldz, int dtype, int out_dtype, void* stream);
int lrp_linear_smallm_dgrad(const void* g, const void* z, const void* W, const void* x, void* out,
                            float* workspace, int M, int N, int K, int64_t ldg, int64_t ldz, float eps,
                            int relevance_in, int relevance_out, int dtype, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * K2/K6  RMSNorm identity rule + residual add.
 *   ref: lxt/efficient/patches.py:111-123, lxt/explicit/functional.py:481-495 (norm),
 *        lxt/explicit/functional.py:430-459 via lxt/explicit/models/llama.py:481,488 (add2)
 * forward:  hsum = h (+ branch) ; y = w' (*) hsum * rstd ; rstd = rsqrt(mean(hsum^2)+eps)
 *           w' = w + w_offset (Gemma3: 1 + w).  branch/hsum_out may be NULL (no residual).
 * backward: Gh = Gres + Gx (*) w' * rstd           (identity rule; Gres or Gx may be NULL)
 *           Gs = Gh (*) hsum/(hsum+eps_add)         (add2 of the residual below; written to Gs_out)
 *           A  = Gs (*) branch/(branch+eps_lin)     (eps-rule scale for the branch's last Linear)
 *           rel_out[row] (optional, fp32) = sum_h hsum*Gh  (latent relevance per token)
 *   branch == NULL means "no residual below" (embedding): Gs_out = Gh, A_out untouched.
 * --------------------------------------------------------------------------------------- */
/* per-head RMSNorm (HF Gemma3Attention q_norm / k_norm: rows of head_dim elements inside a fused projection output [rows, >= heads*d],
 * row pitches ldx / ldy in elements): y = (w + w_offset) (*) x * rstd, rstd[rows*heads] = rsqrt(mean_d(x^2) + eps); the backward treats
 * rstd as a constant (ref lxt/efficient/models/gemma3.py:11-12): out = G (*) (w + w_offset) * rstd.  d * sizeof(T) / 16 must be a power
 * of two <= 64. */
int lrp_head_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int heads, int d, int64_t ldx, int64_t ldy,
                         float eps, float w_offset, int dtype, void* stream);
int lrp_head_rmsnorm_bwd(const void* G, const void* w, const float* rstd, void* out, int64_t rows, int heads, int d, int64_t ldg,
                         int64_t ldo, float w_offset, int dtype, void* stream);
int lrp_add_rmsnorm_fwd(const void* h, const void* branch, const void* w, void* hsum_out,
                        void* y, float* rstd, int M, int H, float eps, float w_offset,
                        int dtype, void* stream);
int lrp_rmsnorm_bwd_add2(const void* Gres, const void* Gx, const void* w, const float* rstd,
                         const void* hsum, const void* branch, void* Gs_out, void* A_out,
                         float* rel_out, int M, int H, float w_offset, float eps_add,
                         float eps_lin, int dtype, void* stream);

/* Gemma-3's row work fused per SITE (round 6, ABI 8; csrc/sandwich.hip).  Every entry reproduces the arithmetic of the launch sequence it
 * replaces, including the roundings to the storage type where that sequence went through memory: bit-identical results.
 *   lrp_sandwich_norm_fwd: a sub-layer output between its post-norm and the next pre-norm (HF Gemma3DecoderLayer: h1 = h + post_attention_layernorm(a),
 *     x2 = pre_feedforward_layernorm(h1); likewise post_feedforward_layernorm / the next layer's input_layernorm; identity rules with detached
 *     rstd, ref lxt/efficient/models/gemma3.py:11-19, lxt/efficient/patches.py:111-123):
 *       hsum_out = res + T(w_post' (*) x rstd_post),  y = w_pre' (*) hsum_out rstd_pre   (w' = w + w_offset; y / w_pre may be NULL)
 *     = lrp_add_rmsnorm_fwd(x, NULL, w_post) + lrp_add_rmsnorm_fwd(res, branch, w_pre) in one pass; the normed branch never goes to memory.
 *     lrp_sandwich_norm_ok(H, dtype): the row fits the kernel's registers (H <= 8192 bf16 / 4096 fp32, H a multiple of 16 bytes).
 *   lrp_sandwich_norm_bwd: Gs_out = T(Gres + Gx w_pre' rstd_pre) (gradient w.r.t. the residual sum; Gres may be NULL),
 *     Ga_out = Gs_out w_post' rstd_post (gradient w.r.t. the sub-layer output)  = two lrp_rmsnorm_bwd_add2 launches.
 *   lrp_qk_norm_rope_fwd: per-head q / k RMSNorm + RoPE on the fused projection output qkv [rows, >= (nq + nk) d] (HF Gemma3Attention q_norm /
 *     k_norm / apply_rotary_pos_emb): qr [rows, nq d], kr [rows, nk d], rstd_q [rows nq], rstd_k [rows nk]; cos / sin tables [seq, d] fp32,
 *     position = row % seq  = 2 x lrp_head_rmsnorm_fwd + 2 x lrp_rope_fwd.
 *   lrp_qkv_bwd_pack: the qkv dgrad's operand A [rows, (nq + 2 nk) d] in one pass: q part = rope^T(dq) (*) wq' rstd_q, k part = rope^T(sum of the
 *     group's query heads of dk_h [rows, nq d]) (*) wk' rstd_k, v part = the group sum of dv_h   = 2 x lrp_gqa_reduce + 2 x lrp_rope_bwd (eps = 0) +
 *     2 x lrp_head_rmsnorm_bwd.  d sizeof(T) / 16 a power of two in 2 .. 64. */
int lrp_sandwich_norm_ok(int H, int dtype);
int lrp_sandwich_norm_fwd(const void* x, const void* res, const void* w_post, const void* w_pre, void* hsum_out, void* y,
                          float* rstd_post, float* rstd_pre, int M, int H, float eps, float w_offset, int dtype, void* stream);
int lrp_sandwich_norm_bwd(const void* Gres, const void* Gx, const void* w_pre, const float* rstd_pre, const void* w_post,
                          const float* rstd_post, void* Gs_out, void* Ga_out, int M, int H, float w_offset, int dtype, void* stream);
int lrp_qk_norm_rope_fwd(const void* qkv, const void* wq, const void* wk, void* qr, void* kr, float* rstd_q, float* rstd_k,
                         const float* cos_t, const float* sin_t, int64_t rows, int seq, int nq, int nk, int d, int64_t ldqkv,
                         int64_t ldq, int64_t ldk, float eps, float w_offset, int dtype, void* stream);
int lrp_qkv_bwd_pack(const void* dq, const void* dk_h, const void* dv_h, const void* wq, const void* wk, const float* rstd_q,
                     const float* rstd_k, const float* cos_t, const float* sin_t, void* A, int64_t rows, int seq, int nq, int nk, int d,
                     int64_t lddq, int64_t lddk, int64_t lddv, int64_t lda, float w_offset, int dtype, void* stream);

/* K7 LayerNorm (BERT/GPT-2/ViT).  ref: lxt/efficient/patches.py:126-142,
 *   lxt/explicit/functional.py:606-635.   y = (x-mean)/std * w + b ; std detached.
 *   backward: u = Gy (*) y/(y+eps_y) (*) w * rstd ; Gx = u - mean_row(u). */
int lrp_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean,
                      float* rstd, int M, int H, float eps, int dtype, void* stream);
int lrp_layernorm_bwd(const void* Gy, const void* y, const void* w, const float* rstd,
                      void* Gx, int M, int H, float eps_y, int dtype, void* stream);
/* full LayerNorm VJP, NO rule (mean and 1/std both differentiated): u = Gy (*) w, xh = (x - mean) rstd,
 * Gx = rstd (u - mean_row(u) - xh mean_row(u (*) xh)); x / mean / rstd as lrp_layernorm_fwd took / wrote them (image tower, see lrp_act_grad) */
int lrp_layernorm_bwd_plain(const void* Gy, const void* x, const void* w, const float* mean, const float* rstd, void* Gx,
                            int M, int H, int dtype, void* stream);


/* ---------------------------------------------------------------------------------------
 * K3/K8  gated MLP element-wise rules.
 *   ref: lxt/efficient/patches.py:145-157 (identity rule on act, uniform rule on the product),
 *        lxt/efficient/rules.py:88-100, lxt/explicit/rules.py:68-78,405-418,
 *        lxt/explicit/models/llama.py:84-86,273-281
 * forward : m = act(g) (*) u        (g,u,m [M,I] with row strides ldg/ldu/ldm)
 * backward: Gact = Gm*u/2 ; Gu = Gm*act/2
 *           Ag = Gact * act/(g + eps_g)            (identity rule (*) gate Linear eps scale;
 *                                                   eps_g = 1e-10 efficient / eps_lin explicit)
 *           Au = Gu * u/(u + eps_lin)              (eps_lin==0 -> Gu)
 * --------------------------------------------------------------------------------------- */
int lrp_gated_act_fwd(const void* g, const void* u, void* m, int M, int I, int64_t ldg,
                      int64_t ldu, int64_t ldm, int act, int dtype, void* stream);
int lrp_gated_act_bwd(const void* Gm, const void* g, const void* u, void* Ag, void* Au,
                      int M, int I, int64_t ldgm, int64_t ldg, int64_t ldu, int64_t ldag,
                      int64_t ldau, float eps_g, float eps_lin, int act, int dtype, void* stream);
/* The same two rules on the INTERLEAVED output of a fused gate/up Linear (the engine's layout: the rows of the fused weight [2 I, H] are
 * ordered in blocks of 64 = [gate rows 32 b .. 32 b + 31 | up rows 32 b .. 32 b + 31], so one 64-column block of gu = W_gu x holds gate AND up
 * of the same 32 intermediate indices -- and so does one wave's accumulator tile of the GEMM):
 *   lrp_gated_act_fwd_il / _bwd_il : the element-wise kernels on gu / Agu [M, 2 I] in that layout (small M, fp32, shapes the fused form below
 *                                    refuses: the caller runs them behind the Linear entry that fits its row count -- lrp_linear_stream_fwd,
 *                                    lrp_gemm_skinny, lrp_gemm_nt / _nn; the pair entries lrp_gemm_gated_fwd / _bwd[_ws] of ABI 3-7, which hard-wired
 *                                    the large-M GEMM, are gone in ABI 8: the host never called them)
 * ref: lxt/efficient/patches.py:145-157, lxt/explicit/models/llama.py:84-86,273-281. */
#define LRP_GATED_IL 32
int lrp_gated_act_fwd_il(const void* gu, void* m, int M, int I, int64_t ldgu, int64_t ldm, int act, int dtype, void* stream);
int lrp_gated_act_bwd_il(const void* Gm, const void* gu, void* Agu, int M, int I, int64_t ldgm, int64_t ldgu, int64_t ldagu,
                         float eps_g, float eps_lin, int act, int dtype, void* stream);
/* FUSED form, M = B S rows (round 6): the rules run in the epilogues of the two GEMMs around them and the forward stashes the backward's
 * COEFFICIENTS instead of g and u.  The gate/up GEMM has g and u of an intermediate index in fp32 registers; it evaluates the activation once
 * and writes   m  = act(g) u,
 *              cg = 1/2 u act(g) / (g + eps_g)    (0 where g + eps_g = 0)     = identity rule on act (*) uniform rule (*) the gate Linear's stabiliser
 *              cu = 1/2 act(g) u / (u + eps_lin)  (1/2 act(g) for eps_lin = 0) = uniform rule (*) the up Linear's stabiliser
 * (eps_g = 1e-10, eps_lin = 0: lxt.efficient, ref lxt/efficient/rules.py:88-100, patches.py:145-157; eps_g = eps_lin = the Linear eps:
 * lxt.explicit, ref lxt/explicit/models/llama.py:84-86,273-281 with rules.py:68-78,405-418 and functional.py:355-364).  The down-projection's
 * dgrad Gm = A_dn W_dn (NN form, W_dn [H, I] as stored) is never written: its epilogue forms Agu = { Gm cg | Gm cu } -- one multiply per element,
 * no transcendental in the backward -- in the interleaved layout above, ready to be the gate/up dgrad's operand.  g and u are never stored.
 *   coef [M, 2 I] bf16, row pitch ldcoef (multiple of 8): PRIVATE to the pair of entry points, in accumulator order -- the 16 bytes
 *   { cg x 4 | cu x 4 } of intermediate indices 4 t .. 4 t + 3 at columns 8 t .. 8 t + 7 of the row.
 *   rs (may be NULL): fp32 [M], the accumulators of the forward are scaled by rs[m] first (K1n below: the folded RMSNorm's 1 / rms).
 *   lrp_gemm_gated_coef_ok(M, I, H, ldx, ldwgu, lda, ldwd, act, dtype) -> 1 when BOTH launches are problems the fused epilogues take (bf16,
 *   I % 32 == 0, >= 190 tiles of 256 x 256 each, SiLU / tanh-GELU); the two entry points return LRP_ESHAPE otherwise (no fallback inside). */
int lrp_gemm_gated_coef_ok(int M, int I, int H, int64_t ldx, int64_t ldwgu, int64_t lda, int64_t ldwd, int act, int dtype);
int lrp_gemm_gated_fwd_coef(const void* x, const void* Wgu, const float* rs, void* coef, void* m, int M, int I, int K, int64_t ldx, int64_t ldw,
                            int64_t ldcoef, int64_t ldm, float eps_g, float eps_lin, int act, int dtype, void* stream);
int lrp_gemm_gated_bwd_coef(const void* Adn, const void* Wdn, const void* coef, void* Agu, int M, int I, int K, int64_t lda, int64_t ldw,
                            int64_t ldcoef, int64_t ldagu, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * K1n  Llama-type RMSNorm FOLDED INTO THE LINEARS AROUND IT (round 5; M = B S rows, bf16, the 256 x 256 ping-pong kernel only).
 *   ref: lxt/efficient/patches.py:111-123 (rms_norm_forward: y = w * x * rsqrt(mean x^2 + eps) with the variance DETACHED = the identity
 *   rule, so its backward is the row scale G_x = G_y * w * rstd), HF modeling_llama's residual sums h = h + attn(..), h = h + mlp(..)
 *   (explicit form: lf.add2, lxt/explicit/models/llama.py:481,488 -- with eps = 0 a plain sum), and the Linears of
 *   lxt/explicit/functional.py:345-364 on both sides.
 *   The norm's weight w is folded into the consuming Linear by the host (W' = W diag(w): (w (.) x rstd) W^T = rstd (x W'^T)), after which
 *     forward :  h1 = h + o Wo^T            and the sums of squares of h1's rows      -> lrp_gemm_res_ssq  (+ lrp_rms_rstd: rstd from the partials)
 *                gu = rstd (.) (h1 W'gu^T), m = act(g) (*) u + coefficient stash      -> lrp_gemm_gated_fwd_coef with rs = rstd
 *                qkv = rstd (.) (h W'qkv^T)                                           -> lrp_gemm_nt_rs
 *     backward:  G_h = rstd (.) (A W') + G_res  (norm's identity rule + residual add) -> lrp_gemm_nn_rs_res
 *   i.e. the stand-alone lrp_add_rmsnorm_fwd / lrp_rmsnorm_bwd_add2 launches (4 x 67 MB round trips per layer at M = 8192, H = 4096) become
 *   16-byte residual loads and 8 row scales in the GEMM epilogues; the normalised activations are never written.
 *   lrp_gemm_norm_fused_ok(M, N, K, lda, ldb, nn, dtype) -> 1 when the entry points take the problem (bf16, N % 256 == 0, K % 64 == 0, at
 *   least 190 output tiles, operands within 32-bit byte offsets); they return LRP_ESHAPE otherwise (no fallback: the caller keeps the
 *   stand-alone kernels for such shapes).
 *   lrp_gemm_res_ssq : out[M,N] = bf16(res + x W^T), W [N,K];  ssq [N / 64][ldssq] fp32: ssq[p][m] = sum over columns 64 p .. 64 p + 63 of the
 *                      ROUNDED out[m][.]^2  (ldssq >= M; every entry of rows < M is written)
 *   lrp_rms_rstd     : rstd[m] = rsqrt(sum_p ssq[p][m] / H + eps)  (partials summed in order: deterministic)
 *   lrp_gemm_nt_rs   : out[M,N] = bf16(rs[m] * (x W^T))
 *   lrp_gemm_nn_rs_res    : out[M,N] = bf16(rs[m] * (s W) + res), W [K,N] as stored (the dgrad form of lrp_gemm_nn); out may alias res
 * --------------------------------------------------------------------------------------- */
int lrp_gemm_norm_fused_ok(int M, int N, int K, int64_t lda, int64_t ldb, int nn, int dtype);
int lrp_gemm_res_ssq(const void* x, const void* W, const void* res, void* out, float* ssq, int M, int N, int K, int64_t ldx, int64_t ldw,
                     int64_t ldres, int64_t ldout, int64_t ldssq, void* raw, int64_t ldraw, int dtype, void* stream);
                     /* raw (may be NULL; round 6): bf16 [M, N], row pitch ldraw -- the Linear's own output x W^T is stored as well: the lxt.explicit
                        placement divides by it in the backward (stabilisers z / (z + eps) of lxt/explicit/functional.py:355-364,430-459) */
int lrp_rms_rstd(const float* ssq, int parts, int64_t ldssq, int M, int H, float eps, float* rstd, void* stream);
int lrp_gemm_nt_rs(const void* x, const void* W, const float* rs, void* out, int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldout,
                   int dtype, void* stream);
/* lrp_gemm_nt_rs_rope (round 6; K5 of SURVEY.md 2.3 folded into K1): the fused QKV forward  out = rs[m] * (x W^T)  with HF's RoPE
 * (apply_rotary_pos_emb: q cos + rotate_half(q) sin; constant tables, un-patched in lxt.efficient; explicit form lxt/explicit/models/llama.py:226-260)
 * applied to the q / k head columns [0, rope_cols) in the epilogue, in fp32 on the un-rounded accumulators; columns >= rope_cols (v) only take the
 * row scale.  cos / sin: fp32 [>= seq, 128], row p = position p (both halves of a row equal); the position of output row m is m % seq.  Heads of
 * 128 columns, M and N multiples of 256, the 256 x 256 ping-pong kernel only: lrp_gemm_nt_rs_rope_ok() -> 1 when the entry point takes the problem,
 * LRP_ESHAPE otherwise (the caller keeps lrp_gemm_nt_rs + lrp_rope_fwd for such shapes).  Weights and outputs stay in the standard head-dim order. */
int lrp_gemm_nt_rs_rope_ok(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldout, int seq, int rope_cols, int head_dim, int dtype);
int lrp_gemm_nt_rs_rope(const void* x, const void* W, const float* rs, const float* cos, const float* sin, void* out, int M, int N, int K,
                        int64_t ldx, int64_t ldw, int64_t ldout, int seq, int rope_cols, int head_dim, int dtype, void* stream);
int lrp_gemm_nn_rs(const void* s, const void* W, const float* rs, void* out, int M, int N, int K, int64_t lds, int64_t ldw, int64_t ldout,
                   int dtype, void* stream);      /* out[M,N] = bf16(rs[m] * (s W)), W [K,N] as stored (rs = 1/2 everywhere: the o-projection's dgrad
                                                     with the uniform rule's factor of the P.V product, lxt/efficient/patches.py:193-203) */
int lrp_gemm_nn_rs_res(const void* s, const void* W, const float* rs, const void* res, void* out, int M, int N, int K, int64_t lds, int64_t ldw,
                       int64_t ldres, int64_t ldout, int dtype, void* stream);

/* stand-alone activation identity rule (BERT/GPT-2 mlp_forward, ref: patches.py:160-168):
 * forward y = act(x); backward A = Gy * act(x)/(x+eps_g)                                  */
int lrp_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream);
int lrp_act_bwd(const void* Gy, const void* x, void* Gx, int64_t n, float eps_g, int act,
                int dtype, void* stream);
/* plain derivative, NO rule: Gx = Gy * act'(x).  The reference's gemma3 map patches nothing in modeling_siglip, so the SigLIP tower of
 * Gemma-3 keeps ordinary gradients through its GELU / LayerNorm (ref: lxt/efficient/models/gemma3.py:14-19). */
int lrp_act_grad(const void* Gy, const void* x, void* Gx, int64_t n, int act, int dtype, void* stream);

/* forward sums of the encoder families: out[m,:] = x[m,:] + y[m % period,:]  (period 1: a bias / token-type row, period S: position
 * embeddings, period M: a residual sum; the summands of lf.add2, ref: lxt/explicit/models/bert.py:249-253,:396).  x, y, out [.,H]
 * contiguous; out may alias x.                                                                                                    */
int lrp_add_bcast(const void* x, const void* y, void* out, int M, int H, int period, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * K5  RoPE (rotate-half convention).  ref: HF apply_rotary_pos_emb;
 *     explicit eps: lxt/explicit/models/llama.py:226-260 (add2 eps=1e-8 on the rotated sum).
 * x [S, n_heads, d] with row stride ldx (elements); cos/sin fp32 [S, d].
 * forward : xr = x*cos + rotate_half(x)*sin
 * backward: Gp = Gr (*) xr/(xr+eps_rope) ; Gx = Gp*cos + rotate_half^T(Gp*sin) ;
 *           A = Gx (*) x/(x+eps_lin)     (xr / x may be NULL when the matching eps is 0)
 * pos0: position of row 0; rows are positions pos0 + (row % S_per_seq).
 * --------------------------------------------------------------------------------------- */
int lrp_rope_fwd(const void* x, void* xr, const float* cos_t, const float* sin_t, int rows,
                 int seq, int n_heads, int d, int64_t ldx, int64_t ldxr, int dtype, void* stream);
int lrp_rope_bwd(const void* Gr, const void* xr, const void* x, void* A, const float* cos_t,
                 const float* sin_t, int rows, int seq, int n_heads, int d, int64_t ldg,
                 int64_t ldxr, int64_t ldx, int64_t lda, float eps_rope, float eps_lin,
                 int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * K4  attention (flash-style, causal or full, GQA, optional sliding window).
 *   ref: lxt/efficient/patches.py:193-203 (divide_gradient 4,4,2 == the two 1/2 below),
 *        lxt/explicit/functional.py:293-322 (softmax, Prop 3.1), :385-408 (QK^T, R/(2s+eps)),
 *        lxt/explicit/rules.py:267-282 (P.V uniform-epsilon), llama.py:379-391.
 * Layout: q [B*S, Hq, d], k/v [B*S, Hkv, d] token-major with row strides ldq/ldk/ldv
 * (slices of a fused QKV GEMM output are consumed in place); o [B*S, Hq, d] (ldo);
 * lse fp32 [B, Hq, S] = log-sum-exp of the scaled scores; d in {16,32,64,128,256}, bf16 also 96.
 * "_t" operands are head-transposed copies [B, H, d, ldt] (ldt >= S, multiple of 16 bytes'
 * worth of elements, pad columns FINITE -- zero them once) made by lrp_transpose_heads.  They are read ONLY by the kernels
 * for which lrp_attn_needs_transposed(dtype, d) returns 1 (fp32, and bf16 head dims without a transpose-read kernel); the
 * bf16 kernels of attention32.hip gather the transposed MFMA operand out of the row-major LDS tile (ds_read_b64_tr_b16)
 * and take NULL for every "_t" pointer.
 * window <= 0: none; window = w: key j visible to query i iff i-w < j <= i (Gemma3 local).
 * q_begin (0 = everything): only query rows >= q_begin are needed / carry relevance -- query blocks
 *   entirely below it are skipped (top-layer sparsity: above the last attention layer only the last
 *   token of a prompt has a non-zero seed); rows below q_begin in outputs are unspecified (fwd, dq).
 *   row_lo / row_hi (int32 [B*S], both or neither; NULL = none): per-query-row key interval -- key j is visible to
 *   query (b, i) only if row_lo[b*S+i] <= j < row_hi[b*S+i], IN ADDITION to (causal, window).  The bf16 kernels of
 *   attention32.hip derive the tiles they visit from the intervals themselves (union over a workgroup's rows; no assumption
 *   on the intervals), so a mask handed over ONLY as intervals (causal = 0) costs what the causal flag costs; the fp32 /
 *   small-head-dim kernels bound their tiles by (causal, window) only.  Every mask HF builds for the supported families is of this
 *   form (left / right padding, packed sequences, Gemma-3's bidirectional image blocks: causal = 0 there).  A row
 *   with an empty interval yields o = 0, lse = -inf and contributes nothing to dQ / dK / dV
 *   (ref: the additive-mask argument of HF eager_attention_forward, which lxt/efficient/patches.py:193-203 wraps).
 * backward (gradient form):
 *   Gho = f Go (*) o/(o+eps_pv), f = 1/2          [lrp_attn_bwd_prep, also D = rowsum(Gho*o)]
 *   dP = Gho V^T ; dV = P^T Gho ; dS3 = P (*) (dP - D)
 *   Ghs = dS3 * scale * s2/(s2+eps_mask) * s/(2 s+eps_qk)   (eps==0 -> 1/2) ; s2 = s*scale
 *   dQ = Ghs K ; dK = Ghs^T Q
 *   dK/dV are produced PER QUERY HEAD ([B*S, Hq, d]) so the grid has Hq-way parallelism and no
 *   atomics; lrp_gqa_reduce sums the heads of a kv group (ref: HF repeat_kv's autograd sum).
 * --------------------------------------------------------------------------------------- */
int lrp_transpose_heads(const void* x, void* xt, int B, int S, int H, int d, int64_t ldx,
                        int64_t ldt, int dtype, void* stream);
/* 1 if the kernels serving (dtype, d) read the head-transposed "_t" copies, 0 if they take every operand from the
 * token-major tensors (bf16, d in {64, 96, 128, 256}: 32x32x16 MFMA kernels, transposed fragments by ds_read_b64_tr_b16
 * out of the row-major LDS tile).  With 0 the "_t" arguments below may be NULL and no lrp_transpose_heads launch is needed. */
int lrp_attn_needs_transposed(int dtype, int d);
int lrp_attn_fwd(const void* q, const void* k, const void* v, const void* v_t, void* o, float* lse,
                 int B, int S, int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv,
                 int64_t ldt, int64_t ldo, float scale, int causal, int window, int q_begin,
                 const int* row_lo, const int* row_hi, int dtype, void* stream);
int lrp_attn_bwd_prep(const void* Go, const void* o, void* Gho, float* D, int B, int S, int Hq,
                      int d, int64_t ldgo, int64_t ldo, int64_t ldgho, float eps_pv,
                      float factor, int dtype, void* stream);
int lrp_attn_bwd_dq(const void* q, const void* k, const void* v, const void* k_t, const void* Gho,
                    const float* lse, const float* D, void* dq, int B, int S, int Hq, int Hkv,
                    int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldt, int64_t ldgho,
                    int64_t lddq, float scale, float eps_mask, float eps_qk, int causal,
                    int window, int q_begin, const int* row_lo, const int* row_hi, int dtype,
                    void* stream);
/* lxt.efficient placement, bf16 d in {64, 96, 128} (lrp_attn_bwd_dq_d_ok): the dQ kernel forms D_i = sum_d Gho_i o_i itself from the rows it
 * holds anyway and WRITES D for lrp_attn_bwd_dkv -- no lrp_attn_bwd_prep pass (Gho = 1/2 x the o-projection's dgrad comes out of
 * lrp_gemm_nn_rs; ref lxt/efficient/patches.py:193-203).  Dense calls only (q_begin = 0). */
int lrp_attn_bwd_dq_d_ok(int dtype, int d);
int lrp_attn_bwd_dq_d(const void* q, const void* k, const void* v, const void* Gho, const void* o, const float* lse, float* D, void* dq,
                      int B, int S, int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldgho, int64_t ldo, int64_t lddq,
                      float scale, int causal, int window, const int* row_lo, const int* row_hi, const float* cos_t, const float* sin_t,
                      int dtype, void* stream);
/* cos_t / sin_t (fp32 [S, d] in the rotate-half convention, i.e. columns c and c + d/2 hold the SAME value -- only the first half is read;
 * NULL = none; d = 64 or 128): RoPE's backward applied to dQ on its way out of the kernel -- and, for dK, inside the
 * group sum: lrp_gqa_reduce_rope(dk_h [rows, Hkv rep d] -> out [rows, Hkv d]) sums the rep query heads of a kv head and applies the transposed
 * rotation (rotate-half pairs (c, c + d/2), position = row % seq) in one pass.  Together they replace the lrp_rope_bwd launch of the
 * lxt.efficient placement (no stabiliser on the rotation: lxt/explicit/models/llama.py:226-260 with eps = 0 = HF's apply_rotary_pos_emb VJP). */
int lrp_gqa_reduce_rope(const void* in, void* out, int64_t rows, int seq, int Hkv, int rep, int d, int64_t ld_in, int64_t ld_out,
                        const float* cos_t, const float* sin_t, int dtype, void* stream);
int lrp_attn_bwd_dkv(const void* q, const void* k, const void* v, const void* q_t, const void* Gho,
                     const void* Gho_t, const float* lse, const float* D, void* dk_h, void* dv_h,
                     int B, int S, int Hq, int Hkv, int d, int64_t ldq, int64_t ldk, int64_t ldv,
                     int64_t ldt, int64_t ldgho, int64_t lddk, int64_t lddv, float scale,
                     float eps_mask, float eps_qk, int causal, int window, int q_begin,
                     const int* row_lo, const int* row_hi, int dtype, void* stream);
/* out[row, hk, :] = sum_{g<rep} in[row, hk*rep+g, :]   (in: Hkv*rep heads, out: Hkv heads) */
int lrp_gqa_reduce(const void* in, void* out, int64_t rows, int Hkv, int rep, int d, int64_t ld_in,
                   int64_t ld_out, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * explicit-API row ops on materialised scores (lf.softmax / lf.add2 / lf.matmul users).
 *   ref: lxt/explicit/functional.py:293-322.  x,p,R [rows, n]; -inf entries of x -> 0.
 *   lrp_softmax_fwd: p = softmax(x/T) ; lrp_softmax_rule_bwd: Rx = (x/T) (*) (Rp - p * sum(Rp))
 * --------------------------------------------------------------------------------------- */
int lrp_softmax_fwd(const void* x, void* p, int64_t rows, int n, float inv_temp, int dtype,
                    void* stream);
int lrp_softmax_rule_bwd(const void* x, const void* p, const void* Rp, void* Rx, int64_t rows,
                         int n, float inv_temp, int dtype, void* stream);
/* add2 rule: s = R/(a+b+eps) ; Ra = s*a ; Rb = s*b (Rb may be NULL). ref: functional.py:430-459 */
int lrp_add2_rule_bwd(const void* a, const void* b, const void* R, void* Ra, void* Rb,
                      int64_t n, float eps, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * K9 / harness.  ref: docs/source/quickstart.rst:120-141, examples/paper/llama.py:45-46
 *   lrp_readout      : R_tok[row] = sum_h emb[row,h]*G[row,h]   (fp32 out)
 *   lrp_head_seed    : last-token head, one launch per prompt batch:
 *        Gxn[b,:] = zfac * W_lm[idx[b],:]  with zfac = z/(z+eps_lin), z = logits[b,idx[b]]
 *        then the final-norm identity rule: Gh_last[b,:] = Gxn (*) w' * rstd[b]
 *   lrp_argmax_rows  : idx[b] = argmax_v logits[b,:] (fp32 logits), val[b] = max
 *   lrp_transpose    : out[b][c,r] = in[b][r,c]  (weight W^T copies, operand re-layouts)
 *   lrp_cast         : dtype conversion fp32 <-> bf16
 * --------------------------------------------------------------------------------------- */
int lrp_readout(const void* emb, const void* G, float* R_tok, int M, int H, int dtype, void* stream);
int lrp_head_seed(const void* W_lm, const float* logits, const int* idx, const void* w_norm,
                  const float* rstd_last, void* Gh_last, int B, int V, int H, int64_t ld_logits,
                  float w_offset, float eps_lin, int dtype, void* stream);
int lrp_argmax_rows(const float* logits, int* idx, float* val, int B, int V, int64_t ld, void* stream);
int lrp_transpose(const void* in, void* out, int rows, int cols, int64_t ld_in, int64_t ld_out,
                  int batch, int64_t s_in, int64_t s_out, int dtype, void* stream);
int lrp_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LRP_HIP_H */
