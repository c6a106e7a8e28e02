// scrf_kernels.h -- launch wrappers of the HIP kernels (scrf_kernels.hip, scrf_mfma.hip).
#ifndef SCRF_KERNELS_H_
#define SCRF_KERNELS_H_

#include "scrf_common.h"

#include <vector>

void launch_windows(hipStream_t st, const float* frames, const uint64_t* sframe_off, ScrfBatchView bv,
                    uint32_t u0, uint32_t u1, uint64_t n_frames, uint32_t W, uint32_t D, uint32_t lctx,
                    uint32_t rctx, int extract, float* X, uint32_t F, uint32_t out_col, int first_only = 0);
void launch_frame_rows(hipStream_t st, ScrfBatchView bv, uint32_t u0, uint32_t u1, uint32_t D,
                       uint64_t n_frames, uint64_t* xrow, int next);
void launch_scores_exact(hipStream_t st, const float* X, uint32_t F, const uint64_t* xrow, uint64_t n_rows,
                         const double* lambda, const ScrfLayout& lay, int is_trans, uint32_t n_out, double* out);
size_t fb_smem_bytes(const ScrfLayout& lay, int NT);
int fb_block_threads(const ScrfLayout& lay);
void launch_fb(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
               const double* S, const double* M, int m_per_frame, double* AD, double* alpha_g, double* beta_g,
               double* XI, double* xi_acc, double* numer, double* zx, int* status, int write_post, int frame_model = 0);
void launch_expf_gemm(hipStream_t st, const double* A, uint32_t n_out, const float* X, uint32_t F,
                      const uint64_t* xrow, uint64_t n_rows, const ScrfLayout& lay, int is_trans,
                      uint64_t rows_per_chunk, uint32_t n_chunks, double* slab);
void launch_reduce_slabs(hipStream_t st, const double* slab, uint32_t n_chunks, uint32_t n_out,
                         const ScrfLayout& lay, const ScrfGemmSpec& sp, double* grad);
void launch_reduce_xiacc(hipStream_t st, const double* xi_acc, uint32_t n_utts, const ScrfLayout& lay,
                         double* grad);
void launch_batch_sums(hipStream_t st, const double* numer, const double* zx, uint32_t n, double* sums3);
void launch_viterbi(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                    const double* S, const double* M, int m_per_frame, int frame_model, uint16_t* bp_b,
                    uint16_t* bp_e, uint32_t* out_labels, uint32_t* out_n, float* out_cost,
                    const float* Wn = nullptr);   // Wn: float(-1 * score) per (segment, label), replaces S
void launch_arcs(hipStream_t st, const ScrfLayout& lay, uint32_t T, int frame_model, const double* S,
                 const double* M, int m_per_frame, float final_w, scrf_arc* arcs);
void launch_sgd_step(hipStream_t st, double* lambda, double* lambda_acc, double* gsa, double* grad, uint32_t n,
                     double lr, int adagrad, double eps);
void launch_scale(hipStream_t st, double* v, uint32_t n, double s, int divide);
void launch_add(hipStream_t st, double* y, const double* x, uint32_t n);

// scrf_segtrans.hip: STDSEG_NO_DUR (one transition matrix per window): workgroup-per-utterance log-domain recursion
size_t fb_segtrans_smem_bytes(const ScrfLayout& lay, int NT);
void launch_zero_initial_rows(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                              uint32_t D, uint32_t L, double* M2);
void launch_fb_segtrans(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                        const uint32_t* prev_lab, const double* S, const double* M2, double* AD, double* alpha_g,
                        double* beta_g, double* XI2, double* numer, double* zx, int* status, int write_post);

uint64_t segtrans_num_arcs(uint32_t T, uint32_t L, uint32_t D);
void launch_arcs_segtrans(hipStream_t st, const ScrfLayout& lay, uint32_t T, const double* S, const double* M2, float final_w,
                          scrf_arc* arcs);
void launch_viterbi_segtrans(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                             const double* S, const double* M2, uint16_t* bp_p, uint16_t* bp_d, uint32_t* out_labels,
                             uint32_t* out_n, float* out_cost);

// scrf_mfma.hip: fp64 MFMA contractions (FAST training precision)
void launch_scores_mfma(hipStream_t st, const float* X, uint32_t F, const uint64_t* xrow, uint64_t n_rows,
                        const double* lambda, const ScrfLayout& lay, const ScrfGemmSpec& sp, uint32_t n_out,
                        double* out, int f32 = 0);
void launch_expf_mfma(hipStream_t st, const double* A, uint32_t n_out, const float* X, uint32_t F,
                      const uint64_t* xrow, uint64_t n_rows, const ScrfLayout& lay, const ScrfGemmSpec& sp,
                      uint64_t rows_per_chunk, uint32_t n_chunks, double* slab, int f32 = 0);
// workgroups launch_expf_mfma starts per row chunk when it takes the 8-wavefront form (one workgroup per CU), else 0:
// the engine sizes the number of row chunks so that the launch fills whole rounds of the chip's CUs
uint32_t expf_mfma_wide_tiles(uint32_t n_out, uint32_t nfun, int f32);

// scrf_dp.hip: wavefront-per-utterance DP and the parallel posterior kernels
int dp_wave_supported(const ScrfLayout& lay);
int atb_supported(const ScrfLayout& lay);
void launch_exp_m(hipStream_t st, const double* M, uint32_t L, uint64_t n_mat, double* E, double* ET,
                  double* mshift);
void launch_dp_wave(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                    const double* S, const double* E, const double* ET, const double* mshift, int m_per_frame,
                    double* AD, double* alpha_g, double* beta_g, double* sd_g, double* zx, int* status);
void launch_post_state(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                       uint64_t n_frames, const uint32_t* next_lab, const double* S, const double* M,
                       int m_per_frame, double* AD, const double* beta_g, const double* zx, double* numer_f,
                       int* status, double* mass_s);
void launch_numer_reduce(hipStream_t st, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* numer_f,
                         double* numer);
void launch_xi_factors(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                       uint64_t n_frames, const double* alpha_g, const double* sd_g, const double* zx, double* A,
                       double* B);
void launch_atb(hipStream_t st, const ScrfLayout& lay, const double* A, const double* B, uint64_t n_frames,
                uint64_t rows_per_chunk, uint32_t n_chunks, double* slab, const double* M0, double* grad,
                const double* bscale = nullptr);   // bscale[f]: factor of row f of B (linear-domain path)
void launch_add_trans_counts(hipStream_t st, const uint32_t* counts, const ScrfLayout& lay, double* grad);
void launch_xi_full(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                    uint64_t n_frames, const uint32_t* next_lab, const double* A, const double* B, const double* E,
                    const double* mshift, double* XI);

// scrf_dplin.hip: scaled linear-domain forward/backward + posteriors (training path, L <= 64)
int dplin_mw_supported(const ScrfLayout& lay);
int dplin_supported(const ScrfLayout& lay);
void launch_true_scores(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                        uint64_t n_frames, const double* S, double* s_true);
void launch_exp_rows(hipStream_t st, double* S, uint64_t n_rows, uint32_t L, double* smax);
void launch_dp_lin(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                   const double* ES, const double* smax, const double* E, const double* ET, const double* mshift,
                   int m_per_frame, const ScrfDpLin& o, double* zx, int* status);
void launch_post_lin(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                     uint64_t n_frames, const uint32_t* next_lab, const double* s_true, const double* M,
                     int m_per_frame, double* ES, const double* smax, const ScrfDpLin& o, const double* zx,
                     double* numer_f, int* status, double* mass_s);
// posterior-mass self-checks per frame (state mass from the posterior kernel vs sum_c exp(alpha + beta - Zx));
// lin: a/b are mantissa vectors with log-scales ga/gb, else log-domain arrays (ga, gb unused)
void launch_mass_check(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                       uint32_t L, int frame_model, int lin, const double* a, const double* ga, const double* b,
                       const double* gb, const double* zx, const double* mass_s, int* status);
void launch_xi_lin(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                   uint64_t n_frames, const ScrfDpLin& o, const double* zx);
void launch_xi_scale(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                     const ScrfDpLin& o, const double* zx);
void launch_lin_to_log(hipStream_t st, uint64_t n_frames, uint32_t L, const double* m, const double* g, double* out);

// scrf_fused.hip: state contractions with the window synthesis fused in (X never materialised)
#define SCRF_FUSED_ROWS_SCORES 256
uint32_t fused_scores_tb(uint32_t W, uint32_t D);   // whole frames per score tile (0: shape not supported)
#define SCRF_FUSED_ROWS_EXPF 76
uint32_t fused_expf_frames(uint32_t D);            // whole frames per expected-count tile (<= 76 rows)
int fused_supported(const ScrfLayout& lay, uint32_t W, int f32 = 0);
// SCRF_PREC_FASTLIN (`la`): the window average leaves both dense contractions (prefix sums of a sixth per-frame
// projection / a sixth group of Z); shapes this returns 0 for run the FAST kernels under that precision
int fused_la_supported(const ScrfLayout& lay, uint32_t W);
// slabs of the fused expected-count kernel: slab [blocks][L][ncol] (dense groups from g0: 0 avg, 1 max) and, when
// ndur > 0, a separate duration slab [blocks][L][ndur] (one-hot duration counts + bias; wave-specialised kernel)
// rows / frames: height of the row tiles the kernel walks (whole frames); tile_list: which of the batch's tile lists
// describes them (1: <= 76 rows, 2: <= 100 rows, built only for batches of an SCRF_PREC_FASTLIN engine)
struct ScrfFusedExpfPlan { int ws; int g0; uint32_t ncol; uint32_t ndur; uint32_t rows; uint32_t frames; int tile_list; uint32_t nz; };   // nz: workgroup columns of the launch (one dense statistic each when > 1)
ScrfFusedExpfPlan fused_expf_plan(const ScrfLayout& lay, uint32_t W, int f32, int la);
uint32_t fused_expf_blocks(const ScrfLayout& lay, uint32_t W, int f32, uint64_t n_tiles, int la);   // workgroups (= slabs) of launch_expf_fused, <= 512
void launch_scores_fused(hipStream_t st, const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* lambda,
                         const double* P, uint64_t n_tiles, double* S, int f32, double* smax = nullptr,
                         double* s_true = nullptr, const uint32_t* labels = nullptr, int la = 0);
size_t fused_tile_table_bytes(uint32_t D, uint32_t TB);
void launch_tile_tables(hipStream_t st, uint32_t D, uint32_t TB, int la, void* rtab);
size_t fused_dur_table_doubles(const ScrfLayout& lay);
void launch_dur_table(hipStream_t st, const ScrfLayout& lay, uint32_t W, const double* lambda, double* dtab);
void launch_avg_prefix(hipStream_t st, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, uint32_t L, double* P);
// k_viterbi on float arc weights with one wavefront per utterance (fast decode; L <= 64, constant M)
int viterbi_fast_supported(const ScrfLayout& lay);
void launch_viterbi_fast(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const float* Wn,
                         const double* M, uint16_t* bp_b, uint16_t* bp_e, uint32_t* out_labels, uint32_t* out_n, float* out_cost);
// decode mode: float arc weights + the list of entries to recompute (ScrfDecodeOut)
void launch_scores_fused_decode(hipStream_t st, const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* lambda,
                                const double* P, uint64_t n_tiles, const ScrfDecodeOut& dz);
// w1[o] = sum of |lambda| over label o's state block (features + bias weight)
void launch_state_l1(hipStream_t st, const double* lambda, const ScrfLayout& lay, double* w1);
// reference-order recomputation of the listed (row, output) arc weights from the raw frames
void launch_decode_fixup(hipStream_t st, const float* frames, uint32_t W, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                         const double* lambda, const ScrfLayout& lay, const uint32_t* cnt, const uint64_t* list,
                         uint32_t cap, float* wneg);
void launch_lin_z(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                  const double* R, double* Z);
int pframe_supported(uint32_t W);
void launch_pframe(hipStream_t st, const float* F, uint32_t W, uint64_t n_frames, const double* lambda,
                   const ScrfLayout& lay, uint32_t n_out, double* P);
void launch_ztf(hipStream_t st, const double* Zm, uint32_t n_out, const float* F, uint32_t W, uint64_t n_frames,
                uint64_t rows_per_chunk, uint32_t n_chunks, double* slab, int narrow = 0);
uint32_t lin_z5_segments(uint32_t n_utts, uint32_t L, uint32_t D, uint32_t t_max, int* seg_len);
void launch_lin_z5(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, uint32_t t_max, uint64_t n_frames,
                   const double* R, double* Z, double* dslab);
void launch_add_p_exp(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                      const double* P, double* S, double* smax, double* s_true);
void launch_post_z(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                   const uint32_t* next_lab, const double* s_true, const double* M, int m_per_frame, double* ES,
                   const double* smax, const ScrfDpLin& o, const double* zx, double* numer_f, int* status, double* Z,
                   double* mass_s, int la = 0, uint32_t t_max = 0, uint64_t n_frames = 0);
void launch_expf_fused(hipStream_t st, const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* R,
                       uint64_t n_tiles, double* slab, double* dslab, int f32, int la);

// ---- STDSEG (scrf_stdseg.hip): lay is the layout over FULL labels (lay.L = nLabs), La = nActualLabs
void launch_stdseg_rowinfo(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                           uint32_t D, uint32_t* row_t, uint32_t* row_d, uint32_t* row_u);
void launch_stdseg_scores(hipStream_t st, const ScrfLayout& lay, uint32_t La, const float* X, uint64_t n_rows,
                          const uint32_t* row_t, const uint32_t* row_d, const double* lambda, double* S, double* MX);
void launch_stdseg_fb(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                      const double* S, const double* MX, double* alpha, double* beta, double* zx, int* status);
void launch_stdseg_post(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                        uint64_t n_rows, const uint32_t* row_t, const uint32_t* row_d, const uint32_t* row_u,
                        const uint32_t* prev_lab, const double* S, const double* MX, const double* alpha, const double* beta,
                        const double* zx, double* G, double* XI, double* mass_s, double* mass_t, double* numer, int* status);
void launch_stdseg_expf(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, uint64_t n_rows,
                        const uint32_t* row_t, const uint32_t* row_d, const uint32_t* row_u, const uint32_t* prev_lab,
                        const float* X, const double* G, const double* XI, double* grad);
// scrf_stdseg_lin.hip: STDSEG with bias-only transitions on the training path: linear-domain recursion against one
// exp(M) table, node arrays duration-major [D][frames][La], transition counts as E o (A^T B) on the MFMA
int stdseg_lin_supported(const ScrfLayout& lay, uint32_t La);
void launch_sl_tables(hipStream_t st, const ScrfLayout& lay, const double* lambda, double* E, double* ET, double* mmax);
void launch_sl_rows(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames, uint32_t D, uint64_t* xrow);
void launch_sl_fb(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, uint64_t n_frames,
                  const double* Sd, const double* E, const double* ET, const double* mmax, double* Ad, double* Bd, double* Am,
                  double* ga, double* zx, int* status);
void launch_sl_post(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                    uint32_t n_utts, uint64_t n_frames, const uint32_t* prev_lab, const double* lambda, const double* Sd,
                    const double* Ad, const double* Bd, const double* ga, const double* zx, double* Rd, double* Bp,
                    double* numer_f, double* numer, int* status);
void launch_sl_trans_counts(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                            uint64_t n_frames, const uint32_t* prev_lab, uint64_t rows_per_chunk, uint32_t n_chunks,
                            const double* Am, const double* Bp, const double* E, const double* mmax, double* slab, double* obs,
                            double* grad);
void launch_stdseg_sums(hipStream_t st, const double* numer, const double* zx, uint32_t u0, uint32_t n, double* sums);
uint64_t stdseg_num_arcs(uint32_t T, uint32_t La, uint32_t D);
void stdseg_row_arc_offsets(uint32_t T, uint32_t La, uint32_t D, std::vector<uint64_t>* off);
void launch_stdseg_arcs(hipStream_t st, const ScrfLayout& lay, uint32_t La, uint32_t T, uint64_t n_rows, const uint64_t* row_arc,
                        const double* S, const double* MX, float final_w, scrf_arc* arcs);
void launch_stdseg_viterbi(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                           const double* S, const double* MX, float* vc, uint16_t* bp, uint32_t* out_labels, uint32_t* out_n,
                           float* out_cost);

// ---- n-state frame model (scrf_nstate.hip): lay.K > 1
void launch_ns_scores(hipStream_t st, const ScrfLayout& lay, const float* X, uint64_t n_frames, const double* lambda, double* S,
                      double* TD, double* TO, double* TE);
void launch_ns_fb(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* S,
                  const double* TD, const double* TO, const double* TE, double* alpha, double* beta, double* zx, int* status);
void launch_ns_post(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint32_t n_utts,
                    uint64_t n_frames, const double* S, const double* TD, const double* TO, const double* TE, const double* alpha,
                    const double* beta, const double* zx, double* G, double* XD, double* XO, double* XE, double* mass_s,
                    double* mass_t, double* numer, int* status);
uint32_t ns_expf_slices(uint64_t n_frames);   // frame slices of the gradient kernel: slab is [slices][lambda_len]
void launch_ns_expf(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                    const float* X, const double* G, const double* XD, const double* XO, const double* XE, double* slab, double* grad);
uint64_t ns_num_arcs(uint32_t T, uint32_t L, uint32_t K);
void launch_ns_arcs(hipStream_t st, const ScrfLayout& lay, uint32_t T, const double* S, const double* TD, const double* TO,
                    const double* TE, float final_w, scrf_arc* arcs);
void launch_ns_viterbi(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* S,
                       const double* TD, const double* TO, const double* TE, float* vc, uint16_t* bp, uint32_t* out_labels,
                       uint32_t* out_n, float* out_cost);

#endif  // SCRF_KERNELS_H_
