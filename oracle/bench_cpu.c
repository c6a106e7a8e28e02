/*
 * bench_cpu.c -- the reference's data-parallel CPU path restated: N POSIX threads over
 * utterances, one gradient buffer per thread, summed in stream order and divided by the
 * number of active streams.  TEST INFRASTRUCTURE ONLY: used by tests/ and by bench.py's
 * cpu_baseline leg (the reported baseline, never the product).
 *
 * Follows trainers/accumulators/CRF_Minibatch_GradAccumulator.cpp:20-97 (thread run
 * loop), :201-322 (accumulateGradient) and io/CRF_FeatureStreamManager.cpp:425-464
 * (child i views a contiguous utterance range).  Each utterance runs the full
 * CRF_NewGradBuilder_StdSeg_NoDur_NoTrans::buildGradient path including segment-window
 * synthesis (orc_windows), like ftr_strm->read() does in the reference.
 *
 * Round 4: (a) a worker keeps its arrays across utterances (orc_workspace), as the reference's threads keep their node
 * vector and builder (nodes/CRF_StateVector.cpp:37-67) -- the previous malloc/free of ~25 MB per utterance made 64
 * threads fight over mmap and page faults; (b) workers can be pinned, one per CPU of a list the caller reads from the
 * host topology (orc_bench_set_cpus: the physical cores of one socket); (c) a second input stream with context frames
 * (the TIMIT demo's transition features, demo/segmental-timit-demo.cfg.in:21-24) for BASELINE config 3.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "scrf_oracle.h"

typedef struct {
  const orc_config* cfg;
  const orc_layout* lay;
  const double* lambda;
  const float* frames;      /* packed [sum_u (T_u)][in_width], lctx = rctx = 0 */
  const uint32_t* labels;   /* packed [sum_u T_u] */
  const uint64_t* frame_off; /* [U+1] */
  uint32_t in_width;
  const float* frames2;     /* second stream or NULL: packed [sum_u (T_u + 2 ctx)][in_width2], windows without the segment recipe */
  uint32_t in_width2, ctx2;
  int cpu;                  /* CPU to pin to, or -1 */
  uint32_t u_begin, u_end;
  double* grad;   /* [lambda_len], zeroed by caller */
  double* numer;  /* [U] */
  double* zx;     /* [U] */
  int err;
  double phase_us[5];   /* featLoad, transMat, alpha, beta, expF of this worker (gradbuilder :155-157) */
} worker_arg;
extern __thread double* orc_phase_us;
static double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}

static void* worker(void* p) {
  worker_arg* a = (worker_arg*)p;
  const orc_config* cfg = a->cfg;
  const uint32_t D = cfg->lab_max_dur, F = cfg->num_feas;
  float* seg = NULL;
  uint64_t seg_cap = 0;
  orc_workspace ws;
  memset(&ws, 0, sizeof(ws));
  if (a->cpu >= 0) {
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(a->cpu, &set);
    pthread_setaffinity_np(pthread_self(), sizeof(set), &set);   /* best effort: a refusal leaves the thread floating */
  }
  const uint32_t w1 = orc_window_width(a->in_width, D, 0, 0, 1);
  orc_phase_us = a->phase_us;
  for (uint32_t u = a->u_begin; u < a->u_end; u++) {
    uint32_t T = (uint32_t)(a->frame_off[u + 1] - a->frame_off[u]);
    uint64_t nseg = orc_num_segs(T, D);
    if (nseg > seg_cap) {
      free(seg);
      seg = (float*)malloc(sizeof(float) * nseg * F);
      seg_cap = nseg;
    }
    const double tw = now_us();
    orc_windows(a->frames + a->frame_off[u] * a->in_width, T, a->in_width, D, 0, 0, 1, seg, F, 0);
    if (a->frames2)   /* utterance u starts at frame_off[u] + 2 ctx u of the padded stream */
      orc_windows(a->frames2 + (a->frame_off[u] + 2ull * a->ctx2 * u) * a->in_width2, T, a->in_width2, D, a->ctx2, a->ctx2, 0,
                  seg, F, w1);
    a->phase_us[0] += now_us() - tw;
    int e = orc_seg_build_gradient_ws(cfg, a->lay, a->lambda, seg, a->labels + a->frame_off[u], T,
                                      a->grad, &a->numer[u], &a->zx[u], &ws);
    if (e != ORC_OK && a->err == ORC_OK) a->err = e;
  }
  free(seg);
  orc_workspace_free(&ws);
  orc_phase_us = NULL;
  return NULL;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Runs forward-backward over utterances [0,U) with n_threads workers; stream s views
 * the contiguous range [s*floor(U/N), (s+1)*floor(U/N)), the last one takes the
 * remainder.  grad_out = sum_s sgrad[s] / n_active.  Returns wall seconds of the
 * threaded region through *seconds. */
static double g_phase_us[5];
/* CPUs the workers of the next orc_bench_fb* calls are pinned to (worker s -> cpus[s % n]); n = 0: no pinning */
static int g_cpus[1024];
static int g_ncpus = 0;
void orc_bench_set_cpus(const int* cpus, int n) {
  g_ncpus = n < 0 ? 0 : (n > 1024 ? 1024 : n);
  for (int i = 0; i < g_ncpus; i++) g_cpus[i] = cpus[i];
}
/* the five phase timers of the last orc_bench_fb call, microseconds summed over the workers */
void orc_bench_phases(double* out5) { memcpy(out5, g_phase_us, sizeof(g_phase_us)); }

int orc_bench_fb(const orc_config* cfg, const double* lambda, const float* frames,
                 const uint32_t* labels, const uint64_t* frame_off, uint32_t U, uint32_t in_width,
                 uint32_t n_threads, double* grad_out, double* numer, double* zx,
                 double* seconds) {
  return orc_bench_fb2(cfg, lambda, frames, NULL, 0, 0, labels, frame_off, U, in_width, n_threads, grad_out, numer, zx, seconds);
}

int orc_bench_fb2(const orc_config* cfg, const double* lambda, const float* frames, const float* frames2,
                  uint32_t in_width2, uint32_t ctx2, const uint32_t* labels, const uint64_t* frame_off, uint32_t U,
                  uint32_t in_width, uint32_t n_threads, double* grad_out, double* numer, double* zx, double* seconds) {
  orc_layout lay;
  int rc = orc_layout_init(cfg, &lay);
  if (rc != ORC_OK) return rc;
  if (n_threads == 0) n_threads = 1;
  if (n_threads > U) n_threads = U ? U : 1;
  double* sgrad = (double*)calloc((size_t)n_threads * lay.lambda_len, sizeof(double));
  worker_arg* args = (worker_arg*)calloc(n_threads, sizeof(worker_arg));
  pthread_t* th = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
  int32_t* active = (int32_t*)calloc(n_threads, sizeof(int32_t));
  uint32_t per = U / n_threads;
  double t0 = now_s();
  for (uint32_t s = 0; s < n_threads; s++) {
    worker_arg* a = &args[s];
    a->cfg = cfg; a->lay = &lay; a->lambda = lambda; a->frames = frames; a->labels = labels;
    a->frame_off = frame_off; a->in_width = in_width;
    a->frames2 = frames2; a->in_width2 = in_width2; a->ctx2 = ctx2;
    a->cpu = g_ncpus ? g_cpus[s % (uint32_t)g_ncpus] : -1;
    a->u_begin = s * per;
    a->u_end = (s == n_threads - 1) ? U : (s + 1) * per;
    a->grad = sgrad + (size_t)s * lay.lambda_len;
    a->numer = numer; a->zx = zx; a->err = ORC_OK;
    active[s] = a->u_end > a->u_begin;
    pthread_create(&th[s], NULL, worker, a);
  }
  memset(g_phase_us, 0, sizeof(g_phase_us));
  for (uint32_t s = 0; s < n_threads; s++) {
    pthread_join(th[s], NULL);
    if (args[s].err != ORC_OK && rc == ORC_OK) rc = args[s].err;
    for (int i = 0; i < 5; i++) g_phase_us[i] += args[s].phase_us[i];
  }
  double t1 = now_s();
  if (seconds) *seconds = t1 - t0;
  if (grad_out) orc_minibatch_reduce(sgrad, n_threads, active, lay.lambda_len, grad_out);
  free(sgrad); free(args); free(th); free(active);
  orc_layout_free(&lay);
  return rc;
}
