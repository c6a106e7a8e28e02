// scrf_stub.cpp -- TEST-ONLY link-time stand-in for libscrf_amd.so (the C ABI of include/scrf_abi.h), so that the host
// layer's multi-rank control flow (asr-craft_amd/host/crf_amd.cpp: rank = stream, share split, all-reduce, / active,
// end-of-epoch protocol, rank-0-only writers, the communicator-id handshake) can run as N real processes without N GPUs.
// It never ships: tests/test_host_multirank.py links bin-less copies of CRFTrain_main.cpp + crf_amd.cpp against it.
//
// What it computes is CANNED, not the model: a batch's gradient is a deterministic function of its utterances and of
// lambda (so a wrong order, share or divisor changes every later step), the optimiser step is the real arithmetic
// (lambda += lr g, lambdaAcc += lambda, g = 0), and the "collective" is files in $SCRF_STUB_COMM_DIR, summed in rank
// order.  tests/test_host_multirank.py holds the same function in Python and replays the reference's protocol
// (trainers/accumulators/CRF_Minibatch_GradAccumulator.cpp:229-241, 296-312) to predict the weight files.
//   SCRF_STUB_FAIL_RANK / SCRF_STUB_FAIL_AT: rank and 1-based scrf_fb_batch call that fails with SCRF_ERR_NUMERIC.
//   SCRF_STUB_COLL_FAIL_RANK / SCRF_STUB_COLL_FAIL_AT: rank and 1-based all-reduce call that itself returns an error
//     WITHOUT publishing (what a HIP error around ncclAllReduce looks like to the host); SCRF_STUB_COLL_DIE=1 makes
//     that rank _exit(9) there instead (a process that is killed).  The collective waits under the same watchdog
//     contract as the engine's: SCRF_COMM_TIMEOUT_S seconds (default 60 here), and a peer's scrf_comm_abort (a marker
//     file, standing in for the asynchronous error ncclCommAbort raises on the peers) ends the wait at once.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "scrf_abi.h"

struct scrf_batch_s {
  std::vector<uint32_t> T;
  std::vector<uint64_t> key;   // per utterance: what the canned gradient depends on
};
struct scrf_engine_s {
  scrf_config cfg;
  uint32_t n = 0;
  std::vector<double> lambda, acc, gsa, grad;
  double sums[3] = {0, 0, 0};
  int rank = 0, world = 1, round = 0, step = 0, fb_calls = 0;
  bool comm = false;
  std::string comm_dir, err;
};
static std::string g_err;

static uint32_t stub_lambda_len(const scrf_config& c) {   // the one-state layout: L (nsf + L ntf)
  const uint32_t nsf = (c.use_state_ftrs ? c.state_fidx_end - c.state_fidx_start + 1 : 0) + (c.use_state_bias ? 1 : 0);
  const uint32_t ntf = (c.use_trans_ftrs ? c.trans_fidx_end - c.trans_fidx_start + 1 : 0) + (c.use_trans_bias ? 1 : 0);
  return c.num_labs * (nsf + c.num_labs * ntf);
}

extern "C" {
int scrf_create(const scrf_config* cfg, scrf_handle* out) {
  scrf_engine_s* h = new scrf_engine_s;
  h->cfg = *cfg;
  h->n = stub_lambda_len(*cfg);
  h->lambda.assign(h->n, 0.0); h->acc.assign(h->n, 0.0); h->gsa.assign(h->n, 0.0); h->grad.assign(h->n, 0.0);
  *out = h;
  return SCRF_OK;
}
int scrf_destroy(scrf_handle h) { delete h; return SCRF_OK; }
const char* scrf_last_error(scrf_handle h) { return h ? h->err.c_str() : g_err.c_str(); }
int scrf_lambda_len(scrf_handle h, uint32_t* n) { *n = h->n; return SCRF_OK; }
#define VEC_IO(name, member)                                                                                          \
  int scrf_set_##name(scrf_handle h, const double* v, uint32_t n) { if (n != h->n) return SCRF_ERR_INVALID; h->member.assign(v, v + n); return SCRF_OK; } \
  int scrf_get_##name(scrf_handle h, double* v, uint32_t n) { if (n != h->n) return SCRF_ERR_INVALID; memcpy(v, h->member.data(), sizeof(double) * n); return SCRF_OK; }
VEC_IO(lambda, lambda)
VEC_IO(lambda_acc, acc)
VEC_IO(grad_sqr_acc, gsa)
int scrf_zero_grad(scrf_handle h) { h->grad.assign(h->n, 0.0); h->sums[0] = h->sums[1] = h->sums[2] = 0.0; return SCRF_OK; }
int scrf_get_grad(scrf_handle h, double* g, uint32_t n) { if (n != h->n) return SCRF_ERR_INVALID; memcpy(g, h->grad.data(), sizeof(double) * n); return SCRF_OK; }
int scrf_get_batch_sums(scrf_handle h, double* s3) { memcpy(s3, h->sums, sizeof(h->sums)); return SCRF_OK; }
static double g_queued_sums[3];
int scrf_queue_batch_sums(scrf_handle h) { memcpy(g_queued_sums, h->sums, sizeof(g_queued_sums)); return SCRF_OK; }
int scrf_take_batch_sums(scrf_handle, double* s3) { memcpy(s3, g_queued_sums, sizeof(g_queued_sums)); return SCRF_OK; }
int scrf_scale_grad(scrf_handle h, double s) { for (double& g : h->grad) g *= s; return SCRF_OK; }
int scrf_div_grad(scrf_handle h, double d) { for (double& g : h->grad) g /= d; return SCRF_OK; }
int scrf_gauss_prior(scrf_handle h, float inv) { for (double& g : h->grad) g -= g * inv; return SCRF_OK; }
int scrf_sgd_step(scrf_handle h, double lr, int use_adagrad, double eps) {
  for (uint32_t i = 0; i < h->n; i++) {
    if (use_adagrad) {
      h->gsa[i] += h->grad[i] * h->grad[i];
      h->lambda[i] += lr * h->grad[i] / sqrt(h->gsa[i] + eps);
    } else {
      h->lambda[i] += lr * h->grad[i];
    }
    h->acc[i] += h->lambda[i];
    h->grad[i] = 0.0;
  }
  return SCRF_OK;
}

int scrf_batch_create(scrf_handle h, const scrf_utt* utts, uint32_t n, uint32_t n_streams, const scrf_stream_recipe* rec, scrf_batch* out) {
  scrf_batch_s* b = new scrf_batch_s;
  for (uint32_t u = 0; u < n; u++) {
    uint64_t key = (uint64_t)utts[u].T * 31u;
    if (utts[u].labels)
      for (uint32_t t = 0; t < utts[u].T; t++) if (utts[u].labels[t] != 0xffffffffu) key += utts[u].labels[t];
    if (n_streams && rec && utts[u].frames[0])
      for (uint32_t i = 0; i < utts[u].T * rec[0].in_width; i++) key += (uint64_t)(int64_t)(utts[u].frames[0][i] * 16.0f);
    b->T.push_back(utts[u].T);
    b->key.push_back(key);
  }
  (void)h;
  *out = b;
  return SCRF_OK;
}
int scrf_batch_destroy(scrf_handle, scrf_batch b) { delete b; return SCRF_OK; }
int scrf_batch_info(scrf_handle, scrf_batch b, uint32_t* nu, uint64_t* nf, uint64_t* ns, uint64_t* na) {
  uint64_t f = 0;
  for (uint32_t t : b->T) f += t;
  if (nu) *nu = (uint32_t)b->T.size();
  if (nf) *nf = f;
  if (ns) *ns = f;
  if (na) *na = 0;
  return SCRF_OK;
}
// canned per-utterance contribution: grad[i] += c_u[i] - floor(8 lambda[i]) / 1024 (multiples of 2^-10: sums are exact
// in any order, so one process with N streams and N processes agree to the bit), numerator -= T / 2, Zx += T / 4
int scrf_fb_batch(scrf_handle h, scrf_batch b, double* numer, double* zx) {
  h->fb_calls++;
  const char* fr = getenv("SCRF_STUB_FAIL_RANK");
  const char* fa = getenv("SCRF_STUB_FAIL_AT");
  if (fr && fa && atoi(fr) == h->rank && atoi(fa) == h->fb_calls) {
    h->err = "stub: injected numeric failure in utterance 0";
    return SCRF_ERR_NUMERIC;
  }
  double nm = 0.0, z = 0.0;
  for (size_t u = 0; u < b->T.size(); u++) {
    for (uint32_t i = 0; i < h->n; i++)
      h->grad[i] += ((double)((b->key[u] + 13ull * i) % 97ull) - 48.0) / 64.0 - floor(h->lambda[i] * 8.0) / 1024.0;
    nm -= b->T[u] * 0.5;
    z += b->T[u] * 0.25;
  }
  h->sums[0] += nm; h->sums[1] += z; h->sums[2] += (double)b->T.size();
  if (numer) *numer = nm;
  if (zx) *zx = z;
  return SCRF_OK;
}

// ---- the "collective": one file per rank and round in SCRF_STUB_COMM_DIR, summed in rank order by every rank
int scrf_comm_unique_id(void* id128) {
  unsigned char* p = (unsigned char*)id128;
  srand((unsigned)(time(nullptr) ^ getpid()));
  for (int i = 0; i < 128; i++) p[i] = (unsigned char)(rand() & 0xff);
  return SCRF_OK;
}
int scrf_comm_init(scrf_handle h, const void* id128, int rank, int n_ranks) {
  const char* d = getenv("SCRF_STUB_COMM_DIR");
  if (!d) { h->err = "stub: SCRF_STUB_COMM_DIR is not set"; return SCRF_ERR_COMM; }
  char tag[17];
  for (int i = 0; i < 8; i++) snprintf(tag + 2 * i, 3, "%02x", ((const unsigned char*)id128)[i]);   // ranks with different ids never meet
  h->comm_dir = std::string(d) + "/" + tag;
  mkdir(h->comm_dir.c_str(), 0777);
  h->rank = rank; h->world = n_ranks; h->comm = true;
  // collective, like ncclCommInitRank: nobody returns before every rank has the id (rank 0 deletes the id file next)
  FILE* f = fopen((h->comm_dir + "/init." + std::to_string(rank)).c_str(), "wb");
  if (!f) { h->err = "stub: cannot write into " + h->comm_dir; return SCRF_ERR_COMM; }
  fclose(f);
  const time_t deadline = time(nullptr) + 60;
  for (int r = 0; r < n_ranks; r++) {
    struct stat st;
    while (stat((h->comm_dir + "/init." + std::to_string(r)).c_str(), &st) != 0) {
      if (time(nullptr) > deadline) { h->err = "stub: rank " + std::to_string(r) + " never initialised"; return SCRF_ERR_COMM; }
      usleep(500);
    }
  }
  return SCRF_OK;
}
int scrf_comm_abort(scrf_handle h) {
  if (h && h->comm) {
    FILE* f = fopen((h->comm_dir + "/abort." + std::to_string(h->rank)).c_str(), "wb");
    if (f) fclose(f);
    h->comm = false;
  }
  return SCRF_OK;
}
static bool peer_aborted(scrf_handle h, int* who) {
  for (int r = 0; r < h->world; r++) {
    struct stat st;
    if (r != h->rank && stat((h->comm_dir + "/abort." + std::to_string(r)).c_str(), &st) == 0) { *who = r; return true; }
  }
  return false;
}
// one exchange: every rank publishes `mine` for the current round and sums all ranks' vectors in rank order
static int stub_exchange(scrf_handle h, const std::vector<double>& mine, std::vector<double>* tot) {
  const size_t len = mine.size();
  char name[64];
  snprintf(name, sizeof(name), "/r%06d.%d", h->round, h->rank);
  const std::string fin = h->comm_dir + name, tmp = fin + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f || fwrite(mine.data(), sizeof(double), len, f) != len || fclose(f) != 0 || rename(tmp.c_str(), fin.c_str()) != 0) { h->err = "stub: cannot publish " + fin; return SCRF_ERR_COMM; }
  tot->assign(len, 0.0);
  std::vector<double> part(len);
  const char* ts = getenv("SCRF_COMM_TIMEOUT_S");
  const time_t deadline = time(nullptr) + (ts && atoi(ts) > 0 ? atoi(ts) : 60);
  for (int r = 0; r < h->world; r++) {
    snprintf(name, sizeof(name), "/r%06d.%d", h->round, r);
    const std::string p = h->comm_dir + name;
    for (;;) {
      FILE* g = fopen(p.c_str(), "rb");
      if (g) {
        const size_t got = fread(part.data(), sizeof(double), len, g);
        fclose(g);
        if (got == len) break;
      }
      int who = -1;
      if (peer_aborted(h, &who)) { h->err = "stub: rank " + std::to_string(who) + " aborted the communicator (asynchronous error)"; h->comm = false; return SCRF_ERR_COMM; }
      if (time(nullptr) > deadline) { h->err = "stub: the collective did not complete within SCRF_COMM_TIMEOUT_S: rank " + std::to_string(r) + " never joined round " + std::to_string(h->round); h->comm = false; return SCRF_ERR_COMM; }
      usleep(500);
    }
    for (size_t i = 0; i < len; i++) (*tot)[i] += part[i];
  }
  h->round++;
  return SCRF_OK;
}
// The engine's protocol: with transition features two blocks per step -- the transition weights of every label, then
// the state weights + the scalars -- in that order on every rank, whichever entry point it uses (scrf_stub_layout tells
// the blocks apart the way ScrfLayout does); otherwise one block.
int scrf_allreduce_grad_ex(scrf_handle h, int active, const double* extra_in, uint32_t n_extra, double* sums4, double* extra_out) {
  if (!h->comm) { h->err = "stub: all-reduce without a communicator"; return SCRF_ERR_COMM; }
  if (getenv("SCRF_STUB_COLL_FAIL_RANK") && atoi(getenv("SCRF_STUB_COLL_FAIL_RANK")) == h->rank &&
      h->step + 1 == atoi(getenv("SCRF_STUB_COLL_FAIL_AT") ? getenv("SCRF_STUB_COLL_FAIL_AT") : "1")) {
    if (getenv("SCRF_STUB_COLL_DIE")) _exit(9);
    h->err = "stub: injected failure inside the collective";
    return SCRF_ERR_HIP;
  }
  h->step++;
  const scrf_config& c = h->cfg;
  const uint32_t nsf = (c.use_state_ftrs ? c.state_fidx_end - c.state_fidx_start + 1 : 0) + (c.use_state_bias ? 1 : 0);
  const uint32_t ntf = (c.use_trans_ftrs ? c.trans_fidx_end - c.trans_fidx_start + 1 : 0) + (c.use_trans_bias ? 1 : 0);
  const uint32_t stride = nsf + c.num_labs * ntf;
  std::vector<double> tot(h->n + 4 + n_extra, 0.0);
  std::vector<double> tail(4 + n_extra);
  tail[0] = h->sums[0]; tail[1] = h->sums[1]; tail[2] = h->sums[2]; tail[3] = active ? 1.0 : 0.0;
  for (uint32_t i = 0; i < n_extra; i++) tail[4 + i] = extra_in[i];
  if (c.use_trans_ftrs) {
    std::vector<double> blk, got;
    for (uint32_t i = 0; i < h->n; i++) if (i % stride >= nsf) blk.push_back(h->grad[i]);     // block 1: transition weights
    int rc = stub_exchange(h, blk, &got);
    if (rc != SCRF_OK) return rc;
    size_t k = 0;
    for (uint32_t i = 0; i < h->n; i++) if (i % stride >= nsf) tot[i] = got[k++];
    blk.clear();
    for (uint32_t i = 0; i < h->n; i++) if (i % stride < nsf) blk.push_back(h->grad[i]);      // block 0: state weights + scalars
    blk.insert(blk.end(), tail.begin(), tail.end());
    rc = stub_exchange(h, blk, &got);
    if (rc != SCRF_OK) return rc;
    k = 0;
    for (uint32_t i = 0; i < h->n; i++) if (i % stride < nsf) tot[i] = got[k++];
    for (size_t j = 0; j < tail.size(); j++) tot[h->n + j] = got[k++];
  } else {
    std::vector<double> mine(h->grad.begin(), h->grad.end()), got;
    mine.insert(mine.end(), tail.begin(), tail.end());
    const int rc = stub_exchange(h, mine, &got);
    if (rc != SCRF_OK) return rc;
    tot = got;
  }
  const double n_active = tot[h->n + 3];
  for (uint32_t i = 0; i < h->n; i++) h->grad[i] = n_active > 0 ? tot[i] / n_active : tot[i];
  for (int i = 0; i < 4; i++) sums4[i] = tot[h->n + i];
  for (uint32_t i = 0; i < n_extra; i++) extra_out[i] = tot[h->n + 4 + i];
  return SCRF_OK;
}
// the fused call: the batch, then the collective; the batch's failure travels as extra[fail_slot]
int scrf_fb_batch_allreduce(scrf_handle h, scrf_batch b, int active, const double* extra_in, uint32_t n_extra, uint32_t fail_slot,
                            double* sums4, double* extra_out, int* fb_status) {
  const int frc = scrf_fb_batch(h, b, nullptr, nullptr);
  if (fb_status) *fb_status = frc;
  const std::string batch_err = h->err;
  double e[4] = {0, 0, 0, 0};
  for (uint32_t i = 0; i < n_extra; i++) e[i] = extra_in[i];
  if (frc != SCRF_OK && n_extra) e[fail_slot] = 1.0;
  const int rc = scrf_allreduce_grad_ex(h, frc == SCRF_OK ? active : 0, e, n_extra, sums4, extra_out);
  if (rc == SCRF_OK && frc != SCRF_OK) h->err = batch_err;
  return rc;
}

int scrf_set_frame_mass_check(scrf_handle, int) { return SCRF_OK; }
int scrf_comm_stats(scrf_handle h, uint64_t* a, uint64_t* b) { if (a) *a = (uint64_t)h->step; if (b) *b = 0; return SCRF_OK; }
// the rest of the ABI the host layer references: not part of the training control flow
#define NOT_HERE(sig) int sig { return SCRF_ERR_INVALID; }
NOT_HERE(scrf_scores(scrf_handle, scrf_batch, uint32_t, double*, double*))
NOT_HERE(scrf_windows(scrf_handle, scrf_batch, uint32_t, float*))
NOT_HERE(scrf_forward_backward(scrf_handle, scrf_batch, uint32_t, uint32_t, double*, double*, double*, double*))
NOT_HERE(scrf_lattice_arcs(scrf_handle, scrf_batch, uint32_t, int, scrf_arc*, uint64_t*, uint32_t*, int32_t*))
NOT_HERE(scrf_viterbi_batch(scrf_handle, scrf_batch, uint32_t*, uint64_t, uint64_t*, float*))
}
