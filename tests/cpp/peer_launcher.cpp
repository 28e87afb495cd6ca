// peer_launcher.cpp — the peer-mapped obs exchange (rsb_obs_peer_*, include/rsb.h) across PROCESSES, the way a C++ host would
// drive it: the launcher forks R rank processes before anything touches HIP; every rank creates its world (GPU rank % device
// count: on a 1-GPU box the ranks share the device, which hipIpc allows and RCCL does not), creates its gathered buffer and
// hands the IPC handle to the launcher; the launcher sends the table of all handles back; every rank connects, then runs control
// steps: ONE launch each, whose epilogue stores the rank's obs rows into every rank's buffer.  After rsb_obs_peer_wait each rank
// compares the gathered block with the shards recomputed locally (every workload quantity is a function of the global env index).
//   peer_launcher <urdf> [ranks=2]      exit 0 + "peer_launcher OK ranks=R"; 77 = no GPU.
#include <hip/hip_runtime_api.h>
#include <signal.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rsb.h"

namespace {
constexpr int kEnvs = 96, kSteps = 4, kFeet = 4;
int g_rank = -1;
#define CHECK(expr)                                                                             \
  do {                                                                                          \
    int st_ = (expr);                                                                           \
    if (st_ != RSB_OK) { std::fprintf(stderr, "rank %d: %s -> %d (%s)\n", g_rank, #expr, st_, rsb_last_error()); return 1; } \
  } while (0)

bool read_all(int fd, void* p, size_t n) { char* c = static_cast<char*>(p); while (n) { ssize_t k = read(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; } return true; }
bool write_all(int fd, const void* p, size_t n) { const char* c = static_cast<const char*>(p); while (n) { ssize_t k = write(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; } return true; }

double uni(uint64_t g, uint64_t j) {
  uint64_t x = (g + 1) * 0x9E3779B97F4A7C15ull + j * 0xD1B54A32D192ED03ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
const float kNominal[12] = {0.03f, 0.4f, -0.8f, -0.03f, 0.4f, -0.8f, 0.03f, -0.4f, 0.8f, -0.03f, -0.4f, 0.8f};

int make_shard(const rsb_model* model, int device, int lo, rsb_world** out) {
  rsb_world* w = nullptr;
  CHECK(rsb_create(model, kEnvs, device, &w));
  int nb, nq, nv, ncol, kmax;
  CHECK(rsb_dims(w, &nb, &nq, &nv, &ncol, &kmax));
  CHECK(rsb_set_timestep(w, 0.0025));
  std::vector<float> kp(nv, 0.f), kd(nv, 0.f), gc((size_t)kEnvs * nq, 0.f), gv((size_t)kEnvs * nv, 0.f);
  for (int i = 6; i < nv; ++i) { kp[i] = 50.f; kd[i] = 0.2f; }
  for (int e = 0; e < kEnvs; ++e) {
    const uint64_t g = (uint64_t)(lo + e);
    float* q = &gc[(size_t)e * nq];
    const double yaw = (2.0 * uni(g, 2) - 1.0) * 3.14159265358979;
    q[0] = (float)(0.2 * uni(g, 0) - 0.1); q[1] = (float)(0.2 * uni(g, 1) - 0.1); q[2] = 0.56f;
    q[3] = (float)std::cos(0.5 * yaw); q[6] = (float)std::sin(0.5 * yaw);
    for (int j = 0; j < 12; ++j) q[7 + j] = kNominal[j];
  }
  CHECK(rsb_set_pd_gains(w, kp.data(), kd.data()));
  CHECK(rsb_set_state(w, gc.data(), gv.data(), nullptr, RSB_HOST));
  *out = w;
  return 0;
}
int set_targets(rsb_world* w, int lo, int k) {
  int nb, nq, nv, ncol, kmax;
  CHECK(rsb_dims(w, &nb, &nq, &nv, &ncol, &kmax));
  std::vector<float> pt((size_t)kEnvs * nq, 0.f);
  for (int e = 0; e < kEnvs; ++e) {
    float* p = &pt[(size_t)e * nq];
    p[3] = 1.f;
    for (int j = 0; j < 12; ++j) p[7 + j] = kNominal[j] + (float)(0.3 * (2.0 * uni((uint64_t)(lo + e), 16 + 12 * k + j) - 1.0));
  }
  CHECK(rsb_set_pd_target(w, pt.data(), nullptr, RSB_HOST));
  return 0;
}

int rank_main(const char* urdf, int rank, int ranks, int up_fd, int down_fd) {
  g_rank = rank;
  const int ndev = rsb_device_count();
  if (ndev < 1) return 77;
  const int device = rank % ndev;
  rsb_model* model = nullptr;
  CHECK(rsb_model_from_urdf_file(urdf, &model));
  rsb_world* w = nullptr;
  if (make_shard(model, device, rank * kEnvs, &w)) return 1;
  char mine[RSB_OBS_HANDLE_BYTES];
  CHECK(rsb_obs_peer_create(w, ranks, rank, nullptr, kFeet, mine));
  if (!write_all(up_fd, mine, sizeof mine)) return 1;
  std::vector<char> table((size_t)ranks * RSB_OBS_HANDLE_BYTES);
  if (!read_all(down_fd, table.data(), table.size())) { std::fprintf(stderr, "rank %d: no handle table\n", rank); return 1; }
  CHECK(rsb_obs_peer_connect(w, table.data()));
  // the other ranks' shards, recomputed here with plain worlds (no exchange): what their rows must be
  std::vector<rsb_world*> twin(ranks, nullptr);
  for (int r = 0; r < ranks; ++r) if (make_shard(model, device, r * kEnvs, &twin[r])) return 1;
  const int od = rsb_obs_dim(w, kFeet);
  std::vector<float> all((size_t)ranks * kEnvs * od), want((size_t)kEnvs * od);
  char token = 1;
  for (int k = 0; k < kSteps; ++k) {
    if (set_targets(w, rank * kEnvs, k)) return 1;
    CHECK(rsb_control_step(w, nullptr, nullptr, 4, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, 0));   // ONE launch; rows go to every rank
    float* gathered = nullptr;
    CHECK(rsb_obs_peer_wait(w, &gathered));
    CHECK(rsb_synchronize(w));
    if (hipMemcpy(all.data(), gathered, all.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { std::fprintf(stderr, "rank %d: hipMemcpy failed\n", rank); return 1; }
    for (int r = 0; r < ranks; ++r) {
      if (set_targets(twin[r], r * kEnvs, k)) return 1;
      CHECK(rsb_integrate(twin[r], 4));
      CHECK(rsb_gather_obs(twin[r], want.data(), nullptr, kFeet, RSB_HOST));
      if (std::memcmp(&all[(size_t)r * kEnvs * od], want.data(), want.size() * sizeof(float)) != 0) {
        std::fprintf(stderr, "rank %d, control step %d: rows of rank %d differ from that shard recomputed locally\n", rank, k, r); return 1;
      }
    }
    // lock-step with the other ranks (the launcher echoes a token when all have checked this step): a rank that ran two steps
    // ahead would overwrite the buffer parity another rank is still reading - the consumer's own pacing in a real loop
    if (!write_all(up_fd, &token, 1) || !read_all(down_fd, &token, 1)) return 1;
  }
  for (auto* t : twin) CHECK(rsb_destroy(t));
  CHECK(rsb_obs_peer_destroy(w));
  CHECK(rsb_destroy(w));
  CHECK(rsb_model_destroy(model));
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: peer_launcher <urdf> [ranks]\n"); return 2; }
  const int ranks = argc > 2 ? std::atoi(argv[2]) : 2;
  if (ranks < 1 || ranks > RSB_MAX_RANKS) return 2;
  std::vector<pid_t> pids(ranks);
  std::vector<int> up(ranks), down(ranks);
  for (int r = 0; r < ranks; ++r) {
    int u[2], d[2];
    if (pipe(u) || pipe(d)) return 2;
    pid_t p = fork();
    if (p == 0) {
      close(u[0]); close(d[1]);
      for (int q = 0; q < r; ++q) { close(up[q]); close(down[q]); }
      _exit(rank_main(argv[1], r, ranks, u[1], d[0]));
    }
    close(u[1]); close(d[0]);
    pids[r] = p; up[r] = u[0]; down[r] = d[1];
  }
  std::vector<char> table((size_t)ranks * RSB_OBS_HANDLE_BYTES);
  bool ok = true;
  for (int r = 0; r < ranks; ++r) ok = ok && read_all(up[r], &table[(size_t)r * RSB_OBS_HANDLE_BYTES], RSB_OBS_HANDLE_BYTES);
  for (int r = 0; r < ranks; ++r) if (ok) write_all(down[r], table.data(), table.size());
  for (int k = 0; k < kSteps && ok; ++k) {
    char t;
    for (int r = 0; r < ranks; ++r) ok = ok && read_all(up[r], &t, 1);
    for (int r = 0; r < ranks; ++r) if (ok) write_all(down[r], &t, 1);
  }
  for (int r = 0; r < ranks; ++r) { close(up[r]); close(down[r]); }
  if (!ok) for (int r = 0; r < ranks; ++r) kill(pids[r], SIGKILL);   // a rank that failed leaves the others waiting for its rows: do not hang
  int rc = ok ? 0 : 1;
  for (int r = 0; r < ranks; ++r) {
    int st = 0;
    waitpid(pids[r], &st, 0);
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128;
    if (code != 0 && (rc == 0 || rc == 1)) rc = code;
  }
  if (rc == 0) std::printf("peer_launcher OK ranks=%d\n", ranks);
  else if (rc == 77) std::printf("peer_launcher: no HIP device\n");
  return rc;
}
