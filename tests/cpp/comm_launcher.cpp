// comm_launcher.cpp — the C-ABI multi-GPU path (rsb_comm_* / rsb_allgather_obs, include/rsb.h) driven the way a C++ host
// would drive it: one PROCESS per GPU, no Python, no MPI.  The launcher forks R rank processes BEFORE anything touches HIP,
// rank 0 creates the RCCL unique id and hands it to the launcher through a pipe, the launcher hands it to the other ranks,
// every rank creates its world on GPU (rank % device count), steps its env shard and calls rsb_allgather_obs; each rank then
// checks the gathered block: its own slice must equal its local obs block bit for bit, and the slices of the other ranks
// must be the obs blocks THEY computed (every workload quantity is a function of the global env index, so rank r can
// recompute any other rank's block on its own GPU and compare).
//
//   comm_launcher <urdf> [ranks]     ranks defaults to min(device count, 2); a 1-GPU box runs ONE rank through the same
//                                    fork / pipe / communicator path (RCCL refuses two ranks on one device).
// Exit code 0 + "comm_launcher OK ranks=R" on success; 77 = no GPU visible (the CPU test treats that as "skipped").
// Upstream counterpart: none (RaiSim is single-process).
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

constexpr int kEnvs = 96;          // per rank
constexpr int kControlSteps = 3;

#define CHECK(expr)                                                                             \
  do {                                                                                          \
    int st_ = (expr);                                                                           \
    if (st_ != RSB_OK) { std::fprintf(stderr, "rank %d: %s -> %d (%s)\n", g_rank, #expr, st_, rsb_last_error()); return 1; } \
  } while (0)
int g_rank = -1;

bool read_all(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n) { ssize_t k = read(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; }
  return true;
}
bool write_all(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n) { ssize_t k = write(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; }
  return true;
}

// splitmix64 -> U[0,1): the state of global env g depends on g only
double uni(uint64_t g, uint64_t j) {
  uint64_t x = (g + 1) * 0x9E3779B97F4A7C15ull + j * 0xD1B54A32D192ED03ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

// creates a world for global envs [lo, lo + kEnvs), steps it kControlSteps control steps; leaves it ready for an obs gather
int make_shard(const rsb_model* model, int device, int lo, rsb_world** out) {
  rsb_world* w = nullptr;
  CHECK(rsb_create(model, kEnvs, device, &w));
  int nb, nq, nv, ncol, kmax;
  CHECK(rsb_dims(w, &nb, &nq, &nv, &ncol, &kmax));
  CHECK(rsb_set_timestep(w, 0.0025));
  std::vector<float> kp(nv, 0.f), kd(nv, 0.f), gc((size_t)kEnvs * nq, 0.f), gv((size_t)kEnvs * nv, 0.f), pt((size_t)kEnvs * nq, 0.f);
  for (int i = 6; i < nv; ++i) { kp[i] = 50.f; kd[i] = 0.2f; }
  const float nominal[12] = {0.03f, 0.4f, -0.8f, -0.03f, 0.4f, -0.8f, 0.03f, -0.4f, 0.8f, -0.03f, -0.4f, 0.8f};
  for (int e = 0; e < kEnvs; ++e) {
    const uint64_t g = (uint64_t)(lo + e);
    float* q = &gc[(size_t)e * nq];
    const double yaw = (2.0 * uni(g, 2) - 1.0) * 3.14159265358979;
    q[0] = (float)(0.2 * uni(g, 0) - 0.1); q[1] = (float)(0.2 * uni(g, 1) - 0.1); q[2] = 0.56f;
    q[3] = (float)std::cos(0.5 * yaw); q[6] = (float)std::sin(0.5 * yaw);
    for (int j = 0; j < 12 && 7 + j < nq; ++j) q[7 + j] = nominal[j];
  }
  CHECK(rsb_set_pd_gains(w, kp.data(), kd.data()));
  CHECK(rsb_set_state(w, gc.data(), gv.data(), nullptr, RSB_HOST));
  for (int k = 0; k < kControlSteps; ++k) {
    for (int e = 0; e < kEnvs; ++e) {
      float* p = &pt[(size_t)e * nq];
      p[3] = 1.f;
      for (int j = 0; j < 12 && 7 + j < nq; ++j) p[7 + j] = nominal[j] + (float)(0.3 * (2.0 * uni((uint64_t)(lo + e), 16 + 12 * k + j) - 1.0));
    }
    CHECK(rsb_set_pd_target(w, pt.data(), nullptr, RSB_HOST));
    CHECK(rsb_integrate(w, 4));
  }
  *out = w;
  return 0;
}

int rank_main(const char* urdf, int rank, int ranks, int id_out_fd, int id_in_fd) {
  g_rank = rank;
  const int ndev = rsb_device_count();
  if (ndev < 1) return 77;
  char id[RSB_COMM_ID_BYTES];
  if (rank == 0) {
    CHECK(rsb_comm_get_unique_id(id));
    if (!write_all(id_out_fd, id, sizeof id)) { std::fprintf(stderr, "rank 0: cannot hand the id to the launcher\n"); return 1; }
  }
  if (!read_all(id_in_fd, id, sizeof id)) { std::fprintf(stderr, "rank %d: no id from the launcher\n", rank); return 1; }
  rsb_model* model = nullptr;
  CHECK(rsb_model_from_urdf_file(urdf, &model));
  const int device = rank % ndev;
  rsb_world* w = nullptr;
  if (make_shard(model, device, rank * kEnvs, &w)) return 1;
  CHECK(rsb_comm_init(w, ranks, rank, id));
  if (rank == 0) {
    // which RCCL is this, and is it the one the library's hand-copied constants were checked against at build time (static_asserts in rsb_comm.hip)?
    int rt = 0, hdr = 0;
    CHECK(rsb_comm_rccl_version(&rt, &hdr));
    std::printf("comm_launcher: RCCL runtime version code %d, header version code %d (0 = no header on the build host)\n", rt, hdr);
    std::fflush(stdout);      // (a rank leaves through _exit: buffered output would be lost)
    if (rt <= 0) { std::fprintf(stderr, "rank 0: ncclGetVersion reported nothing\n"); return 1; }
    if (hdr > 0 && rt / 10000 != hdr / 10000 && rt / 1000 != hdr / 1000) { std::fprintf(stderr, "rank 0: RCCL major version of the runtime (%d) and of the build header (%d) differ\n", rt, hdr); return 1; }
  }
  const int feet = 4, od = rsb_obs_dim(w, feet);
  std::vector<float> all((size_t)ranks * kEnvs * od, -1.f), local((size_t)kEnvs * od);
  CHECK(rsb_allgather_obs(w, nullptr, feet, all.data(), RSB_HOST));
  CHECK(rsb_gather_obs(w, local.data(), nullptr, feet, RSB_HOST));
  if (std::memcmp(&all[(size_t)rank * kEnvs * od], local.data(), local.size() * sizeof(float)) != 0) {
    std::fprintf(stderr, "rank %d: own slice of the gathered block differs from the local obs block\n", rank); return 1;
  }
  // the other ranks' slices: recompute their shards here (shard-invariant workload) and compare bit for bit
  for (int r = 0; r < ranks; ++r) {
    if (r == rank) continue;
    rsb_world* o = nullptr;
    if (make_shard(model, device, r * kEnvs, &o)) return 1;
    std::vector<float> theirs((size_t)kEnvs * od);
    CHECK(rsb_gather_obs(o, theirs.data(), nullptr, feet, RSB_HOST));
    if (std::memcmp(&all[(size_t)r * kEnvs * od], theirs.data(), theirs.size() * sizeof(float)) != 0) {
      std::fprintf(stderr, "rank %d: slice of rank %d differs from that shard recomputed locally\n", rank, r); return 1;
    }
    CHECK(rsb_destroy(o));
  }
  bool nonzero = false;
  for (float x : local) nonzero |= (x != 0.f);
  if (!nonzero) { std::fprintf(stderr, "rank %d: empty obs block\n", rank); return 1; }
  CHECK(rsb_comm_destroy(w));
  CHECK(rsb_destroy(w));
  CHECK(rsb_model_destroy(model));
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: comm_launcher <urdf> [ranks]\n"); return 2; }
  int ranks = argc > 2 ? std::atoi(argv[2]) : 0;
  if (ranks <= 0) {
    // the device count is asked in a child: the launcher itself must never initialise HIP before it forks
    int fd[2];
    if (pipe(fd)) return 2;
    pid_t p = fork();
    if (p == 0) { int n = rsb_device_count(); write_all(fd[1], &n, sizeof n); _exit(0); }
    int n = 0;
    close(fd[1]); read_all(fd[0], &n, sizeof n); close(fd[0]); waitpid(p, nullptr, 0);
    if (n < 1) { std::printf("comm_launcher: no HIP device\n"); return 77; }
    ranks = n < 2 ? 1 : 2;
  }
  std::vector<pid_t> pids(ranks);
  std::vector<int> to_rank(ranks);
  int from0[2];
  if (pipe(from0)) return 2;
  for (int r = 0; r < ranks; ++r) {
    int in[2];
    if (pipe(in)) return 2;
    pid_t p = fork();
    if (p == 0) {
      close(in[1]); close(from0[0]);
      for (int q = 0; q < r; ++q) close(to_rank[q]);
      _exit(rank_main(argv[1], r, ranks, from0[1], in[0]));
    }
    close(in[0]);
    pids[r] = p; to_rank[r] = in[1];
  }
  close(from0[1]);
  char id[RSB_COMM_ID_BYTES];
  bool ok = read_all(from0[0], id, sizeof id);
  for (int r = 0; r < ranks; ++r) { if (ok) write_all(to_rank[r], id, sizeof id); close(to_rank[r]); }
  int rc = ok ? 0 : 1;
  for (int r = 0; r < ranks; ++r) {
    int st = 0;
    waitpid(pids[r], &st, 0);
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128;
    if (code != 0 && (rc == 0 || rc == 1)) rc = code;
  }
  if (rc == 0) std::printf("comm_launcher OK ranks=%d\n", ranks);
  else if (rc == 77) std::printf("comm_launcher: no HIP device\n");
  return rc;
}
