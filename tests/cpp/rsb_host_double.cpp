// rsb_host_double.cpp — a TEST DOUBLE of the rsb_world part of the C-ABI (include/rsb.h) that runs on the host.
//
// TEST INFRASTRUCTURE ONLY.  It exists so that the host side of the drop-in boundary - include/raisim/*.hpp: the per-env
// raisim::World views, their staging / flushing / lazy downloads, the fiber scheduler under VectorizedEnvironment<ENV> - can be
// exercised, raced and profiled in the CPU test tier, where no HIP device exists.  It is linked ONLY into tests/cpp binaries
// (together with the product's own urdf_model.cpp / terrain_io.cpp, which are plain C++), never into librsb.so, never by the
// product: raisimlib_amd has no CPU fallback and rsb_create in librsb.so keeps failing with RSB_E_NO_DEVICE without a GPU.
//
// "Physics" of the double: NOT rigid-body dynamics.  Every integrate() sub-step applies the deterministic toy update
//     u_j += dt (kp_j (p_target_j - q_j) + kd_j (d_target_j - u_j) + tau_ff_j),   q_j += dt u_j          (joints only)
// and the base drifts with its velocity; every env reports one contact per "_foot" primitive and the PD torque as its generalized
// force.  k sub-steps in one call equal k calls of one sub-step bit for bit - what the facade's fused flush relies on and what
// tests/cpp/facade_host_test.cpp checks through the facade.  Calls are counted (rsbd_counters) so that tests can assert how
// many launches / transfers / synchronisations a control step costs.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rsb.h"
#include "rsb_internal.h"

struct rsb_world {
  rsb_model_blob blob;
  int N = 0, kmax = 8;
  double dt = 0.001, time = 0.0;
  std::vector<float> gc, gv, pt, dtg, tff, genf, kp, kd;
  std::vector<int32_t> count;
  std::vector<rsb_contact> contacts;
  std::vector<int> feet;
  bool want_genf = false, query_valid = false;
  std::atomic<int> busy{0};      // detects concurrent entry into one handle (the handle is not re-entrant: rsb.h)
};

extern "C" {
struct rsbd_counter_block { long launches, substeps, masked_launches, uploads, downloads, syncs, exchanges, reentries; };
static rsbd_counter_block g_ctr;
rsbd_counter_block* rsbd_counters(void) { return &g_ctr; }
}

namespace {
struct Enter {
  rsb_world* w;
  explicit Enter(rsb_world* w_) : w(w_) { if (w->busy.fetch_add(1) != 0) ++g_ctr.reentries; }
  ~Enter() { w->busy.fetch_sub(1); }
};

void substep(rsb_world* w, const uint8_t* mask) {
  const int nq = w->blob.nq, nv = w->blob.nv;
  const float dt = (float)w->dt;
  for (int e = 0; e < w->N; ++e) {
    if (mask && !mask[e]) continue;
    float* q = &w->gc[(size_t)e * nq]; float* u = &w->gv[(size_t)e * nv];
    const float* pt = &w->pt[(size_t)e * nq]; const float* dg = &w->dtg[(size_t)e * nv]; const float* tf = &w->tff[(size_t)e * nv];
    float* gf = &w->genf[(size_t)e * nv];
    for (int j = 6; j < nv; ++j) {
      const float tau = w->kp[j] * (pt[j + 1] - q[j + 1]) + w->kd[j] * (dg[j] - u[j]) + tf[j];
      gf[j] = tau;
      u[j] += dt * tau;
      q[j + 1] += dt * u[j];
    }
    for (int j = 0; j < 6; ++j) gf[j] = tf[j];
    for (int c = 0; c < 3; ++c) q[c] += dt * u[c];
    const int nf = (int)w->feet.size();
    w->count[e] = nf;
    for (int k = 0; k < nf; ++k) {
      rsb_contact& ct = w->contacts[(size_t)e * w->kmax + k];
      std::memset(&ct, 0, sizeof ct);
      ct.position[0] = q[0]; ct.position[1] = q[1]; ct.normal[2] = 1.f; ct.impulse[2] = 1.f + 0.001f * (float)k + q[7];
      ct.body = w->blob.col_body[w->feet[k]]; ct.collision = w->feet[k];
    }
  }
}
void launch(rsb_world* w, int nsub, const uint8_t* mask) {
  for (int s = 0; s < nsub; ++s) substep(w, mask);
  w->time += nsub * w->dt; w->query_valid = false;
  ++g_ctr.launches; g_ctr.substeps += nsub; if (mask) ++g_ctr.masked_launches;
}
}  // namespace

extern "C" {

int rsb_device_count(void) { return 0; }
int rsb_create(const rsb_model* m, int num_envs, int /*device*/, rsb_world** out) {
  if (!m || !out || num_envs <= 0) return RSB_E_INVALID;
  rsb_world* w = new rsb_world();
  w->blob = m->blob; w->N = num_envs;
  const size_t N = (size_t)num_envs, nq = m->blob.nq, nv = m->blob.nv;
  w->gc.assign(N * nq, 0.f); w->gv.assign(N * nv, 0.f); w->pt.assign(N * nq, 0.f); w->dtg.assign(N * nv, 0.f); w->tff.assign(N * nv, 0.f);
  w->genf.assign(N * nv, 0.f); w->kp.assign(nv, 0.f); w->kd.assign(nv, 0.f);
  for (size_t e = 0; e < N; ++e) { w->gc[e * nq + 3] = 1.f; w->pt[e * nq + 3] = 1.f; }
  for (int c = 0; c < m->blob.ncol; ++c) { const std::string nm = m->blob.col_name[c]; if (nm.size() >= 5 && nm.compare(nm.size() - 5, 5, "_foot") == 0) w->feet.push_back(c); }
  w->count.assign(N, 0); w->contacts.assign(N * w->kmax, rsb_contact{});
  *out = w;
  return RSB_OK;
}
int rsb_destroy(rsb_world* w) { delete w; return RSB_OK; }
int rsb_synchronize(rsb_world*) { ++g_ctr.syncs; return RSB_OK; }
int rsb_num_envs(const rsb_world* w) { return w->N; }
int rsb_dims(const rsb_world* w, int* nb, int* nq, int* nv, int* ncol, int* kmax) {
  if (nb) *nb = w->blob.nb; if (nq) *nq = w->blob.nq; if (nv) *nv = w->blob.nv; if (ncol) *ncol = w->blob.ncol; if (kmax) *kmax = w->kmax;
  return RSB_OK;
}
int rsb_set_timestep(rsb_world* w, double dt) { w->dt = dt; return RSB_OK; }
double rsb_get_timestep(const rsb_world* w) { return w->dt; }
double rsb_get_world_time(const rsb_world* w) { return w->time; }
int rsb_set_gravity(rsb_world*, const double*) { return RSB_OK; }
int rsb_set_erp(rsb_world*, double) { return RSB_OK; }
int rsb_set_material(rsb_world*, double, double, double) { return RSB_OK; }
int rsb_set_collision_materials(rsb_world*, const double*, const double*, const double*) { return RSB_OK; }
int rsb_set_self_collision(rsb_world*, int) { return RSB_OK; }
int rsb_ignore_collision_between(rsb_world*, int, int) { return RSB_OK; }
int rsb_self_collision_pairs(const rsb_world*, int32_t*, int) { return 0; }
int rsb_set_self_collision_materials(rsb_world*, const double*, const double*, const double*) { return RSB_OK; }
int rsb_set_contact_solver_param(rsb_world*, double, double, double, int, double) { return RSB_OK; }
int rsb_set_integration_scheme(rsb_world*, int) { return RSB_OK; }
int rsb_set_solver_multi_contact(rsb_world*, int, int, int, int) { return RSB_OK; }
int rsb_set_solver_anderson(rsb_world*, int, double) { return RSB_OK; }
int rsb_set_heightmap_contacts(rsb_world*, int, double) { return RSB_OK; }
int rsb_set_capsule_contacts(rsb_world*, int) { return RSB_OK; }
int rsb_set_step_pipelining(rsb_world*, int) { return RSB_OK; }
int rsb_set_early_termination(rsb_world*, int) { return RSB_OK; }
int rsb_set_ground(rsb_world*, double) { return RSB_OK; }
int rsb_set_heightmap(rsb_world*, int, int, double, double, double, double, const float*) { return RSB_OK; }
int rsb_set_control_mode(rsb_world*, int) { return RSB_OK; }
int rsb_enable_generalized_force_output(rsb_world* w, int on) { w->want_genf = on != 0; return RSB_OK; }

int rsb_set_state(rsb_world* w, const float* gc, const float* gv, const uint8_t* mask, int) {
  Enter g(w);
  const int nq = w->blob.nq, nv = w->blob.nv;
  for (int e = 0; e < w->N; ++e) {
    if (mask && !mask[e]) continue;
    if (gc) std::memcpy(&w->gc[(size_t)e * nq], gc + (size_t)e * nq, nq * sizeof(float));
    if (gv) std::memcpy(&w->gv[(size_t)e * nv], gv + (size_t)e * nv, nv * sizeof(float));
  }
  ++g_ctr.uploads; ++g_ctr.syncs; w->query_valid = false;
  return RSB_OK;
}
int rsb_get_state(rsb_world* w, float* gc, float* gv, int) {
  Enter g(w);
  if (gc) std::memcpy(gc, w->gc.data(), w->gc.size() * sizeof(float));
  if (gv) std::memcpy(gv, w->gv.data(), w->gv.size() * sizeof(float));
  ++g_ctr.downloads; ++g_ctr.syncs;
  return RSB_OK;
}
static std::vector<float>* field_of(rsb_world* w, int field) {
  switch (field) {
    case RSB_F_GC: return &w->gc; case RSB_F_GV: return &w->gv; case RSB_F_PTARGET: return &w->pt; case RSB_F_DTARGET: return &w->dtg;
    case RSB_F_TAU_FF: return &w->tff; case RSB_F_GENERALIZED_FORCE: return &w->genf; default: return nullptr;
  }
}
int rsb_get_field(rsb_world* w, int field, float* out, int) {
  Enter g(w);
  std::vector<float>* f = field_of(w, field);
  if (!f || !out) return RSB_E_INVALID;
  std::memcpy(out, f->data(), f->size() * sizeof(float));
  ++g_ctr.downloads; ++g_ctr.syncs;
  return RSB_OK;
}
int rsb_set_pd_gains(rsb_world* w, const float* kp, const float* kd) { Enter g(w); w->kp.assign(kp, kp + w->blob.nv); w->kd.assign(kd, kd + w->blob.nv); return RSB_OK; }
int rsb_set_pd_target(rsb_world* w, const float* p, const float* d, int) {
  Enter g(w);
  if (p) std::memcpy(w->pt.data(), p, w->pt.size() * sizeof(float));
  if (d) std::memcpy(w->dtg.data(), d, w->dtg.size() * sizeof(float));
  ++g_ctr.uploads; ++g_ctr.syncs; w->query_valid = false;
  return RSB_OK;
}
int rsb_set_generalized_force(rsb_world* w, const float* tau, int) { Enter g(w); std::memcpy(w->tff.data(), tau, w->tff.size() * sizeof(float)); ++g_ctr.uploads; ++g_ctr.syncs; w->query_valid = false; return RSB_OK; }
int rsb_integrate(rsb_world* w, int n) { Enter g(w); if (n < 1) return RSB_E_INVALID; launch(w, n, nullptr); return RSB_OK; }
int rsb_integrate_masked(rsb_world* w, int n, const uint8_t* mask, int) { Enter g(w); if (n < 1 || !mask) return RSB_E_INVALID; launch(w, n, mask); ++g_ctr.syncs; return RSB_OK; }
int rsb_integrate1(rsb_world* w) { Enter g(w); w->query_valid = true; ++g_ctr.launches; return RSB_OK; }
int rsb_integrate2(rsb_world* w) { Enter g(w); launch(w, 1, nullptr); return RSB_OK; }
int rsb_get_contacts(rsb_world* w, int32_t* counts, rsb_contact* contacts, int) {
  Enter g(w);
  if (counts) std::memcpy(counts, w->count.data(), w->count.size() * sizeof(int32_t));
  if (contacts) std::memcpy(contacts, w->contacts.data(), w->contacts.size() * sizeof(rsb_contact));
  ++g_ctr.downloads; ++g_ctr.syncs;
  return RSB_OK;
}
// M = identity scaled by (1 + q_joint0), h = the env's velocities: enough to see WHICH state a query saw
int rsb_get_mass_matrix(rsb_world* w, float* M, int) {
  Enter g(w);
  if (!w->query_valid) { rsb::set_error("rsb_get_mass_matrix: call rsb_integrate1 first (state changed since)"); return RSB_E_STATE; }
  const int nv = w->blob.nv, nq = w->blob.nq;
  std::memset(M, 0, (size_t)w->N * nv * nv * sizeof(float));
  for (int e = 0; e < w->N; ++e) for (int i = 0; i < nv; ++i) M[((size_t)e * nv + i) * nv + i] = 1.f + w->gc[(size_t)e * nq + 7];
  ++g_ctr.downloads; ++g_ctr.syncs;
  return RSB_OK;
}
int rsb_get_inverse_mass_matrix(rsb_world* w, float* Mi, int s) { int st = rsb_get_mass_matrix(w, Mi, s); if (st) return st; const int nv = w->blob.nv; for (int e = 0; e < w->N; ++e) for (int i = 0; i < nv; ++i) { float& x = Mi[((size_t)e * nv + i) * nv + i]; x = 1.f / x; } return RSB_OK; }
int rsb_get_nonlinearities(rsb_world* w, float* h, int) {
  Enter g(w);
  if (!w->query_valid) { rsb::set_error("rsb_get_nonlinearities: call rsb_integrate1 first (state changed since)"); return RSB_E_STATE; }
  std::memcpy(h, w->gv.data(), w->gv.size() * sizeof(float));
  ++g_ctr.downloads; ++g_ctr.syncs;
  return RSB_OK;
}
int rsb_host_alloc(size_t bytes, void** out) { if (!out || !bytes) return RSB_E_INVALID; *out = std::malloc(bytes); return *out ? RSB_OK : RSB_E_HIP; }
int rsb_host_free(void* p) { std::free(p); return RSB_OK; }
int rsb_view_exchange(rsb_world* w, const rsb_view_io* io) {
  Enter g(w);
  if (!io || io->n_launches < 0) return RSB_E_INVALID;
  const size_t N = w->N, nq = w->blob.nq, nv = w->blob.nv;
  if (io->p_target) { std::memcpy(w->pt.data(), io->p_target, N * nq * sizeof(float)); ++g_ctr.uploads; }
  if (io->d_target) { std::memcpy(w->dtg.data(), io->d_target, N * nv * sizeof(float)); ++g_ctr.uploads; }
  if (io->tau_ff) { std::memcpy(w->tff.data(), io->tau_ff, N * nv * sizeof(float)); ++g_ctr.uploads; }
  if (io->gc || io->gv) {
    if (!io->state_mask) return RSB_E_INVALID;
    for (size_t e = 0; e < N; ++e) {
      if (!io->state_mask[e]) continue;
      if (io->gc) std::memcpy(&w->gc[e * nq], io->gc + e * nq, nq * sizeof(float));
      if (io->gv) std::memcpy(&w->gv[e * nv], io->gv + e * nv, nv * sizeof(float));
    }
    ++g_ctr.uploads; w->query_valid = false;
  }
  for (int i = 0; i < io->n_launches; ++i) {
    if (io->launch_substeps[i] < 1) return RSB_E_INVALID;
    launch(w, io->launch_substeps[i], io->launch_masks ? io->launch_masks + (size_t)i * N : nullptr);
  }
  if (io->gc_out) { std::memcpy(io->gc_out, w->gc.data(), N * nq * sizeof(float)); ++g_ctr.downloads; }
  if (io->gv_out) { std::memcpy(io->gv_out, w->gv.data(), N * nv * sizeof(float)); ++g_ctr.downloads; }
  if (io->contact_counts) { std::memcpy(io->contact_counts, w->count.data(), N * sizeof(int32_t)); ++g_ctr.downloads; }
  if (io->contacts) { std::memcpy(io->contacts, w->contacts.data(), N * w->kmax * sizeof(rsb_contact)); ++g_ctr.downloads; }
  if (io->generalized_force) { if (!w->want_genf) return RSB_E_INVALID; std::memcpy(io->generalized_force, w->genf.data(), N * nv * sizeof(float)); ++g_ctr.downloads; }
  ++g_ctr.syncs; ++g_ctr.exchanges;
  return RSB_OK;
}

}  // extern "C"
