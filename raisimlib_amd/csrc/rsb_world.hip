// rsb_world.hip — host side of the C-ABI (include/rsb.h): device-resident batched world.
//
// Plays the role of N x { raisim::World + one raisim::ArticulatedSystem + Ground/HeightMap }
// [RECALL; World.hpp / ArticulatedSystem.hpp are absent from /root/reference, SURVEY.md §8b].
// State lives in HBM as row-major [N, dim] float32 rows; rsb_integrate() launches the fused step
// kernel (step_kernel.h) on the handle's stream.  No CPU fallback exists anywhere in this file.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "query_kernel.h"
#include "rsb.h"
#include "rsb_internal.h"
#include "step_launch.h"
#include "env_task.h"
#include "step_types.h"


#include "rsb_world.h"
#include "rsb_spec.h"

namespace rsbw {

// Kernel arguments in device memory: with host-resident kernargs every wave's first scalar loads cross PCIe (measured: step
// kernel prologue 20.6 k cycles instead of 9.2 k, 142 M instead of 149 M env-steps/s).  It is this image's default; set here
// (without overriding the user's choice) for runtimes where it is not - effective when the library loads before HIP starts.
// GPU_MAX_HW_QUEUES: HIP multiplexes its streams onto 4 hardware queues by default; the closed-loop pipeline needs three private streams that
// overlap (two step streams + the action stage's) beside the caller's own - 8 queues make the probe's search for such a set short.
struct DevKernargDefault { DevKernargDefault() { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); setenv("GPU_MAX_HW_QUEUES", "8", 0); } } g_dev_kernarg_default;

int round4(int x) { return (x + 3) & ~3; }

void build_dev_model(const rsb_model_blob& b, DevModel* d) {
  std::memset(d, 0, sizeof *d);
  d->nb = b.nb; d->nq = b.nq; d->nv = b.nv; d->ncol = b.ncol; d->depth = b.depth;
  d->cw = round4(6 + b.depth - 1);
  d->fixed_base = b.fixed_base;
  std::vector<std::vector<int>> kids(b.nb);
  for (int i = 0; i < b.nb; ++i) {
    d->parent[i] = b.parent[i]; d->level[i] = b.level[i]; d->jtype[i] = b.jtype[i];
    if (i > 0) kids[b.parent[i]].push_back(i);
    float* f = d->bodyf[i];
    for (int c = 0; c < 3; ++c) {
      d->axis[i][c] = f[c] = (float)b.axis[i][c];
      d->ptree[i][c] = f[4 + c] = (float)b.ptree[i][c];
      d->com[i][c] = f[17 + c] = (float)b.com[i][c];
    }
    std::memcpy(&f[3], &b.jtype[i], sizeof(int));
    for (int c = 0; c < 9; ++c) d->rtree[i][c] = f[8 + c] = (float)b.rtree[i][c];
    for (int c = 0; c < 6; ++c) d->inertia[i][c] = f[20 + c] = (float)b.inertia[i][c];
    d->mass[i] = f[7] = (float)b.mass[i];
    d->armature[i] = f[26] = (float)b.armature[i];
    f[27] = (float)b.damping[i];
    f[28] = (float)b.effort[i];
    const bool limited = i > 0 && b.q_lower[i] < b.q_upper[i] && b.q_lower[i] > -1e29 && b.q_upper[i] < 1e29;
    f[29] = limited ? (float)b.q_lower[i] : -3e38f;   // joint range (an unlimited joint can never leave it)
    f[30] = limited ? (float)b.q_upper[i] : 3e38f;
  }
  for (int i = 0; i < b.nb * b.depth; ++i) d->anc[i] = -1;
  for (int i = 0; i < b.nb; ++i)
    for (int j = i; j >= 0; j = b.parent[j]) d->anc[i * b.depth + b.level[j]] = j;
  // children lists (the step kernel's up pass gathers over them); the base's children lead the list
  int pos = 0;
  d->max_kid = 0;
  for (int i = 0; i < b.nb; ++i) {
    d->kid_start[i] = pos;
    d->kid_count[i] = (int)kids[i].size();
    for (int c : kids[i]) d->kid_list[pos++] = c;
    if (i >= 1 && d->kid_count[i] > d->max_kid) d->max_kid = d->kid_count[i];
  }
  for (int s = 0; s < b.ncol; ++s) {
    d->col_body[s] = b.col_body[s];
    for (int c = 0; c < 3; ++c) d->col_pos[s][c] = (float)b.col_pos[s][c];
    d->col_pos[s][3] = (float)b.col_radius[s];
  }
}

// candidate pairs of self-collision (oracle: self_pair_ok): primitives on two bodies that are not parent and child, not both
// points, no rim primitives, bodies not excluded by the caller
std::vector<int> enumerate_self_pairs(const rsb_model_blob& b, const std::vector<uint8_t>& ignore) {
  std::vector<int> out;
  for (int i = 0; i < b.ncol; ++i)
    for (int j = i + 1; j < b.ncol; ++j) {
      const int bi = b.col_body[i], bj = b.col_body[j];
      if (bi == bj || b.parent[bi] == bj || b.parent[bj] == bi) continue;
      if (b.col_rim[i] > 0.0 || b.col_rim[j] > 0.0) continue;
      if (!(b.col_radius[i] + b.col_radius[j] > 0.0)) continue;
      if (!ignore.empty() && (ignore[bi * b.nb + bj] || ignore[bj * b.nb + bi])) continue;
      out.push_back(i); out.push_back(j);
    }
  return out;
}

// slots of the height-map narrow phase: one per primitive (every sphere of the model may be near the ground at once - a robot lying in a hollow)
int hm_slots_for(const rsb_model_blob& b) { return std::max(rsbk::kHmSlots, (int)b.ncol); }

// Contact capacity of the kernel class a world is dispatched to (the template's KMAX): 8 for the shallow, few-contact models (the
// quadruped's classes), 16 for kmax > 8 AND for every model deeper than five levels, whose only compiled classes are the large ones
// (do_integrate).  ONE quantity decides layout, dispatch and LDS size: the layout used to follow kmax alone, so a deep model at the
// default kmax 8 ran the KMAX-16 kernel (packed triangular Delassus blocks) on the square layout.
int kcap_of(const rsb_model_blob& b, int kmax) { return (kmax <= 8 && b.depth - 1 <= 4) ? 8 : 16; }

static LdsLayout make_layout_pitch(const rsb_model_blob& b, int kcap, int n_self, int model_pitch) {
  LdsLayout L;
  L.model_pitch = model_pitch;
  const int cw = round4(6 + b.depth - 1);
  int o = 0;
  auto take = [&](int n) { int r = o; o += round4(n); return r; };
  L.t_model = take(b.nb * model_pitch);
  L.t_gain = take(2 * b.nb);
  L.t_parlv = take(b.nb);
  L.t_anc = take(b.nb * b.depth);
  L.t_dir = take(64);
  L.t_col = take(rsbk::kColSlot * b.ncol);
  L.t_kids = take(b.nb);
  L.t_kidx = take(b.nb);
  L.t_spair = take(n_self > 0 ? n_self + 1 : 0);   // (+ one entry that cannot hit, read by the lanes past the list)
  L.shared_total = o;
  o = 0;
  L.q = take(b.nq < 8 ? 8 : b.nq); L.u = take(b.nv < 8 ? 8 : b.nv);
  L.pt = take(b.nq); L.dtg = take(b.nv); L.tf = take(b.nv < 8 ? 8 : b.nv);
  L.body = take(b.nb * rsbk::kBodySlot);
  const bool tri = kcap > 8;   // packed lower-triangular Delassus blocks (step_kernel.h: TRI); the up pass's factors then alias them too
  if (!tri) L.fact = take(b.nb * rsbk::kFactSlot);
  L.wb = take(b.nv);
  L.con = take(kcap * rsbk::kConSlot);
  L.wc = take(3 * kcap * cw);
  L.cv = take(3 * kcap);
  L.gstride = 4 * kcap + 4;   // 3x3 blocks on a 4-float pitch, +4 staggers the banks of consecutive rows
  // the up pass's [nb][28] hand-over slots and the height-map narrow phase's scratch alias the Delassus rows
  // (packed layout: [hand-over slots | joint factors] of the up pass - the factors are last read by the contact columns, before the
  // Delassus phase writes its blocks over both)
  const int gsize = tri ? (kcap * (kcap + 1) / 2) * 12 : 3 * kcap * L.gstride;
  const int upsize = b.nb * rsbk::kUpSlot + (tri ? b.nb * rsbk::kFactSlot : 0);
  L.g = take(std::max({gsize, upsize, (rsbk::kHmRec + 8) * hm_slots_for(b) + RSB_MAX_COLLISIONS + 16}));   // (+ 16: the capsule search's four sample results)
  if (tri) L.fact = L.g + b.nb * rsbk::kUpSlot;
  L.ginv = take(12 * kcap);
  L.lam = take(3 * kcap);
  L.warm = take(6 * b.ncol);
  L.tact = take(b.nv);
  // self-collision: the primitive centres live in the contact columns' space when they fit (dead until the column phase)
  L.cen = L.wc; L.selft = 0;
  if (n_self > 0) {
    if (4 * b.ncol > 3 * kcap * cw) L.cen = take(4 * b.ncol);
    L.selft = take(4 * kcap);
  }
  L.per_env = o + rsbk::kEnvPad;
  return L;
}

// The conflict-free pitch of the model table (step_types.h: kModelPitch) costs 4 floats per body - and a workgroup per CU where the layout sits on a 160-KiB
// cliff (the Atlas-like humanoid at 32 lanes per env: 3 workgroups -> 2, config 5 fell from 38.9 M to 26.4 M env-steps/s in gpurun r06d): it is taken only
// where no lanes-per-env choice loses a workgroup to it.
LdsLayout make_layout(const rsb_model_blob& b, int kcap, int n_self) {
  const LdsLayout tight = make_layout_pitch(b, kcap, n_self, rsbk::kModelSlot), wide = make_layout_pitch(b, kcap, n_self, rsbk::kModelPitch);
  auto wgs = [](const LdsLayout& L, int lpe) {
    const size_t bytes = sizeof(float) * ((size_t)L.shared_total + (size_t)(64 / lpe) * L.per_env);
    return bytes > 160 * 1024 ? 0 : (int)std::min<size_t>(4, (160 * 1024) / bytes);
  };
  for (int lpe : {16, 32, 64}) if (wgs(wide, lpe) != wgs(tight, lpe)) return tight;
  return wide;
}

size_t lds_bytes_for(const rsb_model_blob& b, int kcap, int lpe, int n_self) {
  LdsLayout L = make_layout(b, kcap, n_self);
  return sizeof(float) * ((size_t)L.shared_total + (size_t)(64 / lpe) * L.per_env);
}

// Lanes per env: the group size that keeps the most envs resident on a CU.  The kernel runs at one wave per SIMD
// (register budget), so a CU holds min(4, 160 KiB / workgroup LDS) workgroups of 64/LPE envs each.  ANYmal-like models:
// LPE 16 (4 x 4 envs, 38 KB per workgroup: at N = 4096 one wave on every SIMD of the chip); Atlas-like, kmax 16
// (25 KB per env): every choice holds 4 envs, LPE 64 keeps all four SIMDs busy.
int default_lpe(const rsb_model_blob& b, int kmax, int n_self) {
  const int kcap = kcap_of(b, kmax);
  const int need = b.nb > 16 ? (b.nb > 32 ? 64 : 32) : 16;   // lane = body in the tree passes
  int best = 64, best_envs = 0, best_wgs = 0;
  for (int lpe = need; lpe <= 64; lpe *= 2) {
    const size_t wg = lds_bytes_for(b, kcap, lpe, n_self);
    if (wg > 160 * 1024) continue;
    if (n_self > 30 * lpe) continue;   // the self-collision sweep holds one hit bit per pass of lpe pairs (check_lpe)
    const int wgs = (int)std::min<size_t>(4, (160 * 1024) / wg);
    const int envs = wgs * (64 / lpe);
    // ties go to the layout that keeps more SIMDs busy (more, smaller workgroups)
    if (envs > best_envs || (envs == best_envs && wgs > best_wgs)) { best_envs = envs; best_wgs = wgs; best = lpe; }
  }
  return best;
}

// zero the solver's warm state of the envs whose state was overwritten (mask == NULL: all)
__global__ void warm_clear_kernel(float* warm, const uint8_t* mask, int N, int n6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * n6) return;
  if (!mask || mask[i / n6]) warm[i] = 0.f;
}

__global__ void masked_row_copy(float* dst, const float* src, const uint8_t* mask, int N, int dim) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * dim) return;
  int e = i / dim;
  if (!mask || mask[e]) dst[i] = src[i];
}

__global__ void gather_obs_kernel(float* out, const float* gc, const float* gv, const rsb_contact* contacts,
                                  const int32_t* count, const int32_t* idx, int N, int nq, int nv, int kmax,
                                  int nslots, float inv_dt) {
  int od = nq + nv + 3 * nslots;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * od) return;
  int e = i / od, c = i - e * od;
  float v;
  if (c < nq) v = gc[(size_t)e * nq + c];
  else if (c < nq + nv) v = gv[(size_t)e * nv + c - nq];
  else {
    int sl = (c - nq - nv) / 3, ax = (c - nq - nv) - 3 * sl;
    int want = idx ? idx[sl] : sl;
    v = 0.f;
    int nc = count[e];
    for (int k = 0; k < nc; ++k) {
      const rsb_contact& ct = contacts[(size_t)e * kmax + k];
      if ((ct.collision & ~(RSB_CONTACT_SECOND | RSB_CONTACT_CAPSULE)) == want) v += ct.impulse[ax] * inv_dt;   // (a primitive's contacts with a height map add up)
    }
  }
  out[i] = v;
}

__global__ void reset_terminated_kernel(float* gc, float* gv, const rsb_contact* contacts, int32_t* count,
                                        int32_t* flags, unsigned long long allowed, const float* gc0, const float* gv0,
                                        int rows, uint8_t* done, int N, int nq, int nv, int kmax, float* warm, int n6) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  bool term = (flags[e] & 2) != 0;
  const int nc = count[e];
  for (int k = 0; k < nc; ++k) {
    const int c = contacts[(size_t)e * kmax + k].collision & ~(RSB_CONTACT_SECOND | RSB_CONTACT_CAPSULE);   // (a second flank's / a capsule cylinder's contact counts as its primitive's)
    // an entry of a self-collision (id | RSB_CONTACT_SELF_A / _B) is never a foot on the terrain: terminal, as in the fused
    // epilogue of the step kernel (and a shift by >= 64 would be undefined)
    if (c >= RSB_CONTACT_SELF_A || !((allowed >> c) & 1ull)) term = true;
  }
  if (term) {
    const size_t r = rows == 1 ? 0 : (size_t)e;
    for (int i = 0; i < nq; ++i) gc[(size_t)e * nq + i] = gc0[r * nq + i];
    for (int i = 0; i < nv; ++i) gv[(size_t)e * nv + i] = gv0[r * nv + i];
    for (int i = 0; i < n6; ++i) warm[(size_t)e * n6 + i] = 0.f;
    count[e] = 0;
    flags[e] = 0;
  }
  if (done) done[e] = term ? 1 : 0;
}

// ---- device-resident vectorised env (rsg_anymal task semantics, see rsb.h and env_task.h) ----------------------
// The step itself (action -> targets, sub-steps, reward, termination, reset, next observation) is ONE launch of the step
// kernel (StepArgs::env_*); what remains here is the stand-alone observation and the reset of all envs.
// thread = (env, observation entry): env_ob_entry is the arithmetic the fused epilogue uses (bit-identical observations), so a thread per entry
// only repeats the quaternion -> rotation part per entry - 3 us for 4096 envs where the thread-per-env loop took 13 (it opens every closed-loop run)
__global__ void env_obs_kernel(float* ob, const float* gc, const float* gv, int N, int nq, int nv) {
  const int od = 10 + 2 * (nv - 6);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * od) return;
  const int e = i / od, k = i - e * od;
  const float* q = gc + (size_t)e * nq;
  const float* u = gv + (size_t)e * nv;
  ob[i] = rsbk::env_ob_entry(k, nv - 6, [&](int j) { return q[j]; }, [&](int j) { return u[j]; });
}

__global__ void env_reset_kernel(float* gc, float* gv, int32_t* count, int32_t* flags, const float* gc0, const float* gv0, int rows,
                                 int N, int nq, int nv, float* warm, int n6) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const size_t r = rows == 1 ? 0 : (size_t)e;
  for (int i = 0; i < nq; ++i) gc[(size_t)e * nq + i] = gc0[r * nq + i];
  for (int i = 0; i < nv; ++i) gv[(size_t)e * nv + i] = gv0[r * nv + i];
  for (int i = 0; i < n6; ++i) warm[(size_t)e * n6 + i] = 0.f;
  count[e] = 0; flags[e] = 0;
}

template <int LPE, int KMAX, int CL, int ML>
int launch_step(rsb_world* w, const StepArgs& a, size_t lds_bytes, bool prof) {
  constexpr int EPW = 64 / LPE;
  const int blocks = (w->N + EPW - 1) / EPW;
  // the profiling instance carries the cycle stamps / contact-problem dump / LDS poisoning; production launches use the lean one
  // (the peer-exchange classes, CL bit 2, are built without a profiling twin: profile the exchange-free class instead)
  hipError_t e;
  if constexpr ((CL & (18 | 64)) != 0) {
    if (prof) { rsb::set_error("profiling / debug instrumentation is not built for the peer-exchange, the pipelined and the resident kernel classes: disconnect the exchange (rsb_obs_peer_destroy) / switch pipelining and residency off first"); return RSB_E_UNSUPPORTED; }
    e = rsbk::launch_step_instance<LPE, KMAX, CL, ML, false>(a, blocks, lds_bytes, w->launch_stream);
  } else {
    e = prof ? rsbk::launch_step_instance<LPE, KMAX, CL, ML, true>(a, blocks, lds_bytes, w->launch_stream)
             : rsbk::launch_step_instance<LPE, KMAX, CL, ML, false>(a, blocks, lds_bytes, w->launch_stream);
  }
  HIP_TRY(e);
  return RSB_OK;
}

template <int KMAX, int CL, int ML>
int launch_lpe_class(rsb_world* w, const StepArgs& a, size_t lds_bytes, int lpe, bool prof) {
  if (lpe == 16) return launch_step<16, KMAX, CL, ML>(w, a, lds_bytes, prof);
  if (lpe == 32) return launch_step<32, KMAX, CL, ML>(w, a, lds_bytes, prof);
  return launch_step<64, KMAX, CL, ML>(w, a, lds_bytes, prof);
}
template <int KMAX, int CL, int ML>
int launch_lpe(rsb_world* w, const StepArgs& a, size_t lds_bytes, int lpe, bool prof) {
  if constexpr ((CL & 2) == 0) {     // a pipelined launch (StepArgs::pipe_prog) runs the class's pipelined twin: the plain instances carry none of its code
    if (a.pipe_prog) return launch_lpe_class<KMAX, CL | 16, ML>(w, a, lds_bytes, lpe, prof);
  }
  return launch_lpe_class<KMAX, CL, ML>(w, a, lds_bytes, lpe, prof);
}

int n_self_pairs(const rsb_world* w) { return w->self_collision ? (int)w->self_pairs.size() / 2 : 0; }

// ---- resident launches (StepArgs::res_steps; kernel classes | 64): compiled for the benchmark's two model sizes
template <int CLR>
int launch_resident_class(rsb_world* w, const StepArgs& a, size_t lds_bytes, bool quadruped) {
  if (quadruped) return launch_step<16, 8, CLR, 4>(w, a, lds_bytes, false);
  if constexpr (CLR == (64 | 384)) return RSB_E_UNSUPPORTED;      // (never reached: resident_class refuses it)
  else return launch_step<32, 16, CLR, 12>(w, a, lds_bytes, false);
}
int launch_resident(rsb_world* w, const StepArgs& a, size_t lds_bytes, int cl, bool quadruped) {
  switch (cl) {
    case 64: return launch_resident_class<64>(w, a, lds_bytes, quadruped);
    case 64 | 128: return launch_resident_class<64 | 128>(w, a, lds_bytes, quadruped);
    case 64 | 256: return launch_resident_class<64 | 256>(w, a, lds_bytes, quadruped);
    case 64 | 384: return launch_resident_class<64 | 384>(w, a, lds_bytes, quadruped);
  }
  rsb::set_error("internal: unknown resident kernel class");
  return RSB_E_UNSUPPORTED;
}

}  // namespace rsbw
extern "C" int rsb_model_lds_bytes(const rsb_model* m, int kmax, int self_collision, int lanes_per_env) {
  if (!m || kmax < 1 || kmax > RSB_MAX_CONTACTS || (lanes_per_env != 0 && lanes_per_env != 16 && lanes_per_env != 32 && lanes_per_env != 64)) { rsb::set_error("rsb_model_lds_bytes: bad argument"); return RSB_E_INVALID; }
  const rsb_model_blob& b = m->blob;
  const int n_self = self_collision ? (int)rsbw::enumerate_self_pairs(b, std::vector<uint8_t>()).size() / 2 : 0;
  const int lpe = lanes_per_env ? lanes_per_env : rsbw::default_lpe(b, kmax, n_self);
  return (int)rsbw::lds_bytes_for(b, rsbw::kcap_of(b, kmax), lpe, n_self);
}
namespace rsbw {
int effective_lpe(const rsb_world* w) {
  int lpe = w->lpe > 0 ? w->lpe : default_lpe(w->blob, w->kmax, n_self_pairs(w));
  return lpe;
}

int check_lpe(const rsb_world* w, int lpe) {
  if (lpe != 16 && lpe != 32 && lpe != 64) { rsb::set_error("lanes_per_env must be 16, 32 or 64"); return RSB_E_INVALID; }
  if (lpe < w->blob.nb) { rsb::set_error("lanes_per_env must be >= number of moving bodies (lane = body in the tree passes)"); return RSB_E_INVALID; }
  const int kcap = kcap_of(w->blob, w->kmax);
  if (n_self_pairs(w) > 30 * lpe) { rsb::set_error("too many self-collision candidate pairs for this lanes_per_env (<= 30 per lane): exclude body pairs with rsb_ignore_collision_between or switch self-collision off"); return RSB_E_UNSUPPORTED; }
  if (lds_bytes_for(w->blob, kcap, lpe, n_self_pairs(w)) > 160 * 1024) { rsb::set_error("lanes_per_env too small: the workgroup's envs do not fit in 160 KiB of LDS"); return RSB_E_INVALID; }
  return RSB_OK;
}

// The resident kernel class (CL bits) a launch of this world would run as it is configured now, or -1 with the reason in the error string.
// stage: 0 open loop, 1 linear policy, 2 actor network of greatest width mlp_width.
int resident_class(rsb_world* w, int stage, int mlp_width) {
  auto no = [](const char* why) { rsb::set_error(std::string("no resident launch for this world: ") + why); return -1; };
  const rsb_model_blob& b = w->blob;
  const int lpe = effective_lpe(w), kcap = kcap_of(b, w->kmax), mlv = b.depth - 1;
  if (b.fixed_base) return no("fixed-base systems");
  if (w->peer.connected) return no("the peer-mapped obs exchange is connected");
  if (w->integ_rk4 || w->integ_theta != 1.0) return no("an integration scheme other than SEMI_IMPLICIT");
  if (w->slip_rule == RSB_SLIP_COULOMB) return no("RSB_SLIP_COULOMB");
  if ((w->hm_contacts >= 2 || (w->hm_capsule && w->n_cap > 0)) && w->terrain_type == 1) return no("more than one contact per primitive against a height map");
  if (w->d_prof || w->dbg_env >= 0 || std::getenv("RSB_POISON_LDS")) return no("profiling / debug instrumentation is active");
  if (w->N % (64 / lpe) != 0) return no("the number of envs is not a multiple of the envs per workgroup");
  const bool quad = mlv <= 4 && kcap == 8 && lpe == 16, humanoid = mlv <= 12 && kcap == 16 && lpe == 32;
  if (!quad && !humanoid) return no("compiled for tree depth <= 5 / <= 8 contact slots / 16 lanes per env and for tree depth <= 13 / 16 slots / 32 lanes per env");
  if (stage == 0) return 64;
  if (stage == 1) return 64 | 128;
  if (stage == 2) {
    if (mlp_width <= 128) return 64 | 256;
    if (mlp_width <= 256 && quad) return 64 | 384;
    return no("actor-network widths above 128 (humanoid-sized models) / 256");
  }
  return no("unknown stage");
}

// The per-block tables of the step kernel exactly as they sit in LDS (LdsLayout::t_*): the kernel copies this image with
// float4 loads instead of staging ten tables one latency-bound loop at a time.  Rebuilt when a setter changes what it bakes
// in: PD gains, control mode, contact materials.  (The layout of the shared tables does not depend on the contact capacity.)
std::vector<float> build_lds_image(const rsb_world* w, const LdsLayout& L) {
  const rsb_model_blob& b = w->blob;
  auto dm = std::make_unique<DevModel>();
  build_dev_model(b, dm.get());
  std::vector<float> img((size_t)L.shared_total, 0.f);
  auto put_i = [&](int off, int v) { std::memcpy(&img[off], &v, sizeof(int)); };
  for (int i = 0; i < b.nb; ++i) {
    for (int c = 0; c < rsbk::kModelSlot; ++c) img[L.t_model + i * L.model_pitch + c] = dm->bodyf[i][c];
    if (w->rk4_inner) img[L.t_model + i * L.model_pitch + 28] = 0.f;      // (RUNGE_KUTTA_4's contact step: its generalized force carries inertial terms, the effort clip was applied in the stages)
    const bool pd = w->control_mode == RSB_PD_PLUS_FEEDFORWARD_TORQUE && i >= 1;
    img[L.t_gain + 2 * i] = pd ? w->h_kp[i + 5] : 0.f;
    img[L.t_gain + 2 * i + 1] = pd ? w->h_kd[i + 5] : 0.f;
    put_i(L.t_parlv + i, (b.parent[i] + 1) | (b.level[i] << 8));
    put_i(L.t_kids + i, dm->kid_list[i]);
    put_i(L.t_kidx + i, dm->kid_start[i] | (dm->kid_count[i] << 16));
  }
  for (int i = 0; i < b.nb * b.depth; ++i) put_i(L.t_anc + i, dm->anc[i]);
  for (int k = 0; k < 16; ++k) {   // slip-search bracket around grid point k: {cos, sin((k-1) pi/8), cos, sin((k+1) pi/8)}
    const double lo = ((k + 15) & 15) * 0.125 * M_PI, hi = ((k + 1) & 15) * 0.125 * M_PI;
    img[L.t_dir + 4 * k] = (float)std::cos(lo); img[L.t_dir + 4 * k + 1] = (float)std::sin(lo);
    img[L.t_dir + 4 * k + 2] = (float)std::cos(hi); img[L.t_dir + 4 * k + 3] = (float)std::sin(hi);
  }
  for (int i = 0; i < b.ncol; ++i) {
    float* ct = &img[L.t_col + rsbk::kColSlot * i];
    ct[0] = (float)b.col_pos[i][0]; ct[1] = (float)b.col_pos[i][1]; ct[2] = (float)b.col_pos[i][2]; ct[3] = (float)b.col_radius[i];
    put_i(L.t_col + rsbk::kColSlot * i + 4, b.col_body[i]);
    ct[8] = (float)b.col_axis[i][0]; ct[9] = (float)b.col_axis[i][1]; ct[10] = (float)b.col_axis[i][2]; ct[11] = (float)b.col_rim[i];
    // contact material of the primitive against the terrain: the per-primitive override where one is set, else the world's default
    ct[5] = (float)(w->col_mu[i] >= 0 ? w->col_mu[i] : w->mu);
    ct[6] = (float)(w->col_rest[i] >= 0 ? w->col_rest[i] : w->restitution);
    ct[7] = (float)(w->col_rthr[i] >= 0 ? w->col_rthr[i] : w->res_threshold);
  }
  // self-collision pairs as byte offsets into the centre table (16 B per primitive); the entry behind the list pairs primitive 0 with itself (never a hit)
  for (int k = 0; k < n_self_pairs(w); ++k) put_i(L.t_spair + k, (16 * w->self_pairs[2 * k]) | ((16 * w->self_pairs[2 * k + 1]) << 16));
  return img;
}

// the step kernel's per-block tables (build_lds_image) and the self-collision pairs' materials, re-uploaded when a setter dirtied them.
// Joins (stream_of): callers that must not join later - a closed-loop run with its action stage in flight - call it up front.
int upload_image(rsb_world* w) {
  if (!w->image_dirty) return RSB_OK;
  const int kcap = kcap_of(w->blob, w->kmax);
    std::vector<float> img = build_lds_image(w, make_layout(w->blob, kcap, n_self_pairs(w)));
    HIP_TRY(hipMemcpyAsync(w->d_image, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
    const int np = n_self_pairs(w);
    std::vector<float> mat((size_t)4 * np, 0.f);
    for (int k = 0; k < np; ++k) {
      mat[4 * k] = (float)(w->self_mu[k] >= 0 ? w->self_mu[k] : w->mu);
      mat[4 * k + 1] = (float)(w->self_rest[k] >= 0 ? w->self_rest[k] : w->restitution);
      mat[4 * k + 2] = (float)(w->self_rthr[k] >= 0 ? w->self_rthr[k] : w->res_threshold);
    }
    if (np > 0) {
      if ((size_t)np > w->self_mat_cap) {
        if (w->d_self_mat) HIP_TRY(hipFree(w->d_self_mat));
        w->d_self_mat = nullptr; w->self_mat_cap = 0;
        HIP_TRY(hipMalloc(&w->d_self_mat, (size_t)4 * np * sizeof(float)));
        w->self_mat_cap = (size_t)np;
      }
      HIP_TRY(hipMemcpyAsync(w->d_self_mat, mat.data(), mat.size() * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
    }
    HIP_TRY(hipStreamSynchronize(stream_of(w)));   // img / mat are stack-lifetime buffers
    w->image_dirty = false;
  return RSB_OK;
}

int do_integrate(rsb_world* w, int nsub) {
  HIP_TRY(hipSetDevice(w->device));
  if (w->integ_rk4 && !w->rk4_inner) return rk4_integrate(w, nsub);      // IntegrationScheme::RUNGE_KUTTA_4: host-driven over the query kernels (rsb_rk4.hip)
  const int lpe = effective_lpe(w);
  int st = check_lpe(w, lpe);
  if (st != RSB_OK) return st;
  const int kcap = kcap_of(w->blob, w->kmax);
  StepArgs a;
  std::memset(&a, 0, sizeof a);
  a.model = w->d_model;
  a.gc = w->d_gc; a.gv = w->d_gv; a.ptarget = w->d_pt; a.dtarget = w->d_dt; a.tauff = w->d_tff;
#ifndef RSB_X_READ_ZERO_ROWS      /* (A/B switch: read the rows although they are known to be zero, as rounds 1-4 did) */
  // (a field whose raw device pointer has been handed out - rsb_device_ptr - is read whatever the host thinks it holds: ADVICE r05)
  if (w->dt_zero && !w->raw_dt) a.dtarget = nullptr;
  if (w->tff_zero && !w->raw_tff) a.tauff = nullptr;
#endif
  a.kp = w->d_kp; a.kd = w->d_kd;
  a.contacts = w->d_contacts; a.contact_count = w->d_count; a.flags = w->d_flags; a.iters = w->d_iters;
  a.heights = w->d_heights;
  a.hm_index = w->d_hm_index;
  a.warm = w->warm_start ? w->d_warm : nullptr;
  st = upload_image(w);
  if (st != RSB_OK) return st;
  a.lds_image = w->d_image;
  if (w->fuse.ptarget_src) { a.ptarget = w->fuse.ptarget_src; a.ptarget_store = w->d_pt; }
  if (w->fuse.act) { a.act = w->fuse.act; a.act_mean = w->d_env_mean; a.act_std = w->env_cfg.action_std; a.ptarget_store = w->d_pt; a.tau2_out = w->d_env_tau2; }
  uint8_t* env_done = nullptr;
  if (w->fuse.env_task) {   // rsb_env_step: reward, termination, reset and the next observation in this launch's epilogue
    a.env_reward = w->fuse.env_reward; a.env_ob = w->fuse.env_ob; env_done = w->fuse.env_done;
    a.env_fwd_coeff = w->env_cfg.forward_vel_coeff; a.env_fwd_clip = w->env_cfg.forward_vel_clip;
    a.env_torque_coeff = w->env_cfg.torque_coeff; a.env_terminal_reward = w->env_cfg.terminal_reward;
    a.tau2_out = nullptr;
  }
  a.obs_out = w->fuse.obs_out; a.obs_idx = w->fuse.obs_idx; a.obs_slots = w->fuse.obs_slots;
  const bool peer = w->fuse.peer;
  // a second contact per primitive against a height map (rsb_set_heightmap_contacts): a kernel class of its own, floating base, no peer exchange
  // an integration scheme other than semi-implicit Euler (rsb_set_integration_scheme): likewise a class of its own
  const bool th = w->integ_theta != 1.0;
  if (th && (w->blob.fixed_base || peer || ((w->hm_contacts >= 2 || (w->hm_capsule && w->n_cap > 0)) && w->terrain_type == 1) || w->blob.depth - 1 > 12)) {
    rsb::set_error("integration schemes other than SEMI_IMPLICIT: built for floating-base systems of tree depth <= 13 without the peer-mapped obs exchange and with one contact per primitive");
    return RSB_E_UNSUPPORTED;
  }
  // the classical Coulomb slip rule (rsb_set_slip_rule): a kernel class of its own (bit 32), built for the quadruped-sized models
  const bool coul = w->slip_rule == RSB_SLIP_COULOMB;
  if (coul && (w->blob.fixed_base || peer || th || w->blob.depth - 1 > 4 || kcap != 8 || ((w->hm_contacts >= 2 || (w->hm_capsule && w->n_cap > 0)) && w->terrain_type == 1))) {
    rsb::set_error("RSB_SLIP_COULOMB: built for floating-base systems of tree depth <= 5 with <= 8 contact slots, the default integration scheme, one contact per primitive, no peer-mapped obs exchange");
    return RSB_E_UNSUPPORTED;
  }
  const bool hm2 = (w->hm_contacts >= 2 || (w->hm_capsule && w->n_cap > 0)) && w->terrain_type == 1;   // class-4 kernels: more than one contact per primitive against a height map
  if (hm2 && (w->blob.fixed_base || peer || w->blob.depth - 1 > 12)) {
    rsb::set_error("two contacts per primitive against a height map: built for floating-base systems of tree depth <= 13 without the peer-mapped obs exchange");
    return RSB_E_UNSUPPORTED;
  }
  if (peer) {   // rsb_control_step with the peer-mapped obs exchange connected: rows go to every rank's gathered buffer of this step's parity
    rsb_world::Peer& P = w->peer;
    const uint32_t step = ++P.step;
    const int par = (int)(step & 1u);
    const size_t bufsz = (size_t)P.ranks * w->N * P.od;
    for (int p = 0; p < P.ranks; ++p) {
      float* pb = static_cast<float*>(P.peer_base[p]);
      a.obs_peer[p] = pb + (size_t)par * bufsz;
      a.obs_flag[p] = reinterpret_cast<uint32_t*>(pb + 2 * bufsz) + (size_t)par * RSB_MAX_RANKS + P.rank;
    }
    a.obs_ctr = reinterpret_cast<uint32_t*>(static_cast<float*>(P.base) + 2 * bufsz) + 2 * RSB_MAX_RANKS;
    a.n_obs_peers = P.ranks; a.obs_row0 = P.rank * w->N; a.obs_step = step;
    a.obs_slots = P.slots; a.obs_idx = P.idx.empty() ? nullptr : P.d_idx;
  }
  a.early_term = (w->early_term && w->fuse.have_allowed) ? 1 : 0;
  a.do_reset = w->fuse.do_reset; a.allowed = w->fuse.allowed; a.gc0 = w->fuse.gc0; a.gv0 = w->fuse.gv0; a.reset_rows = w->fuse.rows;
  if (!a.do_reset) { a.gc0 = w->d_gc; a.gv0 = w->d_gv; a.reset_rows = w->N; }   // never dereferenced, but keep the pointers valid
  // resident launch: K control steps in this ONE launch (the caller has checked resident_class)
  int res_cl = -1;
  if (w->fuse.res_steps > 0) {
    const rsb_world::Fuse& f = w->fuse;
    int width = 0;
    if (f.res_stage == 2) for (int l = 0; l <= f.res_mlp.n_layers; ++l) width = std::max(width, (int)f.res_mlp.dims[l]);
    res_cl = resident_class(w, f.res_stage, width);
    if (res_cl < 0 || w->launch_mask) { w->fuse = rsb_world::Fuse(); w->launch_mask = nullptr; if (res_cl >= 0) rsb::set_error("no resident launch with an env mask"); return RSB_E_UNSUPPORTED; }
    a.res_steps = f.res_steps; a.res_full = w->res_full ? 1 : 0;
    a.res_targets = f.res_targets; a.res_period = f.res_period; a.res_first = f.res_first;
    a.res_obs_stride = f.res_obs_stride; a.res_done_stride = f.res_done_stride; a.res_pass_global0 = f.res_pass_global0;
    if (f.res_stage == 1) a.res_pol.lin = f.res_lin;
    if (f.res_stage == 2) a.res_pol.mlp = f.res_mlp;
    if (f.res_stage == 0) { a.ptarget = f.res_targets + (size_t)(f.res_first % f.res_period) * ((size_t)w->N * w->blob.nq); a.ptarget_store = w->d_pt; }
  }
  uint8_t* res_done = w->fuse.res_done;
  const int res_steps = std::max(w->fuse.res_steps, 1);
  const bool closed_loop = w->fuse.closed_loop;      // a step of rsb_closed_loop_run: waits for the action stage's word instead of its predecessor's
  const bool pipe_ok = w->fuse.pipeline && (!w->fuse.env_task || closed_loop);
  const rsb_world::Fuse fuse_in = w->fuse;
  w->fuse = rsb_world::Fuse();
  a.prof = w->d_prof;
  a.dbg = w->dbg_env >= 0 ? w->d_dbg : nullptr;
  a.dbg_env = w->dbg_env;
  a.N = w->N; a.nsub = nsub; a.kmax = w->kmax; a.control_mode = w->control_mode;
  a.nb = w->blob.nb; a.nq = w->blob.nq; a.nv = w->blob.nv; a.ncol = w->blob.ncol; a.depth = w->blob.depth;
  a.cw = round4(6 + w->blob.depth - 1); a.max_kid = w->max_kid; a.fixed_base = w->blob.fixed_base; a.chain = (w->chain && lpe == 16) ? 1 : 0;
  a.dt = (float)w->dt; a.gx = (float)w->gravity[0]; a.gy = (float)w->gravity[1]; a.gz = (float)w->gravity[2];
  a.mu = (float)w->mu; a.erp = (float)w->erp;
  a.alpha_init = (float)w->alpha_init; a.alpha_min = (float)w->alpha_min; a.alpha_decay = (float)w->alpha_decay;
  a.threshold = (float)w->threshold; a.max_iter = w->max_iter; a.section_rounds = w->section_rounds;
  a.multi_depth = w->multi_depth; a.multi_light = w->multi_light; a.multi_freeze_after = w->multi_freeze_after; a.multi_stall_window = w->multi_stall_window;
  a.anderson = w->kmax > 8 ? w->anderson : 0;   // (rsb.h, oracle: worlds with kmax > 8 only - the large kernel classes also serve deep models at kmax <= 8)
  a.anderson_clip = (float)w->anderson_clip;
  a.hm_contacts = w->hm_contacts; a.hm_second_cos = (float)w->hm_second_cos; a.hm_slots = hm_slots_for(w->blob);
  a.hm_capsule = w->hm_capsule ? w->n_cap : 0; a.hm_cap = w->d_cap;
  a.integ_theta = (float)w->integ_theta;
  a.stall_window = w->stall_window; a.stall_factor = (float)w->stall_factor; a.freeze_after = w->freeze_after; a.refine = w->refine; a.settle_tol = (float)w->settle_tol; a.restitution = (float)w->restitution; a.res_threshold = (float)w->res_threshold;
  a.terrain_type = w->terrain_type; a.hm_xs = w->hm_xs; a.hm_ys = w->hm_ys; a.ground_z = (float)w->ground_z;
  if (w->terrain_type == 1) {
    double dx = w->hm_xsize / (w->hm_xs - 1), dy = w->hm_ysize / (w->hm_ys - 1);
    a.hm_x0 = (float)(w->hm_cx - 0.5 * w->hm_xsize); a.hm_y0 = (float)(w->hm_cy - 0.5 * w->hm_ysize);
    a.hm_dx = (float)dx; a.hm_dy = (float)dy; a.hm_inv_dx = (float)(1.0 / dx); a.hm_inv_dy = (float)(1.0 / dy);
    a.hm_max = w->hm_max;
  }
  a.n_self = n_self_pairs(w); a.self_mat = w->d_self_mat;
  a.L = make_layout(w->blob, kcap, a.n_self);
  const size_t lds_bytes = lds_bytes_for(w->blob, kcap, lpe, n_self_pairs(w));
  static const bool poison = std::getenv("RSB_POISON_LDS") != nullptr;  // debug aid, see tests/test_gpu_properties.py
  a.poison_lds = poison ? 1 : 0;
  static const bool prof_fine = std::getenv("RSB_PROF_FINE") != nullptr;  // debug aid: also time searches / Newton steps / epilogues
  a.prof_fine = prof_fine ? 1 : 0;
  a.lds_floats = (int)(lds_bytes / sizeof(float));
  const bool prof = a.prof != nullptr || a.dbg != nullptr || a.poison_lds != 0;
  a.done_out = env_done ? env_done : res_done ? res_done : w->d_done_out;
  a.tau_out = w->want_genf ? w->d_genf : nullptr;
  a.env_mask = w->launch_mask; w->launch_mask = nullptr;
  // ---- pipelined control steps: this launch goes to one of the two private streams, behind a gate that lets it start only when the launch
  // before it (on the other stream) has been dispatched completely - its workgroups wait for their predecessors' envs (open loop) or for the
  // action stage's rows (closed loop), which therefore must all be running or done (no deadlock: a waiting workgroup never keeps a
  // predecessor off the chip).  rsb_pipeline.hip holds the bookkeeping.
  hipStream_t ls = nullptr;
  const bool pipelined = w->pipe_on && pipe_ok && !prof && !peer && !a.env_mask && res_cl < 0;
  if (pipelined) {
    const int blocks = (w->N + (64 / lpe) - 1) / (64 / lpe);
    st = pipe_begin_launch(w, a, blocks, closed_loop, &ls);
    if (st != RSB_OK) return st;
    if (!w->pipe_log_suppress) {      // what a faulted pipeline replays in lock-step (pipe_recover)
      rsb_world::PipeLog e;
      e.f = fuse_in; e.nsub = nsub; e.done_out = w->d_done_out;
      w->pipe_log.push_back(e);
    }
    w->pipe_time_logged += nsub * w->dt;
  } else {
    ls = stream_of(w);
    if (w->pipe_dep) { HIP_TRY(hipStreamWaitEvent(ls, w->pipe_dep, 0)); w->pipe_dep = nullptr; }
  }
  w->launch_stream = ls;
  hipEvent_t e0 = w->ev0, e1 = w->ev1;
  const bool rec = w->timing && (w->launch_index++ % w->timing_stride == 0);
  if (rec && !w->ring0.empty()) { e0 = w->ring0[w->ring_next]; e1 = w->ring1[w->ring_next]; }
  if (rec) HIP_TRY(hipEventRecord(e0, ls));
  // kernel classes by the deepest body level (support-chain capacity of the contact-column / Delassus phases) and by the base (fixed-base systems have a class of their own)
  const int mlv = w->blob.depth - 1;
  // a specialised code object of the class this launch is about to run (rsb_spec.hip): same kernel, the model's dimensions and the world's switches as constants
  hipFunction_t spec_fn = nullptr;
  if (w->spec_mode != RSB_SPEC_OFF && mlv <= 16 && !(prof && (res_cl >= 0 || peer || a.pipe_prog))) {   // (the classes without a profiling twin refuse `prof` below)
    rsbw::SpecClass sc{lpe, kcap, 0, mlv <= 4 ? 4 : mlv <= 12 ? 12 : 16};
    if (res_cl >= 0) sc = mlv <= 4 ? rsbw::SpecClass{16, 8, res_cl, 4} : rsbw::SpecClass{32, 16, res_cl, 12};
    else {
      if (mlv > 4) sc.kmax = 16;
      sc.cl = w->blob.fixed_base ? 1 : (coul && mlv <= 4) ? 32 : (hm2 && mlv <= 12) ? 4 : (th && mlv <= 12) ? 8 : (peer && mlv <= 12) ? 2 : 0;
      if (coul && mlv <= 4) sc.kmax = 8;
      if (!(sc.cl & 2) && a.pipe_prog) sc.cl |= 16;     // the class's pipelined twin (launch_lpe)
    }
    sc.prof = prof ? 1 : 0;
    spec_fn = rsbw::spec_find(w, sc, a);
  }
  if (spec_fn) {
    const int epw = 64 / (res_cl >= 0 ? (mlv <= 4 ? 16 : 32) : lpe);
    st = rsbw::spec_launch(spec_fn, a, (w->N + epw - 1) / epw, lds_bytes, w->launch_stream);
    if (st == RSB_OK) { ++w->spec_launches; if (res_cl >= 0) ++w->res_launches; }
  } else if (res_cl >= 0) {
    st = launch_resident(w, a, lds_bytes, res_cl, mlv <= 4);
    if (st == RSB_OK) { ++w->res_launches; ++w->generic_launches; }
  } else if (mlv <= 4) {
    if (w->blob.fixed_base) st = kcap == 8 ? launch_lpe<8, 1, 4>(w, a, lds_bytes, lpe, prof) : launch_lpe<16, 1, 4>(w, a, lds_bytes, lpe, prof);
    else if (coul) st = launch_lpe<8, 32, 4>(w, a, lds_bytes, lpe, prof);
    else if (hm2) st = kcap == 8 ? launch_lpe<8, 4, 4>(w, a, lds_bytes, lpe, prof) : launch_lpe<16, 4, 4>(w, a, lds_bytes, lpe, prof);
    else if (th) st = kcap == 8 ? launch_lpe<8, 8, 4>(w, a, lds_bytes, lpe, prof) : launch_lpe<16, 8, 4>(w, a, lds_bytes, lpe, prof);
    else if (peer) st = kcap == 8 ? launch_lpe<8, 2, 4>(w, a, lds_bytes, lpe, prof) : launch_lpe<16, 2, 4>(w, a, lds_bytes, lpe, prof);
    else st = kcap == 8 ? launch_lpe<8, 0, 4>(w, a, lds_bytes, lpe, prof) : launch_lpe<16, 0, 4>(w, a, lds_bytes, lpe, prof);
  } else if (mlv <= 12) {
    st = w->blob.fixed_base ? launch_lpe<16, 1, 12>(w, a, lds_bytes, lpe, prof) : hm2 ? launch_lpe<16, 4, 12>(w, a, lds_bytes, lpe, prof)
         : th ? launch_lpe<16, 8, 12>(w, a, lds_bytes, lpe, prof) : peer ? launch_lpe<16, 2, 12>(w, a, lds_bytes, lpe, prof) : launch_lpe<16, 0, 12>(w, a, lds_bytes, lpe, prof);
  } else if (mlv <= 16) {
    st = w->blob.fixed_base ? launch_lpe<16, 1, 16>(w, a, lds_bytes, lpe, prof) : launch_lpe<16, 0, 16>(w, a, lds_bytes, lpe, prof);
  } else {
    rsb::set_error("model outside the compiled kernel classes (tree depth <= 17)");
    return RSB_E_UNSUPPORTED;
  }
  if (st != RSB_OK) return st;
  if (!spec_fn && res_cl < 0) ++w->generic_launches;
  if (pipelined) pipe_end_launch(w, a, ls);
  if (rec) {
    HIP_TRY(hipEventRecord(e1, ls));
    if (!w->ring0.empty()) { w->ring_next = (w->ring_next + 1) % w->ring0.size(); if (w->ring_count < w->ring0.size()) ++w->ring_count; }
  }
  w->world_time += (double)res_steps * nsub * w->dt;
  w->integrate1_valid = false;
  // (the fused epilogue left the observation the NEXT step starts from in the world's own buffer - unless the caller holds raw pointers to the state rows
  //  and may write through them behind the library's back: then every closed-loop run recomputes its first observation, as the lock-step path does)
  w->env_ob_valid = a.env_ob != nullptr && a.env_ob == w->d_env_ob && !w->raw_state;
  return RSB_OK;
}

// does a field the caller uploads hold nothing but zeros?  (host data: looked at; device data: unknown -> no)
bool all_zero(const float* src, size_t n, int space) {
  if (space != RSB_HOST) return false;
  for (size_t i = 0; i < n; ++i) if (src[i] != 0.f) return false;
  return true;
}
int copy_in(rsb_world* w, float* dst, const float* src, size_t n, int space) {
  HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(float), space == RSB_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, stream_of(w)));
  if (space == RSB_HOST) HIP_TRY(hipStreamSynchronize(stream_of(w)));  // the caller may reuse its host buffer
  return RSB_OK;
}
int copy_out(rsb_world* w, void* dst, const void* src, size_t bytes, int space) {
  hipStream_t s = stream_of(w);
  const int fs = fault_status(w);      // a read that joined a faulted pipeline says so ONCE (nothing is copied; the recovered state is there for the next call)
  if (fs != RSB_OK) return fs;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, space == RSB_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
  if (space == RSB_HOST) HIP_TRY(hipStreamSynchronize(s));
  return RSB_OK;
}

int launch_dynamics_query(rsb_world* w, hipStream_t s) {
  const size_t N = w->N, nv = w->blob.nv;
  if (!w->d_M) {
    HIP_TRY(hipMalloc(&w->d_M, N * nv * nv * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_h, N * nv * sizeof(float)));
  }
  if (!w->d_Minv) {
    HIP_TRY(hipMalloc(&w->d_Minv, N * nv * nv * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_Mwork, N * nv * nv * sizeof(float)));
  }
  rsbq::QueryArgs qa;
  qa.model = w->d_model; qa.gc = w->d_gc; qa.gv = w->d_gv; qa.M = w->d_M; qa.h = w->d_h; qa.N = w->N;
  qa.gx = (float)w->gravity[0]; qa.gy = (float)w->gravity[1]; qa.gz = (float)w->gravity[2];
  if (rsbq::launch_query(qa, w->blob.nb, s) != 0) { rsb::set_error("RUNGE_KUTTA_4: query kernel launch failed"); return RSB_E_HIP; }
  hipLaunchKernelGGL(rsbq::rsb_minv_kernel, dim3((w->N + 63) / 64), dim3(64), 0, s, w->d_M, w->d_Mwork, w->d_Minv, w->N, w->blob.nv);
  HIP_TRY(hipGetLastError());
  return RSB_OK;
}
int launch_env_obs(rsb_world* w, float* dst, hipStream_t s) {
  const int total = w->N * (10 + 2 * (w->blob.nv - 6));
  hipLaunchKernelGGL(env_obs_kernel, dim3((total + 255) / 256), dim3(256), 0, s, dst, w->d_gc, w->d_gv, w->N, w->blob.nq, w->blob.nv);
  HIP_TRY(hipGetLastError());
  return RSB_OK;
}

}  // namespace rsbw
using namespace rsbw;

extern "C" {

int rsb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rsb_create(const rsb_model* m, int num_envs, int device, rsb_world** out) {
  if (!m || !out || num_envs <= 0) { rsb::set_error("rsb_create: bad argument"); return RSB_E_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    rsb::set_error("rsb_create: no HIP device is visible (raisimlib_amd has no CPU fallback)");
    return RSB_E_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { rsb::set_error("rsb_create: device index out of range"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(device));
  // every early return below (HIP_TRY) goes through rsb_destroy, which frees whatever has been allocated so far
  struct Destroyer { void operator()(rsb_world* p) const { rsb_destroy(p); } };
  std::unique_ptr<rsb_world, Destroyer> w(new rsb_world());
  w->spec_mode = rsbw::spec_default_mode();
  w->blob = m->blob;
  w->N = num_envs;
  w->device = device;

  const int nq = w->blob.nq, nv = w->blob.nv;
  const size_t N = (size_t)num_envs;
  HIP_TRY(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&w->ev0));
  HIP_TRY(hipEventCreate(&w->ev1));
  auto dm = std::make_unique<DevModel>();
  build_dev_model(w->blob, dm.get());
  HIP_TRY(hipMalloc(&w->d_model, sizeof(DevModel)));
  HIP_TRY(hipMemcpy(w->d_model, dm.get(), sizeof(DevModel), hipMemcpyHostToDevice));
  w->max_kid = dm->max_kid;
  w->chain = w->blob.nb <= 16;
  for (int b = 1; b < w->blob.nb; ++b) if (dm->level[b] >= 2 && dm->parent[b] != b - 1) w->chain = false;
  HIP_TRY(hipMalloc(&w->d_gc, N * nq * sizeof(float)));
  HIP_TRY(hipMalloc(&w->d_gv, N * nv * sizeof(float)));
  HIP_TRY(hipMalloc(&w->d_pt, N * nq * sizeof(float)));
  HIP_TRY(hipMalloc(&w->d_dt, N * nv * sizeof(float)));
  HIP_TRY(hipMalloc(&w->d_tff, N * nv * sizeof(float)));
  HIP_TRY(hipMalloc(&w->d_kp, nv * sizeof(float)));
  HIP_TRY(hipMalloc(&w->d_kd, nv * sizeof(float)));
  HIP_TRY(hipMalloc(&w->d_contacts, N * RSB_MAX_CONTACTS * sizeof(rsb_contact)));
  HIP_TRY(hipMalloc(&w->d_count, N * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&w->d_flags, N * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&w->d_iters, N * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&w->d_obs_idx, RSB_MAX_COLLISIONS * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&w->d_image, (size_t)make_layout_pitch(w->blob, 8, w->blob.ncol * (w->blob.ncol - 1) / 2, rsbk::kModelPitch).shared_total * sizeof(float)));   // (room for every primitive pair)
  w->self_ignore.assign((size_t)w->blob.nb * w->blob.nb, 0);
  w->self_pairs = enumerate_self_pairs(w->blob, w->self_ignore);
  w->self_mu.assign(w->self_pairs.size() / 2, -1.0); w->self_rest = w->self_mu; w->self_rthr = w->self_mu;
  w->h_kp.assign(nv, 0.f); w->h_kd.assign(nv, 0.f);
  w->col_mu.assign(RSB_MAX_COLLISIONS, -1.0); w->col_rest.assign(RSB_MAX_COLLISIONS, -1.0); w->col_rthr.assign(RSB_MAX_COLLISIONS, -1.0);
  HIP_TRY(hipMalloc(&w->d_warm, N * (size_t)rsbk::kWarmRow * sizeof(float)));
  HIP_TRY(hipMemset(w->d_warm, 0, N * (size_t)rsbk::kWarmRow * sizeof(float)));
  HIP_TRY(hipMemset(w->d_gc, 0, N * nq * sizeof(float)));
  HIP_TRY(hipMemset(w->d_gv, 0, N * nv * sizeof(float)));
  HIP_TRY(hipMemset(w->d_pt, 0, N * nq * sizeof(float)));
  HIP_TRY(hipMemset(w->d_dt, 0, N * nv * sizeof(float)));
  HIP_TRY(hipMemset(w->d_tff, 0, N * nv * sizeof(float)));
  HIP_TRY(hipMemset(w->d_kp, 0, nv * sizeof(float)));
  HIP_TRY(hipMemset(w->d_kd, 0, nv * sizeof(float)));
  HIP_TRY(hipMemset(w->d_contacts, 0, N * RSB_MAX_CONTACTS * sizeof(rsb_contact)));
  HIP_TRY(hipMemset(w->d_count, 0, N * sizeof(int32_t)));
  HIP_TRY(hipMemset(w->d_flags, 0, N * sizeof(int32_t)));
  HIP_TRY(hipMemset(w->d_iters, 0, N * sizeof(int32_t)));
  // default state: identity orientation, everything else zero
  std::vector<float> gc(N * nq, 0.f);
  for (size_t e = 0; e < N; ++e) gc[e * nq + 3] = 1.f;
  HIP_TRY(hipMemcpy(w->d_gc, gc.data(), gc.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  *out = w.release();
  return RSB_OK;
}

int rsb_destroy(rsb_world* w) {
  if (!w) return RSB_OK;
  (void)hipSetDevice(w->device);
  if (w->stream) (void)hipStreamSynchronize(stream_of(w));
  (void)rsb_comm_destroy(w);
  (void)rsb_obs_peer_destroy(w);
  void* ptrs[] = {w->d_model, w->d_gc, w->d_gv, w->d_pt, w->d_dt, w->d_tff, w->d_kp, w->d_kd, w->d_heights,
                  w->d_tmp_gc, w->d_tmp_gv, w->d_tmp_mask, w->d_M, w->d_h, w->d_Minv, w->d_Mwork, w->d_obs_idx, w->d_dbg, w->d_prof, w->d_contacts,
                  w->d_env_mean, w->d_env_gc0, w->d_env_gv0, w->d_env_io, w->d_env_ob, w->d_env_reward, w->d_env_tau2, w->d_env_done, w->d_warm, w->d_image, w->d_self_mat, w->d_genf, w->d_hm_index, w->d_launch_mask, w->d_view_masks, w->d_cap, w->d_obs_local, w->d_obs_all,
                  w->d_count, w->d_flags, w->d_iters};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (hipEvent_t e : w->ring0) (void)hipEventDestroy(e);
  for (hipEvent_t e : w->ring1) (void)hipEventDestroy(e);
  if (w->ev0) (void)hipEventDestroy(w->ev0);
  if (w->ev1) (void)hipEventDestroy(w->ev1);
  pipe_destroy(w);
  if (w->d_rk) (void)hipFree(w->d_rk);
  if (w->d_env_act) (void)hipFree(w->d_env_act);
  if (w->d_env_gc0_rows) { (void)hipFree(w->d_env_gc0_rows); (void)hipFree(w->d_env_gv0_rows); }
  if (w->own_stream && w->stream) (void)hipStreamDestroy(w->stream);
  delete w;
  return RSB_OK;
}

int rsb_set_stream(rsb_world* w, void* hip_stream) {
  if (!w) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  if (w->own_stream && w->stream) HIP_TRY(hipStreamDestroy(w->stream));
  w->stream = (hipStream_t)hip_stream;  // NULL is the (legacy) default stream, a valid stream to borrow
  w->own_stream = false;
  return RSB_OK;
}
void* rsb_get_stream(rsb_world* w) { return w ? (void*)stream_of(w) : nullptr; }
int rsb_synchronize(rsb_world* w) {
  if (!w) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  return fault_status(w);
}

int rsb_num_envs(const rsb_world* w) { return w ? w->N : RSB_E_INVALID; }
int rsb_dims(const rsb_world* w, int* nb, int* nq, int* nv, int* ncol, int* kmax) {
  if (!w) return RSB_E_INVALID;
  if (nb) *nb = w->blob.nb;
  if (nq) *nq = w->blob.nq;
  if (nv) *nv = w->blob.nv;
  if (ncol) *ncol = w->blob.ncol;
  if (kmax) *kmax = w->kmax;
  return RSB_OK;
}

int rsb_set_timestep(rsb_world* w, double dt) {
  if (!w || !(dt > 0)) { rsb::set_error("rsb_set_timestep: dt must be positive"); return RSB_E_INVALID; }
  w->dt = dt;
  return RSB_OK;
}
double rsb_get_timestep(const rsb_world* w) { return w ? w->dt : 0.0; }
double rsb_get_world_time(const rsb_world* w) { return w ? w->world_time : 0.0; }
int rsb_set_gravity(rsb_world* w, const double g[3]) {
  if (!w || !g) return RSB_E_INVALID;
  for (int i = 0; i < 3; ++i) w->gravity[i] = g[i];
  return RSB_OK;
}
int rsb_set_erp(rsb_world* w, double erp) { if (!w) return RSB_E_INVALID; w->erp = erp; return RSB_OK; }
int rsb_set_friction(rsb_world* w, double mu) {
  if (!w || mu < 0) { rsb::set_error("rsb_set_friction: mu must be >= 0"); return RSB_E_INVALID; }
  w->mu = mu;
  w->image_dirty = true;
  return RSB_OK;
}
int rsb_set_material(rsb_world* w, double mu, double restitution, double res_threshold) {
  if (!w || mu < 0 || restitution < 0 || restitution > 1 || res_threshold < 0) { rsb::set_error("rsb_set_material: mu >= 0, 0 <= restitution <= 1, res_threshold >= 0"); return RSB_E_INVALID; }
  w->mu = mu; w->restitution = restitution; w->res_threshold = res_threshold;
  w->image_dirty = true;
  return RSB_OK;
}
int rsb_set_collision_materials(rsb_world* w, const double* mu, const double* restitution, const double* res_threshold) {
  if (!w) { rsb::set_error("rsb_set_collision_materials: null world"); return RSB_E_INVALID; }
  for (int i = 0; i < w->blob.ncol; ++i) {
    if (restitution && restitution[i] > 1.0) { rsb::set_error("rsb_set_collision_materials: restitution <= 1"); return RSB_E_INVALID; }
    if ((mu && !std::isfinite(mu[i])) || (restitution && !std::isfinite(restitution[i])) || (res_threshold && !std::isfinite(res_threshold[i]))) {
      rsb::set_error("rsb_set_collision_materials: non-finite material value"); return RSB_E_INVALID;
    }
  }
  for (int i = 0; i < w->blob.ncol; ++i) {
    w->col_mu[i] = mu ? mu[i] : -1.0;
    w->col_rest[i] = restitution ? restitution[i] : -1.0;
    w->col_rthr[i] = res_threshold ? res_threshold[i] : -1.0;
  }
  w->image_dirty = true;
  return RSB_OK;
}
int rsb_set_self_collision(rsb_world* w, int enable) {
  if (!w) { rsb::set_error("rsb_set_self_collision: null world"); return RSB_E_INVALID; }
  w->self_collision = enable != 0;
  w->image_dirty = true;
  return RSB_OK;
}
int rsb_ignore_collision_between(rsb_world* w, int body_a, int body_b) {
  if (!w || body_a < 0 || body_b < 0 || body_a >= w->blob.nb || body_b >= w->blob.nb) { rsb::set_error("rsb_ignore_collision_between: body index out of range"); return RSB_E_INVALID; }
  w->self_ignore[(size_t)body_a * w->blob.nb + body_b] = 1;
  w->self_ignore[(size_t)body_b * w->blob.nb + body_a] = 1;
  w->self_pairs = enumerate_self_pairs(w->blob, w->self_ignore);
  w->self_mu.assign(w->self_pairs.size() / 2, -1.0); w->self_rest = w->self_mu; w->self_rthr = w->self_mu;   // (the pair list changed: overrides are dropped)
  w->image_dirty = true;
  return RSB_OK;
}
int rsb_self_collision_pairs(const rsb_world* w, int32_t* pairs, int capacity) {
  if (!w) { rsb::set_error("rsb_self_collision_pairs: null world"); return RSB_E_INVALID; }
  const int np = (int)w->self_pairs.size() / 2;
  for (int k = 0; pairs && k < np && k < capacity; ++k) { pairs[2 * k] = w->self_pairs[2 * k]; pairs[2 * k + 1] = w->self_pairs[2 * k + 1]; }
  return np;
}
int rsb_set_self_collision_materials(rsb_world* w, const double* mu, const double* restitution, const double* res_threshold) {
  if (!w) { rsb::set_error("rsb_set_self_collision_materials: null world"); return RSB_E_INVALID; }
  const size_t np = w->self_pairs.size() / 2;
  for (size_t k = 0; k < np; ++k) {
    if (restitution && restitution[k] > 1.0) { rsb::set_error("rsb_set_self_collision_materials: restitution <= 1"); return RSB_E_INVALID; }
    if ((mu && !std::isfinite(mu[k])) || (restitution && !std::isfinite(restitution[k])) || (res_threshold && !std::isfinite(res_threshold[k]))) {
      rsb::set_error("rsb_set_self_collision_materials: non-finite material value"); return RSB_E_INVALID;
    }
  }
  for (size_t k = 0; k < np; ++k) {
    w->self_mu[k] = mu ? mu[k] : -1.0;
    w->self_rest[k] = restitution ? restitution[k] : -1.0;
    w->self_rthr[k] = res_threshold ? res_threshold[k] : -1.0;
  }
  w->image_dirty = true;
  return RSB_OK;
}
int rsb_set_contact_solver_param(rsb_world* w, double alpha_init, double alpha_min, double alpha_decay,
                                 int max_iter, double threshold) {
  if (!w || max_iter < 1 || !(threshold >= 0) || !(alpha_init > 0)) { rsb::set_error("rsb_set_contact_solver_param: bad argument"); return RSB_E_INVALID; }
  w->alpha_init = alpha_init; w->alpha_min = alpha_min; w->alpha_decay = alpha_decay;
  w->max_iter = max_iter; w->threshold = threshold;
  return RSB_OK;
}
int rsb_set_solver_stagnation_exit(rsb_world* w, int window, double factor) {
  if (!w || window < 0 || !(factor > 0)) { rsb::set_error("rsb_set_solver_stagnation_exit: window >= 0, factor > 0"); return RSB_E_INVALID; }
  w->stall_window = window; w->stall_factor = factor;
  return RSB_OK;
}
int rsb_set_solver_friction_lag(rsb_world* w, int freeze_after, int refine, double settle_tol) {
  if (!w || freeze_after < 0 || !(settle_tol >= 0.0)) { rsb::set_error("rsb_set_solver_friction_lag: freeze_after >= 0, settle_tol >= 0"); return RSB_E_INVALID; }
  w->freeze_after = freeze_after; w->refine = refine != 0; w->settle_tol = settle_tol;
  return RSB_OK;
}
int rsb_set_solver_multi_contact(rsb_world* w, int depth, int light_passes, int freeze_after, int stall_window) {
  if (!w || depth < 0 || freeze_after < 0 || stall_window < 0) { rsb::set_error("rsb_set_solver_multi_contact: depth, freeze_after, stall_window >= 0"); return RSB_E_INVALID; }
  w->multi_depth = depth; w->multi_light = light_passes != 0; w->multi_freeze_after = freeze_after; w->multi_stall_window = stall_window;
  return RSB_OK;
}
int rsb_set_solver_anderson(rsb_world* w, int first_sweep, double clip) {
  if (!w || first_sweep < 0 || !(clip > 0.0)) { rsb::set_error("rsb_set_solver_anderson: first_sweep >= 0 (0 = off), clip > 0"); return RSB_E_INVALID; }
  w->anderson = first_sweep; w->anderson_clip = clip;
  return RSB_OK;
}
int rsb_set_heightmap_contacts(rsb_world* w, int per_primitive, double min_angle_deg) {
  if (!w || per_primitive < 1 || per_primitive > 2 || !(min_angle_deg > 0.0 && min_angle_deg < 90.0)) {
    rsb::set_error("rsb_set_heightmap_contacts: per_primitive is 1 or 2, 0 < min_angle_deg < 90"); return RSB_E_INVALID;
  }
  w->hm_contacts = per_primitive; w->hm_second_cos = std::cos(min_angle_deg * 3.14159265358979323846 / 180.0);
  return RSB_OK;
}
int rsb_set_capsule_contacts(rsb_world* w, int on) {
  if (!w) { rsb::set_error("rsb_set_capsule_contacts: null world"); return RSB_E_INVALID; }
  if (on && !w->d_cap) {      // the end pairs of the model's capsules and cylinders (rsb_model_blob::col_capsule), once
    std::vector<int32_t> pairs;
    for (int s = 0; s < w->blob.ncol; ++s)
      if (w->blob.col_capsule[s] != 0) { pairs.push_back(s); pairs.push_back(w->blob.col_capsule[s] < 0 ? -1 : w->blob.col_capsule[s] - 1); }   // (s, -1): a box headed by corner s
    w->n_cap = (int)pairs.size() / 2;
    if (w->n_cap > 0) {
      HIP_TRY(hipSetDevice(w->device));
      HIP_TRY(hipMalloc(&w->d_cap, pairs.size() * sizeof(int32_t)));
      HIP_TRY(hipMemcpy(w->d_cap, pairs.data(), pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
  }
  w->hm_capsule = on != 0;
  return RSB_OK;
}
int rsb_set_slip_rule(rsb_world* w, int rule) {
  if (!w || (rule != RSB_SLIP_ENERGY && rule != RSB_SLIP_COULOMB)) { rsb::set_error("rsb_set_slip_rule: RSB_SLIP_ENERGY or RSB_SLIP_COULOMB"); return RSB_E_INVALID; }
  w->slip_rule = rule;
  return RSB_OK;
}
int rsb_set_integration_scheme(rsb_world* w, int scheme) {
  if (!w) return RSB_E_INVALID;
  (void)stream_of(w);
  const bool was_rk4 = w->integ_rk4;
  w->integ_rk4 = false;
  if (scheme == RSB_INTEGRATION_SEMI_IMPLICIT) w->integ_theta = 1.0;
  else if (scheme == RSB_INTEGRATION_EULER) w->integ_theta = 0.0;
  else if (scheme == RSB_INTEGRATION_TRAPEZOID) w->integ_theta = 0.5;
  else if (scheme == RSB_INTEGRATION_RUNGE_KUTTA_4) { w->integ_theta = 1.0; w->integ_rk4 = true; }
  else { w->integ_rk4 = was_rk4; rsb::set_error("rsb_set_integration_scheme: unknown scheme"); return RSB_E_INVALID; }
  return RSB_OK;
}
int rsb_set_early_termination(rsb_world* w, int on) {
  if (!w) return RSB_E_INVALID;
  w->early_term = on != 0;
  return RSB_OK;
}
int rsb_set_solver_warm_start(rsb_world* w, int on) {
  if (!w) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  w->warm_start = on != 0;
  HIP_TRY(hipMemsetAsync(w->d_warm, 0, (size_t)w->N * rsbk::kWarmRow * sizeof(float), stream_of(w)));
  return RSB_OK;
}
int rsb_set_max_contacts(rsb_world* w, int kmax) {
  if (!w || kmax < 1 || kmax > RSB_MAX_CONTACTS) { rsb::set_error("rsb_set_max_contacts: 1..RSB_MAX_CONTACTS"); return RSB_E_INVALID; }
  w->kmax = kmax;
  return RSB_OK;
}
int rsb_set_lanes_per_env(rsb_world* w, int lanes) {
  if (!w) return RSB_E_INVALID;
  if (lanes != 0) { int st = check_lpe(w, lanes); if (st != RSB_OK) return st; }
  w->lpe = lanes;
  return RSB_OK;
}
int rsb_get_lanes_per_env(const rsb_world* w) { return w ? effective_lpe(w) : RSB_E_INVALID; }

int rsb_set_ground(rsb_world* w, double height) {
  if (!w) return RSB_E_INVALID;
  w->terrain_type = 0; w->ground_z = height;
  return RSB_OK;
}
int rsb_set_heightmaps(rsb_world* w, int n_maps, int xs, int ys, double x_size, double y_size, double cx, double cy,
                       const float* heights, const int32_t* env_map) {
  if (!w || !heights || n_maps < 1 || xs < 2 || ys < 2 || !(x_size > 0) || !(y_size > 0) || (n_maps > 1 && !env_map)) {
    rsb::set_error("rsb_set_heightmaps: bad argument");
    return RSB_E_INVALID;
  }
  if (env_map)
    for (int e = 0; e < w->N; ++e)
      if (env_map[e] < 0 || env_map[e] >= n_maps) { rsb::set_error("rsb_set_heightmaps: env_map entry out of range"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  if (w->d_heights) { HIP_TRY(hipFree(w->d_heights)); w->d_heights = nullptr; }
  if (w->d_hm_index) { HIP_TRY(hipFree(w->d_hm_index)); w->d_hm_index = nullptr; }
  const size_t n = (size_t)n_maps * xs * ys;
  HIP_TRY(hipMalloc(&w->d_heights, n * sizeof(float)));
  HIP_TRY(hipMemcpy(w->d_heights, heights, n * sizeof(float), hipMemcpyHostToDevice));
  if (env_map) {
    HIP_TRY(hipMalloc(&w->d_hm_index, (size_t)w->N * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(w->d_hm_index, env_map, (size_t)w->N * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  w->terrain_type = 1; w->hm_xs = xs; w->hm_ys = ys; w->hm_xsize = x_size; w->hm_ysize = y_size; w->hm_cx = cx; w->hm_cy = cy;
  w->hm_max = heights[0];   // highest sample of all maps: the narrow phase skips every sphere whose lowest point lies above it
  for (size_t i = 1; i < n; ++i) if (heights[i] > w->hm_max) w->hm_max = heights[i];
  return RSB_OK;
}
int rsb_set_heightmap(rsb_world* w, int xs, int ys, double x_size, double y_size, double cx, double cy, const float* heights) {
  return rsb_set_heightmaps(w, 1, xs, ys, x_size, y_size, cx, cy, heights, nullptr);
}

int rsb_set_state(rsb_world* w, const float* gc, const float* gv, const uint8_t* mask, int space) {
  if (!w) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  const size_t N = w->N, nq = w->blob.nq, nv = w->blob.nv;
  w->integrate1_valid = false; w->env_ob_valid = false;
  const int n6 = rsbk::kWarmRow;
  if (!mask) {
    if (n6 > 0) hipLaunchKernelGGL(warm_clear_kernel, dim3((N * n6 + 255) / 256), dim3(256), 0, stream_of(w), w->d_warm, (const uint8_t*)nullptr, (int)N, n6);
    if (gc) { int st = copy_in(w, w->d_gc, gc, N * nq, space); if (st) return st; }
    if (gv) { int st = copy_in(w, w->d_gv, gv, N * nv, space); if (st) return st; }
    return RSB_OK;
  }
  const uint8_t* dmask = mask;
  const float *sgc = gc, *sgv = gv;
  if (space == RSB_HOST) {
    if (!w->d_tmp_gc) {
      HIP_TRY(hipMalloc(&w->d_tmp_gc, N * nq * sizeof(float)));
      HIP_TRY(hipMalloc(&w->d_tmp_gv, N * nv * sizeof(float)));
      HIP_TRY(hipMalloc(&w->d_tmp_mask, N));
    }
    HIP_TRY(hipMemcpyAsync(w->d_tmp_mask, mask, N, hipMemcpyHostToDevice, stream_of(w)));
    if (gc) HIP_TRY(hipMemcpyAsync(w->d_tmp_gc, gc, N * nq * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
    if (gv) HIP_TRY(hipMemcpyAsync(w->d_tmp_gv, gv, N * nv * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
    dmask = w->d_tmp_mask; sgc = gc ? w->d_tmp_gc : nullptr; sgv = gv ? w->d_tmp_gv : nullptr;
  }
  if (sgc) hipLaunchKernelGGL(masked_row_copy, dim3((N * nq + 255) / 256), dim3(256), 0, stream_of(w), w->d_gc, sgc, dmask, (int)N, (int)nq);
  if (sgv) hipLaunchKernelGGL(masked_row_copy, dim3((N * nv + 255) / 256), dim3(256), 0, stream_of(w), w->d_gv, sgv, dmask, (int)N, (int)nv);
  if (n6 > 0) hipLaunchKernelGGL(warm_clear_kernel, dim3((N * n6 + 255) / 256), dim3(256), 0, stream_of(w), w->d_warm, dmask, (int)N, n6);
  HIP_TRY(hipGetLastError());
  if (space == RSB_HOST) HIP_TRY(hipStreamSynchronize(stream_of(w)));
  return RSB_OK;
}

int rsb_get_state(rsb_world* w, float* gc, float* gv, int space) {
  if (!w) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  const size_t N = w->N;
  if (gc) { int st = copy_out(w, gc, w->d_gc, N * w->blob.nq * sizeof(float), space); if (st) return st; }
  if (gv) { int st = copy_out(w, gv, w->d_gv, N * w->blob.nv * sizeof(float), space); if (st) return st; }
  return RSB_OK;
}

static int env_row(rsb_world* w, int field, int env, float** base, size_t* dim) {
  if (!w || env < 0 || env >= w->N) { rsb::set_error("env row: env index out of range"); return RSB_E_INVALID; }
  switch (field) {
    case RSB_F_GC: *base = w->d_gc; *dim = w->blob.nq; break;
    case RSB_F_GV: *base = w->d_gv; *dim = w->blob.nv; break;
    case RSB_F_PTARGET: *base = w->d_pt; *dim = w->blob.nq; break;
    case RSB_F_DTARGET: *base = w->d_dt; *dim = w->blob.nv; break;
    case RSB_F_TAU_FF: *base = w->d_tff; *dim = w->blob.nv; break;
    case RSB_F_GENERALIZED_FORCE:
      if (!w->want_genf || !w->d_genf) { rsb::set_error("RSB_F_GENERALIZED_FORCE: call rsb_enable_generalized_force_output first"); return RSB_E_INVALID; }
      *base = w->d_genf; *dim = w->blob.nv; break;
    default: rsb::set_error("env row: unsupported field"); return RSB_E_INVALID;
  }
  return RSB_OK;
}
int rsb_set_env_row(rsb_world* w, int field, int env, const float* data) {
  float* base; size_t dim;
  if (field == RSB_F_GENERALIZED_FORCE) { rsb::set_error("RSB_F_GENERALIZED_FORCE is an output"); return RSB_E_INVALID; }
  int st = env_row(w, field, env, &base, &dim);
  if (st != RSB_OK || !data) return st != RSB_OK ? st : RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipMemcpyAsync(base + (size_t)env * dim, data, dim * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
  if ((field == RSB_F_GC || field == RSB_F_GV) && w->blob.ncol > 0)   // the env's state was overwritten: its solver state is stale
    HIP_TRY(hipMemsetAsync(w->d_warm + (size_t)env * rsbk::kWarmRow, 0, (size_t)rsbk::kWarmRow * sizeof(float), stream_of(w)));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  if (field == RSB_F_DTARGET && !all_zero(data, dim, RSB_HOST)) w->dt_zero = false;
  if (field == RSB_F_TAU_FF && !all_zero(data, dim, RSB_HOST)) w->tff_zero = false;
  w->integrate1_valid = false; w->env_ob_valid = false;
  return RSB_OK;
}
int rsb_get_env_row(rsb_world* w, int field, int env, float* data) {
  float* base; size_t dim;
  int st = env_row(w, field, env, &base, &dim);
  if (st != RSB_OK || !data) return st != RSB_OK ? st : RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipMemcpyAsync(data, base + (size_t)env * dim, dim * sizeof(float), hipMemcpyDeviceToHost, stream_of(w)));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  return RSB_OK;
}

int rsb_get_field(rsb_world* w, int field, float* out, int space) {
  float* base; size_t dim;
  int st = env_row(w, field, 0, &base, &dim);
  if (st != RSB_OK || !out) return st != RSB_OK ? st : RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  return copy_out(w, out, base, (size_t)w->N * dim * sizeof(float), space);
}

int rsb_enable_generalized_force_output(rsb_world* w, int on) {
  if (!w) { rsb::set_error("rsb_enable_generalized_force_output: null world"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  if (on && !w->d_genf) {
    const size_t bytes = (size_t)w->N * w->blob.nv * sizeof(float);
    HIP_TRY(hipMalloc(&w->d_genf, bytes));
    HIP_TRY(hipMemset(w->d_genf, 0, bytes));
  }
  w->want_genf = on != 0;
  return RSB_OK;
}
int rsb_set_control_mode(rsb_world* w, int mode) {
  if (!w || (mode != RSB_FORCE_AND_TORQUE && mode != RSB_PD_PLUS_FEEDFORWARD_TORQUE)) { rsb::set_error("rsb_set_control_mode: unknown mode"); return RSB_E_INVALID; }
  w->control_mode = mode;
  w->image_dirty = true;
  return RSB_OK;
}
int rsb_set_pd_gains(rsb_world* w, const float* kp, const float* kd) {
  if (!w || !kp || !kd) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipMemcpyAsync(w->d_kp, kp, w->blob.nv * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
  HIP_TRY(hipMemcpyAsync(w->d_kd, kd, w->blob.nv * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  w->h_kp.assign(kp, kp + w->blob.nv); w->h_kd.assign(kd, kd + w->blob.nv);
  w->image_dirty = true;
  return RSB_OK;
}
int rsb_set_pd_target(rsb_world* w, const float* p_target, const float* d_target, int space) {
  if (!w) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  const size_t N = w->N;
  if (p_target) { int st = copy_in(w, w->d_pt, p_target, N * w->blob.nq, space); if (st) return st; }
  if (d_target) { int st = copy_in(w, w->d_dt, d_target, N * w->blob.nv, space); if (st) return st; w->dt_zero = all_zero(d_target, N * w->blob.nv, space); }
  return RSB_OK;
}
int rsb_set_generalized_force(rsb_world* w, const float* tau, int space) {
  if (!w || !tau) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  w->tff_zero = all_zero(tau, (size_t)w->N * w->blob.nv, space);
  return copy_in(w, w->d_tff, tau, (size_t)w->N * w->blob.nv, space);
}

int rsb_integrate(rsb_world* w, int n_substeps) {
  if (!w || n_substeps < 1) { rsb::set_error("rsb_integrate: n_substeps must be >= 1"); return RSB_E_INVALID; }
  return do_integrate(w, n_substeps);
}

int rsb_integrate_masked(rsb_world* w, int n_substeps, const uint8_t* mask, int space) {
  if (!w || n_substeps < 1 || !mask) { rsb::set_error("rsb_integrate_masked: bad argument"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  const uint8_t* dmask = mask;
  if (space == RSB_HOST) {
    if (!w->d_launch_mask) HIP_TRY(hipMalloc(&w->d_launch_mask, (size_t)w->N));
    HIP_TRY(hipMemcpyAsync(w->d_launch_mask, mask, (size_t)w->N, hipMemcpyHostToDevice, stream_of(w)));
    HIP_TRY(hipStreamSynchronize(stream_of(w)));   // pageable host memory: the caller may reuse its buffer
    dmask = w->d_launch_mask;
  }
  w->launch_mask = dmask;
  return do_integrate(w, n_substeps);
}

int rsb_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) { rsb::set_error("rsb_host_alloc: bad argument"); return RSB_E_INVALID; }
  HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return RSB_OK;
}
int rsb_host_free(void* p) {
  if (p) HIP_TRY(hipHostFree(p));
  return RSB_OK;
}
int rsb_device_alloc(rsb_world* w, size_t bytes, void** out) {
  if (!w || !out || bytes == 0) { rsb::set_error("rsb_device_alloc: bad argument"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipMalloc(out, bytes));
  return RSB_OK;
}
int rsb_device_free(rsb_world* w, void* p) {
  if (!w) { rsb::set_error("rsb_device_free: bad argument"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  if (p) { HIP_TRY(hipStreamSynchronize(stream_of(w))); HIP_TRY(hipFree(p)); }
  return RSB_OK;
}
int rsb_device_copy(rsb_world* w, void* dst, const void* src, size_t bytes, int kind) {
  if (!w || !dst || !src || (kind != 0 && kind != 1)) { rsb::set_error("rsb_device_copy: bad argument"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  hipStream_t st = stream_of(w);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return fault_status(w);
}

// One flush of the facade's per-env views: uploads, launches and downloads queued on the world's stream, ONE synchronisation at the end.
int rsb_view_exchange(rsb_world* w, const rsb_view_io* io) {
  if (!w || !io || io->n_launches < 0 || (io->n_launches > 0 && !io->launch_substeps)) { rsb::set_error("rsb_view_exchange: bad argument"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  const size_t N = w->N, nq = w->blob.nq, nv = w->blob.nv;
  // host-side cost of the exchange by part (rsb_debug_view_profile): [0] enqueue of the uploads, [1] of the launches, [2] of the downloads, [3] the wait
  const auto tp0 = std::chrono::steady_clock::now();
  auto lap_ns = [](std::chrono::steady_clock::time_point& t) { const auto n = std::chrono::steady_clock::now(); const long long d = std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count(); t = n; return d; };
  auto tp = tp0;
  if (io->p_target) HIP_TRY(hipMemcpyAsync(w->d_pt, io->p_target, N * nq * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
  if (io->d_target) { HIP_TRY(hipMemcpyAsync(w->d_dt, io->d_target, N * nv * sizeof(float), hipMemcpyHostToDevice, stream_of(w))); w->dt_zero = false; }      // (staged rows of the views: not scanned)
  if (io->tau_ff) { HIP_TRY(hipMemcpyAsync(w->d_tff, io->tau_ff, N * nv * sizeof(float), hipMemcpyHostToDevice, stream_of(w))); w->tff_zero = false; }
  if (io->gc || io->gv) {
    if (!io->state_mask) { rsb::set_error("rsb_view_exchange: state rows need state_mask"); return RSB_E_INVALID; }
    if (!w->d_tmp_gc) {
      HIP_TRY(hipMalloc(&w->d_tmp_gc, N * nq * sizeof(float)));
      HIP_TRY(hipMalloc(&w->d_tmp_gv, N * nv * sizeof(float)));
      HIP_TRY(hipMalloc(&w->d_tmp_mask, N));
    }
    HIP_TRY(hipMemcpyAsync(w->d_tmp_mask, io->state_mask, N, hipMemcpyHostToDevice, stream_of(w)));
    if (io->gc) {
      HIP_TRY(hipMemcpyAsync(w->d_tmp_gc, io->gc, N * nq * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
      hipLaunchKernelGGL(masked_row_copy, dim3((N * nq + 255) / 256), dim3(256), 0, stream_of(w), w->d_gc, (const float*)w->d_tmp_gc, (const uint8_t*)w->d_tmp_mask, (int)N, (int)nq);
    }
    if (io->gv) {
      HIP_TRY(hipMemcpyAsync(w->d_tmp_gv, io->gv, N * nv * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
      hipLaunchKernelGGL(masked_row_copy, dim3((N * nv + 255) / 256), dim3(256), 0, stream_of(w), w->d_gv, (const float*)w->d_tmp_gv, (const uint8_t*)w->d_tmp_mask, (int)N, (int)nv);
    }
    hipLaunchKernelGGL(warm_clear_kernel, dim3((N * rsbk::kWarmRow + 255) / 256), dim3(256), 0, stream_of(w), w->d_warm, (const uint8_t*)w->d_tmp_mask, (int)N, rsbk::kWarmRow);
    HIP_TRY(hipGetLastError());
    w->integrate1_valid = false; w->env_ob_valid = false;
  }
  if (io->n_launches > 0 && io->launch_masks) {
    const size_t need = (size_t)io->n_launches * N;
    if (w->view_masks_cap < need) {
      if (w->d_view_masks) HIP_TRY(hipFree(w->d_view_masks));
      w->d_view_masks = nullptr; w->view_masks_cap = 0;
      HIP_TRY(hipMalloc(&w->d_view_masks, need));
      w->view_masks_cap = need;
    }
    HIP_TRY(hipMemcpyAsync(w->d_view_masks, io->launch_masks, need, hipMemcpyHostToDevice, stream_of(w)));
  }
  w->view_prof[0] += lap_ns(tp);
  for (int i = 0; i < io->n_launches; ++i) {
    if (io->launch_substeps[i] < 1) { rsb::set_error("rsb_view_exchange: launch_substeps must be >= 1"); return RSB_E_INVALID; }
    if (io->launch_masks) w->launch_mask = w->d_view_masks + (size_t)i * N;
    const int st = do_integrate(w, io->launch_substeps[i]);
    if (st != RSB_OK) return st;
  }
  w->view_prof[1] += lap_ns(tp);
  if (io->gc_out) HIP_TRY(hipMemcpyAsync(io->gc_out, w->d_gc, N * nq * sizeof(float), hipMemcpyDeviceToHost, stream_of(w)));
  if (io->gv_out) HIP_TRY(hipMemcpyAsync(io->gv_out, w->d_gv, N * nv * sizeof(float), hipMemcpyDeviceToHost, stream_of(w)));
  if (io->contact_counts) HIP_TRY(hipMemcpyAsync(io->contact_counts, w->d_count, N * sizeof(int32_t), hipMemcpyDeviceToHost, stream_of(w)));
  if (io->contacts) HIP_TRY(hipMemcpyAsync(io->contacts, w->d_contacts, N * w->kmax * sizeof(rsb_contact), hipMemcpyDeviceToHost, stream_of(w)));
  if (io->generalized_force) {
    if (!w->want_genf || !w->d_genf) { rsb::set_error("rsb_view_exchange: generalized_force needs rsb_enable_generalized_force_output"); return RSB_E_INVALID; }
    HIP_TRY(hipMemcpyAsync(io->generalized_force, w->d_genf, N * nv * sizeof(float), hipMemcpyDeviceToHost, stream_of(w)));
  }
  w->view_prof[2] += lap_ns(tp);
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  w->view_prof[3] += lap_ns(tp);
  ++w->view_prof[4];
  return RSB_OK;
}
int rsb_debug_view_profile(rsb_world* w, long long out[5], int reset) {
  if (!w) return RSB_E_INVALID;
  if (out) for (int i = 0; i < 5; ++i) out[i] = w->view_prof[i];
  if (reset) for (int i = 0; i < 5; ++i) w->view_prof[i] = 0;
  return RSB_OK;
}

int rsb_set_done_output(rsb_world* w, uint8_t* done_device) {
  if (!w) return RSB_E_INVALID;
  w->d_done_out = done_device;
  return RSB_OK;
}

// integrate1(): collision detection + M, h for the CURRENT state, into query buffers.  The state is
// not advanced; integrate2() then runs the fused step (which recomputes the same quantities from the
// unchanged state, so the pair is equivalent to integrate()).
int rsb_integrate1(rsb_world* w) {
  if (!w) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  const size_t N = w->N, nv = w->blob.nv;
  if (!w->d_M) {
    HIP_TRY(hipMalloc(&w->d_M, N * nv * nv * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_h, N * nv * sizeof(float)));
  }
  rsbq::QueryArgs qa;
  qa.model = w->d_model; qa.gc = w->d_gc; qa.gv = w->d_gv; qa.M = w->d_M; qa.h = w->d_h; qa.N = w->N;
  qa.gx = (float)w->gravity[0]; qa.gy = (float)w->gravity[1]; qa.gz = (float)w->gravity[2];
  int st = rsbq::launch_query(qa, w->blob.nb, stream_of(w));
  if (st != 0) { rsb::set_error("integrate1: query kernel launch failed"); return RSB_E_HIP; }
  w->integrate1_valid = true;
  return RSB_OK;
}
int rsb_integrate2(rsb_world* w) {
  if (!w) return RSB_E_INVALID;
  return do_integrate(w, 1);
}

int rsb_get_contacts(rsb_world* w, int32_t* counts, rsb_contact* contacts, int space) {
  if (!w) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  const size_t N = w->N;
  if (counts) { int st = copy_out(w, counts, w->d_count, N * sizeof(int32_t), space); if (st) return st; }
  if (contacts) { int st = copy_out(w, contacts, w->d_contacts, N * w->kmax * sizeof(rsb_contact), space); if (st) return st; }
  return RSB_OK;
}
int rsb_get_mass_matrix(rsb_world* w, float* M, int space) {
  if (!w || !M) return RSB_E_INVALID;
  if (!w->integrate1_valid) { rsb::set_error("rsb_get_mass_matrix: call rsb_integrate1 first (state changed since)"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  return copy_out(w, M, w->d_M, (size_t)w->N * w->blob.nv * w->blob.nv * sizeof(float), space);
}
int rsb_get_inverse_mass_matrix(rsb_world* w, float* Minv, int space) {
  if (!w || !Minv) return RSB_E_INVALID;
  if (!w->integrate1_valid) { rsb::set_error("rsb_get_inverse_mass_matrix: call rsb_integrate1 first (state changed since)"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  const size_t n = (size_t)w->N * w->blob.nv * w->blob.nv;
  if (!w->d_Minv) {
    HIP_TRY(hipMalloc(&w->d_Minv, n * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_Mwork, n * sizeof(float)));
  }
  hipLaunchKernelGGL(rsbq::rsb_minv_kernel, dim3((w->N + 63) / 64), dim3(64), 0, stream_of(w), w->d_M, w->d_Mwork, w->d_Minv, w->N, w->blob.nv);
  HIP_TRY(hipGetLastError());
  return copy_out(w, Minv, w->d_Minv, n * sizeof(float), space);
}
int rsb_get_nonlinearities(rsb_world* w, float* h, int space) {
  if (!w || !h) return RSB_E_INVALID;
  if (!w->integrate1_valid) { rsb::set_error("rsb_get_nonlinearities: call rsb_integrate1 first (state changed since)"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  return copy_out(w, h, w->d_h, (size_t)w->N * w->blob.nv * sizeof(float), space);
}
int rsb_get_flags(rsb_world* w, int32_t* flags, int space) {
  if (!w || !flags) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  return copy_out(w, flags, w->d_flags, (size_t)w->N * sizeof(int32_t), space);
}
int rsb_get_solver_iterations(rsb_world* w, int32_t* iters, int space) {
  if (!w || !iters) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  return copy_out(w, iters, w->d_iters, (size_t)w->N * sizeof(int32_t), space);
}

}  // extern "C"
namespace {
// The index list is tiny and normally the same every control step: a pageable H2D copy per call would put a
// host-side staging round trip on the stream each step, so it is re-uploaded only when it changes.
int upload_obs_idx(rsb_world* w, const int32_t* idx, int n) {
  for (int i = 0; i < n; ++i)
    if (idx[i] < 0 || idx[i] >= w->blob.ncol) { rsb::set_error("collision index out of range"); return RSB_E_INVALID; }
  if ((int)w->obs_idx_host.size() != n || std::memcmp(w->obs_idx_host.data(), idx, n * sizeof(int32_t)) != 0) {
    w->obs_idx_host.assign(idx, idx + n);
    HIP_TRY(hipMemcpyAsync(w->d_obs_idx, w->obs_idx_host.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, stream_of(w)));
    HIP_TRY(hipStreamSynchronize(stream_of(w)));
  }
  return RSB_OK;
}
}  // namespace
extern "C" {

int rsb_obs_dim(const rsb_world* w, int n_force_slots) {
  if (!w || n_force_slots < 0 || n_force_slots > RSB_MAX_COLLISIONS) return RSB_E_INVALID;
  return w->blob.nq + w->blob.nv + 3 * n_force_slots;
}
int rsb_gather_obs(rsb_world* w, float* out, const int32_t* collision_indices, int n_force_slots, int space) {
  if (!w || !out || n_force_slots < 0 || n_force_slots > RSB_MAX_COLLISIONS) { rsb::set_error("rsb_gather_obs: bad argument"); return RSB_E_INVALID; }
  if (space != RSB_DEVICE && space != RSB_HOST) { rsb::set_error("rsb_gather_obs: space must be RSB_DEVICE or RSB_HOST"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  if (space == RSB_HOST) {   // slow path (C hosts without a device allocator, tests): gathered on the device, then copied out
    const size_t local = (size_t)w->N * (w->blob.nq + w->blob.nv + 3 * n_force_slots);
    if (w->obs_local_cap < local) {
      if (w->d_obs_local) HIP_TRY(hipFree(w->d_obs_local));
      w->d_obs_local = nullptr; w->obs_local_cap = 0;
      HIP_TRY(hipMalloc(&w->d_obs_local, local * sizeof(float)));
      w->obs_local_cap = local;
    }
    const int st = rsb_gather_obs(w, w->d_obs_local, collision_indices, n_force_slots, RSB_DEVICE);
    if (st != RSB_OK) return st;
    return copy_out(w, out, w->d_obs_local, local * sizeof(float), RSB_HOST);
  }
  const int32_t* didx = nullptr;
  if (collision_indices && n_force_slots > 0) {
    int st = upload_obs_idx(w, collision_indices, n_force_slots);
    if (st != RSB_OK) return st;
    didx = w->d_obs_idx;
  }
  const int od = w->blob.nq + w->blob.nv + 3 * n_force_slots;
  const size_t total = (size_t)w->N * od;
  hipLaunchKernelGGL(gather_obs_kernel, dim3((total + 255) / 256), dim3(256), 0, stream_of(w), out, w->d_gc, w->d_gv,
                     w->d_contacts, w->d_count, didx, w->N, w->blob.nq, w->blob.nv, w->kmax, n_force_slots,
                     (float)(1.0 / w->dt));
  HIP_TRY(hipGetLastError());
  return RSB_OK;
}

int rsb_reset_terminated(rsb_world* w, const int32_t* allowed_collisions, int n_allowed, const float* gc0,
                         const float* gv0, int rows, uint8_t* done, int space) {
  if (!w || !gc0 || !gv0 || (rows != 1 && rows != w->N) || n_allowed < 0 || (n_allowed > 0 && !allowed_collisions)) {
    rsb::set_error("rsb_reset_terminated: bad argument");
    return RSB_E_INVALID;
  }
  HIP_TRY(hipSetDevice(w->device));
  unsigned long long allowed = 0;
  for (int i = 0; i < n_allowed; ++i) {
    if (allowed_collisions[i] < 0 || allowed_collisions[i] >= w->blob.ncol) { rsb::set_error("rsb_reset_terminated: collision index out of range"); return RSB_E_INVALID; }
    allowed |= 1ull << allowed_collisions[i];
  }
  const size_t N = w->N, nq = w->blob.nq, nv = w->blob.nv;
  const float *dgc0 = gc0, *dgv0 = gv0;
  uint8_t* ddone = done;
  if (space == RSB_HOST) {
    if (!w->d_tmp_gc) {
      HIP_TRY(hipMalloc(&w->d_tmp_gc, N * nq * sizeof(float)));
      HIP_TRY(hipMalloc(&w->d_tmp_gv, N * nv * sizeof(float)));
      HIP_TRY(hipMalloc(&w->d_tmp_mask, N));
    }
    HIP_TRY(hipMemcpyAsync(w->d_tmp_gc, gc0, (size_t)rows * nq * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
    HIP_TRY(hipMemcpyAsync(w->d_tmp_gv, gv0, (size_t)rows * nv * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
    dgc0 = w->d_tmp_gc; dgv0 = w->d_tmp_gv; ddone = done ? w->d_tmp_mask : nullptr;
  }
  hipLaunchKernelGGL(reset_terminated_kernel, dim3((N + 255) / 256), dim3(256), 0, stream_of(w), w->d_gc, w->d_gv, w->d_contacts,
                     w->d_count, w->d_flags, allowed, dgc0, dgv0, rows, ddone, (int)N, (int)nq, (int)nv, w->kmax, w->d_warm, rsbk::kWarmRow);
  HIP_TRY(hipGetLastError());
  w->integrate1_valid = false; w->env_ob_valid = false;
  if (space == RSB_HOST) {
    if (done) HIP_TRY(hipMemcpyAsync(done, w->d_tmp_mask, N, hipMemcpyDeviceToHost, stream_of(w)));
    HIP_TRY(hipStreamSynchronize(stream_of(w)));
  }
  return RSB_OK;
}

// One control step of a vectorised env, enqueued with a single call: PD targets in, n_substeps x integrate(),
// observation block out, terminated envs reset.  All pointers are device pointers; nothing synchronises.
int rsb_control_step(rsb_world* w, const float* p_target, const float* d_target, int n_substeps, float* obs_out,
                     const int32_t* force_collisions, int n_force_slots, const int32_t* allowed_collisions,
                     int n_allowed, const float* gc0, const float* gv0, int rows) {
  if (!w || n_substeps < 1 || n_force_slots < 0 || n_force_slots > RSB_MAX_COLLISIONS || n_allowed < 0 ||
      (n_allowed > 0 && !allowed_collisions) || ((gc0 != nullptr) != (gv0 != nullptr)) || (gc0 && rows != 1 && rows != w->N)) {
    rsb::set_error("rsb_control_step: bad argument");
    return RSB_E_INVALID;
  }
  HIP_TRY(hipSetDevice(w->device));
  if (d_target) { int st = copy_in(w, w->d_dt, d_target, (size_t)w->N * w->blob.nv, RSB_DEVICE); if (st) return st; w->dt_zero = false; }
  rsb_world::Fuse f;
  f.ptarget_src = p_target;   // read in place by the launch, which also refreshes the world's own copy
  f.pipeline = p_target != nullptr && d_target == nullptr;   // (rsb_set_step_pipelining: control steps that upload nothing may overlap)
  if (w->peer.connected) {
    if (w->blob.fixed_base || w->blob.depth - 1 > 12) { rsb::set_error("rsb_control_step: the peer-mapped obs exchange is compiled for floating-base models of tree depth <= 13"); return RSB_E_UNSUPPORTED; }
    if (obs_out && n_force_slots != w->peer.slots) { rsb::set_error("rsb_control_step: obs_out must use the force slots the peer exchange was created with"); return RSB_E_INVALID; }
    if (obs_out) {   // ... and the same primitives in them: the launch writes ONE obs row layout to the caller's block and to the peers'
      for (int i = 0; i < n_force_slots; ++i) {
        const int32_t want = w->peer.idx.empty() ? i : w->peer.idx[i], got = force_collisions ? force_collisions[i] : i;
        if (want != got) { rsb::set_error("rsb_control_step: force_collisions differ from the list the peer exchange was created with"); return RSB_E_INVALID; }
      }
    }
    f.peer = true;
  }
  if (obs_out) {
    f.obs_out = obs_out; f.obs_slots = n_force_slots;
    if (force_collisions && n_force_slots > 0) {
      int st = upload_obs_idx(w, force_collisions, n_force_slots);
      if (st != RSB_OK) return st;
      f.obs_idx = w->d_obs_idx;
    }
  }
  if (gc0) {
    unsigned long long allowed = 0;
    for (int i = 0; i < n_allowed; ++i) {
      if (allowed_collisions[i] < 0 || allowed_collisions[i] >= w->blob.ncol) { rsb::set_error("rsb_control_step: collision index out of range"); return RSB_E_INVALID; }
      allowed |= 1ull << allowed_collisions[i];
    }
    f.do_reset = 1; f.have_allowed = 1; f.allowed = allowed; f.gc0 = gc0; f.gv0 = gv0; f.rows = rows;
  }
  w->fuse = f;
  return rsb_integrate(w, n_substeps);
}

// K control steps of the open loop (rsb.h): ONE resident launch when residency is on and the world's class has a resident twin, else K control steps
int rsb_control_steps(rsb_world* w, int n_steps, const float* p_targets, int period, long long first, int n_substeps, float* obs_out,
                      long long obs_step_stride, const int32_t* force_collisions, int n_force_slots, const int32_t* allowed_collisions, int n_allowed,
                      const float* gc0, const float* gv0, int rows, uint8_t* done_out, long long done_step_stride) {
  if (!w || n_steps < 1 || !p_targets || period < 1 || first < 0 || n_substeps < 1 || obs_step_stride < 0 || done_step_stride < 0 || n_force_slots < 0 ||
      n_force_slots > RSB_MAX_COLLISIONS || n_allowed < 0 || (n_allowed > 0 && !allowed_collisions) || ((gc0 != nullptr) != (gv0 != nullptr)) ||
      (gc0 && rows != 1 && rows != w->N)) {
    rsb::set_error("rsb_control_steps: bad argument");
    return RSB_E_INVALID;
  }
  HIP_TRY(hipSetDevice(w->device));
  const size_t slice = (size_t)w->N * w->blob.nq;
  if (!(w->res_on && resident_class(w, 0, 0) >= 0)) {
    uint8_t* const saved = w->d_done_out;
    int st = RSB_OK;
    for (int j = 0; j < n_steps && st == RSB_OK; ++j) {
      if (done_out) w->d_done_out = done_out + (size_t)j * (size_t)done_step_stride;
      st = rsb_control_step(w, p_targets + (size_t)((first + j) % period) * slice, nullptr, n_substeps, obs_out ? obs_out + (size_t)j * (size_t)obs_step_stride : nullptr,
                            force_collisions, n_force_slots, allowed_collisions, n_allowed, gc0, gv0, rows);
    }
    w->d_done_out = saved;
    return st;
  }
  rsb_world::Fuse f;
  if (obs_out) {
    f.obs_out = obs_out; f.obs_slots = n_force_slots;
    if (force_collisions && n_force_slots > 0) {
      int st = upload_obs_idx(w, force_collisions, n_force_slots);
      if (st != RSB_OK) return st;
      f.obs_idx = w->d_obs_idx;
    }
  }
  if (gc0) {
    unsigned long long allowed = 0;
    for (int i = 0; i < n_allowed; ++i) {
      if (allowed_collisions[i] < 0 || allowed_collisions[i] >= w->blob.ncol) { rsb::set_error("rsb_control_steps: collision index out of range"); return RSB_E_INVALID; }
      allowed |= 1ull << allowed_collisions[i];
    }
    f.do_reset = 1; f.have_allowed = 1; f.allowed = allowed; f.gc0 = gc0; f.gv0 = gv0; f.rows = rows;
  }
  f.res_steps = n_steps; f.res_stage = 0; f.res_targets = p_targets; f.res_period = period; f.res_first = first;
  f.res_obs_stride = obs_out ? obs_step_stride : 0; f.res_done_stride = done_out ? done_step_stride : 0; f.res_done = done_out;   // (a stride without its buffer must not walk the world's own done flags)
  w->fuse = f;
  return do_integrate(w, n_substeps);
}
int rsb_set_step_residency(rsb_world* w, int on) {
  if (!w) { rsb::set_error("rsb_set_step_residency: null world"); return RSB_E_INVALID; }
  w->res_on = on != 0;
  return RSB_OK;
}
int rsb_step_residency_enabled(const rsb_world* w) { return w && w->res_on ? 1 : 0; }
int rsb_step_residency_status(rsb_world* w, int stage) {
  if (!w) { rsb::set_error("rsb_step_residency_status: null world"); return 0; }
  return resident_class(w, stage, 128) >= 0 ? 1 : 0;
}
long long rsb_step_residency_launches(const rsb_world* w) { return w ? w->res_launches : 0; }
int rsb_debug_resident_full_writes(rsb_world* w, int on) {
  if (!w) return RSB_E_INVALID;
  w->res_full = on != 0;
  return RSB_OK;
}

// ---- device-resident vectorised env ------------------------------------------------------------------------------
int rsb_env_configure(rsb_world* w, const rsb_env_config* cfg, const float* action_mean, const float* gc_init,
                      const float* gv_init) {
  if (!w || !cfg || !action_mean || !gc_init || !gv_init || cfg->n_substeps < 1 || cfg->n_foot < 0 || cfg->n_foot > RSB_MAX_COLLISIONS) {
    rsb::set_error("rsb_env_configure: bad argument");
    return RSB_E_INVALID;
  }
  unsigned long long allowed = 0;
  for (int i = 0; i < cfg->n_foot; ++i) {
    if (cfg->foot_collisions[i] < 0 || cfg->foot_collisions[i] >= w->blob.ncol) { rsb::set_error("rsb_env_configure: foot collision index out of range"); return RSB_E_INVALID; }
    allowed |= 1ull << cfg->foot_collisions[i];
  }
  HIP_TRY(hipSetDevice(w->device));
  const size_t N = w->N, nq = w->blob.nq, nv = w->blob.nv, nj = nv - 6, od = 10 + 2 * nj;
  if (!w->d_env_mean) {
    HIP_TRY(hipMalloc(&w->d_env_mean, nj * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_env_gc0, nq * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_env_gv0, nv * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_env_io, N * od * sizeof(float)));      // staging for host-side action / observation buffers
    HIP_TRY(hipMalloc(&w->d_env_ob, N * od * sizeof(float)));      // the fused step's observation (never the action's buffer: other waves may still have to read their actions)
    HIP_TRY(hipMalloc(&w->d_env_reward, N * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_env_tau2, N * sizeof(float)));
    HIP_TRY(hipMemset(w->d_env_tau2, 0, N * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_env_done, N));
    HIP_TRY(hipMalloc(&w->d_env_act, N * nj * sizeof(float)));     // the closed loop's action rows (rsb_closed_loop_run)
    HIP_TRY(hipMemset(w->d_env_act, 0, N * nj * sizeof(float)));
  }
  HIP_TRY(hipMemcpyAsync(w->d_env_mean, action_mean, nj * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
  HIP_TRY(hipMemcpyAsync(w->d_env_gc0, gc_init, nq * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
  HIP_TRY(hipMemcpyAsync(w->d_env_gv0, gv_init, nv * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  w->env_cfg = *cfg; w->env_allowed = allowed; w->env_ready = true;
  return RSB_OK;
}
int rsb_env_set_reset_states(rsb_world* w, const float* gc0, const float* gv0, int space) {
  if (!w || ((gc0 != nullptr) != (gv0 != nullptr)) || (space != RSB_HOST && space != RSB_DEVICE)) { rsb::set_error("rsb_env_set_reset_states: bad argument"); return RSB_E_INVALID; }
  if (!w->env_ready) { rsb::set_error("rsb_env_set_reset_states: call rsb_env_configure first"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  hipStream_t s = stream_of(w);
  const size_t N = w->N, nq = w->blob.nq, nv = w->blob.nv;
  if (!gc0) {
    HIP_TRY(hipStreamSynchronize(s));
    if (w->d_env_gc0_rows) { HIP_TRY(hipFree(w->d_env_gc0_rows)); HIP_TRY(hipFree(w->d_env_gv0_rows)); }
    w->d_env_gc0_rows = w->d_env_gv0_rows = nullptr;
    return RSB_OK;
  }
  if (!w->d_env_gc0_rows) {
    HIP_TRY(hipMalloc(&w->d_env_gc0_rows, N * nq * sizeof(float)));
    HIP_TRY(hipMalloc(&w->d_env_gv0_rows, N * nv * sizeof(float)));
  }
  const hipMemcpyKind kind = space == RSB_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  HIP_TRY(hipMemcpyAsync(w->d_env_gc0_rows, gc0, N * nq * sizeof(float), kind, s));
  HIP_TRY(hipMemcpyAsync(w->d_env_gv0_rows, gv0, N * nv * sizeof(float), kind, s));
  HIP_TRY(hipStreamSynchronize(s));
  return RSB_OK;
}
int rsb_env_dims(const rsb_world* w, int* ob_dim, int* action_dim) {
  if (!w) return RSB_E_INVALID;
  if (ob_dim) *ob_dim = 10 + 2 * (w->blob.nv - 6);
  if (action_dim) *action_dim = w->blob.nv - 6;
  return RSB_OK;
}
static int env_check(rsb_world* w, const char* who) {
  if (!w) return RSB_E_INVALID;
  if (!w->env_ready) { rsb::set_error(std::string(who) + ": call rsb_env_configure first"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  return RSB_OK;
}
int rsb_env_reset(rsb_world* w) {
  int st = env_check(w, "rsb_env_reset"); if (st != RSB_OK) return st;
  hipLaunchKernelGGL(env_reset_kernel, dim3((w->N + 255) / 256), dim3(256), 0, stream_of(w), w->d_gc, w->d_gv, w->d_count,
                     w->d_flags, w->d_env_gc0_rows ? w->d_env_gc0_rows : w->d_env_gc0, w->d_env_gc0_rows ? w->d_env_gv0_rows : w->d_env_gv0,
                     w->d_env_gc0_rows ? w->N : 1, w->N, w->blob.nq, w->blob.nv, w->d_warm, rsbk::kWarmRow);
  HIP_TRY(hipGetLastError());
  w->integrate1_valid = false; w->env_ob_valid = false;
  w->cl_passes = 0;      // (a closed-loop run's global step index - the noise slice of the reference stage - restarts with the episodes)
  return RSB_OK;
}
int rsb_env_observe(rsb_world* w, float* ob, int space) {
  int st = env_check(w, "rsb_env_observe"); if (st != RSB_OK) return st;
  if (!ob) return RSB_E_INVALID;
  const size_t od = 10 + 2 * (size_t)(w->blob.nv - 6);
  float* dob = space == RSB_DEVICE ? ob : w->d_env_io;
  st = launch_env_obs(w, dob, stream_of(w));
  if (st != RSB_OK) return st;
  if (space == RSB_HOST) return copy_out(w, ob, dob, (size_t)w->N * od * sizeof(float), RSB_HOST);
  return RSB_OK;
}
int rsb_env_step(rsb_world* w, const float* action, float* reward, uint8_t* done, float* ob_next, int space) {
  int st = env_check(w, "rsb_env_step"); if (st != RSB_OK) return st;
  if (!action) return RSB_E_INVALID;
  const int N = w->N, nv = w->blob.nv, nj = nv - 6;
  const size_t od = 10 + 2 * (size_t)nj;
  const float* dact = action;
  if (space == RSB_HOST) {
    HIP_TRY(hipMemcpyAsync(w->d_env_io, action, (size_t)N * nj * sizeof(float), hipMemcpyHostToDevice, stream_of(w)));
    dact = w->d_env_io;
  }
  float* drew = space == RSB_DEVICE ? reward : (reward ? w->d_env_reward : nullptr);
  uint8_t* ddone = space == RSB_DEVICE ? done : (done ? w->d_env_done : nullptr);
  float* dob = ob_next ? (space == RSB_DEVICE ? ob_next : w->d_env_ob) : nullptr;
  {
    // ONE launch per vectorised step: action -> PD targets in the prologue, control_dt / simulation_dt sub-steps, then reward,
    // termination (non-foot contact or non-finite state), reset and the next observation in the epilogue
    rsb_world::Fuse f;
    f.act = dact;
    f.have_allowed = 1; f.allowed = w->env_allowed;
    f.do_reset = 1; f.gc0 = w->d_env_gc0; f.gv0 = w->d_env_gv0; f.rows = 1;
    if (w->d_env_gc0_rows) { f.gc0 = w->d_env_gc0_rows; f.gv0 = w->d_env_gv0_rows; f.rows = w->N; }      // rsb_env_set_reset_states
    f.env_task = true; f.env_reward = drew; f.env_ob = dob; f.env_done = ddone;
    w->fuse = f;
  }
  st = do_integrate(w, w->env_cfg.n_substeps);
  if (st != RSB_OK) return st;
  w->integrate1_valid = false; w->env_ob_valid = false;
  if (space == RSB_HOST) {
    if (reward) HIP_TRY(hipMemcpyAsync(reward, w->d_env_reward, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, stream_of(w)));
    if (done) HIP_TRY(hipMemcpyAsync(done, w->d_env_done, (size_t)N, hipMemcpyDeviceToHost, stream_of(w)));
    if (ob_next) HIP_TRY(hipMemcpyAsync(ob_next, w->d_env_ob, (size_t)N * od * sizeof(float), hipMemcpyDeviceToHost, stream_of(w)));
    HIP_TRY(hipStreamSynchronize(stream_of(w)));
  }
  return RSB_OK;
}

void* rsb_device_ptr(rsb_world* w, int field) {
  if (!w) return nullptr;
  switch (field) {
    case RSB_F_GC: w->raw_state = true; w->env_ob_valid = false; return w->d_gc;      // (sticky: the caller may write through the pointer at any time)
    case RSB_F_GV: w->raw_state = true; w->env_ob_valid = false; return w->d_gv;
    case RSB_F_PTARGET: return w->d_pt;
    case RSB_F_DTARGET: w->dt_zero = false; w->raw_dt = true; return w->d_dt;       // (the caller may write through the pointer: the rows are read from now on, whatever a later upload of zeros says)
    case RSB_F_TAU_FF: w->tff_zero = false; w->raw_tff = true; return w->d_tff;
    case RSB_F_CONTACT_COUNT: return w->d_count;
    case RSB_F_CONTACTS: return w->d_contacts;
    case RSB_F_FLAGS: return w->d_flags;
    default: return nullptr;
  }
}

// Debug aid: the next rsb_integrate() launches dump env `env`'s contact problem of the LAST sub-step
// (Delassus matrix G [3nc,3nc], free contact velocity c [3nc], solved impulses lam [3nc], contact frame
// coordinates [t1 t2 n]); rsb_debug_read_contact_problem copies it to the host.  env < 0 disables.
int rsb_debug_select_env(rsb_world* w, int env) {
  if (!w || env >= w->N) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  const size_t n = 1 + 3 * RSB_MAX_CONTACTS * 3 * RSB_MAX_CONTACTS + 6 * RSB_MAX_CONTACTS;
  if (!w->d_dbg) HIP_TRY(hipMalloc(&w->d_dbg, n * sizeof(float)));
  HIP_TRY(hipMemsetAsync(w->d_dbg, 0, n * sizeof(float), stream_of(w)));
  w->dbg_env = env;
  return RSB_OK;
}
int rsb_debug_read_contact_problem(rsb_world* w, int* nc, float* G, float* c, float* lam) {
  if (!w || !nc || !w->d_dbg) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  const size_t n = 1 + 3 * RSB_MAX_CONTACTS * 3 * RSB_MAX_CONTACTS + 6 * RSB_MAX_CONTACTS;
  std::vector<float> buf(n);
  HIP_TRY(hipMemcpyAsync(buf.data(), w->d_dbg, n * sizeof(float), hipMemcpyDeviceToHost, stream_of(w)));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  const int k = (int)buf[0], n3 = 3 * k;
  *nc = k;
  if (G) std::memcpy(G, buf.data() + 1, sizeof(float) * n3 * n3);
  if (c) std::memcpy(c, buf.data() + 1 + n3 * n3, sizeof(float) * n3);
  if (lam) std::memcpy(lam, buf.data() + 1 + n3 * n3 + n3, sizeof(float) * n3);
  return RSB_OK;
}

// Debug aid: per-phase cycle stamps (s_memtime) of workgroup 0 in the last sub-step of each launch.
int rsb_debug_phase_cycles(rsb_world* w, int enable, long long* out16) {
  if (!w) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  const size_t nprof = 16 + 16 * (size_t)w->N;  // 16 phase stamps + 16 words per workgroup (upper bound: one env per wave)
  if (enable && !w->d_prof) { HIP_TRY(hipMalloc(&w->d_prof, nprof * sizeof(long long))); HIP_TRY(hipMemset(w->d_prof, 0, nprof * sizeof(long long))); }
  if (out16 && w->d_prof) {
    HIP_TRY(hipMemcpyAsync(out16, w->d_prof, 16 * sizeof(long long), hipMemcpyDeviceToHost, stream_of(w)));
    HIP_TRY(hipStreamSynchronize(stream_of(w)));
  }
  if (!enable && w->d_prof) { HIP_TRY(hipStreamSynchronize(stream_of(w))); HIP_TRY(hipFree(w->d_prof)); w->d_prof = nullptr; }
  return RSB_OK;
}

// per-workgroup profile of the last launch: out [4 * n_blocks] = {total cycles, Gauss-Seidel cycles, sum over
// sub-steps of the wave's max sweep count, max contact count}
int rsb_debug_wave_profile(rsb_world* w, long long* out, int n_blocks) {
  if (!w || !out || !w->d_prof || n_blocks > w->N) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipMemcpyAsync(out, w->d_prof + 16, 16 * (size_t)n_blocks * sizeof(long long), hipMemcpyDeviceToHost, stream_of(w)));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  return RSB_OK;
}

int rsb_enable_timing(rsb_world* w, int on) {
  if (!w || on < 0) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  for (hipEvent_t e : w->ring0) (void)hipEventDestroy(e);
  for (hipEvent_t e : w->ring1) (void)hipEventDestroy(e);
  w->ring0.clear(); w->ring1.clear(); w->ring_next = w->ring_count = 0;
  w->launch_index = 0;
  w->timing = on != 0;
  if (on > 1) {
    w->ring0.resize(on); w->ring1.resize(on);
    for (int i = 0; i < on; ++i) { HIP_TRY(hipEventCreate(&w->ring0[i])); HIP_TRY(hipEventCreate(&w->ring1[i])); }
  }
  return RSB_OK;
}
int rsb_set_timing_stride(rsb_world* w, int stride) {
  if (!w || stride < 1) return RSB_E_INVALID;
  w->timing_stride = stride; w->launch_index = 0;
  return RSB_OK;
}
int rsb_read_kernel_ms(rsb_world* w, float* ms, int n) {
  if (!w || !ms || n < 0) return RSB_E_INVALID;
  if (w->ring0.empty()) { rsb::set_error("rsb_read_kernel_ms: no timing ring (rsb_enable_timing(w, n) with n > 1)"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  const size_t have = w->ring_count, cap = w->ring0.size();
  const size_t take = (size_t)n < have ? (size_t)n : have;
  for (size_t i = 0; i < take; ++i) {   // oldest of the last `take` launches first
    const size_t slot = (w->ring_next + cap - take + i) % cap;
    HIP_TRY(hipEventElapsedTime(&ms[i], w->ring0[slot], w->ring1[slot]));
  }
  return (int)take;
}
int rsb_last_kernel_ms(rsb_world* w, float* ms) {
  if (!w || !ms) return RSB_E_INVALID;
  if (!w->timing) { rsb::set_error("rsb_last_kernel_ms: timing is disabled (rsb_enable_timing)"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipEventSynchronize(w->ev1));
  HIP_TRY(hipEventElapsedTime(ms, w->ev0, w->ev1));
  return RSB_OK;
}

}  // extern "C"
