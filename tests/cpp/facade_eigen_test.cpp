// facade_eigen_test.cpp - the facade's Eigen-typed boundary under a compiler.
// Built with -I tests/cpp/eigen_stub (the test infrastructure's stand-in for Eigen3, the reference's one declared dependency:
// /root/reference/.travis.yml:7) so that RAISIM_HAS_EIGEN is defined and include/raisim/*.hpp compile their Eigen branch:
//   part 1 (no GPU): the stand-in itself and the facade's math types against hand-computed values
//   part 2 (GPU)   : raisim::VectorizedEnvironment<ENVIRONMENT> over tests/cpp/anymal_env_eigen/Environment.hpp - written the way upstream's
//                    rsg_anymal environment is (Eigen expressions, Eigen::Ref arguments, termination by body index) - equals the
//                    device-resident env configured with the same rule (feet AND knees sit on the shanks) over 80 control steps with resets,
//                    the 4 integrate() calls of a control step still being ONE launch.
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "anymal_env_eigen/Environment.hpp"
#include "raisim/VectorizedEnvironment.hpp"

#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static int stub_and_math() {
  // the comma initialiser with scalars and vectors, tail / segment / row / transpose views, cast, cwiseProduct, matrix * vector
  Eigen::VectorXd a(5), b(3);
  b << 1.0, 2.0, 3.0;
  a << 9.0, b, 7.0;
  CHECK(a[0] == 9.0 && a[1] == 1.0 && a[3] == 3.0 && a[4] == 7.0 && a.size() == 5);
  a.tail(2).setConstant(4.0);
  CHECK(a[3] == 4.0 && a[4] == 4.0 && a[2] == 2.0);
  a.segment(1, 2) = b.head(2) * 10.0;
  CHECK(a[1] == 10.0 && a[2] == 20.0);
  Eigen::VectorXd c = b.cwiseProduct(b);
  c += b;
  CHECK(c[2] == 12.0 && std::fabs(b.squaredNorm() - 14.0) < 1e-12 && std::fabs(b.norm() - std::sqrt(14.0)) < 1e-12);
  Eigen::Matrix<float, Eigen::Dynamic, 1> f = b.cast<float>();
  CHECK(f[1] == 2.0f && f.size() == 3);
  raisim::Mat<3, 3> R;
  raisim::Vec<4> q;
  q[0] = std::cos(0.35); q[1] = 0.0; q[2] = 0.0; q[3] = std::sin(0.35);      // yaw 0.7
  raisim::quatToRotMat(q, R);
  CHECK(std::fabs(R(0, 0) - std::cos(0.7)) < 1e-12 && std::fabs(R(1, 0) - std::sin(0.7)) < 1e-12 && std::fabs(R(2, 2) - 1.0) < 1e-12);
  Eigen::Vector3d v; v << 1.0, 0.0, 0.5;
  Eigen::Vector3d w = R.e().transpose() * v;        // world -> body
  CHECK(std::fabs(w[0] - std::cos(0.7)) < 1e-12 && std::fabs(w[1] + std::sin(0.7)) < 1e-12 && std::fabs(w[2] - 0.5) < 1e-12);
  Eigen::Vector3d r2 = R.e().row(2).transpose();
  CHECK(r2[0] == 0.0 && r2[2] == 1.0);
  raisim::Vec<4> q2;
  raisim::rotMatToQuat(R, q2);
  CHECK(std::fabs(q2[0] - q[0]) < 1e-12 && std::fabs(q2[3] - q[3]) < 1e-12);
  // a row of a row-major float matrix as Eigen::Ref<EigenVec>, written through
  raisim::EigenRowMajorMat M(2, 3);
  auto fill = [](Eigen::Ref<raisim::EigenVec> row, float x) { for (int i = 0; i < row.size(); ++i) row[i] = x + (float)i; };
  fill(raisim::rowOf(M.data() + 3, 3), 5.f);
  CHECK(M(1, 0) == 5.f && M(1, 2) == 7.f && M(0, 1) == 0.f);
  // VecDyn <- Eigen and back
  raisim::VecDyn d = b;
  CHECK(d.size() == 3 && d[2] == 3.0);
  Eigen::VectorXd e2 = d.e();
  CHECK(e2[0] == 1.0 && e2.size() == 3);
  // the stream-style message macros
  bool threw = false;
  try { RSFATAL_IF(e2.size() == 3, "size " << e2.size() << " of " << 3); } catch (const std::runtime_error& ex) { threw = std::string(ex.what()) == "size 3 of 3"; }
  CHECK(threw);
  RSINFO_IF(false, "not printed " << 1);
  RSWARN_IF(false, "not printed " << 2);
  std::printf("eigen stand-in + facade math OK\n");
  return 0;
}

int main(int argc, char** argv) {
  if (stub_and_math() != 0) return 1;
  if (argc < 2) { std::printf("facade_eigen_test OK (no URDF given: host part only)\n"); return 0; }
  if (rsb_device_count() <= 0) { std::printf("no HIP device: the Eigen-typed environment compiled, its GPU run is skipped\nfacade_eigen_test OK (host part)\n"); return 0; }
  try {
    const std::string urdf = argv[1];
    const std::string resourceDir = urdf.substr(0, urdf.find_last_of('/'));
    const int NE = 64;
    const std::string yaml =
        "num_envs: 64\nnum_threads: 8\nsimulation_dt: 0.0025\ncontrol_dt: 0.01\nrender: false\naction_std: 0.3\n"
        "reward:\n  forwardVel:\n    coeff: 0.3\n  torque:\n    coeff: -4e-5\n";
    raisim::VectorizedEnvironment<raisim::ENVIRONMENT> venv(resourceDir, yaml, /*normalizeObservation=*/false);
    CHECK(venv.getNumOfEnvs() == NE && venv.getObDim() == 34 && venv.getActionDim() == 12);
    raisim::VecEnvConfig dc;
    dc.num_envs = NE; dc.torque_reward_coeff = -4e-5; dc.forward_vel_reward_coeff = 0.3;
    dc.gc_init = {0, 0, 0.57, 1.0, 0.0, 0.0, 0.0, 0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8};
    dc.foot_collision_suffixes = {"_foot", "_knee"};      // upstream's rule is by BODY: every primitive of the four shanks may touch
    raisim::DeviceVectorizedEnvironment denv(urdf, dc);
    denv.init();
    std::vector<float> a((size_t)NE * 12), r1(NE), r2(NE), o1((size_t)NE * 34), o2((size_t)NE * 34);
    std::unique_ptr<bool[]> d1(new bool[NE]), d2(new bool[NE]);
    unsigned sd = 4242u;
    int ndone = 0, nparted = 0;
    std::vector<bool> parted(NE, false);
    const long l0 = venv.batch()->viewLaunches();
    const int STEPS = 80;
    for (int it = 0; it < STEPS; ++it) {
      // (by BODY the knees may touch: it takes harder kicks than in facade_test.cpp until a thigh or the trunk reaches the ground)
      for (auto& x : a) { sd = sd * 1664525u + 1013904223u; x = ((sd >> 8) / 16777216.0f - 0.5f) * (it % 4 == 3 ? 16.0f : 3.0f); }
      venv.step(a.data(), NE, 12, r1.data(), d1.get());
      denv.step(a.data(), NE, 12, r2.data(), d2.get());
      venv.observe(o1.data(), NE, 34, false);
      denv.observe(o2.data(), NE, 34);
      // The environment scales its actions in DOUBLE (upstream's Eigen expressions), the device env in float: targets differ in the last bit, and a
      // robot kicked onto its knees amplifies that.  An env's trajectory is compared (1e-4) for as long as it coincides - every env for the first
      // 20 control steps -; one that has parted (> 1e-3, or a different done flag) is counted and left alone afterwards.
      for (int e = 0; e < NE; ++e) {
        ndone += d1[e] ? 1 : 0;
        if (parted[e]) continue;
        float worst = std::fabs(r1[e] - r2[e]);
        for (int k = 0; k < 34; ++k) worst = std::fmax(worst, std::fabs(o1[(size_t)e * 34 + k] - o2[(size_t)e * 34 + k]));
        if (d1[e] != d2[e] || worst > 1e-3f) { CHECK(it >= 20); parted[e] = true; ++nparted; continue; }
        CHECK(worst < 1e-4f || it >= 20);
      }
    }
    CHECK(venv.batch()->viewLaunches() - l0 == STEPS);       // the 4 integrate() calls of a control step: ONE fused launch for all 64 envs
    CHECK(ndone > 0);
    CHECK(nparted <= NE / 4);
    std::printf("%d of %d envs parted from the float-scaled device env after step 20 (double action scaling)\n", nparted, NE);
    std::printf("VectorizedEnvironment<ENVIRONMENT (Eigen-typed, termination by body)>: %d envs x %d control steps, %d resets, equal to the device-resident env\n", NE, STEPS, ndone);
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  std::printf("facade_eigen_test OK\n");
  return 0;
}
