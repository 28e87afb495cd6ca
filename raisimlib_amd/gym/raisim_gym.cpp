// raisim_gym.cpp — the Python entry of the drop-in: a pybind11 module with upstream's raisim_gym surface
// [RECALL raisimGymTorch/env/raisim_gym.cpp; absent from /root/reference, SURVEY.md §8b]:
//
//     from raisimlib_amd.lib.<module> import RaisimGymEnv          # upstream: from raisimGymTorch.env.bin.rsg_anymal import RaisimGymEnv
//     env = RaisimGymEnv(resource_dir, cfg_yaml_text)
//     env.reset(); env.observe(ob, update_statistics); env.step(action, reward, done)     # numpy buffers, written IN PLACE
//
// It instantiates raisim::VectorizedEnvironment<ENVIRONMENT> (include/raisim/VectorizedEnvironment.hpp) over the user's
// UNMODIFIED Environment.hpp - the file named by -DRSG_ENVIRONMENT_HEADER at build time (raisimlib_amd/gym/build_gym.py),
// exactly as upstream compiles one module per environment folder.  The N environments' Worlds are the replicas of one
// GPU batch: every World::integrate() inside Environment::step() is ONE launch of the HIP step kernel for all of them.
//
// Upstream takes Eigen::Ref<EigenRowMajorMat> / EigenVec / EigenBoolVec (pybind11/eigen.h); Eigen is not installed here, so
// the bindings take C-contiguous numpy arrays of the same dtypes and shapes (float32 [num_envs, dim], float32 [num_envs],
// bool [num_envs]) through the buffer protocol and write them in place - the Python side cannot tell the difference.
// Every in-place argument is bound .noconvert(): an array of another dtype or layout raises TypeError, as upstream's Eigen::Ref
// bindings do, instead of being converted to a temporary copy that would swallow the results.
// A second class, DeviceRaisimGymEnv, binds raisim::DeviceVectorizedEnvironment (rsg_anymal's task compiled into the library;
// pointer arguments are device addresses, e.g. torch.Tensor.data_ptr()).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>

#ifndef RSG_ENVIRONMENT_HEADER
#error "build with -DRSG_ENVIRONMENT_HEADER=\"<path to the environment's Environment.hpp>\" (raisimlib_amd/gym/build_gym.py does)"
#endif
#include RSG_ENVIRONMENT_HEADER
#include "raisim/VectorizedEnvironment.hpp"

#ifndef RSG_MODULE_NAME
#define RSG_MODULE_NAME raisim_gym
#endif

namespace py = pybind11;
using FMat = py::array_t<float, py::array::c_style>;
using BVec = py::array_t<bool, py::array::c_style>;
using VecEnv = raisim::VectorizedEnvironment<raisim::ENVIRONMENT>;

namespace {
void need(bool ok, const char* what) { if (!ok) throw std::invalid_argument(what); }
float* mat(FMat& a, int rows, int cols, const char* what) {
  need(a.ndim() == 2 && a.shape(0) == rows && a.shape(1) == cols && a.writeable(), what);
  return a.mutable_data();
}
float* vec(FMat& a, int n, const char* what) {
  need(a.size() == n && a.writeable(), what);
  return a.mutable_data();
}
}  // namespace

PYBIND11_MODULE(RSG_MODULE_NAME, m) {
  m.doc() = "raisim_gym-style module over the MI355X-native batched simulator (raisimlib_amd)";
  py::class_<VecEnv>(m, "RaisimGymEnv")
      .def(py::init<std::string, std::string>(), py::arg("resourceDir"), py::arg("cfg"))
      .def(py::init<std::string, std::string, bool>(), py::arg("resourceDir"), py::arg("cfg"), py::arg("normalizeObservation"))
      .def("init", &VecEnv::init)
      .def("reset", &VecEnv::reset)
      .def("observe", [](VecEnv& e, FMat ob, bool updateStatistics) {
             e.observe(mat(ob, e.getNumOfEnvs(), e.getObDim(), "observe: ob must be a writeable C-contiguous float32 [num_envs, obDim] array"),
                       e.getNumOfEnvs(), e.getObDim(), updateStatistics);
           }, py::arg("ob").noconvert(), py::arg("updateStatistics") = true)
      .def("step", [](VecEnv& e, FMat action, FMat reward, BVec done) {
             need(action.ndim() == 2 && action.shape(0) == e.getNumOfEnvs() && action.shape(1) == e.getActionDim(), "step: action must be float32 [num_envs, actionDim]");
             need(done.size() == e.getNumOfEnvs() && done.writeable(), "step: done must be a writeable bool [num_envs] array");
             float* r = vec(reward, e.getNumOfEnvs(), "step: reward must be a writeable float32 [num_envs] array");
             py::gil_scoped_release nogil;      // the N Environment::step() bodies and their launches run without the GIL
             e.step(action.data(), e.getNumOfEnvs(), e.getActionDim(), r, done.mutable_data());
           }, py::arg("action").noconvert(), py::arg("reward").noconvert(), py::arg("done").noconvert())
      .def("setSeed", &VecEnv::setSeed)
      .def("close", &VecEnv::close)
      .def("isTerminalState", [](VecEnv& e, BVec t) {
             need(t.size() == e.getNumOfEnvs() && t.writeable(), "isTerminalState: bool [num_envs] array expected");
             e.isTerminalState(t.mutable_data());
           }, py::arg("terminalState").noconvert())
      .def("setSimulationTimeStep", &VecEnv::setSimulationTimeStep)
      .def("setControlTimeStep", &VecEnv::setControlTimeStep)
      .def("getObDim", &VecEnv::getObDim)
      .def("getActionDim", &VecEnv::getActionDim)
      .def("getNumOfEnvs", &VecEnv::getNumOfEnvs)
      .def("turnOnVisualization", &VecEnv::turnOnVisualization)
      .def("turnOffVisualization", &VecEnv::turnOffVisualization)
      .def("stopRecordingVideo", &VecEnv::stopRecordingVideo)
      .def("startRecordingVideo", &VecEnv::startRecordingVideo)
      .def("curriculumUpdate", &VecEnv::curriculumUpdate)
      .def("getObStatistics", [](VecEnv& e, FMat mean, FMat var) {
             float count = 0.f;
             e.getObStatistics(vec(mean, e.getObDim(), "getObStatistics: mean must be float32 [obDim]"), vec(var, e.getObDim(), "getObStatistics: var must be float32 [obDim]"), count);
             return count;       // (upstream passes count by reference; a Python float cannot be written in place)
           }, py::arg("mean").noconvert(), py::arg("var").noconvert())
      .def("setObStatistics", [](VecEnv& e, FMat mean, FMat var, float count) {
             need(mean.size() == e.getObDim() && var.size() == e.getObDim(), "setObStatistics: float32 [obDim] arrays expected");
             e.setObStatistics(mean.data(), var.data(), count);
           })
      .def("getRewardInfo", &VecEnv::getRewardInfo)
      // what upstream does not have: how many kernel launches the batch has issued (tests: N envs x k integrate() calls -> ONE fused launch)
      .def("viewLaunches", [](VecEnv& e) { return e.batch() ? e.batch()->viewLaunches() : 0L; })
      // ... and where step() spends the host's time (ns, accumulated; reset = True clears the counters): tools/prof_template_path.py
      .def("stepProfile", [](VecEnv& e, bool reset) {
             py::dict d;
             const auto& p = e.stepProfile();
             d["total_ns"] = p.total_ns; d["flush_ns"] = p.flush_ns; d["steps"] = p.steps; d["flushes"] = p.flushes;
             if (e.batch()) {
               d["flush_prep_ns"] = e.batch()->flushPrepNs(); d["flush_exchange_ns"] = e.batch()->flushExchangeNs();
               long long v[5] = {0, 0, 0, 0, 0};
               rsb_debug_view_profile(e.batch()->handle(), v, reset ? 1 : 0);
               d["exchange_upload_enqueue_ns"] = v[0]; d["exchange_launch_enqueue_ns"] = v[1]; d["exchange_download_enqueue_ns"] = v[2]; d["exchange_wait_ns"] = v[3]; d["exchanges"] = v[4];
             }
             if (reset) e.resetStepProfile();
             return d;
           }, py::arg("reset") = false);

  py::class_<raisim::VecEnvConfig>(m, "VecEnvConfig")
      .def(py::init<>())
      .def_readwrite("num_envs", &raisim::VecEnvConfig::num_envs)
      .def_readwrite("simulation_dt", &raisim::VecEnvConfig::simulation_dt)
      .def_readwrite("control_dt", &raisim::VecEnvConfig::control_dt)
      .def_readwrite("action_std", &raisim::VecEnvConfig::action_std)
      .def_readwrite("p_gain", &raisim::VecEnvConfig::p_gain)
      .def_readwrite("d_gain", &raisim::VecEnvConfig::d_gain)
      .def_readwrite("forward_vel_reward_coeff", &raisim::VecEnvConfig::forward_vel_reward_coeff)
      .def_readwrite("torque_reward_coeff", &raisim::VecEnvConfig::torque_reward_coeff)
      .def_readwrite("terminal_reward", &raisim::VecEnvConfig::terminal_reward)
      .def_readwrite("gc_init", &raisim::VecEnvConfig::gc_init)
      .def_readwrite("device", &raisim::VecEnvConfig::device);

  using DevEnv = raisim::DeviceVectorizedEnvironment;
  py::class_<DevEnv>(m, "DeviceRaisimGymEnv")
      .def(py::init<std::string, raisim::VecEnvConfig>(), py::arg("urdfPath"), py::arg("cfg"))
      .def("init", &DevEnv::init)
      .def("reset", &DevEnv::reset)
      .def("observe", [](DevEnv& e, FMat ob, bool u) { e.observe(mat(ob, e.getNumOfEnvs(), e.getObDim(), "observe: float32 [num_envs, obDim]"), e.getNumOfEnvs(), e.getObDim(), u); },
           py::arg("ob").noconvert(), py::arg("updateStatistics") = false)
      .def("step", [](DevEnv& e, FMat action, FMat reward, BVec done) {
             need(action.ndim() == 2 && action.shape(0) == e.getNumOfEnvs() && action.shape(1) == e.getActionDim(), "step: action must be float32 [num_envs, actionDim]");
             need(done.size() == e.getNumOfEnvs() && done.writeable(), "step: done must be a writeable bool [num_envs] array");
             e.step(action.data(), e.getNumOfEnvs(), e.getActionDim(), vec(reward, e.getNumOfEnvs(), "step: reward float32 [num_envs]"), done.mutable_data());
           }, py::arg("action").noconvert(), py::arg("reward").noconvert(), py::arg("done").noconvert())
      // device-resident loop: the arguments are DEVICE addresses (torch.Tensor.data_ptr()); nothing crosses PCIe, nothing synchronises
      .def("observeDevice", [](DevEnv& e, std::uintptr_t ob) { e.observeDevice(reinterpret_cast<float*>(ob)); })
      .def("stepDevice", [](DevEnv& e, std::uintptr_t action, std::uintptr_t reward, std::uintptr_t done, std::uintptr_t ob_next) {
             e.stepDevice(reinterpret_cast<const float*>(action), reinterpret_cast<float*>(reward), reinterpret_cast<uint8_t*>(done), reinterpret_cast<float*>(ob_next));
           }, py::arg("action"), py::arg("reward"), py::arg("done"), py::arg("ob_next") = 0)
      .def("getObDim", &DevEnv::getObDim)
      .def("getActionDim", &DevEnv::getActionDim)
      .def("getNumOfEnvs", &DevEnv::getNumOfEnvs);
}
