// Python module `_elf`: same class and method names as the reference's module of that name
// (src_cpp/elf/Pybind.cc:27-112, pybind_module.cc:13-15), over the device-resident engine of facade.h.
// Host C++ only (g++); the GPU work sits behind the C-ABI of include/elfhip.h.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "facade.h"
#include "options_logging.h"

namespace py = pybind11;
using namespace elfhip_py;

PYBIND11_MODULE(_elf, m) {
  m.doc() = "MI355X-native stand-in for ELF's _elf extension (batch collector facade over libelfhip.so)";
  auto ref = py::return_value_policy::reference_internal;

  py::enum_<ReplyStatus>(m, "ReplyStatus")
      .value("SUCCESS", SUCCESS)
      .value("FAILED", FAILED)
      .value("UNKNOWN", UNKNOWN)
      .export_values();

  py::class_<Context>(m, "Context")
      .def("wait", &Context::wait, py::arg("timeout_usec") = 0, ref, py::call_guard<py::gil_scoped_release>())
      .def("step", &Context::step, py::arg("success") = SUCCESS, py::call_guard<py::gil_scoped_release>())
      .def("start", &Context::start)
      .def("stop", &Context::stop)
      .def("version", &Context::version)
      .def("allocateSharedMem", &Context::allocateSharedMem, ref)
      .def("createSharedMemOptions", &Context::createSharedMemOptions);

  py::class_<Size>(m, "Size").def("vec", &Size::vec, ref);

  py::class_<SharedMemOptions>(m, "SharedMemOptions")
      .def("idx", &SharedMemOptions::getIdx)
      .def("batchsize", &SharedMemOptions::getBatchSize)
      .def("label", &SharedMemOptions::getLabel, ref)
      .def("setTimeout", &SharedMemOptions::setTimeout);

  py::class_<SharedMem>(m, "SharedMem")
      .def("__getitem__", &SharedMem::get, ref)
      .def("getSharedMemOptions", &SharedMem::getSharedMemOptions, ref)
      .def("effective_batchsize", &SharedMem::getEffectiveBatchSize)
      .def("info", &SharedMem::info);

  py::class_<AnyP>(m, "AnyP").def("info", &AnyP::info).def("field", &AnyP::field, ref).def("set", &AnyP::setAddress);

  py::class_<FuncMapBase>(m, "FuncMapBase")
      .def("batchsize", &FuncMapBase::getBatchSize)
      .def("name", &FuncMapBase::getName, ref)
      .def("sz", &FuncMapBase::getSize, ref)
      .def("type_name", &FuncMapBase::getTypeName)
      .def("type_size", &FuncMapBase::getSizeOfType);

  py::class_<SearchAlgoOptions>(m, "SearchAlgoOptions")
      .def(py::init<>())
      .def_readwrite("use_prior", &SearchAlgoOptions::use_prior)
      .def_readwrite("c_puct", &SearchAlgoOptions::c_puct)
      .def_readwrite("unexplored_q_zero", &SearchAlgoOptions::unexplored_q_zero)
      .def_readwrite("root_unexplored_q_zero", &SearchAlgoOptions::root_unexplored_q_zero);

  py::class_<TSOptions>(m, "TSOptions")
      .def(py::init<>())
      .def_readwrite("max_num_moves", &TSOptions::max_num_moves)
      .def_readwrite("num_threads", &TSOptions::num_threads)
      .def_readwrite("num_rollouts_per_thread", &TSOptions::num_rollouts_per_thread)
      .def_readwrite("num_rollouts_per_batch", &TSOptions::num_rollouts_per_batch)
      .def_readwrite("verbose", &TSOptions::verbose)
      .def_readwrite("persistent_tree", &TSOptions::persistent_tree)
      .def_readwrite("pick_method", &TSOptions::pick_method)
      .def_readwrite("log_prefix", &TSOptions::log_prefix)
      .def_readwrite("virtual_loss", &TSOptions::virtual_loss)
      .def_readwrite("verbose_time", &TSOptions::verbose_time)
      .def_readwrite("alg_opt", &TSOptions::alg_opt)
      .def_readwrite("root_epsilon", &TSOptions::root_epsilon)
      .def_readwrite("root_alpha", &TSOptions::root_alpha);

  // src_cpp/elf/Pybind.cc:108-112: the submodules the reference's Python package subclasses and star-imports
  auto m_logging = m.def_submodule("_logging");
  register_logging(m_logging);
  auto m_options = m.def_submodule("_options");
  register_options(m_options);
}
