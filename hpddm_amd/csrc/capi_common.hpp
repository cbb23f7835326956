// Shared helpers of the C ABI translation units: exceptions never cross the boundary.
#pragma once
#include "common.hpp"
#include <string>

struct HpddmHipSubdomain;
namespace hpddm_hip {
std::string &last_error();
struct LocalSolver;
LocalSolver &local_solver_of(HpddmHipSubdomain *S);
} // namespace hpddm_hip

#define HH_TRY(...)                                 \
  try {                                             \
    __VA_ARGS__                                     \
  } catch (const std::exception &e_) {              \
    ::hpddm_hip::last_error() = e_.what();          \
    return -1;                                      \
  } catch (...) {                                   \
    ::hpddm_hip::last_error() = "unknown exception"; \
    return -2;                                      \
  }
