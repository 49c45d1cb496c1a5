// Generic build of the window combine + run-time dispatch.
#define CURDLE_COMBINE_NAME curdle_window_combine_generic
#include "window_combine_impl.h"

extern "C" void curdle_window_combine_bmi2(const void*, int, const int*, uint64_t[18]);

extern "C" void curdle_window_combine(const void* winsums_xyzz, int nw, const int* dbls, uint64_t out[18]) {
  static const bool fast = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
  if (fast)
    curdle_window_combine_bmi2(winsums_xyzz, nw, dbls, out);
  else
    curdle_window_combine_generic(winsums_xyzz, nw, dbls, out);
}
