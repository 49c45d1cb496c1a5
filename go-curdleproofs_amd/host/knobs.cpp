#include "knobs.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

namespace curdle {
namespace knobs {
namespace {
const char* const kNames[COUNT] = {
    "WINDOW_BITS", "SEG_LEN", "REDUCE_SEG", "SCATTER", "HOST_CHUNKS", "MAX_MSMS_PER_PASS",
    "MULTI_DEVICE_MIN", "MAIN_STREAMS", "TWO_KERNEL_MAX", "QUAD_MAX_LANES", "BATCH_CHUNK", "BATCH_PRODUCERS",
    "BATCH_GROUP", "DEVICE_ACC", "HOST_DECODE", "VERIFY_EAGER", "VERIFY_TRACE", "PROVER_FOLD_BASES",
    "ACC_PRIO", "REDUCE_PRIO", "AUX_PRIO", "HOST_FOLD"};
std::atomic<long long> g_val[COUNT];
std::once_flag g_once;
void load() {
  for (int i = 0; i < COUNT; i++) {
    char env[64] = "CURDLE_";
    strncat(env, kNames[i], sizeof(env) - 8);
    const char* e = getenv(env);
    long long v = -1;
    if (e && *e) {
      v = atoll(e);
      if (v < 0) v = -1;
    }
    g_val[i].store(v, std::memory_order_relaxed);
  }
}
}  // namespace

long long get(Id id) {
  std::call_once(g_once, load);
  return g_val[id].load(std::memory_order_relaxed);
}
const char* name(Id id) { return kNames[id]; }
int set(const char* nm, long long value) {
  if (!nm) return -1;
  std::call_once(g_once, load);
  if (!strncmp(nm, "CURDLE_", 7)) nm += 7;
  for (int i = 0; i < COUNT; i++)
    if (!strcmp(nm, kNames[i])) {
      g_val[i].store(value < 0 ? -1 : value, std::memory_order_relaxed);
      return 0;
    }
  return -1;
}
}  // namespace knobs
}  // namespace curdle
