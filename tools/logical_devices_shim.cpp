// LD_PRELOAD shim: ONE physical GPU posing as several LOGICAL devices, to find multi-device bugs
// that `devices = {0, 0}` cannot show (with both contexts on GPU 0 a missing
// hipSetDevice before an allocation, launch or event on a worker / verifier / producer thread is
// invisible).  The shim reports CURDLE_LOGICAL_DEVICES devices (default 2), maps every one of them
// to physical device 0, remembers for every stream, event and device allocation the logical device
// that was current on the creating thread, and checks every use the way a real second GPU would
// punish it:
//   * a kernel launch on a stream of another logical device than the calling thread's current one
//   * an event recorded on a stream of another logical device than the event's
//   * an asynchronous copy / memset whose device-side pointer belongs to another logical device
//     than the stream it is queued on (legal in HIP with peer access, never intended here)
//   * a launch with more than 64 KiB of dynamic LDS of a kernel whose hipFuncSetAttribute opt-in was
//     not made while THIS logical device was current (the attribute is per device)
// A violation is printed ("[logical-devices] VIOLATION ...") and the call fails with
// hipErrorInvalidResourceHandle, so the library fails loudly; the summary at exit says how many
// launches were checked on which logical devices.
//   g++ -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/logical_devices_shim.cpp -o gpurun_out/logical_devices_shim.so -ldl
//   LD_PRELOAD=gpurun_out/logical_devices_shim.so CURDLE_TEST_DEVICES=0,1 python -m pytest tests/test_multi_device.py
// Test infrastructure: never part of libcurdlemsm.so.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <unordered_map>
#include <utility>

namespace {
int n_logical() {
  static const int n = [] {
    const char* e = getenv("CURDLE_LOGICAL_DEVICES");
    int v = e ? atoi(e) : 2;
    return v < 1 ? 1 : (v > 16 ? 16 : v);
  }();
  return n;
}
thread_local int tl_cur = 0;
std::mutex g_mu;
std::unordered_map<void*, int> g_streams, g_events;
std::map<uintptr_t, std::pair<size_t, int>> g_allocs;  // base -> (size, logical device)
std::set<std::pair<const void*, int>> g_lds_optin;     // (kernel, logical device)
std::atomic<long> g_violations{0}, g_launches[16], g_copies{0}, g_records{0};

// The HIP runtime may sit in a dlopen'ed (RTLD_LOCAL) dependency tree -- Python's import of torch, the
// binding's dlopen of libcurdlemsm.so -- which RTLD_NEXT does not search: ask the loaded library itself.
template <class F>
F real(const char* name) {
  static void* lib = [] {
    void* h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW);
    return h;
  }();
  void* f = lib ? dlsym(lib, name) : dlsym(RTLD_NEXT, name);
  if (!f) f = dlsym(RTLD_NEXT, name);
  if (!f) {
    fprintf(stderr, "[logical-devices] cannot resolve %s\n", name);
    abort();
  }
  return reinterpret_cast<F>(f);
}
#define REAL(name) static auto fn = real<decltype(&name)>(#name)

hipError_t violation(const char* what, int have, int want) {
  g_violations++;
  fprintf(stderr, "[logical-devices] VIOLATION: %s (object of logical device %d used under logical device %d)\n", what, want, have);
  return hipErrorInvalidResourceHandle;
}
// logical device of a stream (the null stream belongs to the current device); -1 = not ours
int stream_dev(hipStream_t s) {
  if (!s) return tl_cur;
  std::lock_guard<std::mutex> g(g_mu);
  auto it = g_streams.find((void*)s);
  return it == g_streams.end() ? -1 : it->second;
}
int ptr_dev(const void* p) {
  std::lock_guard<std::mutex> g(g_mu);
  auto it = g_allocs.upper_bound((uintptr_t)p);
  if (it == g_allocs.begin()) return -1;
  --it;
  return (uintptr_t)p < it->first + it->second.first ? it->second.second : -1;
}
bool bad_device(int d) { return d < 0 || d >= n_logical(); }

struct Summary {
  ~Summary() {
    fprintf(stderr, "[logical-devices] %d logical devices on one GPU; launches checked per device:", n_logical());
    for (int d = 0; d < n_logical(); d++) fprintf(stderr, " %ld", g_launches[d].load());
    fprintf(stderr, "; copies checked %ld, event records checked %ld; violations: %ld\n", g_copies.load(), g_records.load(),
            g_violations.load());
  }
} g_summary;
}  // namespace

extern "C" {
// ---- device enumeration and selection ----
hipError_t hipGetDeviceCount(int* n) {
  REAL(hipGetDeviceCount);
  int real_n = 0;
  hipError_t e = fn(&real_n);
  if (e != hipSuccess || real_n < 1) return e;
  *n = n_logical();
  return hipSuccess;
}
hipError_t hipSetDevice(int d) {
  REAL(hipSetDevice);
  if (bad_device(d)) return hipErrorInvalidDevice;
  hipError_t e = fn(0);
  if (e == hipSuccess) tl_cur = d;
  return e;
}
hipError_t hipGetDevice(int* d) {
  REAL(hipGetDevice);
  int r = 0;
  hipError_t e = fn(&r);
  if (e == hipSuccess) *d = tl_cur;
  return e;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int d) {
  REAL(hipGetDevicePropertiesR0600);
  return bad_device(d) ? hipErrorInvalidDevice : fn(p, 0);
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int d) {
  REAL(hipDeviceGetAttribute);
  return bad_device(d) ? hipErrorInvalidDevice : fn(v, a, 0);
}
hipError_t hipDeviceGet(hipDevice_t* dev, int ordinal) {
  REAL(hipDeviceGet);
  return bad_device(ordinal) ? hipErrorInvalidDevice : fn(dev, 0);
}
hipError_t hipDeviceGetPCIBusId(char* s, int len, int d) {
  REAL(hipDeviceGetPCIBusId);
  return bad_device(d) ? hipErrorInvalidDevice : fn(s, len, 0);
}
hipError_t hipDeviceCanAccessPeer(int* can, int d, int p) {
  if (bad_device(d) || bad_device(p)) return hipErrorInvalidDevice;
  *can = d != p ? 1 : 0;
  return hipSuccess;
}
// everything else that takes a device ordinal (PyTorch's start-up walks them for every device it counts)
#define MAP_DEV1(name, T1)                                     \
  hipError_t name(T1 a, hipDevice_t d) {                       \
    REAL(name);                                                \
    return bad_device(d) ? hipErrorInvalidDevice : fn(a, 0);   \
  }
MAP_DEV1(hipDeviceGetUuid, hipUUID*)
MAP_DEV1(hipDeviceTotalMem, size_t*)
MAP_DEV1(hipDeviceGetDefaultMemPool, hipMemPool_t*)
MAP_DEV1(hipDeviceGetMemPool, hipMemPool_t*)
MAP_DEV1(hipDevicePrimaryCtxRetain, hipCtx_t*)
#undef MAP_DEV1
hipError_t hipDeviceComputeCapability(int* major, int* minor, hipDevice_t d) {
  REAL(hipDeviceComputeCapability);
  return bad_device(d) ? hipErrorInvalidDevice : fn(major, minor, 0);
}
hipError_t hipDeviceGetName(char* name, int len, hipDevice_t d) {
  REAL(hipDeviceGetName);
  return bad_device(d) ? hipErrorInvalidDevice : fn(name, len, 0);
}
hipError_t hipDeviceSetMemPool(int d, hipMemPool_t pool) {
  REAL(hipDeviceSetMemPool);
  return bad_device(d) ? hipErrorInvalidDevice : fn(0, pool);
}
hipError_t hipDeviceGetP2PAttribute(int* v, hipDeviceP2PAttr attr, int a, int b) {
  if (bad_device(a) || bad_device(b)) return hipErrorInvalidDevice;
  *v = attr == hipDevP2PAttrPerformanceRank ? 0 : 1;
  return hipSuccess;
}
hipError_t hipExtGetLinkTypeAndHopCount(int a, int b, uint32_t* linktype, uint32_t* hops) {
  if (bad_device(a) || bad_device(b)) return hipErrorInvalidDevice;
  *linktype = 4;  // HSA_AMD_LINK_INFO_TYPE_XGMI
  *hops = 1;
  return hipSuccess;
}
hipError_t hipDevicePrimaryCtxGetState(hipDevice_t d, unsigned* flags, int* active) {
  REAL(hipDevicePrimaryCtxGetState);
  return bad_device(d) ? hipErrorInvalidDevice : fn(0, flags, active);
}
hipError_t hipDevicePrimaryCtxRelease(hipDevice_t d) {
  REAL(hipDevicePrimaryCtxRelease);
  return bad_device(d) ? hipErrorInvalidDevice : fn(0);
}
hipError_t hipDevicePrimaryCtxReset(hipDevice_t d) {
  REAL(hipDevicePrimaryCtxReset);
  return bad_device(d) ? hipErrorInvalidDevice : fn(0);
}
hipError_t hipDevicePrimaryCtxSetFlags(hipDevice_t d, unsigned flags) {
  REAL(hipDevicePrimaryCtxSetFlags);
  return bad_device(d) ? hipErrorInvalidDevice : fn(0, flags);
}
hipError_t hipDeviceGetGraphMemAttribute(int d, hipGraphMemAttributeType attr, void* value) {
  REAL(hipDeviceGetGraphMemAttribute);
  return bad_device(d) ? hipErrorInvalidDevice : fn(0, attr, value);
}
hipError_t hipDeviceSetGraphMemAttribute(int d, hipGraphMemAttributeType attr, void* value) {
  REAL(hipDeviceSetGraphMemAttribute);
  return bad_device(d) ? hipErrorInvalidDevice : fn(0, attr, value);
}
hipError_t hipDeviceEnablePeerAccess(int p, unsigned) { return bad_device(p) ? hipErrorInvalidDevice : hipSuccess; }
hipError_t hipDeviceDisablePeerAccess(int p) { return bad_device(p) ? hipErrorInvalidDevice : hipSuccess; }

// ---- streams ----
static hipError_t tag_stream(hipError_t e, hipStream_t* s) {
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> g(g_mu);
    g_streams[(void*)*s] = tl_cur;
  }
  return e;
}
hipError_t hipStreamCreate(hipStream_t* s) {
  REAL(hipStreamCreate);
  return tag_stream(fn(s), s);
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned f) {
  REAL(hipStreamCreateWithFlags);
  return tag_stream(fn(s, f), s);
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int p) {
  REAL(hipStreamCreateWithPriority);
  return tag_stream(fn(s, f, p), s);
}
hipError_t hipStreamDestroy(hipStream_t s) {
  REAL(hipStreamDestroy);
  {
    std::lock_guard<std::mutex> g(g_mu);
    g_streams.erase((void*)s);
  }
  return fn(s);
}

int hipGetStreamDeviceId(hipStream_t s) {
  const int d = stream_dev(s);
  return d >= 0 ? d : tl_cur;
}

// ---- events ----
static hipError_t tag_event(hipError_t e, hipEvent_t* ev) {
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> g(g_mu);
    g_events[(void*)*ev] = tl_cur;
  }
  return e;
}
hipError_t hipEventCreate(hipEvent_t* ev) {
  REAL(hipEventCreate);
  return tag_event(fn(ev), ev);
}
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned f) {
  REAL(hipEventCreateWithFlags);
  return tag_event(fn(ev, f), ev);
}
hipError_t hipEventDestroy(hipEvent_t ev) {
  REAL(hipEventDestroy);
  {
    std::lock_guard<std::mutex> g(g_mu);
    g_events.erase((void*)ev);
  }
  return fn(ev);
}
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t s) {
  REAL(hipEventRecord);
  int ed = -1;
  {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_events.find((void*)ev);
    if (it != g_events.end()) ed = it->second;
  }
  const int sd = stream_dev(s);
  if (ed >= 0 && sd >= 0) {
    g_records++;
    if (ed != sd) return violation("hipEventRecord: event and stream of different devices", sd, ed);
  }
  return fn(ev, s);
}

// ---- memory ----
hipError_t hipMalloc(void** p, size_t n) {
  static auto fn = real<hipError_t (*)(void**, size_t)>("hipMalloc");  // the header overloads the name for C++
  hipError_t e = fn(p, n);
  if (e == hipSuccess && *p) {
    std::lock_guard<std::mutex> g(g_mu);
    g_allocs[(uintptr_t)*p] = {n, tl_cur};
  }
  return e;
}
hipError_t hipFree(void* p) {
  REAL(hipFree);
  if (p) {
    std::lock_guard<std::mutex> g(g_mu);
    g_allocs.erase((uintptr_t)p);
  }
  return fn(p);
}
static hipError_t check_ptr_on_stream(const char* what, const void* p, hipStream_t s) {
  const int pd = ptr_dev(p), sd = stream_dev(s);
  if (pd >= 0 && sd >= 0) {
    g_copies++;
    if (pd != sd) return violation(what, sd, pd);
  }
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s) {
  REAL(hipMemcpyAsync);
  hipError_t e = check_ptr_on_stream("hipMemcpyAsync: destination of another device than the stream", dst, s);
  if (e == hipSuccess) e = check_ptr_on_stream("hipMemcpyAsync: source of another device than the stream", src, s);
  return e != hipSuccess ? e : fn(dst, src, n, kind, s);
}
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s) {
  REAL(hipMemsetAsync);
  hipError_t e = check_ptr_on_stream("hipMemsetAsync: memory of another device than the stream", dst, s);
  return e != hipSuccess ? e : fn(dst, v, n, s);
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind) {
  REAL(hipMemcpy);
  // the null stream of the CURRENT device
  hipError_t e = check_ptr_on_stream("hipMemcpy: destination of another device than the current one", dst, nullptr);
  if (e == hipSuccess) e = check_ptr_on_stream("hipMemcpy: source of another device than the current one", src, nullptr);
  return e != hipSuccess ? e : fn(dst, src, n, kind);
}

hipError_t hipMemcpyPeerAsync(void* dst, int dd, const void* src, int sd, size_t n, hipStream_t s) {
  REAL(hipMemcpyPeerAsync);
  return bad_device(dd) || bad_device(sd) ? hipErrorInvalidDevice : fn(dst, 0, src, 0, n, s);
}
hipError_t hipMemcpyPeer(void* dst, int dd, const void* src, int sd, size_t n) {
  REAL(hipMemcpyPeer);
  return bad_device(dd) || bad_device(sd) ? hipErrorInvalidDevice : fn(dst, 0, src, 0, n);
}

// ---- kernels ----
hipError_t hipFuncSetAttribute(const void* func, hipFuncAttribute attr, int value) {
  REAL(hipFuncSetAttribute);
  hipError_t e = fn(func, attr, value);
  if (e == hipSuccess && attr == hipFuncAttributeMaxDynamicSharedMemorySize) {
    std::lock_guard<std::mutex> g(g_mu);
    g_lds_optin.insert({func, tl_cur});
  }
  return e;
}
hipError_t hipLaunchKernel(const void* func, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t s) {
  REAL(hipLaunchKernel);
  const int sd = stream_dev(s);
  if (sd >= 0) {
    g_launches[tl_cur]++;
    if (sd != tl_cur) return violation("hipLaunchKernel: stream of another device than the calling thread's current one", tl_cur, sd);
    if (shmem > 65536) {
      std::lock_guard<std::mutex> g(g_mu);
      if (!g_lds_optin.count({func, tl_cur})) {
        g_violations++;
        fprintf(stderr, "[logical-devices] VIOLATION: launch with %zu bytes of dynamic LDS on logical device %d without the "
                        "kernel's hipFuncSetAttribute opt-in on that device\n", shmem, tl_cur);
        return hipErrorInvalidValue;
      }
    }
  }
  return fn(func, grid, block, args, shmem, s);
}
}
