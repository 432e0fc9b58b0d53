// api.hip — library services of libudt_kernels.so: version, status strings, the shared zero page used
// by the LDS-DMA gathers, and per-op-class HIP-event timing for bench.py's roofline object.
#include "common.h"

#include <mutex>
#include <stdio.h>
#include <string.h>
#include <vector>

namespace {

std::mutex g_mu;
int g_last_hip_error = 0;

struct ProfRec {
  hipEvent_t start, stop;
  int cls;
  char tag[96];
};
struct TraceRow { int cls; float ms; char tag[96]; };
std::vector<TraceRow> g_trace;
bool g_trace_on = false;
uint32_t g_prof_mask = 0;
std::vector<ProfRec> g_recs;        // recorded, not yet folded
std::vector<ProfRec> g_pool;        // reusable event pairs
double g_total_ms[UDT_PROF_NCLASS] = {0};
int64_t g_launches[UDT_PROF_NCLASS] = {0};

}  // namespace

int udt_set_hip_error(hipError_t e) {
  if (e == hipSuccess) return UDT_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  g_last_hip_error = (int)e;
  return UDT_ERR_HIP;
}

// one zero page per device (the library is used one process per GPU, but nothing here assumes it)
const uint16_t* udt_zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  static uint16_t* pages[16] = {nullptr};
  std::lock_guard<std::mutex> lk(g_mu);
  if (!pages[dev]) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, 4096);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return nullptr; }
    e = hipMemset(p, 0, 4096);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return nullptr; }
    pages[dev] = reinterpret_cast<uint16_t*>(p);
  }
  return pages[dev];
}

void udt_prof_tag(void* rec, const char* tag) {
  if (!rec || !tag) return;
  std::lock_guard<std::mutex> lk(g_mu);
  const size_t idx = reinterpret_cast<size_t>(rec) - 1;
  if (idx < g_recs.size()) { strncpy(g_recs[idx].tag, tag, sizeof(g_recs[idx].tag) - 1); }
}

UdtProfScope::UdtProfScope(int cls_, hipStream_t s_) : cls(cls_), s(s_), rec(nullptr) {
  if (!(g_prof_mask & (1u << cls))) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r;
  if (!g_pool.empty()) {
    r = g_pool.back();
    g_pool.pop_back();
  } else {
    if (hipEventCreate(&r.start) != hipSuccess) return;
    if (hipEventCreate(&r.stop) != hipSuccess) return;
  }
  r.cls = cls;
  r.tag[0] = 0;
  (void)hipEventRecord(r.start, s);
  g_recs.push_back(r);
  rec = reinterpret_cast<void*>(g_recs.size());   // 1-based index
}

UdtProfScope::~UdtProfScope() {
  if (!rec) return;
  std::lock_guard<std::mutex> lk(g_mu);
  const size_t idx = reinterpret_cast<size_t>(rec) - 1;
  if (idx < g_recs.size()) (void)hipEventRecord(g_recs[idx].stop, s);
}

static void fold_records_locked() {
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.stop);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      g_total_ms[r.cls] += (double)ms;
      g_launches[r.cls] += 1;
      if (g_trace_on) {
        TraceRow t; t.cls = r.cls; t.ms = ms; memcpy(t.tag, r.tag, sizeof(t.tag));
        g_trace.push_back(t);
      }
    }
    g_pool.push_back(r);
  }
  g_recs.clear();
}

extern "C" const char* udt_version(void) { return "udt_kernels 0.1.0 (gfx950)"; }

extern "C" const char* udt_status_string(int status) {
  switch (status) {
    case UDT_OK: return "ok";
    case UDT_ERR_BAD_SHAPE: return "unsupported shape / alignment";
    case UDT_ERR_BAD_ARG: return "bad argument (null pointer or inconsistent flags)";
    case UDT_ERR_WORKSPACE: return "workspace too small";
    case UDT_ERR_HIP: return "HIP runtime error (see udt_last_hip_error)";
    case UDT_ERR_NO_DEVICE: return "no gfx950 device";
    case UDT_ERR_ASYNC: return "a stream-K launch timed out waiting for a partner workgroup (not co-resident)";
    default: return "unknown status";
  }
}

extern "C" int udt_last_hip_error(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_last_hip_error;
}

extern "C" int udt_device_arch_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

extern "C" int udt_prof_enable(uint32_t class_mask) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof_mask = class_mask;
  return UDT_OK;
}

extern "C" int udt_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  fold_records_locked();
  for (int i = 0; i < UDT_PROF_NCLASS; ++i) {
    g_total_ms[i] = 0.0;
    g_launches[i] = 0;
  }
  return UDT_OK;
}

extern "C" int udt_prof_trace(int32_t on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_trace_on = on != 0;
  if (!on) g_trace.clear();
  return UDT_OK;
}

extern "C" int udt_prof_dump(const char* path) {
  if (!path) return UDT_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  fold_records_locked();
  FILE* f = fopen(path, "w");
  if (!f) return UDT_ERR_BAD_ARG;
  fprintf(f, "class,ms,tag\n");
  for (auto& t : g_trace) fprintf(f, "%d,%.6f,%s\n", t.cls, t.ms, t.tag);
  fclose(f);
  return UDT_OK;
}

extern "C" int udt_prof_get(int32_t op_class, double* total_ms, int64_t* launches) {
  if (op_class < 0 || op_class >= UDT_PROF_NCLASS || !total_ms || !launches) return UDT_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  fold_records_locked();
  *total_ms = g_total_ms[op_class];
  *launches = g_launches[op_class];
  return UDT_OK;
}
