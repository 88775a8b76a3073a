// prof.cpp — optional in-situ kernel timing with HIP events on the launch stream.
// When enabled (dzn_profile_enable(1)) every instrumented launch is bracketed by events on the stream it is launched on;
// dzn_profile_collect() synchronises, reads the elapsed times and aggregates them per kernel class together with the
// ALGORITHMIC flops / bytes the launch site declared.  bench.py derives the `roofline` object from these records.
//
// r3: the profiler's own cost mattered at small launch sizes (BASELINE configs[1]: ~1850 launches of ~30 us per step —
// two hipEventRecord per launch plus hipEventCreate on first use were 14 % of the step).  Now
//   * events come from a pool that dzn_profile_reserve() / the previous collect fills OUTSIDE the timed region;
//   * back-to-back launches share their boundary event: the end event of one launch is the start event of the next when
//     the next begin follows within CHAIN_US of host time on the same stream (one record per launch instead of two; the
//     ~1 us inter-kernel gap is then charged to the following kernel — an over-, never an under-estimate of kernel time);
//     a longer host gap (a synchronisation, a copy, the next step) starts a fresh pair;
//   * class names are interned, records are plain structs.
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {
struct Rec {
  int a, b;          // indices into g_events
  int cls;
  double flops, bytes;
};
constexpr double CHAIN_US = 30.0;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_events;   // events in use by the current records
std::vector<hipEvent_t> g_pool;
std::vector<std::string> g_names;
std::unordered_map<std::string, int> g_name_id;
int g_last_event = -1;              // index of the most recent end event
hipStream_t g_last_stream = nullptr;
std::chrono::steady_clock::time_point g_last_time;
std::mutex g_mu;      // (r5) the host stage may run in a second thread beside the engine (pipeline.diarize_many)

int new_event() {
  hipEvent_t e = nullptr;
  if (!g_pool.empty()) {
    e = g_pool.back();
    g_pool.pop_back();
  } else if (hipEventCreate(&e) != hipSuccess) {
    return -1;
  }
  g_events.push_back(e);
  return (int)g_events.size() - 1;
}

int intern(const char* cls) {
  auto it = g_name_id.find(cls);
  if (it != g_name_id.end()) return it->second;
  g_names.emplace_back(cls);
  g_name_id.emplace(g_names.back(), (int)g_names.size() - 1);
  return (int)g_names.size() - 1;
}
}  // namespace

bool prof_enabled() { return g_on; }

int prof_begin(hipStream_t st, const char* cls, double flops, double bytes) {
  if (!g_on) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  const auto now = std::chrono::steady_clock::now();
  const bool chain = g_last_event >= 0 && g_last_stream == st &&
                     std::chrono::duration<double, std::micro>(now - g_last_time).count() < CHAIN_US;
  if (chain) {
    r.a = g_last_event;
  } else {
    r.a = new_event();
    if (r.a < 0) return -1;
    (void)hipEventRecord(g_events[r.a], st);
  }
  r.b = -1;
  r.cls = intern(cls);
  r.flops = flops;
  r.bytes = bytes;
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void prof_end(int id, hipStream_t st) {
  if (id < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (id < 0 || id >= (int)g_recs.size()) return;
  const int b = new_event();
  if (b < 0) return;
  (void)hipEventRecord(g_events[b], st);
  g_recs[id].b = b;
  g_last_event = b;
  g_last_stream = st;
  g_last_time = std::chrono::steady_clock::now();
}

extern "C" int dzn_profile_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  g_last_event = -1;
  return DZN_OK;
}

// pre-create events so that the timed region never calls hipEventCreate
extern "C" int dzn_profile_reserve(int32_t n_events) {
  std::lock_guard<std::mutex> lk(g_mu);
  while ((int)g_pool.size() < n_events) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return DZN_E_HIP;
    g_pool.push_back(e);
  }
  return DZN_OK;
}

extern "C" int dzn_profile_collect(dzn_prof_entry* out, int32_t cap, int32_t* n) {
  if (hipDeviceSynchronize() != hipSuccess) return DZN_E_HIP;
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, dzn_prof_entry> agg;
  for (Rec& r : g_recs) {
    float ms = 0.f;
    if (r.a < 0 || r.b < 0 || hipEventElapsedTime(&ms, g_events[r.a], g_events[r.b]) != hipSuccess) ms = 0.f;
    dzn_prof_entry& e = agg[g_names[r.cls]];
    if (e.launches == 0) {
      memset(&e, 0, sizeof(e));
      strncpy(e.name, g_names[r.cls].c_str(), sizeof(e.name) - 1);
    }
    e.launches += 1;
    e.ms += ms;
    e.flops += r.flops;
    e.bytes += r.bytes;
  }
  g_recs.clear();
  for (hipEvent_t e : g_events) g_pool.push_back(e);
  g_events.clear();
  g_last_event = -1;
  int i = 0;
  for (auto& kv : agg) {
    if (out && i < cap) out[i] = kv.second;
    ++i;
  }
  if (n) *n = i;
  return DZN_OK;
}
