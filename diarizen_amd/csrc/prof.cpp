// prof.cpp — optional in-situ kernel timing with HIP events on the launch stream.
// When enabled (dzn_profile_enable(1)) every instrumented launch is bracketed by two events on
// the stream it is launched on; dzn_profile_collect() synchronises, reads the elapsed times and
// aggregates them per kernel class together with the ALGORITHMIC flops / bytes the launch site
// declared.  bench.py derives the `roofline` object from these records.
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace {
struct Rec {
  hipEvent_t a, b;
  std::string cls;
  double flops, bytes;
};
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

bool prof_enabled() { return g_on; }

int prof_begin(hipStream_t st, const char* cls, double flops, double bytes) {
  if (!g_on) return -1;
  Rec r;
  r.a = get_event();
  r.b = get_event();
  if (!r.a || !r.b) return -1;
  r.cls = cls;
  r.flops = flops;
  r.bytes = bytes;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void prof_end(int id, hipStream_t st) {
  if (id < 0 || id >= (int)g_recs.size()) return;
  (void)hipEventRecord(g_recs[id].b, st);
}

extern "C" int dzn_profile_enable(int32_t on) {
  g_on = on != 0;
  return DZN_OK;
}

extern "C" int dzn_profile_collect(dzn_prof_entry* out, int32_t cap, int32_t* n) {
  if (hipDeviceSynchronize() != hipSuccess) return DZN_E_HIP;
  std::map<std::string, dzn_prof_entry> agg;
  for (Rec& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0.f;
    dzn_prof_entry& e = agg[r.cls];
    if (e.launches == 0) {
      memset(&e, 0, sizeof(e));
      strncpy(e.name, r.cls.c_str(), sizeof(e.name) - 1);
    }
    e.launches += 1;
    e.ms += ms;
    e.flops += r.flops;
    e.bytes += r.bytes;
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  int i = 0;
  for (auto& kv : agg) {
    if (out && i < cap) out[i] = kv.second;
    ++i;
  }
  if (n) *n = i;
  return DZN_OK;
}
