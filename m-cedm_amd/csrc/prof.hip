// prof.hip -- event-pair profiler behind mcedm_prof_enable / mcedm_prof_report.
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "prof.hpp"

namespace mcedm {

struct Rec { std::string name; double flops, bytes; hipEvent_t a, b; };
static bool g_on = false;
static std::vector<Rec> g_recs;
static std::vector<hipEvent_t> g_pool;
static std::mutex g_mu;

bool prof_enabled() { return g_on; }

static thread_local const KernelVariants* t_variants = nullptr;
const KernelVariants* current_variants() { return t_variants; }
VariantScope::VariantScope(const KernelVariants* kv) : prev(t_variants) { t_variants = kv; }
VariantScope::~VariantScope() { t_variants = prev; }

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

int prof_begin(const char* name, double flops, double bytes, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r{name, flops, bytes, get_event(), get_event()};
  if (!r.a || !r.b) return -1;
  if (hipEventRecord(r.a, s) != hipSuccess) return -1;
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void prof_end(int idx, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx >= 0 && idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].b, s);
}

}  // namespace mcedm

using namespace mcedm;

extern "C" int mcedm_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  return MCEDM_OK;
}

// Waits for every recorded event, aggregates per kernel name, clears the records and writes a JSON array
// [{"name":..., "launches":n, "total_ms":t, "flops":f, "bytes":b}, ...] (flops/bytes are sums) into buf.
extern "C" int mcedm_prof_report(char* buf, size_t buflen) {
  MCEDM_REQUIRE(buf && buflen > 2, "prof_report: bad buffer");
  std::lock_guard<std::mutex> lk(g_mu);
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (Rec& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      Agg& a = agg[r.name];
      a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  std::string out = "[";
  bool first = true;
  for (auto& kv : agg) {
    char line[768];
    snprintf(line, sizeof(line), "%s{\"name\": \"%s\", \"launches\": %ld, \"total_ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
             first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
    out += line;
    first = false;
  }
  out += "]";
  if (out.size() + 1 > buflen) { set_error("prof_report: buffer too small (%zu needed)", out.size() + 1); return MCEDM_ERR_WORKSPACE; }
  memcpy(buf, out.c_str(), out.size() + 1);
  return MCEDM_OK;
}
