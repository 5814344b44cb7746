// prof.hpp -- optional per-launch timing with HIP events on the launch stream (bench.py roofline leg).
// Off by default; when off a ProfScope costs one predictable branch.
#pragma once
#include "common.hpp"

namespace mcedm {

bool prof_enabled();
int prof_begin(const char* name, double flops, double bytes, hipStream_t s);   // returns record index or -1
void prof_end(int idx, hipStream_t s);

struct ProfScope {
  int idx;
  hipStream_t s;
  ProfScope(const char* name, double flops, double bytes, hipStream_t stream)
      : idx(prof_enabled() ? prof_begin(name, flops, bytes, stream) : -1), s(stream) {}
  ~ProfScope() { if (idx >= 0) prof_end(idx, s); }
};

}  // namespace mcedm
