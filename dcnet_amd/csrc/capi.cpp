// Error reporting, version and the optional HIP-event kernel profiler of libdcnet_hip.so.
#include "common.h"
#include "prof.h"
#include <string.h>
#include <vector>

static thread_local char g_err[512] = "";

void dcn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dcn_last_error(void) { return g_err; }
extern "C" int dcn_version(void) { return DCN_ABI_VERSION; }

// ---- streams ---------------------------------------------------------------------------------
// level: -1 = highest, 0 = normal, +1 = lowest priority the device offers.  The weight-gradient GEMMs run
// on a lowest-priority stream: their workgroups are dispatched into whatever the data-gradient chain on
// the caller's stream leaves idle (tile-quantisation tails) instead of competing with it.
extern "C" void* dcn_stream_create(int level) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { dcn_set_error("stream_create: no priority range"); return nullptr; }
  const int prio = level < 0 ? greatest : (level > 0 ? least : (least + greatest) / 2);
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio) != hipSuccess) { dcn_set_error("stream_create: hipStreamCreateWithPriority failed"); return nullptr; }
  return (void*)s;
}

extern "C" int dcn_stream_destroy(void* stream) {
  DCN_CHECK_ARG(stream, "stream_destroy: null stream");
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? DCN_OK : DCN_ERR_LAUNCH;
}

extern "C" int dcn_stream_priority_range(int* least, int* greatest) {
  DCN_CHECK_ARG(least && greatest, "stream_priority_range: null pointer");
  return hipDeviceGetStreamPriorityRange(least, greatest) == hipSuccess ? DCN_OK : DCN_ERR_LAUNCH;
}

// ---- profiler ---------------------------------------------------------------------------------
namespace {
struct Rec { hipEvent_t a, b; int tag; double work, bytes; };
std::vector<Rec> g_recs;
size_t g_used = 0;
bool g_on = false;
constexpr size_t kMaxRecs = 1 << 16;
}  // namespace

int prof_begin(int tag, double work, hipStream_t s, double bytes) {
  if (!g_on || g_used >= kMaxRecs) return -1;
  if (g_used == g_recs.size()) {
    Rec r{};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
    g_recs.push_back(r);
  }
  Rec& r = g_recs[g_used];
  r.tag = tag; r.work = work; r.bytes = bytes;
  (void)hipEventRecord(r.a, s);
  return (int)g_used++;
}

void prof_end(int id, hipStream_t s) {
  if (id >= 0) (void)hipEventRecord(g_recs[id].b, s);
}

// on != 0: start a fresh recording window; on == 0: stop recording (records are kept for collect)
extern "C" int dcn_prof_enable(int on) {
  if (on) g_used = 0;
  g_on = on != 0;
  return DCN_OK;
}

// Host-synchronising: waits for every recorded event, then sums per tag.  counts/ms/work: [DCN_PROF_TAGS].
extern "C" int dcn_prof_collect(int64_t* counts, double* ms, double* work, double* bytes) {
  DCN_CHECK_ARG(counts && ms && work, "prof_collect: null pointer");
  for (int t = 0; t < DCN_PROF_TAGS; ++t) { counts[t] = 0; ms[t] = 0.0; work[t] = 0.0; if (bytes) bytes[t] = 0.0; }
  for (size_t i = 0; i < g_used; ++i) {
    Rec& r = g_recs[i];
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    if (r.tag >= 0 && r.tag < DCN_PROF_TAGS) { counts[r.tag]++; ms[r.tag] += t; work[r.tag] += r.work; if (bytes) bytes[r.tag] += r.bytes; }
  }
  return DCN_OK;
}

// Per-launch records of the last window (host-synchronising): tag / milliseconds / algorithmic work / algorithmic bytes of up to
// `max` launches in launch order.  Returns the number of records written (or a negative error).  Lets the caller price every
// LAUNCH against the roofline that binds it (max of its MFMA time and its HBM time) instead of a per-tag average.
extern "C" int dcn_prof_records(int32_t* tags, double* ms, double* work, double* bytes, int max) {
  DCN_CHECK_ARG(tags && ms && work && bytes && max >= 0, "prof_records: null pointer");
  int n = 0;
  for (size_t i = 0; i < g_used && n < max; ++i) {
    Rec& r = g_recs[i];
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    tags[n] = r.tag; ms[n] = t; work[n] = r.work; bytes[n] = r.bytes; ++n;
  }
  return n;
}
