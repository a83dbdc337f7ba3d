// Session plumbing: error channel, arena parsing, workspace, profiler, taps.
#include "engine.h"

#include <cstring>

static thread_local std::string g_last_error;
void asr_set_error(const std::string& msg) { g_last_error = msg; }
const std::string& asr_get_error() { return g_last_error; }

void asr_require_device(int device_id) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw AsrError{ASR_ERR_NO_DEVICE, "no HIP device visible: the MI355X engine has no CPU fallback"};
  if (device_id < 0 || device_id >= n)
    throw AsrError{ASR_ERR_INVALID, "device_id " + std::to_string(device_id) + " out of range (" + std::to_string(n) + " devices)"};
  HIP_CHECK(hipSetDevice(device_id));
}

// ---------------------------------------------------------------------------------------- Arena
namespace {
struct ArenaHeader {
  char magic[8];
  uint32_t version, n_tensors;
  uint64_t data_offset, total_bytes;
};
struct ArenaRecord {
  char name[80];
  uint32_t dtype, ndim;
  int64_t shape[4];
  uint64_t offset;
};
static_assert(sizeof(ArenaHeader) == 32, "arena header layout");
static_assert(sizeof(ArenaRecord) == 128, "arena record layout");
}  // namespace

void Arena::load(const void* src, size_t nbytes, int mem, hipStream_t s) {
  ASR_REQUIRE(src && nbytes >= sizeof(ArenaHeader), "arena: empty");
  ArenaHeader hdr;
  if (mem == 0) {
    memcpy(&hdr, src, sizeof(hdr));
  } else {
    HIP_CHECK(hipMemcpy(&hdr, src, sizeof(hdr), hipMemcpyDeviceToHost));
  }
  ASR_REQUIRE(memcmp(hdr.magic, "ASRARENA", 8) == 0, "arena: bad magic");
  ASR_REQUIRE(hdr.version == 1, "arena: unsupported version %u", hdr.version);
  ASR_REQUIRE(hdr.total_bytes == nbytes, "arena: size mismatch (header %llu, given %llu)",
              (unsigned long long)hdr.total_bytes, (unsigned long long)nbytes);
  const size_t table = (size_t)hdr.n_tensors * sizeof(ArenaRecord);
  ASR_REQUIRE(sizeof(hdr) + table <= hdr.data_offset && hdr.data_offset <= nbytes, "arena: corrupt manifest");
  std::vector<ArenaRecord> recs(hdr.n_tensors);
  if (mem == 0) {
    memcpy(recs.data(), (const unsigned char*)src + sizeof(hdr), table);
    HIP_CHECK(hipMalloc((void**)&base, nbytes));
    owned = true;
    HIP_CHECK(hipMemcpyAsync(base, src, nbytes, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
  } else {
    HIP_CHECK(hipMemcpy(recs.data(), (const unsigned char*)src + sizeof(hdr), table, hipMemcpyDeviceToHost));
    base = (unsigned char*)const_cast<void*>(src);
    owned = false;
  }
  bytes = nbytes;
  static const int elt[4] = {4, 2, 4, 2};
  for (const auto& r : recs) {
    ASR_REQUIRE(r.dtype < 4 && r.ndim <= 4, "arena: bad record");
    TensorRef t;
    t.dtype = (int)r.dtype;
    t.ndim = (int)r.ndim;
    for (int i = 0; i < 4; ++i) t.shape[i] = r.shape[i];
    ASR_REQUIRE(r.offset % 256 == 0 && r.offset + (uint64_t)t.numel() * elt[r.dtype] <= nbytes, "arena: tensor out of range");
    t.ptr = base + r.offset;
    char nm[81];
    memcpy(nm, r.name, 80);
    nm[80] = 0;
    tensors[nm] = t;
  }
}

void Arena::release() {
  if (owned && base) (void)hipFree(base);
  base = nullptr;
  tensors.clear();
}

const TensorRef& Arena::get(const std::string& name) const {
  auto it = tensors.find(name);
  if (it == tensors.end()) throw AsrError{ASR_ERR_NOT_FOUND, "arena: tensor '" + name + "' missing"};
  return it->second;
}

const TensorRef& Arena::get(const std::string& name, int dtype, std::initializer_list<int64_t> shape) const {
  const TensorRef& t = get(name);
  bool ok = t.dtype == dtype && t.ndim == (int)shape.size();
  int i = 0;
  for (int64_t d : shape) { if (ok && t.shape[i] != d) ok = false; ++i; }
  if (!ok) {
    std::string want, got;
    for (int64_t d : shape) want += std::to_string(d) + ",";
    for (int j = 0; j < t.ndim; ++j) got += std::to_string(t.shape[j]) + ",";
    throw AsrError{ASR_ERR_INVALID, "arena: tensor '" + name + "' has dtype " + std::to_string(t.dtype) + " shape (" + got +
                                        "), expected dtype " + std::to_string(dtype) + " shape (" + want + ")"};
  }
  return t;
}

// ---------------------------------------------------------------------------------------- DeviceBuffer
void DeviceBuffer::reserve(size_t bytes, hipStream_t s) {
  if (bytes <= cap) return;
  if (ptr) {
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipFree(ptr));
    ptr = nullptr;
    cap = 0;
  }
  const size_t want = (bytes + 255) & ~(size_t)255;
  HIP_CHECK(hipMalloc(&ptr, want));
  HIP_CHECK(hipMemsetAsync(ptr, 0, want, s));
  cap = want;
}

void DeviceBuffer::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  cap = 0;
}

// ---------------------------------------------------------------------------------------- Profiler
int Profiler::cls(const char* name) {
  for (size_t i = 0; i < names.size(); ++i)
    if (names[i] == name) return (int)i;
  names.push_back(name);
  total_ms.push_back(0.0);
  launches.push_back(0);
  return (int)names.size() - 1;
}

hipEvent_t Profiler::get_event() {
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}

void Profiler::begin(int c, hipStream_t s) {
  Pending p{c, get_event(), get_event()};
  HIP_CHECK(hipEventRecord(p.a, s));
  pending.push_back(p);
}

void Profiler::end(hipStream_t s) { HIP_CHECK(hipEventRecord(pending.back().b, s)); }

void Profiler::collect() {
  for (auto& p : pending) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
    total_ms[p.cls] += ms;
    launches[p.cls] += 1;
    pool.push_back(p.a);
    pool.push_back(p.b);
  }
  pending.clear();
}

void Profiler::reset() {
  for (auto& v : total_ms) v = 0.0;
  for (auto& v : launches) v = 0;
}

void Profiler::release() {
  for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : pool) (void)hipEventDestroy(e);
  pending.clear();
  pool.clear();
}

// ---------------------------------------------------------------------------------------- taps
void asr_session::save_tap(const char* name, const void* src, int64_t rows, int64_t cols, int64_t ld_src, int elt) {
  if (!taps_enabled) return;
  Tap& t = taps[name];
  t.rows = rows;
  t.cols = cols;
  t.elt = elt;
  t.buf.reserve((size_t)rows * cols * elt, stream);
  HIP_CHECK(hipMemcpy2DAsync(t.buf.ptr, (size_t)cols * elt, src, (size_t)ld_src * elt, (size_t)cols * elt, (size_t)rows,
                             hipMemcpyDeviceToDevice, stream));
}

// ---- tenancy table (engine.h)
#include <atomic>
#include <chrono>
namespace {
constexpr int TENANT_SLOTS = 512;
struct TenantSlot { std::atomic<int> used{0}, device{-1}, busy{0}; std::atomic<int64_t> last_ns{0}; };
TenantSlot g_tenants[TENANT_SLOTS];
int64_t tenant_now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

void asr_tenant_attach(asr_session* s) {
  if (!s || s->tenant_slot >= 0) return;
  for (int i = 0; i < TENANT_SLOTS; ++i) {
    int expect = 0;
    if (g_tenants[i].used.compare_exchange_strong(expect, 1)) {
      g_tenants[i].device.store(s->device); g_tenants[i].busy.store(0); g_tenants[i].last_ns.store(0);
      s->tenant_slot = i;
      return;
    }
  }          // (table full: the session stays anonymous -- it is then never counted, which only ever errs towards the faster path)
}
asr_session::~asr_session() {
  if (tenant_slot >= 0) { g_tenants[tenant_slot].device.store(-1); g_tenants[tenant_slot].used.store(0); tenant_slot = -1; }
}
TenantScope::TenantScope(asr_session* s_) : s(s_) {
  if (s && s->tenant_slot >= 0) { g_tenants[s->tenant_slot].device.store(s->device); g_tenants[s->tenant_slot].busy.fetch_add(1); }
}
TenantScope::~TenantScope() {
  if (s && s->tenant_slot >= 0) { g_tenants[s->tenant_slot].last_ns.store(tenant_now_ns()); g_tenants[s->tenant_slot].busy.fetch_sub(1); }
}
int asr_tenant_live_others(const asr_session* s) {
  int n = 0;
  for (int i = 0; i < TENANT_SLOTS; ++i)
    if (i != s->tenant_slot && g_tenants[i].used.load() && g_tenants[i].device.load() == s->device) ++n;
  return n;
}
int asr_tenant_busy_others(const asr_session* s, double window_ms) {
  const int64_t now = tenant_now_ns(), win = (int64_t)(window_ms * 1e6);
  int n = 0;
  for (int i = 0; i < TENANT_SLOTS; ++i) {
    if (i == s->tenant_slot || !g_tenants[i].used.load() || g_tenants[i].device.load() != s->device) continue;
    const int64_t last = g_tenants[i].last_ns.load();
    if (g_tenants[i].busy.load() > 0 || (last != 0 && now - last < win)) ++n;
  }
  return n;
}

// ---- foreign-kernel gate (engine.h)
#include <condition_variable>
#include <mutex>
namespace {
constexpr int GATE_DEVS = 64;
std::mutex g_gate_mu;
std::condition_variable g_gate_cv;
int g_foreign[GATE_DEVS], g_cluster[GATE_DEVS];
int64_t g_gate_stats[GATE_DEVS][4];              // foreign sections opened, cluster passes diverted, sections that had to wait, cluster passes admitted
inline int gate_slot(int device) { return device >= 0 && device < GATE_DEVS ? device : GATE_DEVS - 1; }
}  // namespace

ClusterScope::ClusterScope(int device) : dev(gate_slot(device)), ok(false) {
  std::lock_guard<std::mutex> lk(g_gate_mu);
  if (g_foreign[dev] > 0) { ++g_gate_stats[dev][1]; return; }
  ++g_cluster[dev]; ++g_gate_stats[dev][3];
  ok = true;
}
ClusterScope::~ClusterScope() {
  if (!ok) return;
  { std::lock_guard<std::mutex> lk(g_gate_mu); --g_cluster[dev]; }
  g_gate_cv.notify_all();
}

extern "C" int asr_device_foreign_begin(int device_id) {
  return asr_guard([&] {
    ASR_REQUIRE(device_id >= 0 && device_id < GATE_DEVS, "device_foreign_begin: device %d", device_id);
    std::unique_lock<std::mutex> lk(g_gate_mu);
    ++g_foreign[device_id]; ++g_gate_stats[device_id][0];
    if (g_cluster[device_id] > 0) ++g_gate_stats[device_id][2];
    g_gate_cv.wait(lk, [&] { return g_cluster[device_id] == 0; });
  });
}
extern "C" int asr_device_foreign_end(int device_id) {
  return asr_guard([&] {
    ASR_REQUIRE(device_id >= 0 && device_id < GATE_DEVS, "device_foreign_end: device %d", device_id);
    std::lock_guard<std::mutex> lk(g_gate_mu);
    ASR_REQUIRE(g_foreign[device_id] > 0, "device_foreign_end: no foreign section is open on device %d", device_id);
    --g_foreign[device_id];
  });
}
extern "C" int asr_device_foreign_stats(int device_id, int64_t* out4) {
  return asr_guard([&] {
    ASR_REQUIRE(device_id >= 0 && device_id < GATE_DEVS && out4, "device_foreign_stats: bad argument");
    std::lock_guard<std::mutex> lk(g_gate_mu);
    for (int i = 0; i < 4; ++i) out4[i] = g_gate_stats[device_id][i];
  });
}
