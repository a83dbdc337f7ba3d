// Session plumbing shared by the model families: weight arena, grow-only HBM workspace,
// HIP-event profiler, debug taps.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

// ---- weight arena (layout written by arena.py) -----------------------------------------
//   header  : "ASRARENA" | u32 version | u32 n_tensors | u64 data_offset | u64 total_bytes
//   records : n_tensors x 128 B  { char name[80]; u32 dtype; u32 ndim; i64 shape[4]; u64 offset }
//   data    : tensors, each 256-byte aligned, offsets relative to the start of the arena
enum { ARENA_F32 = 0, ARENA_BF16 = 1, ARENA_I32 = 2, ARENA_F16 = 3 };

struct TensorRef {
  void* ptr = nullptr;
  int dtype = 0;
  int ndim = 0;
  int64_t shape[4] = {0, 0, 0, 0};
  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
  }
};

struct Arena {
  unsigned char* base = nullptr;   // device
  size_t bytes = 0;
  bool owned = false;
  std::map<std::string, TensorRef> tensors;

  void load(const void* src, size_t nbytes, int mem, hipStream_t s);
  void release();
  const TensorRef& get(const std::string& name) const;
  const TensorRef& get(const std::string& name, int dtype, std::initializer_list<int64_t> shape) const;
  bool has(const std::string& name) const { return tensors.count(name) != 0; }
};

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t cap = 0;
  // grow-only; new memory is zero-filled so padded rows/columns are finite
  void reserve(size_t bytes, hipStream_t s);
  void release();
  template <typename T> T* as() const { return reinterpret_cast<T*>(ptr); }
};

struct Profiler {
  bool enabled = false;
  std::vector<std::string> names;
  std::vector<double> total_ms;
  std::vector<int64_t> launches;
  struct Pending { int cls; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;

  int cls(const char* name);
  void begin(int c, hipStream_t s);
  void end(hipStream_t s);
  void collect();          // after the stream is synchronised
  void reset();
  void release();
  hipEvent_t get_event();
};

struct ProfScope {
  Profiler& p; hipStream_t s; bool on;
  ProfScope(Profiler& p_, const char* name, hipStream_t s_) : p(p_), s(s_), on(p_.enabled) { if (on) p.begin(p.cls(name), s); }
  ~ProfScope() { if (on) p.end(s); }
};

struct Tap {
  DeviceBuffer buf;
  int64_t rows = 0, cols = 0;
  int elt = 4;
};

struct asr_session {
  int kind = 0;               // 1 = sensevoice, 2 = whisper
  int device = 0;
  int tenant_slot = -1;       // this session's entry in the process-wide tenancy table (asr_tenant_attach; released by the destructor)
  int precision = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Arena arena;
  Profiler prof;
  bool taps_enabled = false;
  std::map<std::string, Tap> taps;
  virtual ~asr_session();
  void save_tap(const char* name, const void* src, int64_t rows, int64_t cols, int64_t ld_src, int elt);
};

template <typename F>
int asr_guard(F&& f) {
  try {
    f();
    return ASR_OK;
  } catch (const AsrError& e) {
    asr_set_error(e.msg);
    return e.code;
  } catch (const std::exception& e) {
    asr_set_error(std::string("internal error: ") + e.what());
    return ASR_ERR_INVALID;
  }
}

void asr_require_device(int device_id);

// ---- who else uses this GPU (process-wide). The cluster kernels of the streaming path (and the block kernel) hold every CU for a whole launch and wait on
// sibling workgroups; next to another session's kernels that costs the other tenant its CUs (Qwen3-ASR 3 415 -> 319 audio-s/s beside 1 024 cluster
// workgroups) and can void the launch. Every session registers at creation; every compute entry point holds a TenantScope while it runs.
void asr_tenant_attach(asr_session* s);                                         // after s->device is set
struct TenantScope { asr_session* s; explicit TenantScope(asr_session* s_); ~TenantScope(); };
int asr_tenant_live_others(const asr_session* s);                              // other sessions that exist on s->device
int asr_tenant_busy_others(const asr_session* s, double window_ms);            // ... that are inside a compute call now, or left one less than window_ms ago

// ---- foreign kernels on this GPU: RCCL collectives (or anything else the host program launches outside this library) must never share the chip with a
// cluster kernel (sanm_block8 / sanm_tiles / stream_layers / stream_dec: every workgroup of the grid must be resident and they spin on each other's counters --
// an RCCL kernel parked on a few CUs is exactly the co-tenant that splits a cluster). The rule, process-wide and per device:
//   * asr_device_foreign_begin(dev) (C ABI) opens a foreign section and BLOCKS until no cluster pass is in flight on `dev`; asr_device_foreign_end(dev) closes it
//     (the caller makes sure its kernels have finished by then);
//   * a compute call that would launch cluster kernels asks ClusterScope first: while a foreign section is open it takes its cluster-free path instead
//     (four launches per SANM block, per-layer launches of the streaming step -- same results), it never waits, so the two sides cannot deadlock.
struct ClusterScope {
  int dev; bool ok;                 // ok: cluster kernels may be launched until this scope ends (the call returns with its stream drained)
  explicit ClusterScope(int device);
  ~ClusterScope();
};
