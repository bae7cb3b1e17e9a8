// Handle lifecycle, error reporting, memory and timing helpers of the C-ABI.
#include "td_common.h"

#include <algorithm>
#include <mutex>
#include <unordered_map>

thread_local std::string td_global_error;

int td_fail(td_handle* h, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  td_global_error = buf;
  if (h) h->error = buf;
  return code;
}

namespace {
std::mutex g_handles_mu;
std::vector<td_handle*> g_handles;      // live handles of the process (td_create / td_destroy)
}

int td_order_after_others(td_handle* h) {
  std::lock_guard<std::mutex> lock(g_handles_mu);
  for (td_handle* o : g_handles) {
    if (o == h || o->device != h->device || o->stream == h->stream || !o->order_event) continue;
    // nothing outstanding there: nothing to order after (a record + wait on an idle CU-masked
    // stream cost ~50 us of stream time each: its queue has to be scheduled to signal the event)
    if (hipStreamQuery(o->stream) == hipSuccess) continue;
    (void)hipGetLastError();                 // (hipErrorNotReady is the other expected answer)
    TD_HIP(h, hipEventRecord(o->order_event, o->stream));
    TD_HIP(h, hipStreamWaitEvent(h->stream, o->order_event, 0));
  }
  return TD_OK;
}

// The pool behind td_alloc_async / td_free_async: freed blocks wait, with an event that marks the
// point in stream order after which they are free, for the next request of the same size.  (HIP's
// own stream-ordered allocator was tried first: hipFreeAsync took 0.3 ms a call on this stack --
// 21 ms for the 67 statistics objects of a leave-one-out sweep -- where an event record takes
// microseconds.)
namespace {
struct PoolBlock {
  void* p;
  size_t bytes;
  hipEvent_t ev;
  int device;
};
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool;                       // free blocks, most recently freed last
std::unordered_map<int, std::vector<hipEvent_t>> g_pool_events;   // spare event objects, per device
std::unordered_map<void*, size_t> g_pool_sizes;      // live blocks handed out by td_alloc_async
size_t g_pool_bytes = 0;
constexpr size_t kPoolMaxBytes = (size_t)4 << 30;
constexpr size_t kPoolMaxBlocks = 4096;

void pool_drain_locked() {
  for (PoolBlock& b : g_pool) {
    (void)hipSetDevice(b.device);
    hipEventSynchronize(b.ev);
    hipFree(b.p);
    hipEventDestroy(b.ev);
  }
  g_pool.clear();
  g_pool_bytes = 0;
  for (auto& kv : g_pool_events)
    for (hipEvent_t e : kv.second) hipEventDestroy(e);
  g_pool_events.clear();
}
}  // namespace

int td_alloc_async(td_handle* h, size_t bytes, void** out) {
  *out = nullptr;
  bytes = (size_t)td_round_up((int64_t)(bytes ? bytes : 1), 256);
  // (a process may hold handles on several devices: the allocation, and the events of this device's
  // blocks, belong to h->device, not to whatever device happens to be current)
  TD_HIP(h, hipSetDevice(h->device));
  {
    std::unique_lock<std::mutex> lock(g_pool_mu);
    for (size_t i = g_pool.size(); i-- > 0;) {
      if (g_pool[i].device != h->device || g_pool[i].bytes != bytes) continue;
      const PoolBlock b = g_pool[i];
      // the block is free from the point its last owner recorded; this stream starts after it
      // (the event object may be recorded again later: a wait refers to the record it saw)
      const hipError_t e = hipStreamWaitEvent(h->stream, b.ev, 0);
      if (e != hipSuccess) {               // the block stays pooled
        (void)hipGetLastError();
        return td_fail(h, TD_ERR_HIP, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
      }
      g_pool.erase(g_pool.begin() + (long)i);
      g_pool_bytes -= bytes;
      g_pool_sizes[b.p] = bytes;
      g_pool_events[b.device].push_back(b.ev);
      *out = b.p;
      return TD_OK;
    }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {
    // memory held by the pool may be what is missing
    std::unique_lock<std::mutex> lock(g_pool_mu);
    pool_drain_locked();
    lock.unlock();
    (void)hipGetLastError();
    TD_HIP(h, hipSetDevice(h->device));
    e = hipMalloc(out, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return td_fail(h, TD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  std::lock_guard<std::mutex> lock(g_pool_mu);
  g_pool_sizes[*out] = bytes;
  return TD_OK;
}

int td_free_async(td_handle* h, void* p, bool own_stream_only) {
  if (!p) return TD_OK;
  TD_HIP(h, hipSetDevice(h->device));
  if (!own_stream_only) TD_TRY(td_order_after_others(h));
  std::unique_lock<std::mutex> lock(g_pool_mu);
  const auto it = g_pool_sizes.find(p);
  if (it == g_pool_sizes.end()) {                    // not from td_alloc_async
    lock.unlock();
    TD_HIP(h, hipStreamSynchronize(h->stream));
    TD_HIP(h, hipFree(p));
    return TD_OK;
  }
  PoolBlock b;
  b.p = p; b.bytes = it->second; b.device = h->device; b.ev = nullptr;
  g_pool_sizes.erase(it);
  // From here on the block is neither pooled nor owned by the caller: whatever fails below, it goes
  // back to the driver the slow way (wait for the stream, hipFree) and is never lost.
  std::vector<hipEvent_t>& spare = g_pool_events[h->device];       // events are per device
  hipError_t e = hipSuccess;
  if (!spare.empty()) {
    b.ev = spare.back();
    spare.pop_back();
  } else {
    e = hipEventCreateWithFlags(&b.ev, hipEventDisableTiming);
    if (e != hipSuccess) b.ev = nullptr;
  }
  if (e == hipSuccess) e = hipEventRecord(b.ev, h->stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (b.ev) spare.push_back(b.ev);
    lock.unlock();
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(p);
    return td_fail(h, TD_ERR_HIP, "td_free_async: event record failed (%s); the block was freed synchronously",
                   hipGetErrorString(e));
  }
  g_pool.push_back(b);
  g_pool_bytes += b.bytes;
  // over the limits: the oldest blocks go back to the driver (this waits for them)
  while (g_pool_bytes > kPoolMaxBytes || g_pool.size() > kPoolMaxBlocks) {
    const PoolBlock old = g_pool.front();
    g_pool.erase(g_pool.begin());
    g_pool_bytes -= old.bytes;
    hipEventSynchronize(old.ev);
    hipFree(old.p);
    g_pool_events[old.device].push_back(old.ev);
  }
  return TD_OK;
}

int td_scratch(td_handle* h, size_t bytes, void** out) {
  if (bytes > h->scratch_bytes) {
    if (h->scratch) {
      TD_HIP(h, hipStreamSynchronize(h->stream));
      TD_HIP(h, hipFree(h->scratch));
      h->scratch = nullptr;
      h->scratch_bytes = 0;
    }
    size_t want = bytes + bytes / 4;
    hipError_t e = hipMalloc(&h->scratch, want);
    if (e != hipSuccess) {
      want = bytes;
      e = hipMalloc(&h->scratch, want);
    }
    if (e != hipSuccess)
      return td_fail(h, TD_ERR_NOMEM, "scratch allocation of %zu bytes failed: %s", bytes,
                     hipGetErrorString(e));
    h->scratch_bytes = want;
  }
  *out = h->scratch;
  return TD_OK;
}

int td_workspace(td_handle* h, size_t bytes, void** out) {
  if (bytes > h->work_bytes) {
    if (h->work) {
      TD_HIP(h, hipStreamSynchronize(h->stream));
      TD_HIP(h, hipFree(h->work));
      h->work = nullptr;
      h->work_bytes = 0;
    }
    hipError_t e = hipMalloc(&h->work, bytes);
    if (e != hipSuccess)
      return td_fail(h, TD_ERR_NOMEM, "workspace allocation of %zu bytes failed: %s", bytes,
                     hipGetErrorString(e));
    h->work_bytes = bytes;
  }
  *out = h->work;
  return TD_OK;
}

int td_upload_async(td_handle* h, const void* host, size_t bytes, void* dev_dst) {
  if (bytes == 0) return TD_OK;
  td_handle::PinSlot& slot = h->pin[h->pin_next];
  h->pin_next = (h->pin_next + 1) % td_handle::kPinSlots;
  if (!slot.ev) TD_HIP(h, hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
  if (slot.used) TD_HIP(h, hipEventSynchronize(slot.ev));
  if (bytes > slot.bytes) {
    if (slot.p) TD_HIP(h, hipHostFree(slot.p));
    slot.p = nullptr;
    slot.bytes = 0;
    // generous first size: growing a slot later means hipHostFree + hipHostMalloc, which
    // wait for the device -- with 16 slots and a 94 KB work table rotating through them that
    // stalled the host behind the running accumulate kernel on most calls
    const size_t want = bytes * 2 < (256u << 10) ? (256u << 10) : bytes * 2;
    TD_HIP(h, hipHostMalloc(&slot.p, want, hipHostMallocDefault));
    slot.bytes = want;
  }
  memcpy(slot.p, host, bytes);
  TD_HIP(h, hipMemcpyAsync(dev_dst, slot.p, bytes, hipMemcpyHostToDevice, h->stream));
  TD_HIP(h, hipEventRecord(slot.ev, h->stream));
  slot.used = true;
  return TD_OK;
}

int td_table_upload(td_handle* h, const void* host, size_t bytes, const void** dev) {
  *dev = nullptr;
  if (bytes == 0) return TD_OK;
  td_handle::TableSlot* lru = &h->tables[0];
  for (auto& slot : h->tables) {
    if (slot.host.size() == bytes && memcmp(slot.host.data(), host, bytes) == 0) {
      slot.stamp = ++h->table_clock;
      *dev = slot.dev;
      return TD_OK;
    }
    if (slot.stamp < lru->stamp) lru = &slot;
  }
  if (lru->cap < bytes) {
    // (hipFree waits for the device: kernels still reading the old block finish first)
    if (lru->dev) TD_HIP(h, hipFree(lru->dev));
    lru->dev = nullptr; lru->cap = 0; lru->host.clear();
    const size_t want = td_round_up(bytes * 2 < 4096 ? 4096 : bytes * 2, 256);
    TD_HIP(h, hipMalloc(&lru->dev, want));
    lru->cap = want;
  }
  // overwriting a block that earlier kernels of this stream read is safe in stream order
  lru->host.clear();                       // not a valid cache entry if the upload fails
  TD_TRY(td_upload_async(h, host, bytes, lru->dev));
  lru->host.assign(reinterpret_cast<const char*>(host), reinterpret_cast<const char*>(host) + bytes);
  lru->stamp = ++h->table_clock;
  *dev = lru->dev;
  return TD_OK;
}

int td_profile_mark(td_handle* h, bool start, double samples) {
  if (!h->profile) return TD_OK;
  if (h->prof_used == h->prof_events.size()) {
    hipEvent_t ev = nullptr;
    TD_HIP(h, hipEventCreate(&ev));
    h->prof_events.push_back(ev);
  }
  TD_HIP(h, hipEventRecord(h->prof_events[h->prof_used++], h->stream));
  if (start) h->prof_samples += samples;
  return TD_OK;
}

extern "C" {

int td_version(void) { return 1; }

int td_profile_enable(td_handle* h, int on) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "handle is NULL");
  TD_HIP(h, hipStreamSynchronize(h->stream));
  h->profile = on != 0;
  h->prof_used = 0;
  h->prof_samples = 0;
  return TD_OK;
}

int td_profile_read(td_handle* h, int64_t* launches, double* total_ms, double* samples) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "handle is NULL");
  TD_HIP(h, hipStreamSynchronize(h->stream));
  double ms = 0.0;
  for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
    float t = 0.f;
    TD_HIP(h, hipEventElapsedTime(&t, h->prof_events[i], h->prof_events[i + 1]));
    ms += t;
  }
  if (launches) *launches = (int64_t)(h->prof_used / 2);
  if (total_ms) *total_ms = ms;
  if (samples) *samples = h->prof_samples;
  h->prof_used = 0;
  h->prof_samples = 0;
  return TD_OK;
}

int td_device_count(int* count) {
  if (!count) return td_fail(nullptr, TD_ERR_INVALID, "count is NULL");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return td_fail(nullptr, TD_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  return TD_OK;
}

int td_create(int device_id, td_handle** out) {
  if (!out) return td_fail(nullptr, TD_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return td_fail(nullptr, TD_ERR_HIP,
                   "no HIP device available (%s): the MI355X hot path cannot run on CPU",
                   e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device_id < 0 || device_id >= n)
    return td_fail(nullptr, TD_ERR_INVALID, "device_id %d out of range [0, %d)", device_id, n);
  td_handle* h = new td_handle();
  h->device = device_id;
  TD_HIP(h, hipSetDevice(device_id));
  hipDeviceProp_t prop;
  TD_HIP(h, hipGetDeviceProperties(&prop, device_id));
  h->cu_count = prop.multiProcessorCount;
  TD_HIP(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  TD_HIP(h, hipEventCreate(&h->ev_start));
  TD_HIP(h, hipEventCreate(&h->ev_stop));
  TD_HIP(h, hipEventCreateWithFlags(&h->order_event, hipEventDisableTiming));
  TD_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->dev_flag), sizeof(int) * 64));
  TD_HIP(h, hipMemset(h->dev_flag, 0, sizeof(int) * 64));
  {
    std::lock_guard<std::mutex> lock(g_handles_mu);
    g_handles.push_back(h);
  }
  *out = h;
  return TD_OK;
}

int td_destroy(td_handle* h) {
  if (!h) return TD_OK;
  {
    std::lock_guard<std::mutex> lock(g_handles_mu);
    g_handles.erase(std::remove(g_handles.begin(), g_handles.end(), h), g_handles.end());
  }
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  bool last = false;
  {
    std::lock_guard<std::mutex> lock(g_handles_mu);
    last = g_handles.empty();
  }
  if (last) {                      // the pool of td_alloc_async goes with the last handle
    std::lock_guard<std::mutex> lock(g_pool_mu);
    pool_drain_locked();
  }
  if (h->scratch) hipFree(h->scratch);
  if (h->work) hipFree(h->work);
  for (auto& slot : h->pin) {
    if (slot.p) hipHostFree(slot.p);
    if (slot.ev) hipEventDestroy(slot.ev);
  }
  for (auto& slot : h->tables)
    if (slot.dev) hipFree(slot.dev);
  if (h->dev_flag) hipFree(h->dev_flag);
  if (h->chan_max) hipFree(h->chan_max);
  if (h->dev_flags) hipFree(h->dev_flags);
  if (h->cg_packets) hipFree(h->cg_packets);
  for (hipEvent_t e : h->async_events) if (e) hipEventDestroy(e);
  if (h->cg_status) hipFree(h->cg_status);
  if (h->host_flags) hipHostFree(h->host_flags);
  if (h->ev_start) hipEventDestroy(h->ev_start);
  if (h->ev_stop) hipEventDestroy(h->ev_stop);
  if (h->order_event) hipEventDestroy(h->order_event);
  if (h->own_stream) hipStreamDestroy(h->own_stream);
  delete h;
  return TD_OK;
}

int td_set_cu_count(td_handle* h, int cu_count) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "td_set_cu_count: NULL handle");
  TD_REQUIRE(h, cu_count > 0, "td_set_cu_count: %d CUs", cu_count);
  h->cu_count = cu_count;
  return TD_OK;
}

int td_set_accumulate_mode(td_handle* h, int mode) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "td_set_accumulate_mode: NULL handle");
  TD_REQUIRE(h, mode >= TD_ACC_F16X2 && mode <= TD_ACC_F32, "td_set_accumulate_mode: unknown mode %d", mode);
  h->acc_mode = mode;
  return TD_OK;
}

const char* td_last_error(const td_handle* h) {
  return h ? h->error.c_str() : td_global_error.c_str();
}

int td_stream_create_masked(int device_id, int cu_first, int cu_count, void** stream_out) {
  if (!stream_out) return td_fail(nullptr, TD_ERR_INVALID, "td_stream_create_masked: NULL output");
  *stream_out = nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess)
    return td_fail(nullptr, TD_ERR_HIP, "td_stream_create_masked: no device %d", device_id);
  const int n_cu = prop.multiProcessorCount;
  if (cu_first < 0 || cu_count <= 0 || cu_first + cu_count > n_cu)
    return td_fail(nullptr, TD_ERR_INVALID, "CU range [%d, %d) outside the device's %d CUs", cu_first,
                   cu_first + cu_count, n_cu);
  std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
  for (int cu = cu_first; cu < cu_first + cu_count; ++cu) mask[cu >> 5] |= 1u << (cu & 31);
  if (hipSetDevice(device_id) != hipSuccess)
    return td_fail(nullptr, TD_ERR_HIP, "hipSetDevice(%d) failed", device_id);
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess)
    return td_fail(nullptr, TD_ERR_HIP, "hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
  *stream_out = st;
  return TD_OK;
}

int td_stream_destroy(void* hip_stream) {
  if (!hip_stream) return TD_OK;
  hipStreamSynchronize(reinterpret_cast<hipStream_t>(hip_stream));
  return hipStreamDestroy(reinterpret_cast<hipStream_t>(hip_stream)) == hipSuccess
             ? TD_OK : td_fail(nullptr, TD_ERR_HIP, "hipStreamDestroy failed");
}

int td_set_stream(td_handle* h, void* hip_stream) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "handle is NULL");
  if (h->stream == reinterpret_cast<hipStream_t>(hip_stream)) return TD_OK;
  TD_HIP(h, hipStreamSynchronize(h->stream));
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);   // NULL is HIP's default stream
  return TD_OK;
}

int td_use_own_stream(td_handle* h) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "handle is NULL");
  TD_HIP(h, hipStreamSynchronize(h->stream));
  h->stream = h->own_stream;
  return TD_OK;
}

int td_synchronize(td_handle* h) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "handle is NULL");
  TD_HIP(h, hipStreamSynchronize(h->stream));
  return TD_OK;
}

int td_malloc(td_handle* h, size_t bytes, void** dev_ptr) {
  if (!h || !dev_ptr) return td_fail(h, TD_ERR_INVALID, "td_malloc: NULL argument");
  *dev_ptr = nullptr;
  if (bytes == 0) return TD_OK;
  hipError_t e = hipMalloc(dev_ptr, bytes);
  if (e != hipSuccess)
    return td_fail(h, TD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  return TD_OK;
}

int td_free(td_handle* h, void* dev_ptr) {
  if (!dev_ptr) return TD_OK;
  if (h) TD_HIP(h, hipStreamSynchronize(h->stream));
  TD_HIP(h, hipFree(dev_ptr));
  return TD_OK;
}

int td_memcpy_h2d(td_handle* h, void* dst_dev, const void* src_host, size_t bytes) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "handle is NULL");
  if (bytes == 0) return TD_OK;
  TD_HIP(h, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, h->stream));
  TD_HIP(h, hipStreamSynchronize(h->stream));  // host buffer is the caller's: do not outlive the call
  return TD_OK;
}

int td_memcpy_d2h(td_handle* h, void* dst_host, const void* src_dev, size_t bytes) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "handle is NULL");
  if (bytes == 0) return TD_OK;
  TD_HIP(h, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
  TD_HIP(h, hipStreamSynchronize(h->stream));
  return TD_OK;
}

int td_memset(td_handle* h, void* dst_dev, int value, size_t bytes) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "handle is NULL");
  if (bytes == 0) return TD_OK;
  TD_HIP(h, hipMemsetAsync(dst_dev, value, bytes, h->stream));
  return TD_OK;
}

int td_timer_start(td_handle* h) {
  if (!h) return td_fail(nullptr, TD_ERR_INVALID, "handle is NULL");
  TD_HIP(h, hipEventRecord(h->ev_start, h->stream));
  return TD_OK;
}

int td_timer_stop(td_handle* h, float* elapsed_ms) {
  if (!h || !elapsed_ms) return td_fail(h, TD_ERR_INVALID, "td_timer_stop: NULL argument");
  TD_HIP(h, hipEventRecord(h->ev_stop, h->stream));
  TD_HIP(h, hipEventSynchronize(h->ev_stop));
  TD_HIP(h, hipEventElapsedTime(elapsed_ms, h->ev_start, h->ev_stop));
  return TD_OK;
}

}  // extern "C"
