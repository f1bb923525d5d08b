// Shared host-side plumbing of the HIP library: error propagation to the C ABI,
// RAII device buffers, kernel-time accounting.  (The reference aborts on any
// failed CHECK / CUDA_CALL, gnnflow/csrc/logging.h:9-46; here every failure is an
// exception that the C boundary turns into a GF_ERR_* code.)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gnnflow_hip.h"

namespace gf {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

void set_last_error(const std::string& msg);

#define GF_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      throw ::gf::Error(_e == hipErrorOutOfMemory ? GF_ERR_OUT_OF_MEMORY          \
                                                  : GF_ERR_HIP,                   \
                        std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + \
                            __FILE__ + ":" + std::to_string(__LINE__) + ")");     \
    }                                                                             \
  } while (0)

#define GF_REQUIRE(cond, msg)                                            \
  do {                                                                   \
    if (!(cond)) throw ::gf::Error(GF_ERR_INVALID_ARGUMENT, (msg));      \
  } while (0)

// Runs `fn`, mapping exceptions to status codes + gf_last_error().
template <typename Fn>
int guarded(Fn&& fn) noexcept {
  try {
    fn();
    return GF_OK;
  } catch (const Error& e) {
    set_last_error(e.what());
    return e.code;
  } catch (const std::bad_alloc&) {
    set_last_error("host allocation failed");
    return GF_ERR_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return GF_ERR_INVALID_ARGUMENT;
  }
}

// Grow-only device allocation.  grow(n, keep, stream) reallocates to >= n bytes
// (geometric) and, if keep > 0, copies the first `keep` bytes device-to-device.
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : ptr_(o.ptr_), bytes_(o.bytes_) {
    o.ptr_ = nullptr;
    o.bytes_ = 0;
  }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) {
      release();
      ptr_ = o.ptr_;
      bytes_ = o.bytes_;
      o.ptr_ = nullptr;
      o.bytes_ = 0;
    }
    return *this;
  }
  ~DeviceBuffer() { release(); }

  void* data() const { return ptr_; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(ptr_); }
  size_t bytes() const { return bytes_; }

  // returns true when the buffer was reallocated
  bool reserve(size_t n, size_t keep = 0, hipStream_t stream = nullptr,
               bool zero_new = false) {
    if (n <= bytes_) return false;
    size_t cap = bytes_ ? bytes_ : 256;
    while (cap < n) cap *= 2;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) {
      // retry with the exact size before giving up
      cap = n;
      GF_HIP(hipMalloc(&p, cap));
    }
    if (keep > 0 && ptr_) {
      GF_HIP(hipMemcpyAsync(p, ptr_, keep, hipMemcpyDeviceToDevice, stream));
    }
    if (zero_new && cap > keep) {
      GF_HIP(hipMemsetAsync(static_cast<char*>(p) + keep, 0, cap - keep, stream));
    }
    if (ptr_) {
      GF_HIP(hipStreamSynchronize(stream));
      (void)hipFree(ptr_);
    }
    ptr_ = p;
    bytes_ = cap;
    return true;
  }

  void release() {
    if (ptr_) (void)hipFree(ptr_);
    ptr_ = nullptr;
    bytes_ = 0;
  }

 private:
  void* ptr_ = nullptr;
  size_t bytes_ = 0;
};

// Device buffers that were replaced while kernels queued on a stream may still use them:
// each waits behind an event recorded on that stream and is freed by a later collect().
class RetiredBuffers {
 public:
  RetiredBuffers() = default;
  RetiredBuffers(const RetiredBuffers&) = delete;
  RetiredBuffers& operator=(const RetiredBuffers&) = delete;
  ~RetiredBuffers() {
    for (Item& it : items_) {
      (void)hipEventSynchronize(it.done);
      (void)hipEventDestroy(it.done);
    }
  }
  void retire(DeviceBuffer&& b, hipStream_t stream) {
    if (!b.data()) return;
    Item it;
    it.buf = std::move(b);
    GF_HIP(hipEventCreateWithFlags(&it.done, hipEventDisableTiming));
    GF_HIP(hipEventRecord(it.done, stream));
    items_.push_back(std::move(it));
  }
  void collect() {
    for (size_t i = 0; i < items_.size();) {
      if (hipEventQuery(items_[i].done) == hipSuccess) {
        (void)hipEventDestroy(items_[i].done);
        items_[i] = std::move(items_.back());
        items_.pop_back();
      } else {
        ++i;
      }
    }
  }

 private:
  struct Item { DeviceBuffer buf; hipEvent_t done = nullptr; };
  std::vector<Item> items_;
};

// Pinned host staging buffer (grow-only).
class PinnedBuffer {
 public:
  PinnedBuffer() = default;
  PinnedBuffer(const PinnedBuffer&) = delete;
  PinnedBuffer& operator=(const PinnedBuffer&) = delete;
  ~PinnedBuffer() { if (ptr_) (void)hipHostFree(ptr_); }
  void* data() const { return ptr_; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(ptr_); }
  void reserve(size_t n) {
    if (n <= bytes_) return;
    size_t cap = bytes_ ? bytes_ : 4096;
    while (cap < n) cap *= 2;
    void* p = nullptr;
    GF_HIP(hipHostMalloc(&p, cap, hipHostMallocDefault));
    if (ptr_) (void)hipHostFree(ptr_);
    ptr_ = p;
    bytes_ = cap;
  }
 private:
  void* ptr_ = nullptr;
  size_t bytes_ = 0;
};

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    GF_HIP(hipGetDevice(&prev));
    if (prev != dev) GF_HIP(hipSetDevice(dev));
    else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- kernel time accounting (bench.py roofline; include/gnnflow_hip.h) --------
enum ProfileSlot { kProfSearch = 0, kProfEmit = 1, kProfGather = 2, kProfScan = 3,
                   kProfLru = 4, kProfSlots = 5 };
bool profile_enabled();
// Brackets one kernel launch with HIP events on `stream` when profiling is on.
struct ProfileScope {
  int slot;
  hipStream_t stream;
  hipEvent_t start = nullptr;
  ProfileScope(int slot, hipStream_t stream);
  ~ProfileScope();
};

// Kernel-attached timing: profile_begin() says whether this launch of `slot` is to be timed
// and hands out two events for hipExtLaunchKernelGGL(start, stop) — they carry the
// dispatch's own begin / end timestamps, so the interval is the kernel's execution time
// (what rocprofv3 reports) without the stream-marker overhead of events recorded around it;
// profile_end() queues the pair for gf_profile_get.
bool profile_begin(int slot, hipEvent_t* start, hipEvent_t* stop);
void profile_end(int slot, hipEvent_t start, hipEvent_t stop);

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace gf
