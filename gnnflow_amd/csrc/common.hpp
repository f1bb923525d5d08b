// Shared host-side plumbing of the HIP library: error propagation to the C ABI,
// RAII device buffers, kernel-time accounting.  (The reference aborts on any
// failed CHECK / CUDA_CALL, gnnflow/csrc/logging.h:9-46; here every failure is an
// exception that the C boundary turns into a GF_ERR_* code.)
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <utility>
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
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

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

// A device buffer that grows by reallocation while it is small and IN PLACE once it is large:
// from kInPlaceBytes on, a range of virtual addresses is reserved (hipMemAddressReserve) and
// physical memory is mapped behind what is already there as the buffer grows (hipMemCreate /
// hipMemMap, 1 GiB pieces) — no reallocation, no copy of the old contents, no second copy of
// the buffer alive while it grows, and the address no longer changes.  Growing a multi-GB
// hipMalloc'd pool by reallocation was measured at 120-530 ms per step on the MAG-shaped build
// (8.6 -> 17 GB) and needs old + new at once.  Small buffers stay on hipMalloc: with the pools
// of the 21 MB REDDIT-shaped graph mapped this way the hash-partitioned replay ran at 76 us
// per step instead of 52 (unexplained; sampling on the multi-GB graphs is unaffected).
// GNNFLOW_VMM_POOLS=0: always reallocate.
class GrowBuffer {
 public:
  static constexpr size_t kInPlaceBytes = size_t(1) << 30;
  static constexpr size_t kPiece = size_t(1) << 30;

  GrowBuffer() = default;
  GrowBuffer(const GrowBuffer&) = delete;
  GrowBuffer& operator=(const GrowBuffer&) = delete;
  ~GrowBuffer() { release(); }

  // max_bytes: the most it may ever be asked to hold (the reservation costs nothing)
  void init(size_t max_bytes, int device) {
    release();
    max_bytes_ = max_bytes;
    device_ = device;
  }

  void* data() const { return vmm_ ? static_cast<void*>(base_) : fallback_.data(); }
  template <typename T> T* as() const { return reinterpret_cast<T*>(data()); }
  size_t bytes() const { return vmm_ ? mapped_ : fallback_.bytes(); }
  bool in_place() const { return vmm_; }

  // makes [0, n) usable; the first `keep` bytes keep their contents
  void reserve(size_t n, size_t keep, hipStream_t stream) {
    if (!vmm_) {
      if (n <= fallback_.bytes()) return;
      if (n < in_place_bytes() || !start_in_place()) {
        fallback_.reserve(n, keep, stream);
        return;
      }
      // the move into the reserved range: the one copy this buffer will ever see again
      map_up_to(n);
      if (keep && fallback_.data()) {
        GF_HIP(hipMemcpyAsync(base_, fallback_.data(), keep, hipMemcpyDeviceToDevice, stream));
        GF_HIP(hipStreamSynchronize(stream));
      }
      fallback_.release();
      return;
    }
    if (n > mapped_) map_up_to(n);
  }

  void release() {
    if (vmm_) {
      size_t off = 0;
      for (auto& h : chunks_) {
        (void)hipMemUnmap(base_ + off, kPiece);
        (void)hipMemRelease(h);
        off += kPiece;
      }
      chunks_.clear();
      if (base_) (void)hipMemAddressFree(base_, va_);
    }
    fallback_.release();
    vmm_ = false;
    base_ = nullptr;
    va_ = mapped_ = 0;
  }

 private:
  static size_t in_place_bytes() {
    static const size_t v = [] {
      const char* e = std::getenv("GNNFLOW_VMM_MIN_BYTES");   // experiments / tests
      return e ? static_cast<size_t>(std::atoll(e)) : kInPlaceBytes;
    }();
    return v;
  }

  bool start_in_place() {
    static const bool enabled = [] {
      const char* v = std::getenv("GNNFLOW_VMM_POOLS");
      return !(v && std::atoi(v) == 0);
    }();
    if (!enabled || max_bytes_ < in_place_bytes()) return false;
    va_ = align_up(max_bytes_, kPiece);
    void* base = nullptr;
    if (hipMemAddressReserve(&base, va_, kPiece, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      va_ = 0;
      return false;
    }
    base_ = static_cast<char*>(base);
    vmm_ = true;
    return true;
  }

  // Pieces of ONE size: with pieces of different sizes in one reservation hipMemSetAccess
  // fails with "invalid argument" on ROCm 7.2 (scripts/micro/vmm_probe2.hip).
  void map_up_to(size_t n) {
    if (n > va_)
      throw Error(GF_ERR_OUT_OF_MEMORY, "device pool: more than the reserved address range");
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device_;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t goal = align_up(n, kPiece);
    while (mapped_ < goal) {
      hipMemGenericAllocationHandle_t h;
      GF_HIP(hipMemCreate(&h, kPiece, &prop, 0));
      GF_HIP(hipMemMap(base_ + mapped_, kPiece, 0, h, 0));
      GF_HIP(hipMemSetAccess(base_ + mapped_, kPiece, &acc, 1));
      chunks_.push_back(h);
      mapped_ += kPiece;
    }
  }

  bool vmm_ = false;
  char* base_ = nullptr;
  size_t va_ = 0, mapped_ = 0, max_bytes_ = 0;
  int device_ = 0;
  std::vector<hipMemGenericAllocationHandle_t> chunks_;
  DeviceBuffer fallback_;
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


}  // namespace gf
