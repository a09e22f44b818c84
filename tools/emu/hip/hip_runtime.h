// tools/emu/hip/hip_runtime.h — a stand-in for <hip/hip_runtime.h> that compiles the kernel sources of libtokenmonster_hip.so for the
// HOST, so that their LOGIC can be exercised where there is no GPU (this container; the CPU half of the test suite).
//
// DEVELOPMENT AID / TEST INFRASTRUCTURE.  Nothing in the product reaches this: tokenmonster_amd/_native.py loads
// libtokenmonster_hip.so and only that; tools/emu/build_emu.py compiles the same .hip files a second time, against this header, into
// tools/emu/libtokenmonster_emu.so, which only the emulation leg of the tests loads (tests/conftest.py under TM_EMU=1).  It says
// nothing about speed and nothing about what the gfx950 compiler makes of the code: parity claims rest on the -m gpu tests alone.
//
// Execution model: a launch runs its workgroups one after the other on the calling thread (one launch at a time, process wide).  Every
// work-item of a workgroup is a FIBER with its own stack; fibers run until they reach a point where the hardware would have made the
// lanes meet — a wave-level operation (__ballot, __shfl, __any, readfirstlane, wave_barrier) or __syncthreads — and wait there until
// all live work-items of the wavefront (or the workgroup) have arrived.  Between two such points a fiber runs alone, so anything a
// kernel gets right only because its lanes run in lockstep WITHOUT saying so (no barrier, no cross-lane operation) shows up here as a
// wrong result, and a cross-lane operation that not all live lanes of a wavefront reach (divergent control flow) as a reported deadlock.
// Atomics are plain read-modify-writes (one fiber runs at a time).  "Device" memory is host memory with a guard page behind every
// allocation; streams and events complete immediately.
#pragma once
#define TM_EMU 1

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>

// ---- qualifiers -------------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) (&(x))

// ---- vector types -----------------------------------------------------------------------------------------------------------------
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
  dim3(int x_) : x((unsigned)x_), y(1), z(1) {}
  dim3(unsigned long x_) : x((unsigned)x_), y(1), z(1) {}
  dim3(unsigned long long x_) : x((unsigned)x_), y(1), z(1) {}
};

// ---- the fiber scheduler (emu_runtime.cpp) --------------------------------------------------------------------------------------
namespace emu {
struct Idx { unsigned x, y, z; };
struct Wave {
  unsigned live = 0, arrived = 0;
  unsigned long long gen = 0, live_mask = 0;
  unsigned long long ballot[2] = {0, 0};
  unsigned long long xch[2][64];
};
struct Group { unsigned live = 0, arrived = 0; unsigned long long gen = 0; };
struct Ctx {
  Idx tid, bid, bdim, gdim;
  unsigned lane;
  Wave* wave;
  Group* group;
};
extern thread_local Ctx* cur;
void wave_sync();                 // all live lanes of the wavefront meet
void group_sync();                // __syncthreads
unsigned long long ballot(bool p);
unsigned long long exchange(unsigned long long v, unsigned src);     // value of lane src & 63
unsigned long long first_lane(unsigned long long v);                 // value of the lowest live lane
struct Launch { virtual void run() = 0; virtual ~Launch() {} };
void launch_impl(dim3 grid, dim3 block, Launch& l);
template <class F>
inline void launch(dim3 grid, dim3 block, F f) {
  struct L : Launch { F f; explicit L(F g) : f(g) {} void run() override { f(); } } l(f);
  launch_impl(grid, block, l);
}
// LDS addresses (tm_device.h: TM_LDS_ADDR / TM_LDS_PTR): offsets from a base that a kernel may pin with TM_LDS_OBJECTS
extern thread_local uintptr_t lds_base;
void lds_objects(const void* a, size_t na, const void* b, size_t nb);
inline uint32_t lds_addr(const void* p) { return (uint32_t)((uintptr_t)p - lds_base); }
inline void* lds_ptr(uint32_t a) { return (void*)(lds_base + (uintptr_t)a); }
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)

// ---- cross-lane and synchronisation --------------------------------------------------------------------------------------------
static inline void __syncthreads() { emu::group_sync(); }
#define __builtin_amdgcn_wave_barrier() emu::wave_sync()
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
static inline unsigned long long __ballot(int p) { return emu::ballot(p != 0); }
#define __builtin_amdgcn_ballot_w64(p) emu::ballot(p)
static inline int __any(int p) { return emu::ballot(p != 0) != 0ull; }
static inline int __all(int p) { return emu::ballot(p == 0) == 0ull; }
template <class T>
static inline T __shfl(T v, int src) {
  static_assert(sizeof(T) <= 8 && std::is_trivially_copyable<T>::value, "__shfl: at most 8 bytes");
  unsigned long long u = 0;
  std::memcpy(&u, &v, sizeof(T));
  u = emu::exchange(u, (unsigned)src);
  T r;
  std::memcpy(&r, &u, sizeof(T));
  return r;
}
template <class T>
static inline T __shfl_xor(T v, int mask) { return __shfl(v, (int)(emu::cur->lane ^ (unsigned)mask)); }
template <class T>
static inline T emu_readfirstlane(T v) {
  unsigned long long u = 0;
  std::memcpy(&u, &v, sizeof(T));
  u = emu::first_lane(u);
  T r;
  std::memcpy(&r, &u, sizeof(T));
  return r;
}
#define __builtin_amdgcn_readfirstlane(x) emu_readfirstlane(x)
static inline uint32_t emu_mbcnt(uint32_t mask, uint32_t acc, unsigned lo) {     // bits of mask below the lane, in the low / high half
  const unsigned lane = emu::cur->lane;
  uint32_t below;
  if (lo) below = lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u);
  else below = lane <= 32 ? 0u : ((1u << (lane - 32)) - 1u);
  return acc + (uint32_t)__builtin_popcount(mask & below);
}
#define __builtin_amdgcn_mbcnt_lo(mask, acc) emu_mbcnt(mask, acc, 1)
#define __builtin_amdgcn_mbcnt_hi(mask, acc) emu_mbcnt(mask, acc, 0)
static inline uint32_t emu_udot4(uint32_t a, uint32_t b, uint32_t c) {
  return c + (a & 255u) * (b & 255u) + ((a >> 8) & 255u) * ((b >> 8) & 255u) + ((a >> 16) & 255u) * ((b >> 16) & 255u) + (a >> 24) * (b >> 24);
}
#define __builtin_amdgcn_udot4(a, b, c, clamp) emu_udot4(a, b, c)

// ---- integer helpers ------------------------------------------------------------------------------------------------------------
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return x ? __builtin_ctzll(x) + 1 : 0; }
static inline int __ffs(unsigned x) { return x ? __builtin_ctz(x) + 1 : 0; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

// ---- atomics: one fiber runs at a time ---------------------------------------------------------------------------------------
template <class T, class U> static inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
template <class T, class U, class V> static inline T atomicCAS(T* p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
static inline void __threadfence() {}
static inline void __threadfence_system() {}

// ---- the runtime API the library uses (emu_runtime.cpp) ----------------------------------------------------------------------
typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorOutOfMemory = 2, hipErrorInsufficientDriver = 35, hipErrorNoDevice = 100,
               hipErrorInvalidDevice = 101 } hipError_t;
typedef struct emuStream* hipStream_t;
typedef struct emuEvent* hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 } hipMemoryType;
typedef struct { hipMemoryType type; int device; void* devicePointer; void* hostPointer; } hipPointerAttribute_t;
typedef enum { hipDeviceAttributeMultiprocessorCount = 0 } hipDeviceAttribute_t;
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostRegisterDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };

hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceGetPCIBusId(char* bdf, int len, int dev);
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int dev);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void** p, size_t n);
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) { return hipHostMalloc((void**)p, n, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipHostRegister(void* p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemcpyPeer(void* dst, int dst_dev, const void* src, int src_dev, size_t n);
hipError_t hipMemcpyPeerAsync(void* dst, int dst_dev, const void* src, int src_dev, size_t n, hipStream_t st = nullptr);
hipError_t hipDeviceCanAccessPeer(int* can, int dev, int peer);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipMemset(void* p, int v, size_t n);
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipMemcpyToSymbol(void* sym, const void* src, size_t n);
hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t n);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
