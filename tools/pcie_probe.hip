// tools/pcie_probe.hip — development aid (round 6): what the host link of the box gives, so that the host-to-host rate of tm_tokenize_pipeline
// can be priced against it.  hipcc -O3 --offload-arch=gfx950 tools/pcie_probe.hip -o gpurun_out/pcie_probe && gpurun_out/pcie_probe
//   (1) copy engine H2D / D2H of page-locked memory, one stream, chunk size swept      (2) both directions at once
//   (3) a KERNEL reading page-locked host memory (16-byte loads) into HBM, grid swept   (4) a kernel writing HBM -> host
//   (5) the copy engine's H2D while a memory-bound kernel keeps the CUs busy
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
// four loads in flight per lane before the first store
__global__ void k_copy16x4(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride],
          d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
__global__ void k_busy(uint4* __restrict__ a, uint64_t n16, int rounds) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (int r = 0; r < rounds; r++)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { uint4 x = a[i]; x.x += 1; a[i] = x; }
}

int main() {
  const uint64_t N = 1ull << 30, NOUT = 560ull << 20;
  uint8_t *h_in, *h_out, *d_in, *d_out, *d_busy;
  CK(hipHostMalloc((void**)&h_in, N, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&h_out, N, hipHostMallocDefault));
  CK(hipMalloc((void**)&d_in, N));
  CK(hipMalloc((void**)&d_out, N));
  CK(hipMalloc((void**)&d_busy, 2ull << 30));
  for (uint64_t i = 0; i < N; i += 4096) { h_in[i] = (uint8_t)i; h_out[i] = 0; }
  hipStream_t s0, s1, s2;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto run = [&](const char* what, uint64_t bytes, auto&& body) {
    double best = 1e9, sum = 0;
    for (int r = 0; r < 6; r++) {
      CK(hipDeviceSynchronize());
      const double t0 = now_ms();
      body();
      CK(hipDeviceSynchronize());
      const double dt = now_ms() - t0;
      if (r) { sum += dt; if (dt < best) best = dt; }
    }
    printf("%-78s best %7.2f ms = %6.2f GB/s   mean %7.2f ms\n", what, best, bytes / best / 1e6, sum / 5);
    fflush(stdout);
  };
  char name[200];
  for (uint64_t chunk : {2ull << 20, 8ull << 20, 32ull << 20, 128ull << 20, 1024ull << 20}) {
    snprintf(name, sizeof name, "(1) H2D 1 GiB pinned, copy engine, one stream, %4llu MiB chunks", (unsigned long long)(chunk >> 20));
    run(name, N, [&] { for (uint64_t o = 0; o < N; o += chunk) CK(hipMemcpyAsync(d_in + o, h_in + o, chunk, hipMemcpyHostToDevice, s0)); });
  }
  run("(1) H2D 1 GiB pinned, copy engine, two streams alternating, 32 MiB chunks", N, [&] {
    int k = 0;
    for (uint64_t o = 0; o < N; o += 32ull << 20, k++) CK(hipMemcpyAsync(d_in + o, h_in + o, 32ull << 20, hipMemcpyHostToDevice, k & 1 ? s1 : s0));
  });
  run("(1) H2D 1 GiB pinned, copy engine, four streams alternating, 32 MiB chunks", N, [&] {
    static hipStream_t s3 = nullptr;
    if (!s3) CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    hipStream_t ss[4] = {s0, s1, s2, s3};
    int k = 0;
    for (uint64_t o = 0; o < N; o += 32ull << 20, k++) CK(hipMemcpyAsync(d_in + o, h_in + o, 32ull << 20, hipMemcpyHostToDevice, ss[k & 3]));
  });
  for (uint64_t chunk : {2ull << 20, 16ull << 20, 560ull << 20}) {
    snprintf(name, sizeof name, "(1) D2H 560 MiB pinned, copy engine, one stream, %4llu MiB chunks", (unsigned long long)(chunk >> 20));
    run(name, NOUT, [&] { for (uint64_t o = 0; o < NOUT; o += chunk) CK(hipMemcpyAsync(h_out + o, d_out + o, chunk, hipMemcpyDeviceToHost, s1)); });
  }
  run("(2) H2D 1 GiB (32 MiB chunks) + D2H 560 MiB (17.5 MiB chunks) at once; rate of the H2D bytes", N, [&] {
    for (uint64_t k = 0; k < 32; k++) {
      CK(hipMemcpyAsync(d_in + (k << 25), h_in + (k << 25), 32ull << 20, hipMemcpyHostToDevice, s0));
      CK(hipMemcpyAsync(h_out + k * (NOUT / 32), d_out + k * (NOUT / 32), NOUT / 32, hipMemcpyDeviceToHost, s1));
    }
  });
  for (int grid : {64, 256, 1024, 4096}) {
    snprintf(name, sizeof name, "(3) kernel reads host memory -> HBM, 1 GiB, 16-byte loads, %4d x 256 threads", grid);
    run(name, N, [&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, s0, (const uint4*)h_in, (uint4*)d_in, N / 16); });
    snprintf(name, sizeof name, "(3) the same, four loads in flight per lane,                  %4d x 256 threads", grid);
    run(name, N, [&] { hipLaunchKernelGGL(k_copy16x4, dim3(grid), dim3(256), 0, s0, (const uint4*)h_in, (uint4*)d_in, N / 16); });
  }
  run("(3) kernel reads host memory, 32 launches of 32 MiB on one stream (256 x 256)", N, [&] {
    for (uint64_t k = 0; k < 32; k++) hipLaunchKernelGGL(k_copy16x4, dim3(256), dim3(256), 0, s0, (const uint4*)(h_in + (k << 25)), (uint4*)(d_in + (k << 25)), (32ull << 20) / 16);
  });
  for (int grid : {256, 4096}) {
    snprintf(name, sizeof name, "(4) kernel writes HBM -> host memory, 560 MiB, 16-byte stores, %4d x 256 threads", grid);
    run(name, NOUT, [&] { hipLaunchKernelGGL(k_copy16x4, dim3(grid), dim3(256), 0, s0, (const uint4*)d_out, (uint4*)h_out, NOUT / 16); });
  }
  run("(3+1) kernel reads 1 GiB host -> HBM (256 x 256) while the copy engine takes 560 MiB D2H", N, [&] {
    hipLaunchKernelGGL(k_copy16x4, dim3(256), dim3(256), 0, s0, (const uint4*)h_in, (uint4*)d_in, N / 16);
    for (uint64_t k = 0; k < 32; k++) CK(hipMemcpyAsync(h_out + k * (NOUT / 32), d_out + k * (NOUT / 32), NOUT / 32, hipMemcpyDeviceToHost, s1));
  });
  // (5) a memory-bound kernel over 2 GiB keeps every CU busy for ~25 ms while the copy engine works
  {
    CK(hipDeviceSynchronize());
    double t0 = now_ms();
    hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, s2, (uint4*)d_busy, (2ull << 30) / 16, 24);
    CK(hipDeviceSynchronize());
    printf("(5) busy kernel alone: %.2f ms\n", now_ms() - t0);
  }
  run("(5) H2D 1 GiB (32 MiB chunks, copy engine) under the busy kernel: time of BOTH", N, [&] {
    hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, s2, (uint4*)d_busy, (2ull << 30) / 16, 24);
    for (uint64_t o = 0; o < N; o += 32ull << 20) CK(hipMemcpyAsync(d_in + o, h_in + o, 32ull << 20, hipMemcpyHostToDevice, s0));
  });
  {
    // the H2D's own time under the busy kernel
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < 3; r++) {
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, s2, (uint4*)d_busy, (2ull << 30) / 16, 24);
      CK(hipEventRecord(a, s0));
      for (uint64_t o = 0; o < N; o += 32ull << 20) CK(hipMemcpyAsync(d_in + o, h_in + o, 32ull << 20, hipMemcpyHostToDevice, s0));
      CK(hipEventRecord(b, s0));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      printf("(5) H2D 1 GiB under the busy kernel, by events: %.2f ms = %.2f GB/s\n", ms, N / ms / 1e6);
    }
    for (int r = 0; r < 3; r++) {
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, s2, (uint4*)d_busy, (2ull << 30) / 16, 24);
      CK(hipEventRecord(a, s0));
      hipLaunchKernelGGL(k_copy16x4, dim3(256), dim3(256), 0, s0, (const uint4*)h_in, (uint4*)d_in, N / 16);
      CK(hipEventRecord(b, s0));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      printf("(5) kernel read of 1 GiB host memory under the busy kernel, by events: %.2f ms = %.2f GB/s\n", ms, N / ms / 1e6);
    }
  }
  return 0;
}
