// tools/gather_bench.hip — development aid: what a divergent 16-byte gather costs on gfx950, as k_match_branch issues them.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o gpurun_out/gather_bench && gpurun_out/gather_bench
// Every wavefront runs ROUNDS dependent rounds of one global_load_dwordx4 per lane (the next address comes out of the data, as in a trie
// walk); 32 wavefronts per CU, all CUs.  Varied: how many lanes of a wavefront are ACTIVE (the others gather one shared entry, or are masked
// off), the size of the table (L1-resident 16 KiB, L2-resident 3 MiB, 13 MiB = beyond the 4 MiB L2 of an XCD), and a second load per round
// from the same 32 bytes / from another line.  Output: CU clocks per wavefront-round per CU, i.e. the throughput price of one gather
// instruction (2.4 GHz assumed), to be compared with ~64 (one line look-up per clock) and ~16 (four 16-byte lanes per clock).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ROUNDS = 256;

// mode: 0 = idle lanes gather entry 0 (what K1 does), 1 = idle lanes masked off, extra: 0 none, 1 same 32 bytes, 2 another line
template <int MODE, int EXTRA>
__global__ __launch_bounds__(256, 8) void k_gather(const uint4* __restrict__ tab, uint32_t mask, int active, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  uint32_t acc = 0;
  const bool act = lane < active;
  for (int r = 0; r < ROUNDS; r++) {
    const uint32_t a = act ? ((idx & mask) | 1u) : 0u;           // (odd entries are the walk's; entry 0 is the shared idle entry)
    uint4 e = make_uint4(0, 0, 0, 0);
    if (MODE == 0 || act) e = tab[a];
    if (EXTRA == 1) { const uint4 f = tab[a ^ 1u]; acc += f.y; }
    if (EXTRA == 2) { const uint4 f = tab[(a ^ 0x100u)]; acc += f.y; }
    idx = idx * 1664525u + e.x + 1013904223u;                   // depends on the data: the next round cannot start early
    acc += e.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int EXTRA>
static int run(const uint4* d_tab, uint32_t entries, int active, uint32_t* d_out, int n_cu, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = n_cu * 8 * 8;                   // 8 workgroups of 4 wavefronts per CU resident, 8 turns
  k_gather<MODE, EXTRA><<<blocks, 256>>>(d_tab, entries - 1, active, d_out);
  CK(hipEventRecord(e0));
  k_gather<MODE, EXTRA><<<blocks, 256>>>(d_tab, entries - 1, active, d_out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_rounds_per_cu = (double)blocks * 4 * ROUNDS / n_cu;
  const double clocks = ms * 1e-3 * 2.4e9 / wave_rounds_per_cu;
  printf("%-34s table %6.2f MiB  active lanes %2d: %7.3f ms  %6.1f clocks per wavefront-round per CU  (%.2f per active lane)\n", what, entries * 16.0 / 1048576.0, active, ms,
         clocks, clocks / active);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const uint32_t max_entries = 1u << 20;             // 16 MiB
  std::vector<uint4> h(max_entries);
  uint32_t s = 12345;
  for (auto& e : h) { s = s * 1103515245u + 12345u; e = make_uint4(s, s >> 3, s >> 7, s >> 11); }
  uint4* d_tab; uint32_t* d_out;
  CK(hipMalloc(&d_tab, (size_t)max_entries * 16)); CK(hipMalloc(&d_out, 64));
  CK(hipMemcpy(d_tab, h.data(), (size_t)max_entries * 16, hipMemcpyHostToDevice));
  printf("%s, %d CUs\n", prop.name, n_cu);
  for (uint32_t entries : {1u << 10, 1u << 16, 3u << 16, 1u << 20}) {       // 16 KiB, 1 MiB, 3 MiB, 16 MiB
    const uint32_t pow2 = entries & (entries - 1) ? (1u << 17) : entries;     // (mask needs a power of two: 3 MiB -> 2 MiB)
    for (int active : {64, 32, 16, 4}) {
      if (run<0, 0>(d_tab, pow2, active, d_out, n_cu, "idle lanes gather a shared entry")) return 1;
      if (run<1, 0>(d_tab, pow2, active, d_out, n_cu, "idle lanes masked off")) return 1;
    }
    if (run<0, 1>(d_tab, pow2, 64, d_out, n_cu, "+ second load, same 32 bytes")) return 1;
    if (run<0, 2>(d_tab, pow2, 64, d_out, n_cu, "+ second load, another line")) return 1;
  }
  return 0;
}
