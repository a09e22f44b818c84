// tools/devel/tm_devel.h — DEVELOPMENT ONLY: included by tokenmonster_amd/csrc/tm_kernels.hip when a tools/ build defines TM_DEVEL
// (phases of k_match_branch can be switched off through tm_debug_flags / TM_DBG: RESULTS ARE WRONG BY DESIGN, only the time counts; tools/pmc_phases.sh,
// tools/variant_ab.sh ablate) or TM_PHASE_TIMERS (per-phase wall cycles of a wavefront, tools/phase_profile.py).  The product build never sees this file.
#pragma once
#ifdef TM_DEVEL
#define TM_DBG_ON(x) (x)
constexpr int kDebugMask = ~0;
#define TM_K1_EXTRA_LDS ((debug_flags() & 512) ? 4096 : 0)
#define TM_DBG_INITIAL (getenv("TM_DBG") ? atoi(getenv("TM_DBG")) : 0)
#else
#define TM_DBG_ON(x) false
#endif
#ifdef TM_PHASE_TIMERS
// development aid (never defined in the product build): per-phase wall cycles of one wavefront, summed over all of them
__device__ unsigned long long g_phase[64 * 32];
#define PH_INIT unsigned long long ph_t = __builtin_readcyclecounter(); unsigned long long ph_a[8] = {0}; int ph_c[16] = {0};
#define PH(i) { const unsigned long long ph_n = __builtin_readcyclecounter(); ph_a[i] += ph_n - ph_t; ph_t = ph_n; }
#define PH_COUNT(i, n) ph_c[i] += (int)(n);
#define PH_INC(i) ph_c[i]++;
#define PH_FLUSH { if (lane == 0) { unsigned long long* gp = g_phase + (blockIdx.x & 63) * 32; for (int q = 0; q < 8; q++) atomicAdd(&gp[q], ph_a[q]); for (int q = 8; q < 16; q++) atomicAdd(&gp[q], (unsigned long long)ph_c[q]); } }
#else
#define PH_INIT
#define PH_INC(i)
#define PH_FLUSH
#define PH(i)
#define PH_COUNT(i, n)
#endif

#ifdef TM_PHASE_TIMERS
#define TM_DEVEL_PHASES_ENTRY \
int tm_debug_phases(unsigned long long* out, int reset) { \
  static unsigned long long z[64 * 32]; \
  if (out) { \
    if (hipMemcpyFromSymbol(z, HIP_SYMBOL(tmh::g_phase), sizeof z) != hipSuccess) return -1; \
    for (int i = 0; i < 32; i++) { out[i] = 0; for (int b = 0; b < 64; b++) out[i] += z[b * 32 + i]; } \
  } \
  if (reset) { for (auto& x : z) x = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(tmh::g_phase), z, sizeof z) != hipSuccess) return -1; } \
  return 0; \
} \

#endif
