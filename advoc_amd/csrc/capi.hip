// Library-level entry points of libadvoc_hip.so (version / error strings).
#include "common.h"

namespace advoc {
static thread_local hipError_t g_last_hip_error = hipSuccess;
void note_hip_error(hipError_t e) { g_last_hip_error = e; }
}  // namespace advoc

extern "C" const char* advoc_last_hip_error(void) {
  return hipGetErrorString(advoc::g_last_hip_error);
}

extern "C" int advoc_abi_version(void) { return ADVOC_ABI_VERSION; }

extern "C" const char* advoc_target_arch(void) { return "gfx950"; }

extern "C" const char* advoc_error_string(int code) {
  switch (code) {
    case ADVOC_OK: return "ok";
    case ADVOC_ERR_BAD_SHAPE: return "bad shape: inconsistent or non-positive dimensions";
    case ADVOC_ERR_UNSUPPORTED: return "unsupported: outside what the gfx950 kernels implement";
    case ADVOC_ERR_HIP: return "HIP runtime error (launch failed)";
    case ADVOC_ERR_NULL: return "null pointer";
    default: return "unknown advoc error code";
  }
}

// ---- diagnostic switches: environment read once, re-read on request (tuning.h) ----
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "tuning.h"

namespace advoc {
namespace {
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
Tuning read_env() {
  Tuning t;
  t.igemm_splitk = env_int("ADVOC_IGEMM_SPLITK", 1);
  t.igemm_tail = env_int("ADVOC_IGEMM_TAIL", 1);
  t.igemm_x6 = env_int("ADVOC_IGEMM_X6", 1);
  t.igemm_x6_wide = -1;      // (r6: fixed -- ADVOC_IGEMM_X6_WIDE had no test)
  t.igemm_bk = 0;      // (r6: fixed -- the switch ADVOC_IGEMM_BK had no test; its other settings are history, NOTEBOOK.md)
  t.igemm_x6_n32 = 1;      // (r6: fixed -- the switch ADVOC_IGEMM_X6_N32 had no test; its other settings are history, NOTEBOOK.md)
  t.igemm_x6_tile = 0;      // (r6: fixed -- the switch ADVOC_IGEMM_X6_TILE had no test; its other settings are history, NOTEBOOK.md)
  t.igemm_tile = 0;      // (r6: fixed -- the switch ADVOC_IGEMM_TILE had no test; its other settings are history, NOTEBOOK.md)
  t.igemm_korder = -1;      // (r6: fixed -- the switch ADVOC_IGEMM_KORDER had no test; its other settings are history, NOTEBOOK.md)
  t.wgrad_x6 = env_int("ADVOC_WGRAD_X6", 1);
  t.wgrad_h3 = env_int("ADVOC_WGRAD_H3", 1);
  t.wgrad_h3_min_m = env_int("ADVOC_WGRAD_H3_MIN_M", 128);
  t.wgrad_h3_tile = env_int("ADVOC_WGRAD_H3_TILE", 0);
  t.wgrad_h3_rounds = 0;      // (r6: fixed -- the switch ADVOC_WGRAD_H3_ROUNDS had no test; its other settings are history, NOTEBOOK.md)
  t.wgrad_h3_rows = env_int("ADVOC_WGRAD_H3_ROWS", 1);
  t.wgrad_h3_ordered = env_int("ADVOC_WGRAD_H3_ORDERED", 1);
  t.h3 = env_int("ADVOC_H3", 1);
  t.h3_tile = env_int("ADVOC_H3_TILE", 0);
#ifdef ADVOC_DIAG      // switches that make results WRONG (timing experiments, tools/micro): only in `make DIAG=1` builds
  t.h3_skip_prep = env_int("ADVOC_H3_SKIP_PREP", 0);
#else
  t.h3_skip_prep = 0;
#endif
  t.h3_min_tiles = env_int("ADVOC_H3_MIN_TILES", 4);
  t.h3_patch = env_int("ADVOC_H3_PATCH", 1);
  t.h3_patch_min_wgs = env_int("ADVOC_H3_PATCH_MIN_WGS", 256);
  t.h3_patch_s2 = 1;      // (r6: fixed -- the switch ADVOC_H3_PATCH_S2 had no test; its other settings are history, NOTEBOOK.md)
  t.h3_patch_rem = 1;      // (r6: fixed -- the switch ADVOC_H3_PATCH_REM had no test; its other settings are history, NOTEBOOK.md)
  t.h3_patch_n32 = 1;      // (r6: fixed -- the switch ADVOC_H3_PATCH_N32 had no test; its other settings are history, NOTEBOOK.md)
  t.h3_patch_persist = env_int("ADVOC_H3_PATCH_PERSIST", 2);
#ifdef ADVOC_DIAG
  t.h3_patch_ablate = env_int("ADVOC_H3_PATCH_ABLATE", 0);
#else
  t.h3_patch_ablate = 0;
#endif
  t.h3_patch_2wg = env_int("ADVOC_H3_PATCH_2WG", 0);
  t.h3_patch_2wg_delay = 100;      // (r6: fixed -- the switch ADVOC_H3_PATCH_2WG_DELAY had no test; its other settings are history, NOTEBOOK.md)
  t.reserve_cus = env_int("ADVOC_RESERVE_CUS", 0);
  if (t.reserve_cus < 0) t.reserve_cus = 0;
  t.thin_wgrad_bias = 1;      // (r6: fixed -- the switch ADVOC_THIN_WGRAD_BIAS had no test; its other settings are history, NOTEBOOK.md)
  t.fused_taps = env_int("ADVOC_FUSED_TAPS", 1);
  t.thin_fwd_spec = 1;      // (r6: fixed -- the switch ADVOC_THIN_FWD_SPEC had no test; its other settings are history, NOTEBOOK.md)
  t.thin_wgrad_nt = 4;      // (r6: fixed -- the switch ADVOC_THIN_WGRAD_NT had no test; its other settings are history, NOTEBOOK.md)
  t.h3_deep_wgs_per_cu = env_int("ADVOC_H3_DEEP_WGS_PER_CU", 2);
  t.h3_deep_split_div = env_int("ADVOC_H3_DEEP_SPLIT_DIV", 8);
  if (t.h3_deep_wgs_per_cu < 1) t.h3_deep_wgs_per_cu = 1;
  if (t.h3_deep_split_div < 2) t.h3_deep_split_div = 2;
  t.h3_rem_ws = env_int("ADVOC_H3_REM_WS", 1);
  t.h3_rem_wgs_per_cu = env_int("ADVOC_H3_REM_WGS_PER_CU", 2);
  t.h3_rem_split_div = env_int("ADVOC_H3_REM_SPLIT_DIV", 8);
  t.h3_deep_stages = env_int("ADVOC_H3_DEEP_STAGES", 3);      // (r6: 3 -- 35 launches per step, -10 % each: profiles/r06_ab_deep_env.txt)
  t.h3_deep_split = env_int("ADVOC_H3_DEEP_SPLIT", 0);
  t.h3_deep_plan = 1;      // (r6: fixed -- the switch ADVOC_H3_DEEP_PLAN had no test; its other settings are history, NOTEBOOK.md)
  t.emit_dx = 1;      // (r6: fixed -- the switch ADVOC_EMIT_DX had no test; its other settings are history, NOTEBOOK.md)
  t.h3_patch_s1n128 = 1;      // (r6: fixed -- the switch ADVOC_H3_PATCH_S1N128 had no test; its other settings are history, NOTEBOOK.md)
  if (t.h3_rem_wgs_per_cu < 1) t.h3_rem_wgs_per_cu = 1;
  if (t.h3_rem_split_div < 2) t.h3_rem_split_div = 2;
  return t;
}
// two slots + an index: a reload publishes a complete new table; readers never see a half-written one
Tuning g_tuning[2];
std::atomic<int> g_tuning_idx{-1};
std::mutex g_tuning_mu;
}  // namespace

const Tuning& tuning() {
  int i = g_tuning_idx.load(std::memory_order_acquire);
  if (i < 0) {
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    i = g_tuning_idx.load(std::memory_order_acquire);
    if (i < 0) {
      g_tuning[0] = read_env();
      g_tuning_idx.store(0, std::memory_order_release);
      i = 0;
    }
  }
  return g_tuning[i];
}
}  // namespace advoc

extern "C" void advoc_tuning_reload(void) {
  std::lock_guard<std::mutex> lk(advoc::g_tuning_mu);
  const int cur = advoc::g_tuning_idx.load(std::memory_order_acquire);
  const int nxt = cur == 0 ? 1 : 0;
  advoc::g_tuning[nxt] = advoc::read_env();
  advoc::g_tuning_idx.store(nxt, std::memory_order_release);
}
