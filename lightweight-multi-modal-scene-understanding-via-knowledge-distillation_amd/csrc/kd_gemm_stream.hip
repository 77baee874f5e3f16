// kd_gemm_stream.hip -- dispatch of the weight-resident streaming GEMM kernels (kd_gemm_stream_kernel.h) and the
// instantiations of the forward shapes; the data-gradient shapes are instantiated in kd_gemm_stream_bwd*.hip.
#include "kd_gemm_stream_kernel.h"

#include <atomic>
#include <cstdlib>

#ifdef KD_STREAM_DBG
namespace kd_stream { __device__ unsigned long long kd_stream_dbg[8]; }
#endif

using namespace kd_stream;

// data-gradient instantiations (kd_gemm_stream_bwd0.hip / _bwd2.hip): 1 launched, 0 no such instance
int kd_stream_bwd0_dispatch(const GemmArgs& g, int kb, int nb, dim3 grid, hipStream_t st);             // PRO2, EPI0
int kd_stream_bwd2_dispatch(const GemmArgs& g, int kb, int nb, int pro, int epi, dim3 grid, hipStream_t st);   // PRO2/4, EPI2/3

namespace {

std::atomic<int> g_stream_on{-1};

int stream_mode() {          // 0 off, 1 only the forward shapes that win in isolation, 2 every covered shape (default)
  int v = g_stream_on.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("KD_GEMM_STREAM");
    v = (e && e[0] == '0') ? 0 : ((e && e[0] == '1') ? 1 : ((e && e[0] == '3') ? 3 : 2));
    g_stream_on.store(v, std::memory_order_relaxed);
  }
  return v;
}

struct StreamCfg { int kb, nb, ntiles; };

constexpr size_t LDS_MAX = 160 * 1024;

// Which (K, N, prologue, epilogue) launches have a streaming instance, and its tiling.
bool stream_cfg(int K, int N, int pro, int epi, bool add, StreamCfg& c) {
  const int mode = stream_mode();
  if (mode == 0 || K % 32 != 0 || N % 32 != 0) return false;
  const int kb = K / 32, nbt = N / 32;
  const bool fwd = (pro == 0 || pro == 1 || pro == 3) && (epi == 0 || epi == 1 || epi == 5) && !(pro == 3 && epi == 5);
  const bool bwd = (pro == 2 && (epi == 0 || epi == 2)) || (pro == 4 && epi == 2 && kb == 4 && nbt == 4);
  if (!fwd && !bwd) return false;
  if (bwd && mode == 3) return false;                 // mode 3: forward shapes only (A/B of the data-gradient kernels)
  // K = 32, 64, 128.  Round 4 re-measured K = 192 / 256 / 384 in the two-register-set form (profiles/r04_stream_wide_k.txt): in isolation only
  // the K = 192 -> N = 32 data gradient won (1444 -> 1314 us), and inside the KD step not even that (1163 us tiled, 1173-1279 us streaming):
  // every K >= 192 reduction stays on the tiled kernel.
  if (!(kb == 1 || kb == 2 || kb == 4)) return false;
  if (pro == 3 && kb != 2) return false;
  // widest column tile (in 32-column blocks) that divides N and whose three W planes + tables fit the LDS
  int nb = 0;
  for (int cand = 4; cand >= 1; cand >>= 1)
    if (nbt % cand == 0 && stream_lds_bytes(K, 32 * cand, pro) <= LDS_MAX) { nb = cand; break; }
  if (nb == 0) return false;
  if (pro == 2 && epi == 2) {
    const int sig = kb * 100 + nb * 10 + (add ? 1 : 0);
    if (sig == 120 || sig == 141 || sig == 211 || sig == 220 || sig == 241 || sig == 411 || sig == 420 || sig == 441) return false;
  }
  if (pro == 2 && epi == 0 && kb == 4 && nb == 2) return false;   // (measured 255 vs 243 us in the KD step: the tiled kernel stays)
  const int ntiles = nbt / nb;
  // every column tile streams A again (the tiles of one slab run side by side on the same XCD, so most of it is an L2
  // hit): accept two tiles always, more only when A is the small side of the launch
  if (ntiles > 2 && (int64_t)ntiles * K > 2 * (int64_t)N) return false;
  if (ntiles > 8) return false;
  c = {kb, nb, ntiles};
  // Measured on an MI355X (profiles/r02_stream_vs_tiled.txt, r02_stream_dgrad_ab.txt): the forward shapes run 1.0-1.45x the
  // tiled kernel, the K <= 128 data gradients 1.0-1.2x (LiDAR layer 2: 8.2 -> 7.3 ms); K >= 192 loses in every form tried
  // and stays tiled.  "Every covered shape" is the default; mode 1 keeps the narrow forward-only rule of the first version.
  if (mode >= 2) return true;
  return fwd && ntiles == 1 && kb <= 4 && (nb <= 2 || (kb == 4 && nb == 4) || (pro == 3 && epi == 1));
}

int stream_grid(int64_t M, int ntiles) {
  // one workgroup per CU (the W planes take most of the LDS), fewer when there are not enough 32-row slabs for every wave
  int64_t want = (M + 32 * SW - 1) / (32 * SW);
  int cap = 256 / ntiles;
  if (cap < 1) cap = 1;
  return (int)(want < cap ? want : cap);
}

template <int KB, int KC, int NB, bool DB>
int fwd_dispatch(const GemmArgs& g, int pro, int epi, dim3 grid, hipStream_t st) {
  const bool add = g.addend != nullptr;
#define KD_SCASE(P_, E_)                                                              \
  if (pro == P_ && epi == E_) {                                                       \
    return add ? stream_launch_one<KB, KC, NB, P_, E_, DB, true>(g, grid, st)         \
               : stream_launch_one<KB, KC, NB, P_, E_, DB, false>(g, grid, st);       \
  }
  if constexpr (!(KB == 2 && !DB)) { KD_SCASE(0, 0) KD_SCASE(0, 1) KD_SCASE(1, 0) KD_SCASE(1, 1) KD_SCASE(0, 5) KD_SCASE(1, 5) }
  if constexpr (KB == 2 && !DB) {
    if (!add && pro == 3 && epi == 0) return stream_launch_one<KB, KC, NB, 3, 0, DB, false>(g, grid, st);
    if (!add && pro == 3 && epi == 1) return stream_launch_one<KB, KC, NB, 3, 1, DB, false>(g, grid, st);
  }
#undef KD_SCASE
  return 0;
}

}  // namespace

int kd_gemm_stream_stat_rows(int64_t M, int K, int N, int pro, int epi, bool add) {
  StreamCfg c;
  if (!stream_cfg(K, N, pro, epi, add, c)) return 0;
  return stream_grid(M, c.ntiles);          // one row per workgroup (the waves' sums are combined in LDS)
}

int kd_gemm_stream_launch(GemmArgs& g, int pro, int epi, hipStream_t st) {
  StreamCfg c;
  if (!stream_cfg(g.K, g.N, pro, epi, g.addend != nullptr, c)) return 0;
  const dim3 grid(stream_grid(g.M, c.ntiles), c.ntiles);
  int rc = 0;
  if (pro == 2 && epi == 0) rc = kd_stream_bwd0_dispatch(g, c.kb, c.nb, grid, st);
  else if (pro == 2 || pro == 4) rc = kd_stream_bwd2_dispatch(g, c.kb, c.nb, pro, epi, grid, st);
  else {
    // (K / 32, chunk / 32, N tile / 32, two register sets).  Measured per shape (tools/bench_stream, M = 32 frames): K = 128 and
    // the BatchNorm-prologue K = 64 layers run 5-15 % faster with 32-column chunks loaded one chunk ahead; the LiDAR layer-1
    // launch (PRO3: the streamed tensor is the 16-byte point) keeps the single whole-slab load.
#define KD_SHAPE(KB_, KC_, NB_, DB_) if (c.kb == KB_ && c.nb == NB_) rc = fwd_dispatch<KB_, KC_, NB_, DB_>(g, pro, epi, grid, st);
    KD_SHAPE(1, 1, 1, false) KD_SHAPE(1, 1, 2, false) KD_SHAPE(1, 1, 4, false)
    if (pro == 3) { KD_SHAPE(2, 2, 1, false) KD_SHAPE(2, 2, 2, false) KD_SHAPE(2, 2, 4, false) }
    else { KD_SHAPE(2, 1, 1, true) KD_SHAPE(2, 1, 2, true) KD_SHAPE(2, 1, 4, true) }
    KD_SHAPE(4, 1, 1, true) KD_SHAPE(4, 1, 2, true) KD_SHAPE(4, 1, 4, true)
#undef KD_SHAPE
  }
  if (rc < 0) return rc;                                                                                       // (error already set)
  if (rc == 1) { const int e = kd_check_launch("kd_gemm_stream"); if (e) return -(e > 0 ? e : -e) - 1000; }   // < 0: error
  if (rc == 0) {          // stream_cfg promised this shape (and its statistics-slab row count) to the callers: never fall back silently
    kd_set_error("kd_gemm_stream: no instance for K=%d N=%d pro=%d epi=%d%s", g.K, g.N, pro, epi, g.addend ? " with addend" : "");
    return -2000;
  }
  return rc;
}

extern "C" int kd_set_gemm_stream(int mode) { return g_stream_on.exchange(mode < 0 ? 0 : (mode > 3 ? 2 : mode)); }   // 0 off, 1 selective, 2 all, 3 forward only; returns the previous mode
#ifdef KD_STREAM_DBG
extern "C" int kd_stream_dbg_read(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(kd_stream::kd_stream_dbg), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(kd_stream::kd_stream_dbg), z, sizeof(z)); }
  return 0;
}
#endif
