// C-ABI glue: error channel, version, and the attention entry points that pick between the exact (fp32 arithmetic)
// kernels and the MFMA (bf16) kernels.  Declarations: include/bevbert_hip.h.
#include <stdarg.h>

#include "attn_common.h"

static thread_local char g_err[512] = "";

void bb_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

BEVBERT_API const char* bevbert_last_error(void) { return g_err; }
// A failed stream capture (hipErrorStreamCaptureInvalidated and friends) leaves its code in the runtime's sticky
// last-error slot: the NEXT launch check of any library in the process (PyTorch checks hipGetLastError after every
// kernel launch) would report it as its own.  Callers that recover from a failed capture drain the slot here; returns
// the number of stale errors dropped.
BEVBERT_API int bevbert_hip_error_reset(void) {
  int n = 0;
  while (hipGetLastError() != hipSuccess && n < 16) ++n;
  return n;
}

// Zero ``bytes`` bytes of device memory on the stream: a memset command (a memset node in a captured step) instead of a fill
// kernel that takes CUs from the work it runs beside -- the 0.96 GB gradient arena is cleared with it every step.
BEVBERT_API int bevbert_zero(void* p, int64_t bytes, hipStream_t stream) {
  BB_REQUIRE(p != nullptr && bytes >= 0, "zero: null pointer or negative size");
  if (bytes == 0) return BB_OK;
  hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, stream);
  if (e != hipSuccess) {
    bb_set_error("zero: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return BB_ELAUNCH;
  }
  return BB_OK;
}

BEVBERT_API int bevbert_version(void) { return 111; }  // 0.1.1: step salt, device-resident learning rate

static const uint32_t* g_step_salt = nullptr;
const uint32_t* bb_step_salt() { return g_step_salt; }
BEVBERT_API int bevbert_set_step_salt(const void* device_word) {
  g_step_salt = static_cast<const uint32_t*>(device_word);
  return BB_OK;
}
BEVBERT_API const char* bevbert_arch(void) { return "gfx950"; }

int attn_simple_fwd(const AttnArgs& a, int dtype, hipStream_t st);
int attn_simple_bwd(const AttnArgs& a, int dtype, hipStream_t st);
int attn_delta(const AttnArgs& a, float* delta, int dtype, hipStream_t st);
int attn_mfma_fwd(const AttnArgs& a, hipStream_t st);
int attn_mfma_bwd(const AttnArgs& a, hipStream_t st);
int attn_mfma_bwd1(const AttnArgs& a, hipStream_t st);
bool attn_mfma_bwd1_supported(const AttnArgs& a);
int attn_fwd2(const AttnArgs& a, hipStream_t st);
bool attn_fwd2_supported(const AttnArgs& a);
int attn_drop_bits(const AttnArgs& a, uint64_t* bits_f, uint64_t* bits_b, uint32_t* bits_l, hipStream_t st);
int attn_fwd4(const AttnArgs& a, const uint32_t* bits_l, hipStream_t st);
bool attn_fwd4_supported(const AttnArgs& a, const uint32_t* bits_l);
int attn_bwd2(const AttnArgs& a, hipStream_t st);
bool attn_bwd2_supported(const AttnArgs& a);
int attn_bwd3(const AttnArgs& a, hipStream_t st);
int attn_f32_fwd(const AttnArgs& a, hipStream_t st);
int attn_f32_bwd(const AttnArgs& a, hipStream_t st);
bool attn_f32_supported(const AttnArgs& a, int dtype, bool bwd);
bool attn_bwd3_supported(const AttnArgs& a);
int attn_small_fwd(const AttnArgs& a, hipStream_t st);
bool attn_small_fwd_supported(const AttnArgs& a);
int attn_small_bwd(const AttnArgs& a, hipStream_t st);
bool attn_small_bwd_supported(const AttnArgs& a);
int attn_small_bwd2(const AttnArgs& a, hipStream_t st);
bool attn_small_bwd2_supported(const AttnArgs& a);
int attn_short_fwd(const AttnArgs& a, bool bits_ready, hipStream_t st);
bool attn_short_fwd_supported(const AttnArgs& a, bool bits_ready);
int attn_short_bwd(const AttnArgs& a, hipStream_t st);
bool attn_short_bwd_supported(const AttnArgs& a);

// The keep-bit workspace of a call holds the matrix three times: [forward layout | backward layout | per-lane layout of
// attn_fwd4.hip], see attn_fwd2.hip
static int64_t bits_words_one(int B, int nh, int Lq, int Lk) {
  return (int64_t)B * nh * ((Lq + 127) / 128 * 8) * ((Lk + 63) / 64) * 16;
}

// Which kernel the last bevbert_attn_fwd / _bwd call of this thread was dispatched to (the choice depends on shape,
// dtype, dropout and the environment knobs): test / bench introspection, bevbert_attn_last_path().
static thread_local const char* g_attn_path[2] = {"", ""};
#define ATTN_PATH(dir, name, expr) (g_attn_path[dir] = (name), (expr))
BEVBERT_API const char* bevbert_attn_last_path(int backward) { return g_attn_path[backward ? 1 : 0]; }

static bool small_kernels_on() {
  const char* v = getenv("BEVBERT_ATTN_SMALL");
  return v && v[0] == '1';
}

static bool small_bwd2_on() {
  const char* v = getenv("BEVBERT_ATTN_SMALL_BWD");
  return !(v && v[0] == '0');
}

// impl: 0 = auto (bf16 -> MFMA, f32 -> exact), 1 = force exact kernels, 2 = force MFMA (bf16 only),
//       3 = MFMA with the two-kernel backward even where the single-pass backward applies (tests, A/B measurements)
static int pick_impl(int dtype, int impl) {
  if (impl == 0) return dtype == BB_BF16 ? 2 : 1;
  return impl;
}

static int fill_common(AttnArgs& a, const void* q, const void* k, const void* v, const float* key_mask,
                       const float* bias, const int64_t* strides, int B, int nh, int Lq, int Lk, int head_dim,
                       float scale, float drop_p, uint64_t seed, uint64_t offset) {
  BB_REQUIRE(head_dim == ATTN_D, "attention: head_dim=%d unsupported (kernels are specialised for 64)", head_dim);
  BB_REQUIRE(B > 0 && nh > 0 && Lq > 0 && Lk > 0, "attention: empty problem B=%d nh=%d Lq=%d Lk=%d", B, nh, Lq, Lk);
  BB_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention: dropout p=%f", drop_p);
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.key_mask = key_mask; a.bias = bias;
  a.ldq = strides[0]; a.ldk = strides[1]; a.ldv = strides[2]; a.ldo = strides[3];
  a.bsq = strides[4]; a.bsk = strides[5]; a.bsv = strides[6]; a.bso = strides[7];
  a.B = B; a.nh = nh; a.Lq = Lq; a.Lk = Lk; a.scale = scale;
  a.drop_p = drop_p; a.keep_scale = 1.0f / (1.0f - drop_p); a.drop_thr = bb_drop_threshold(drop_p); a.drop_key = bb_site_key(seed, offset); a.salt = bb_step_salt();
  a.Lk2 = (Lk + 1) & ~1;
  a.nq16 = (Lq + 127) / 128 * 8;
  a.nk64 = (Lk + 63) / 64;
  BB_REQUIRE((double)B * nh * Lq * a.Lk2 < 4294967296.0, "attention: more than 2^32 score elements per launch");
  return BB_OK;
}

BEVBERT_API int bevbert_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                                 const float* key_mask, const float* bias, const int64_t* strides, int B, int nh,
                                 int Lq, int Lk, int head_dim, float scale, int dtype, int impl, float drop_p,
                                 uint64_t seed, uint64_t offset, uint64_t* drop_bits, int bits_ready,
                                 hipStream_t stream) {
  AttnArgs a;
  int rc = fill_common(a, q, k, v, key_mask, bias, strides, B, nh, Lq, Lk, head_dim, scale, drop_p, seed, offset);
  if (rc != BB_OK) return rc;
  a.o = o; a.lse = lse; a.drop_bits = drop_bits;
  a.drop_bits_b = drop_bits ? drop_bits + bits_words_one(B, nh, Lq, Lk) : nullptr;
  BB_REQUIRE(dtype == BB_F32 || dtype == BB_BF16, "attn_fwd: dtype %d unsupported", dtype);
  const int im = pick_impl(dtype, impl);
  if (im == 2 || im == 3) {
    BB_REQUIRE(dtype == BB_BF16, "attn_fwd: the MFMA path takes bf16 tensors");
    // BEVBERT_ATTN_FWD=1: the round-2 forward (hashes the dropout mask inline) for A/B measurements and as the on-GPU
    // cross-check of the second-generation kernel
    static const bool gen1 = [] { const char* v = getenv("BEVBERT_ATTN_FWD"); return v && v[0] == '1'; }();
    // BEVBERT_ATTN_SMALL=1 (read per call): short key sequences without a graph bias go to the one-tile-set kernels of
    // attn_small.hip.  Measured (r03y, B = 64, 80 x 80, p = 0.1): 17.6 us against 12.3 + 5.9 us (tiled forward + bit
    // generation), backward 30.6 against 27.7 us -- no gain, so the tiled kernels stay the default.
    // round 6: key sequences up to 96 without a graph bias (text, panoramas, global map, BEV <- text): attn_short.hip, one
    // round of workgroups per launch; BEVBERT_ATTN_SHORT=0 falls through to the kernels of rounds 2-5
    if (!gen1 && !small_kernels_on() && attn_short_fwd_supported(a, bits_ready != 0))
      return ATTN_PATH(0, "attn_short_fwd", attn_short_fwd(a, bits_ready != 0, stream));
    if (!gen1 && small_kernels_on() && attn_small_fwd_supported(a)) return ATTN_PATH(0, "attn_small_fwd", attn_small_fwd(a, stream));
    // Small score matrices with dropout (text 80 x 80, panoramas 36 x 36, the global map): their kernels are bound by
    // launch latency, the inline hash of the round-2 forward hides in it, and that forward leaves the keep bits behind
    // for the backward anyway -- a separate bit-generation launch per site only adds launches (35 of 71 per three steps).
    const bool small = drop_p > 0.f && (int64_t)Lq * Lk < 32768 && Lk <= 256 && !bits_ready;   // (Lk > 256: the 7+1-wave
    // backward wants the backward-layout bits, which only bevbert_attn_drop_bits writes)
    if (!gen1 && !small && attn_fwd2_supported(a)) {
      uint32_t* bits_l = drop_bits ? reinterpret_cast<uint32_t*>(drop_bits + 2 * bits_words_one(B, nh, Lq, Lk)) : nullptr;
      if (drop_p > 0.f && !bits_ready) {
        rc = attn_drop_bits(a, a.drop_bits, a.drop_bits_b, bits_l, stream);
        if (rc != BB_OK) return rc;
      }
      if (attn_fwd4_supported(a, bits_l)) return ATTN_PATH(0, "attn_fwd4", attn_fwd4(a, bits_l, stream));
      return ATTN_PATH(0, "attn_fwd2", attn_fwd2(a, stream));
    }
    return ATTN_PATH(0, "attn_mfma_fwd", attn_mfma_fwd(a, stream));
  }
  // exact arithmetic: fp32 tensors on the fp32 matrix instructions (attn_f32.hip); bf16 storage / BEVBERT_ATTN_F32=simple on
  // the wave-per-row kernels
  if (attn_f32_supported(a, dtype, false)) return ATTN_PATH(0, "attn_f32_fwd", attn_f32_fwd(a, stream));
  return ATTN_PATH(0, "attn_simple_fwd", attn_simple_fwd(a, dtype, stream));
}

BEVBERT_API int bevbert_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                 const float* lse, float* delta_ws, void* dq, void* dk, void* dv, float* dbias,
                                 const float* key_mask, const float* bias, const int64_t* strides, int B, int nh,
                                 int Lq, int Lk, int head_dim, float scale, int dtype, int impl, float drop_p,
                                 uint64_t seed, uint64_t offset, const uint64_t* drop_bits, hipStream_t stream) {
  AttnArgs a;
  int rc = fill_common(a, q, k, v, key_mask, bias, strides, B, nh, Lq, Lk, head_dim, scale, drop_p, seed, offset);
  if (rc != BB_OK) return rc;
  a.drop_bits = const_cast<uint64_t*>(drop_bits);
  a.drop_bits_b = drop_bits ? a.drop_bits + bits_words_one(B, nh, Lq, Lk) : nullptr;
  BB_REQUIRE(dtype == BB_F32 || dtype == BB_BF16, "attn_bwd: dtype %d unsupported", dtype);
  BB_REQUIRE(lse != nullptr && delta_ws != nullptr, "attn_bwd: lse and the (B,nh,Lq) delta workspace are required");
  a.o = const_cast<void*>(o); a.dout = dout; a.lse = const_cast<float*>(lse); a.delta = delta_ws;
  a.dq = dq; a.dk = dk; a.dv = dv; a.dbias = dbias;
  const int im = pick_impl(dtype, impl);
  if (im == 2 || im == 3) {   // the MFMA dQ kernel computes delta itself (and publishes it for the dK/dV kernel)
    BB_REQUIRE(dtype == BB_BF16, "attn_bwd: the MFMA path takes bf16 tensors");
    // one pass over the scores when all keys of a (batch, head) fit one workgroup (attn_bwd1.hip); BEVBERT_ATTN_BWD=split
    // forces the two-kernel path (A/B measurements)
    static const bool split = [] { const char* v = getenv("BEVBERT_ATTN_BWD"); return v && v[0] == 's'; }();
    // BEVBERT_ATTN_BWD=1: the round-2 single-pass kernel where the 7+1-wave kernel (attn_bwd2.hip) would run
    static const bool gen1 = [] { const char* v = getenv("BEVBERT_ATTN_BWD"); return v && v[0] == '1'; }();
    // query and key sequences up to 96 (80 x 80 text, 36 x 36 panoramas, 17 x 80 / 80 x 17 map <-> text): independent
    // query-owner / key-owner waves, attn_small.hip.  BEVBERT_ATTN_SMALL_BWD=0 keeps the single-pass kernel (A/B).
    if (!split && !gen1 && im == 2 && !small_kernels_on() && attn_short_bwd_supported(a))
      return ATTN_PATH(1, "attn_short_bwd", attn_short_bwd(a, stream));
    if (!split && !gen1 && im == 2 && small_bwd2_on() && attn_small_bwd2_supported(a)) return ATTN_PATH(1, "attn_small_bwd2", attn_small_bwd2(a, stream));
    if (!split && !gen1 && small_kernels_on() && im == 2 && attn_small_bwd_supported(a)) return ATTN_PATH(1, "attn_small_bwd", attn_small_bwd(a, stream));
    // BEVBERT_ATTN_BWD3=0: the round-3 loop of the 7+1-wave kernel (attn_bwd2.hip) where the round-5 one (attn_bwd3.hip)
    // would run -- A/B measurements and the on-GPU cross-check
    static const bool gen3 = [] { const char* v = getenv("BEVBERT_ATTN_BWD3"); return !(v && v[0] == '0'); }();
    if (!split && !gen1 && gen3 && im == 2 && attn_bwd3_supported(a)) return ATTN_PATH(1, "attn_bwd3", attn_bwd3(a, stream));
    if (!split && !gen1 && im == 2 && attn_bwd2_supported(a)) return ATTN_PATH(1, "attn_bwd2", attn_bwd2(a, stream));
    if (!split && im == 2 && attn_mfma_bwd1_supported(a)) return ATTN_PATH(1, "attn_mfma_bwd1", attn_mfma_bwd1(a, stream));
    return ATTN_PATH(1, "attn_mfma_bwd", attn_mfma_bwd(a, stream));
  }
  rc = attn_delta(a, delta_ws, dtype, stream);
  if (rc != BB_OK) return rc;
  if (attn_f32_supported(a, dtype, true)) return ATTN_PATH(1, "attn_f32_bwd", attn_f32_bwd(a, stream));
  return ATTN_PATH(1, "attn_simple_bwd", attn_simple_bwd(a, dtype, stream));
}

// Test hook: materialise the dropout keep-mask the kernels derive from (seed, offset + element index).
__global__ void keep_mask_kernel(uint8_t* out, size_t n, uint32_t key, uint32_t thr, const uint32_t* salt) {
  key = bb_salted(key, salt);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = (uint8_t)bb_keep(key, (uint32_t)i, thr);
}
// Size of the keep-bit matrix of an attention call (64-bit words), see attn_common.h.
BEVBERT_API int64_t bevbert_attn_drop_bits_words(int B, int nh, int Lq, int Lk) {
  return 3 * bits_words_one(B, nh, Lq, Lk);
}

// Fill the keep-bit workspace of one attention call ahead of its forward (any stream: the mask is a pure function of
// (seed, offset, step salt, element index)); pass bits_ready = 1 to bevbert_attn_fwd afterwards.
BEVBERT_API int bevbert_attn_drop_bits(uint64_t* drop_bits, int B, int nh, int Lq, int Lk, float drop_p, uint64_t seed,
                                       uint64_t offset, hipStream_t stream) {
  BB_REQUIRE(drop_bits != nullptr && drop_p > 0.f && drop_p < 1.f, "attn_drop_bits: workspace and 0 < p < 1 required");
  AttnArgs a;
  const int64_t st[8] = {64, 64, 64, 64, 64, 64, 64, 64};
  int rc = fill_common(a, nullptr, nullptr, nullptr, nullptr, nullptr, st, B, nh, Lq, Lk, ATTN_D, 1.f, drop_p, seed, offset);
  if (rc != BB_OK) return rc;
  const int64_t one = bits_words_one(B, nh, Lq, Lk);
  return attn_drop_bits(a, drop_bits, drop_bits + one, reinterpret_cast<uint32_t*>(drop_bits + 2 * one), stream);
}

BEVBERT_API int bevbert_dropout_keep_mask(uint8_t* out, int64_t n, float drop_p, uint64_t seed, uint64_t offset,
                                          hipStream_t stream) {
  if (n <= 0) return BB_OK;
  size_t nb = ((size_t)n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(keep_mask_kernel, dim3(nb), dim3(256), 0, stream, out, (size_t)n, bb_site_key(seed, offset),
                     bb_drop_threshold(drop_p), bb_step_salt());
  BB_CHECK_LAUNCH("dropout_keep_mask");
  return BB_OK;
}
