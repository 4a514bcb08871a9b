// Do matrix (v_mfma_f32_16x16x32_bf16) and vector instructions overlap on a gfx950 SIMD, and what does a vector instruction
// cost?  Every measured instruction is an `asm volatile` statement (the first version of this probe let the compiler see the
// vector chains: it packed pairs of v_fma_f32 into v_pk_fma_f32 and moved accumulators between the register files, and the
// numbers described its output, not the instructions named).  One workgroup per CU; cycles from s_memtime around ROUNDS rounds;
// a round = NM matrix instructions (8 independent AGPR accumulators) and / or NV vector instructions (8 independent chains).
//   A. one wave per SIMD: matrix only; each vector kind only; matrix + vector kind interleaved (1 matrix : 5 vector)
//   B. two waves per SIMD: waves 0-3 matrix only for the whole run, waves 4-7 one vector kind for a QUARTER of the rounds (so the
//      vector waves live entirely inside the matrix waves' run): their time per round against the same waves running alone
//   C. two waves per SIMD, both running the interleaved stream of A, half the rounds each: SIMD time per round
// Build: hipcc --offload-arch=gfx950 -O2 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ROUNDS 2000
#define NM 8
#define NV 40

enum { V_NONE, V_FMA, V_PK_FMA, V_EXP, V_ADD, V_MAX3, V_CVT_PK, V_AND, V_PK_SHL16, V_ADD_CO, V_CNDMASK_S, NKIND };
static const char* kind_name[] = {"-", "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_add_f32", "v_max3_f32", "v_cvt_pk_bf16_f32",
                                  "v_and_b32", "v_pk_lshlrev_b16", "v_add_co_u32 (vcc)", "v_cndmask_b32 (sgpr pair mask)"};

template <int K>
__device__ __forceinline__ void vinst(uint32_t (&v)[16], int i, uint32_t k, unsigned long long mask) {
  uint32_t& x = v[i & 7];
  unsigned long long& x2 = *reinterpret_cast<unsigned long long*>(&v[(2 * i) & 14]);
  if (K == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
  if (K == V_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"((unsigned long long)k << 32 | k));
  if (K == V_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (K == V_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(k));
  if (K == V_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
  if (K == V_CVT_PK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(k));
  if (K == V_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k));
  if (K == V_PK_SHL16) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(x));
  if (K == V_ADD_CO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(k) : "vcc");
  if (K == V_CNDMASK_S) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "s"(mask));
}

template <bool DO_M, int K>
__device__ __forceinline__ void body(float* out, int rounds) {
  f32x4 acc[NM];
  for (int i = 0; i < NM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  uint32_t v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0x3f800000u + threadIdx.x * 64u + i;     // floats a little above 1
  const uint32_t k = 0x3f7fff00u;                                                // a float a little below 1
  const unsigned long long mask = 0x5555555555555555ull;
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      if (DO_M) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < NV / NM; ++i) vinst<K>(v, j * (NV / NM) + i, k, mask);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += __builtin_bit_cast(float, v[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K>
__global__ __launch_bounds__(512) void probe(int mode, float* out, unsigned long long* cyc) {
  const int w = threadIdx.x >> 6;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mode == 0) body<true, V_NONE>(out, ROUNDS);                // matrix only
  if (mode == 1) body<false, K>(out, ROUNDS);                    // vector kind only
  if (mode == 2) body<true, K>(out, ROUNDS);                     // interleaved in one wave
  if (mode == 3) { if (w < 4) body<true, V_NONE>(out, ROUNDS); else body<false, K>(out, ROUNDS / 4); }
  if (mode == 4) body<true, K>(out, ROUNDS / 2);                 // two waves per SIMD, BOTH interleaved, half the rounds each
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

template <int K>
static void run(float* out, unsigned long long* cyc, double mfma_alone) {
  unsigned long long h[8];
  double t[5] = {0, 0, 0, 0, 0};
  for (int mode = (K == V_NONE ? 0 : 1); mode < (K == V_NONE ? 1 : 5); ++mode) {
    const int threads = mode >= 3 ? 512 : 256;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemset(cyc, 0, 64);
      hipLaunchKernelGGL(probe<K>, dim3(256), dim3(threads), 0, 0, mode, out, cyc);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    }
    t[mode] = mode == 3 ? (double)h[7] / (ROUNDS / 4) : (double)h[0] / ROUNDS;
    if (mode == 3) t[0] = (double)h[0] / ROUNDS;      // the matrix wave of the same SIMD, per round
    if (mode == 4) t[4] = (double)(h[0] > h[4] ? h[0] : h[4]) / ROUNDS;   // SIMD time per round of (matrix + vector) work, two waves sharing it
  }
  if (K == V_NONE) {
    printf("matrix only: %.1f cycles per round of %d v_mfma_f32_16x16x32_bf16 = %.2f each\n", t[0], NM, t[0] / NM);
    return;
  }
  printf("%-32s alone %6.1f cycles / %d = %5.2f each | interleaved with %d matrix instructions in one wave: %6.1f (sum of the two alone: %6.1f)"
         " | beside a matrix-only wave on the same SIMD: %6.1f per round = %.2f x alone (the matrix wave: %.1f per round)"
         " | the interleaved rounds split over TWO waves of the SIMD: %6.1f per round\n",
         kind_name[K], t[1], NV, t[1] / NV, NM, t[2], t[1] + mfma_alone, t[3], t[3] / t[1], t[0], t[4]);
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  // matrix only first: its time per round is the reference of the "sum" column
  unsigned long long h[8];
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL(probe<V_NONE>, dim3(256), dim3(256), 0, 0, 0, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  }
  const double m = (double)h[0] / ROUNDS;
  run<V_NONE>(out, cyc, m);
  run<V_FMA>(out, cyc, m); run<V_PK_FMA>(out, cyc, m); run<V_EXP>(out, cyc, m); run<V_ADD>(out, cyc, m); run<V_MAX3>(out, cyc, m);
  run<V_CVT_PK>(out, cyc, m); run<V_AND>(out, cyc, m); run<V_PK_SHL16>(out, cyc, m); run<V_ADD_CO>(out, cyc, m); run<V_CNDMASK_S>(out, cyc, m);
  return 0;
}
