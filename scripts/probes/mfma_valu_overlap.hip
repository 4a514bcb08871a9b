// Do matrix (v_mfma_f32_16x16x32_bf16) and vector instructions overlap on a gfx950 SIMD?  One workgroup per CU, cycles from
// s_memtime around a loop of ITER rounds; per round a wave issues NM independent matrix instructions and / or NV vector
// instructions (8 independent dependency chains each, so neither side stalls on its own latency).
//   mode 0: matrix only            mode 1: vector only (v_fma_f32)      mode 2: both, interleaved in ONE wave
//   mode 3: two waves per SIMD, waves 0-3 matrix only, waves 4-7 vector only
//   mode 4: two waves per SIMD, both interleaved (half the rounds each)
//   mode 5: vector only, v_exp_f32 instead of v_fma_f32 (cost of the transcendental)
//   mode 6: vector only, v_pk_fma_f32           mode 7: matrix + v_exp_f32 interleaved, one wave
//   mode 8 / 9: as 0 / 2 with v_mfma_f32_32x32x16_bf16 (4 per round = the same flops)
//   mode 10: two waves per SIMD, waves 4-7 (the YOUNGER ones) matrix only, waves 0-3 vector only
//   mode 11 / 12 / 13: as 0 / 2 / 3 with the accumulators in the AGPR file (inline asm, "a" constraint)
// Build: hipcc --offload-arch=gfx950 -O2 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITER 2000
#define NM 8
#define NV 40

template <bool DO_M, int VKIND>   // VKIND: 0 none, 1 fma, 2 exp, 3 pk_fma
__device__ __forceinline__ void body(float* out, int rounds) {
  f32x4 acc[NM];
  for (int i = 0; i < NM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float v[8];
  f32x2 v2[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; v2[i] = (f32x2){v[i], v[i] + 1.f}; }
  const float k0 = 0.999f, k1 = 1e-3f;
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      if (DO_M) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV / NM; ++i) {
        const int c = (j * (NV / NM) + i) & 7;
        if (VKIND == 1) v[c] = __builtin_fmaf(v[c], k0, k1);
        if (VKIND == 2) v[c] = __builtin_amdgcn_exp2f(v[c]);
        if (VKIND == 3) v2[c] = __builtin_elementwise_fma(v2[c], (f32x2){k0, k0}, (f32x2){k1, k1});
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i] + v2[i][0] + v2[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool DO_M, bool DO_V>
__device__ __forceinline__ void body_agpr(float* out, int rounds) {
  f32x4 acc[NM];
  for (int i = 0; i < NM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float k0 = 0.999f, k1 = 1e-3f;
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      if (DO_M) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < NV / NM; ++i) {
        const int c = (j * (NV / NM) + i) & 7;
        if (DO_V) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(k0), "v"(k1));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool DO_V>
__device__ __forceinline__ void body32(float* out, int rounds) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float k0 = 0.999f, k1 = 1e-3f;
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV / 4; ++i) {
        const int c = (j * (NV / 4) + i) & 7;
        if (DO_V) v[c] = __builtin_fmaf(v[c], k0, k1);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void probe(int mode, float* out, unsigned long long* cyc) {
  const int w = threadIdx.x >> 6;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  switch (mode) {
    case 0: body<true, 0>(out, ITER); break;
    case 1: body<false, 1>(out, ITER); break;
    case 2: body<true, 1>(out, ITER); break;
    case 3: if (w < 4) body<true, 0>(out, ITER); else body<false, 1>(out, ITER); break;
    case 4: body<true, 1>(out, ITER / 2); break;
    case 5: body<false, 2>(out, ITER); break;
    case 6: body<false, 3>(out, ITER); break;
    case 7: body<true, 2>(out, ITER); break;
    case 8: body32<false>(out, ITER); break;
    case 9: body32<true>(out, ITER); break;
    case 10: if (w >= 4) body<true, 0>(out, ITER); else body<false, 1>(out, ITER); break;
    case 11: body_agpr<true, false>(out, ITER); break;
    case 12: body_agpr<true, true>(out, ITER); break;
    case 13: if (w < 4) body_agpr<true, false>(out, ITER); else body_agpr<false, true>(out, ITER); break;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc; unsigned long long h[8];
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  const char* names[] = {"matrix only, 1 wave/SIMD", "v_fma only, 1 wave/SIMD", "matrix + v_fma interleaved, 1 wave/SIMD",
                         "2 waves/SIMD: one matrix-only, one v_fma-only", "2 waves/SIMD, both interleaved, half the rounds each",
                         "v_exp_f32 only, 1 wave/SIMD", "v_pk_fma_f32 only, 1 wave/SIMD", "matrix + v_exp interleaved, 1 wave/SIMD",
                         "32x32x16 matrix only (4 per round), 1 wave/SIMD", "32x32x16 matrix + v_fma interleaved, 1 wave/SIMD",
                         "2 waves/SIMD: the older v_fma-only, the younger matrix-only",
                         "matrix only, accumulators in AGPRs, 1 wave/SIMD", "matrix (AGPR accumulators) + v_fma interleaved, 1 wave/SIMD",
                         "2 waves/SIMD: one matrix-only (AGPR accumulators), one v_fma-only"};
  for (int mode = 0; mode < 14; ++mode) {
    const int threads = (mode == 3 || mode == 4 || mode == 10 || mode == 13) ? 512 : 256;
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(cyc, 0, 64);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(256), dim3(threads), 0, 0, mode, out, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      if (rep == 1)
        printf("mode %d (%s): %d rounds x (%d matrix + %d vector): wave 0 %.1f ticks/round, wave %d %.1f ticks/round; kernel %.1f us -> %.1f ns/round\n",
               mode, names[mode], ITER, NM, NV, (double)h[0] / ITER, threads / 64 - 1, (double)h[threads / 64 - 1] / ITER, ms * 1e3,
               ms * 1e6 / ITER);
    }
  }
  return 0;
}
