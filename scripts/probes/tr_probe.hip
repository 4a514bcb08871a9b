// Probe of ds_read_b64_tr_b16 (gfx950): which (source lane, element) does lane n / element j receive?
// Every lane passes the address of ITS OWN 4 consecutive 16-bit words (lane i -> words 4i..4i+3, value = word index),
// so a returned value v names source lane v / 4, element v % 4.   Build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(&lds[4 * l]));
  uint64_t raw = __builtin_bit_cast(uint64_t, v);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(raw >> (16 * j));
}
int main() {
  uint16_t* d; uint16_t h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int n = 0; n < 64; ++n) {
    printf("lane %2d:", n);
    for (int j = 0; j < 4; ++j) {
      const int v = h[n * 4 + j];
      printf("  (L%2d,e%d)", v / 4, v % 4);
      const int g = n >> 4, c = n & 15;
      if (v / 4 != 16 * g + 4 * j + (c >> 2) || v % 4 != (c & 3)) ok = 0;
    }
    printf("\n");
  }
  printf("EXPECTED_MAPPING %s  (lane n elem j <- lane 16*(n>>4) + 4*j + ((n&15)>>2), elem n&3)\n", ok ? "CONFIRMED" : "DIFFERENT");
  return 0;
}
