// Which of a SIMD's pipes run side by side on gfx950?  One workgroup per CU, W waves per SIMD (W = 1, 2); every wave runs
// a long loop of ONE kind of instruction (by wave role) or a mix; time per iteration in shader cycles (hipEvents, 2.4 GHz assumed
// only for the cycle column).  Roles: E = 8 independent v_exp_f32, F = 32 independent v_fma_f32, P = 16 v_pk_fma_f32 (32 values),
// M = 4 independent v_mfma_f32_32x32x16_bf16, X = E and F in the same wave, Y = M and F in the same wave.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/pipes.hip -o pytorch-models_amd/csrc/build/mb_pipes && gpurun -- ./pytorch-models_amd/csrc/build/mb_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void body_E(float (&e)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
}
__device__ __forceinline__ void body_F(float (&f)[32], float a, float b) {
#pragma unroll
  for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(a), "v"(b));
}
__device__ __forceinline__ void body_P(f32x2 (&p)[16], f32x2 a, f32x2 b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a), "v"(b));
}
__device__ __forceinline__ void body_M(f32x16 (&m)[4], bf16x8 a, bf16x8 b) {
#pragma unroll
  for (int i = 0; i < 4; ++i) m[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, m[i], 0, 0, 0);
}

// role per wave: roles[wave]; 'E','F','P','M','X' (E+F), 'Y' (M+F), 'Z' (M+E), '-' idle
__global__ __launch_bounds__(512) void pipes(const char* roles, int iters, float* sink) {
  const int wave = threadIdx.x >> 6;
  const char role = roles[wave];
  float e[8], f[32];
  f32x2 p[16];
  f32x16 m[4];
  for (int i = 0; i < 8; ++i) e[i] = -1.0f - i * 1e-3f - threadIdx.x * 1e-6f;
  for (int i = 0; i < 32; ++i) f[i] = 1.0f + i * 1e-3f;
  for (int i = 0; i < 16; ++i) p[i] = f32x2{1.0f + i, 2.0f + i};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) m[i][j] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x & 7)); b[j] = (__bf16)0.002f; }
  const float fa = 0.999f, fb = 1e-3f;
  const bool doE = role == 'E' || role == 'X' || role == 'Z', doF = role == 'F' || role == 'X' || role == 'Y';
  const bool doP = role == 'P', doM = role == 'M' || role == 'Y' || role == 'Z';
  if (role != '-') {
    for (int it = 0; it < iters; ++it) {
      if (doM) body_M(m, a, b);
      if (doE) { body_E(e); for (int i = 0; i < 8; ++i) asm volatile("v_sub_f32 %0, 0, %0" : "+v"(e[i])); }  // keep exp2's argument negative
      if (doF) body_F(f, fa, fb);
      if (doP) body_P(p, f32x2{fa, fa}, f32x2{fb, fb});
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += e[i];
  for (int i = 0; i < 32; ++i) s += f[i];
  for (int i = 0; i < 16; ++i) s += p[i][0] + p[i][1];
  for (int i = 0; i < 4; ++i) s += m[i][0] + m[i][7];
  if (s == 12345.678f) sink[0] = s;
}

// vector-ALU roles only, up to 4 waves per SIMD (1024 threads): does the per-wave issue rate or the pipe bound the loop?
// A = 32 v_add_f32 (two register operands), F = 32 v_fma_f32 (three), P = 16 v_pk_fma_f32, E = 8 v_exp_f32 (+ 8 v_sub)
__global__ __launch_bounds__(1024) void valu_waves(const char* roles, int iters, float* sink) {
  const int wave = threadIdx.x >> 6;
  const char role = roles[wave];
  float e[8], f[32];
  f32x2 p[16];
  for (int i = 0; i < 8; ++i) e[i] = -1.0f - i * 1e-3f - threadIdx.x * 1e-6f;
  for (int i = 0; i < 32; ++i) f[i] = 1.0f + i * 1e-3f;
  for (int i = 0; i < 16; ++i) p[i] = f32x2{1.0f + i, 2.0f + i};
  const float fa = 0.999f, fb = 1e-3f;
  if (role != '-') {
    for (int it = 0; it < iters; ++it) {
      if (role == 'E') { body_E(e); for (int i = 0; i < 8; ++i) asm volatile("v_sub_f32 %0, 0, %0" : "+v"(e[i])); }
      if (role == 'F') body_F(f, fa, fb);
      if (role == 'A') {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fb));
      }
      if (role == 'P') body_P(p, f32x2{fa, fa}, f32x2{fb, fb});
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += e[i];
  for (int i = 0; i < 32; ++i) s += f[i];
  for (int i = 0; i < 16; ++i) s += p[i][0] + p[i][1];
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  const int iters = 20000;
  char* d_roles; float* sink;
  (void)hipMalloc(&d_roles, 8); (void)hipMalloc(&sink, 4);
  // waves 0-3 sit on SIMDs 0-3, waves 4-7 are their partners
  const char* cfgs[] = {"EEEE----", "FFFF----", "PPPP----", "MMMM----", "XXXX----", "YYYY----", "ZZZZ----",
                        "EEEEEEEE", "FFFFFFFF", "MMMMMMMM", "EEEEFFFF", "MMMMFFFF", "MMMMEEEE", "MMMMPPPP", "XXXXXXXX", "YYYYYYYY"};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("roles (waves 0-3 | their SIMD partners 4-7)   us     cycles/iteration @2.4 GHz   [E: 8 v_exp = 128 at quarter rate; F: 32 v_fma = 128; P: 16 v_pk_fma = 64; M: 4 mfma 32x32x16 = 128; the 8 v_sub beside E = 32]\n");
  for (const char* c : cfgs) {
    (void)hipMemcpy(d_roles, c, 8, hipMemcpyHostToDevice);
    pipes<<<256, 512>>>(d_roles, iters, sink);  // same length: the clocks are up when the timed launch starts
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    pipes<<<256, 512>>>(d_roles, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s   %9.1f   %8.1f\n", c, ms * 1e3, ms * 1e-3 * 2.4e9 / iters);
  }
  printf("\nvector-ALU roles, 1 .. 4 waves per SIMD (1024-thread workgroup, wave w on SIMD w %% 4):   us   cycles/iteration @2.4 GHz   cycles per instruction and SIMD\n");
  char* d16; (void)hipMalloc(&d16, 16);
  for (char r : {'A', 'F', 'P', 'E'})
    for (int n = 1; n <= 4; ++n) {
      char cfg[17];
      for (int w = 0; w < 16; ++w) cfg[w] = w < 4 * n ? r : '-';
      cfg[16] = 0;
      (void)hipMemcpy(d16, cfg, 16, hipMemcpyHostToDevice);
      valu_waves<<<256, 1024>>>(d16, iters, sink);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      valu_waves<<<256, 1024>>>(d16, iters, sink);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double cyc = ms * 1e-3 * 2.4e9 / iters;
      const int per_iter = r == 'P' ? 16 : r == 'E' ? 16 : 32;
      printf("%s   %9.1f   %8.1f   %6.2f\n", cfg, ms * 1e3, cyc, cyc / (per_iter * n));
    }
  return 0;
}
