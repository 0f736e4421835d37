// mfma_shapes.hip -- what the matrix pipe sustains on RANDOM bf16 data with operands resident in registers: the 16x16x32 tile this
// library's kernels use against the 32x32x16 tile (half the operand register reads and half the LDS / L2 fragment bytes per FLOP).
// Answers one question for DESIGN.md 4.5: is the ~1.9 GHz the chip grants inside the K loops a property of the FLOPs, or of the
// operand traffic that comes with them?   Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_shapes tools/experiments/mfma_shapes.hip && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// NA x NB independent accumulators per wave: A fragment i with B fragment j, as a register-tiled GEMM inner loop does.
template <int NA, int NB>
__global__ __launch_bounds__(256, 2) void k16(const bf16x8_t* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  bf16x8_t a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = src[(blockIdx.x * 16 + i) * 64 + lane];
  for (int j = 0; j < NB; ++j) b[j] = src[(blockIdx.x * 16 + 8 + j) * 64 + lane];
  f32x4_t acc[NA][NB];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NA, int NB>
__global__ __launch_bounds__(256, 2) void k32(const bf16x8_t* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  bf16x8_t a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = src[(blockIdx.x * 16 + i) * 64 + lane];
  for (int j = 0; j < NB; ++j) b[j] = src[(blockIdx.x * 16 + 8 + j) * 64 + lane];
  f32x16_t acc[NA][NB];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename K>
static int run(const char* name, K kern, double flop_per_mfma, int n_mfma, const bf16x8_t* src, float* out, unsigned long long* clk, int blocks) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, out, 2000, clk);      // warm
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  unsigned long long c[2] = {0, 0};
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; sum += ms;
    CK(hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost));
  }
  const double flops = (double)blocks * 4 * n_mfma * (double)iters * flop_per_mfma;
  printf("%-34s %8.3f ms (best of 5; mean %.3f)  %7.1f TFLOP/s  shader clock in the loop %.0f MHz (last run)\n", name, best, sum / 5, flops / best * 1e-9,
         c[1] ? (double)c[0] / ((double)c[1] / 100.0) : 0.0);
  return 0;
}

int main(int argc, char** argv) {
  const int zeros = argc > 1 && atoi(argv[1]) == 0;        // `mfma_shapes 0`: all-zero operands (the data-dependent part of the power)
  const int blocks = 512;                                   // two 4-wave workgroups per CU: 2 waves per SIMD, as in the kernels
  std::vector<unsigned short> h((size_t)blocks * 16 * 64 * 8);
  srand(1);
  for (auto& v : h) { float f = zeros ? 0.f : (float)(rand() & 0xffff) / 65536.0f - 0.5f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8_t* src; float* out; unsigned long long* clk;
  CK(hipMalloc(&src, h.size() * 2)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&clk, 16));
  CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  printf("%s operands, %d workgroups of 4 waves, 20000 loop turns\n", zeros ? "all-zero" : "random", blocks);
  for (int round = 0; round < 2; ++round) {
    if (run("16x16x32 bf16, 8 x 4 accumulators", k16<8, 4>, 16384.0, 32, src, out, clk, blocks)) return 1;
    if (run("32x32x16 bf16, 4 x 2 accumulators", k32<4, 2>, 32768.0, 8, src, out, clk, blocks)) return 1;
    if (run("16x16x32 bf16, 4 x 4 accumulators", k16<4, 4>, 16384.0, 16, src, out, clk, blocks)) return 1;
    if (run("32x32x16 bf16, 2 x 2 accumulators", k32<2, 2>, 32768.0, 4, src, out, clk, blocks)) return 1;
  }
  return 0;
}
