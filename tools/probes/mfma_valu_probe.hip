// Does vector arithmetic issue in the shadow of a matrix instruction?  One SIMD, one or two waves, four streams:
//   0: 16 x v_mfma_f32_32x32x16_bf16 (accumulator in vector registers)          2: 0 and 1 interleaved (6 fma after each mfma)
//   1: 16 x 6 v_fma_f32                                                          3: as 2 with the accumulator in accumulation registers
// Prints shader cycles per group of 16 (median over waves).  build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define FMA6 "v_fma_f32 %3, %3, %9, %3\n v_fma_f32 %4, %4, %9, %4\n v_fma_f32 %5, %5, %9, %5\n v_fma_f32 %6, %6, %9, %6\n v_fma_f32 %7, %7, %9, %7\n v_fma_f32 %8, %8, %9, %8\n"
// two exponentials and four fma, independent / the softmax chain of one score pair: fma -> exp -> add -> add -> cvt
#define EXP2FMA4 "v_exp_f32 %3, %3\n v_fma_f32 %4, %4, %9, %4\n v_fma_f32 %5, %5, %9, %5\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %9, %7\n v_fma_f32 %8, %8, %9, %8\n"
#define CHAIN "v_fma_f32 %3, %5, %9, %7\n v_fma_f32 %4, %6, %9, %7\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_add_f32 %8, %8, %3\n v_add_f32 %8, %8, %4\n v_cvt_pk_f16_f32 %5, %3, %4\n"
#define EXP6_F32 "v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n v_exp_f32 %8, %8\n"
#define EXP6_F16 "v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n v_exp_f16 %8, %8\n"
#define MF "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n"
// the softmax of one score pair WITHOUT the transcendental unit: exponent = rint via a magic add, fraction through a
// cubic in packed fp16, 2^n built from the integer part, row sum by a dot product (13 instructions, none of them v_exp)
#define POLY "v_pk_fma_f32 %[A], %[S], %[M], %[N]\n v_pk_add_f32 %[B], %[A], %[G]\n v_pk_add_f32 %[C], %[B], %[H]\n v_pk_add_f32 %[D], %[A], %[C]\n" \
             "v_cvt_pk_f16_f32 %[e], %[x0], %[x1]\n v_pk_fma_f16 %[p], %[e], %[x2], %[x3]\n v_pk_fma_f16 %[p], %[p], %[e], %[x4]\n v_pk_fma_f16 %[p], %[p], %[e], %[x5]\n" \
             "v_perm_b32 %[n], %[x0], %[x1], %[x2]\n v_pk_max_i16 %[n], %[n], %[x3]\n v_pk_lshlrev_b16 %[n], 10, %[n]\n v_pk_mul_f16 %[p], %[p], %[n]\n v_dot2_f32_f16 %[l], %[p], %[x4], %[l]\n"
#define MFN "v_mfma_f32_32x32x16_bf16 %[acc], %[a], %[b], %[acc]\n"
#define R16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void probe(long long* cycles, float* sink, int iters) {
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  u32x4 a = {threadIdx.x, 1, 2, 3}, b = {4, 5, 6, threadIdx.x};
  float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, c = 0.999f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0)
      asm volatile(R16(MF) : "+v"(acc) : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(c));
    else if constexpr (MODE == 1)
      asm volatile(R16(FMA6) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 2)
      asm volatile(R16(MF FMA6) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 4)
      asm volatile(R16(MF EXP2FMA4) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 5)
      asm volatile(R16(MF CHAIN) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 6)
      asm volatile(R16(CHAIN) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 7)
      asm volatile(R16(EXP2FMA4) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 10)
      asm volatile(R16(EXP6_F32) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 11)
      asm volatile(R16(EXP6_F16) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 12)
      asm volatile(R16(MF EXP6_F32) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 13)
      asm volatile(R16(MF EXP6_F16) : "+v"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
    else if constexpr (MODE == 8 || MODE == 9) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      static f2 A, B, C, D;  // (values do not matter: throughput only)
      f2 S = {x0, x1}, M = {c, c}, N = {x2, x3}, G = {x4, x4}, H = {x5, x5};
      unsigned e = 1, pp = 2, n = 3;
      float l = 0.f;
      f2 rA = S, rB = S, rC = S, rD = S;
      if constexpr (MODE == 8)
        asm volatile(R16(POLY) : [A] "+v"(rA), [B] "+v"(rB), [C] "+v"(rC), [D] "+v"(rD), [e] "+v"(e), [p] "+v"(pp), [n] "+v"(n), [l] "+v"(l)
                     : [S] "v"(S), [M] "v"(M), [N] "v"(N), [G] "v"(G), [H] "v"(H), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [x4] "v"(x4), [x5] "v"(x5));
      else
        asm volatile(R16(MFN POLY) : [acc] "+v"(acc), [A] "+v"(rA), [B] "+v"(rB), [C] "+v"(rC), [D] "+v"(rD), [e] "+v"(e), [p] "+v"(pp), [n] "+v"(n), [l] "+v"(l)
                     : [a] "v"(a), [b] "v"(b), [S] "v"(S), [M] "v"(M), [N] "v"(N), [G] "v"(G), [H] "v"(H), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [x4] "v"(x4), [x5] "v"(x5));
      x0 += l + rD[0] + __uint_as_float(pp);
    } else
      asm volatile(R16(MF FMA6) : "+a"(acc), "+v"(a), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(c));
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
  if ((threadIdx.x & 63) == 0) cycles[w] = t1 - t0;
  float s = x0 + x1 + x2 + x3 + x4 + x5;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
static void run(const char* name, int threads, int iters) {
  const int grid = 256, waves = grid * threads / 64;
  long long* d;
  float* sink;
  hipMalloc(&d, waves * sizeof(long long));
  hipMalloc(&sink, 4);
  for (int rep = 0; rep < 3; ++rep) probe<MODE><<<grid, threads>>>(d, sink, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(waves);
  hipMemcpy(h.data(), d, waves * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-44s %d wave(s)/SIMD: %7.1f cycles per 16\n", name, threads / 256, (double)h[waves / 2] / iters);
  hipFree(d);
  hipFree(sink);
}

int main() {
  const int iters = 2000;
  for (int threads : {256, 512}) {
    run<0>("16 mfma 32x32x16 (acc in VGPRs)", threads, iters);
    run<1>("16 x 6 v_fma_f32", threads, iters);
    run<2>("16 x (mfma + 6 fma), acc in VGPRs", threads, iters);
    run<3>("16 x (mfma + 6 fma), acc in AGPRs", threads, iters);
    run<7>("16 x (2 exp + 4 fma)", threads, iters);
    run<4>("16 x (mfma + 2 exp + 4 fma)", threads, iters);
    run<6>("16 x softmax chain of a score pair (7 instr)", threads, iters);
    run<5>("16 x (mfma + softmax chain of a score pair)", threads, iters);
    run<10>("16 x 6 v_exp_f32", threads, iters);
    run<11>("16 x 6 v_exp_f16", threads, iters);
    run<12>("16 x (mfma + 6 v_exp_f32)", threads, iters);
    run<13>("16 x (mfma + 6 v_exp_f16)", threads, iters);
    run<8>("16 x exp-free chain of a score pair (13 instr)", threads, iters);
    run<9>("16 x (mfma + exp-free chain of a score pair)", threads, iters);
  }
  return 0;
}
