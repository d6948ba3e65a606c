// Which bf16 MFMA shape should k_match_bf16 use on MI355X?  (round 5; MI355X_MICROARCH.md "DVFS give-back" item 7: the chip can
// hold a higher clock on v_mfma_f32_16x16x32_bf16 than on v_mfma_f32_32x32x16_bf16 at equal cycles per flop.)
// One wave per SIMD, the kernel's own phase: 32 video rows x 32 audio columns x 3 features, K = 41 padded; the resident operand
// of six row tiles in AGPRs, the streamed operand and two ping-ponged accumulator sets in VGPRs, random bf16 data, the epilogue
// of the previous phase (one v_add3_u32 + one v_alignbit_b32 per accumulator row of 64 pairs: 32 VALU per phase) spread over the
// MFMA gaps of the phase or left out.
//   s32      9 x 32x32x16                          K = 48   288 MFMA cycles per phase (the shipped kernel's shape)
//   s16k64   24 x 16x16x32                         K = 64   384
//   s16k48   12 x 16x16x32 + 12 x 16x16x16         K = 48   288 IF the K = 16 instruction runs at the full rate (8 cycles)
//   s32k8    6 x 32x32x16 + 6 x 32x32x8            K = 48   288 IF the K = 8 instruction runs at the full rate (16 cycles)
//   t16 / t8 a loop of nothing but v_mfma_f32_16x16x16_bf16 / v_mfma_f32_32x32x8_bf16: their cycles
// Output per variant: shader cycles per phase (s_memtime), wall ns per phase (HIP events), the clock that follows, and the
// in-kernel clock from s_memtime / s_memrealtime (100 MHz).
// build: hipcc -O3 --offload-arch=gfx950 shape_bench.hip -o shape_bench; run: ./shape_bench [blocks = 256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

enum { S32 = 0, S16K64 = 1, S16K48 = 2, S32K8 = 3, T16 = 4, T8 = 5 };

__device__ __forceinline__ void epi2(uint32_t& codes, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f) {
  uint32_t t;
  asm volatile("v_add3_u32 %1, %2, %3, %4\n\tv_alignbit_b32 %0, %0, %1, 30\n\tv_add3_u32 %1, %5, %6, %7\n\tv_alignbit_b32 %0, %0, %1, 30"
               : "+v"(codes), "=&v"(t) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f));
}
__device__ __forceinline__ void epi1(uint32_t& codes, uint32_t a, uint32_t b, uint32_t c) {
  uint32_t t;
  asm volatile("v_add3_u32 %1, %2, %3, %4\n\tv_alignbit_b32 %0, %0, %1, 30" : "+v"(codes), "=&v"(t) : "v"(a), "v"(b), "v"(c));
}

// accumulators of one phase as 48 floats: [feature][16]; the 16x16 shapes see them as [feature][sub-tile 4][4]
struct Acc { float v[3][16]; };

template <int kShape, bool kEpi>
__global__ __launch_bounds__(256, 1) void k_shape(const uint4* __restrict__ frag, int iters, unsigned long long* __restrict__ out, float* sink) {
  const int lane = threadIdx.x & 63;
  // resident operand: 6 row tiles x 36 registers (whatever the shape: 32 rows x 48 k x 3 features of bf16)
  bf16x8 A[6][9];
#pragma unroll
  for (int rt = 0; rt < 6; ++rt)
#pragma unroll
    for (int q = 0; q < 9; ++q)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(A[rt][q]) : "v"(frag + (rt * 9 + q) * 64 + lane) : "memory");
  bf16x8 B[12];                                    // streamed operand (K = 64 needs 12 quads, the others 9)
#pragma unroll
  for (int q = 0; q < 12; ++q)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(B[q]) : "v"(frag + (54 + q) * 64 + lane) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // every loaded quad stays live up to here: a dead load result may otherwise share registers with a later address temporary
  // and land on top of it (the compiler does not know these statements are loads)
#pragma unroll
  for (int q = 0; q < 12; ++q) asm volatile("" : "+v"(B[q]));
  f32x16 acc0[3], acc1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { acc0[j] = f32x16{0}; acc1[j] = f32x16{0}; }
  uint32_t codes = lane;

  auto phase = [&](const bf16x8 (&Art)[9], f32x16 (&acc)[3], const f32x16 (&accp)[3]) {
    if constexpr (kShape == S32) {
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const int j = m % 3, s = m / 3;
        if (s == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[j]) : "a"(Art[3 * j + s]), "v"(B[3 * j + s]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "a"(Art[3 * j + s]), "v"(B[3 * j + s]));
        __builtin_amdgcn_sched_barrier(0);
        if (kEpi && m >= 1) {
          const int g = 17 - 2 * m;
          epi2(codes, __float_as_uint(accp[0][g]), __float_as_uint(accp[1][g ^ 1]), __float_as_uint(accp[2][g ^ 2]),
               __float_as_uint(accp[0][g - 1]), __float_as_uint(accp[1][(g - 1) ^ 1]), __float_as_uint(accp[2][(g - 1) ^ 2]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (kShape == S32K8) {
      // per feature: 2 x 32x32x16 (k 0..31) + 2 x 32x32x8 (k 32..47); operands of the K = 8 instruction are register pairs
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const int j = m % 3, s = m / 3;
        if (s == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[j]) : "a"(Art[3 * j]), "v"(B[3 * j]));
        else if (s == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "a"(Art[3 * j + 1]), "v"(B[3 * j + 1]));
        else {
          const bf16x4 a = s == 2 ? __builtin_shufflevector(Art[3 * j + 2], Art[3 * j + 2], 0, 1, 2, 3) : __builtin_shufflevector(Art[3 * j + 2], Art[3 * j + 2], 4, 5, 6, 7);
          const bf16x4 b = s == 2 ? __builtin_shufflevector(B[3 * j + 2], B[3 * j + 2], 0, 1, 2, 3) : __builtin_shufflevector(B[3 * j + 2], B[3 * j + 2], 4, 5, 6, 7);
          asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "a"(a), "v"(b));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kEpi && m >= 1 && m <= 8) {
          const int g = 17 - 2 * m;
          epi2(codes, __float_as_uint(accp[0][g]), __float_as_uint(accp[1][g ^ 1]), __float_as_uint(accp[2][g ^ 2]),
               __float_as_uint(accp[0][g - 1]), __float_as_uint(accp[1][(g - 1) ^ 1]), __float_as_uint(accp[2][(g - 1) ^ 2]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // 16x16 shapes: sub-tile t = (row half, column half) of the 32 x 32 phase; accumulator quad [j][t]
      f32x4 (&a4)[3][4] = reinterpret_cast<f32x4 (&)[3][4]>(acc);
      const f32x4 (&p4)[3][4] = reinterpret_cast<const f32x4 (&)[3][4]>(accp);
      constexpr int kSteps = kShape == S16K64 ? 2 : 2;
#pragma unroll
      for (int s = 0; s < kSteps; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int m = (s * 4 + t) * 3 + j;                 // 0 .. 23
            const int rh = t >> 1, ch = t & 1;
            // operand quads: resident Art[3 j + ...] holds (row half rh, k step) pieces; streamed B likewise -- which quad feeds
            // which instruction does not matter for timing, only that every instruction has its own registers
            if (s == 0) {
              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(a4[j][t]) : "a"(Art[3 * j + rh]), "v"(B[3 * j + ch]));
            } else if (kShape == S16K64) {
              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[j][t]) : "a"(Art[(3 * j + 2 + rh) % 9]), "v"(B[9 + (j + ch) % 3]));
            } else {
              const bf16x4 a = rh ? __builtin_shufflevector(Art[3 * j + 2], Art[3 * j + 2], 4, 5, 6, 7) : __builtin_shufflevector(Art[3 * j + 2], Art[3 * j + 2], 0, 1, 2, 3);
              const bf16x4 b = ch ? __builtin_shufflevector(B[3 * j + 2], B[3 * j + 2], 4, 5, 6, 7) : __builtin_shufflevector(B[3 * j + 2], B[3 * j + 2], 0, 1, 2, 3);
              asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(a4[j][t]) : "a"(a), "v"(b));
            }
            __builtin_amdgcn_sched_barrier(0);
            // 16 accumulator rows of the previous phase over the gaps behind instructions 2 .. 17: one row each
            if (kEpi && m >= 2 && m < 18) {
              const int g = 17 - m, tq = g >> 2, e = g & 3;
              epi1(codes, __float_as_uint(p4[0][tq][e]), __float_as_uint(p4[1][tq][e ^ 1]), __float_as_uint(p4[2][tq][e ^ 2]));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
    }
  };

  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  if constexpr (kShape == T16 || kShape == T8) {
    // the tail instruction alone: 24 (12) independent accumulator quads, back to back
    f32x4 (&a4)[3][4] = reinterpret_cast<f32x4 (&)[3][4]>(acc0);
    f32x4 (&b4)[3][4] = reinterpret_cast<f32x4 (&)[3][4]>(acc1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 3; ++rep)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if constexpr (kShape == T16) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const bf16x4 a = __builtin_shufflevector(A[rep][3 * j], A[rep][3 * j], 0, 1, 2, 3), b = __builtin_shufflevector(B[3 * j], B[3 * j], 0, 1, 2, 3);
              asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(a4[j][t]) : "a"(a), "v"(b));
              asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(b4[j][t]) : "a"(a), "v"(b));
            }
          } else {
            const bf16x4 a = __builtin_shufflevector(A[rep][3 * j], A[rep][3 * j], 0, 1, 2, 3), b = __builtin_shufflevector(B[3 * j], B[3 * j], 0, 1, 2, 3);
            asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+v"(acc0[j]) : "a"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+v"(acc1[j]) : "a"(a), "v"(b));
          }
        }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rt = 0; rt < 6; rt += 2) { phase(A[rt], acc0, acc1); phase(A[rt + 1], acc1, acc0); }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc0[0]), "+v"(acc0[1]), "+v"(acc0[2]));
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  if (lane == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
  float sacc = __uint_as_float(codes);
#pragma unroll
  for (int j = 0; j < 3; ++j) sacc += acc0[j][lane & 15] + acc1[j][(lane + 3) & 15];
  if (sacc == 12345.678f) sink[0] = sacc;
}

template <int kShape, bool kEpi>
void run(const char* name, double mfma_cycles, double flop_per_phase, const uint4* d_frag, unsigned long long* d_out, float* d_sink, int blocks, int iters) {
  float ms = 0; double cyc = 0, rt = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_shape<kShape, kEpi>), dim3(blocks), dim3(256), 0, 0, d_frag, iters, d_out, d_sink);
    (void)hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c, r;
    for (size_t k = 0; k < h.size(); k += 2) { c.push_back((double)h[k]); r.push_back((double)h[k + 1]); }
    std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
    cyc = c[c.size() / 2]; rt = r[r.size() / 2];
  }
  const double phases = (double)iters * 6;
  const double ns = ms * 1e6 / phases;
  if (flop_per_phase == 0) {                        // tail-instruction loops: mfma_cycles = instructions per iteration here
    printf("%-10s blocks=%-4d cycles per instruction=%6.2f  wall ns per instruction=%6.2f  clock %.3f GHz  in-kernel clock %.3f GHz\n", name, blocks,
           cyc / (iters * mfma_cycles), ms * 1e6 / (iters * mfma_cycles), cyc / (ms * 1e6), cyc / rt * 0.1);
    return;
  }
  printf("%-10s epilogue=%-3s blocks=%-4d cycles/phase=%8.2f (MFMA alone: %3.0f)  wall ns/phase=%7.2f  clock(cycles/wall) %.3f GHz  in-kernel clock %.3f GHz  "
         "%6.1f TFLOP/s on 256 CUs counting 246 flop per pair\n",
         name, kEpi ? "yes" : "no", blocks, cyc / phases, mfma_cycles, ns, (cyc / phases) / ns, cyc / rt * 0.1,
         flop_per_phase * 1024.0 * 1024.0 / ns * 1e-3);   // 1024 pairs per phase and wave, 1024 waves on 256 CUs
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  setvbuf(stdout, nullptr, _IONBF, 0);
  std::vector<uint32_t> h(80 * 64 * 4);
  uint32_t s = 12345;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; uint32_t a = 0x3C00 + ((s >> 8) & 0x3FF), b = 0xBC00 + ((s >> 20) & 0x3FF); x = a | (b << 16); }   // random bf16, magnitude ~0.01
  uint4* d_frag; (void)hipMalloc(&d_frag, h.size() * 4); (void)hipMemcpy(d_frag, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  unsigned long long* d_out; (void)hipMalloc(&d_out, blocks * 8 * 8);
  float* d_sink; (void)hipMalloc(&d_sink, 4);
  const double useful = 246.0;                     // algorithmic flop per pair; a phase is 1024 pairs
  for (int round = 0; round < 2; ++round) {        // the first round also warms the chip up: quote the second
    printf("---- round %d\n", round);
    run<S32, false>("s32", 288, useful, d_frag, d_out, d_sink, blocks, iters);
    run<S32, true>("s32", 288, useful, d_frag, d_out, d_sink, blocks, iters);
    run<S16K64, false>("s16k64", 384, useful, d_frag, d_out, d_sink, blocks, iters);
    run<S16K64, true>("s16k64", 384, useful, d_frag, d_out, d_sink, blocks, iters);
    run<S16K48, false>("s16k48", 288, useful, d_frag, d_out, d_sink, blocks, iters);
    run<S16K48, true>("s16k48", 288, useful, d_frag, d_out, d_sink, blocks, iters);
    run<S32K8, false>("s32k8", 288, useful, d_frag, d_out, d_sink, blocks, iters);
    run<S32K8, true>("s32k8", 288, useful, d_frag, d_out, d_sink, blocks, iters);
    run<T16, false>("t16", 72, 0, d_frag, d_out, d_sink, blocks, iters);
    run<T8, false>("t8", 18, 0, d_frag, d_out, d_sink, blocks, iters);
  }
  return 0;
}
