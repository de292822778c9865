// Issue cost of the instructions the field arithmetic is made of, on gfx950: 8 independent chains per lane of ONE opcode, 4 waves per
// SIMD (4096 blocks of 64), wave-cycles per instruction from the wall time at the clock rocm-smi reports.
//   v_mad_u64_u32 (the product columns), v_lshrrev_b64 / v_alignbit_b32 + v_lshrrev_b32 (carry extraction), v_and_b32, v_add_u32
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_microbench.hip -o tools/bin/valu_rate_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32;
typedef unsigned long long u64;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP> __global__ void __launch_bounds__(64) k_rate(u32 *io, int K) {
  const u32 t = blockIdx.x * 64 + threadIdx.x;
  u64 a[8];
  u32 b[8];
  for (int k = 0; k < 8; k++) { a[k] = ((u64)io[t * 16 + k] << 32) | io[t * 16 + 8 + k]; b[k] = io[t * 16 + k] | 1u; }
  for (int i = 0; i < K; i++) {
#define MAD(j) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[j]) : "v"(b[j]), "v"(b[(j + 1) & 7]) : "s10", "s11");
#define SHR64(j) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(a[j]));
#define ALIGN(j) asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(b[j]) : "v"(b[(j + 1) & 7]));
#define SHR32(j) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(b[j]));
#define AND32(j) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(b[j]));
#define ADD32(j) asm volatile("v_add_u32 %0, %1, %0" : "+v"(b[j]) : "v"(b[(j + 1) & 7]));
#define ADD64(j) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
    if (OP == 0) { REP8(MAD) REP8(MAD) }
    if (OP == 1) { REP8(SHR64) REP8(SHR64) }
    if (OP == 2) { REP8(ALIGN) REP8(ALIGN) }
    if (OP == 3) { REP8(SHR32) REP8(SHR32) }
    if (OP == 4) { REP8(AND32) REP8(AND32) }
    if (OP == 5) { REP8(ADD32) REP8(ADD32) }
    if (OP == 6) { REP8(ADD64) REP8(ADD64) }
    if (OP == 7) { REP8(MAD) REP8(AND32) }          // the mix of a product column
#define MADNOP(j) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n\ts_nop 0" : "+v"(a[j]) : "v"(b[j]), "v"(b[(j + 1) & 7]) : "s10", "s11");
    if (OP == 8) { REP8(MADNOP) REP8(MADNOP) }      // round 4: what the compiler's s_nop 0 behind an asm statement costs (instruction count: 16 of 32 are nops)
  }
  u32 r = 0;
  for (int k = 0; k < 8; k++) r ^= (u32)a[k] ^ (u32)(a[k] >> 32) ^ b[k];
  io[t * 16] = r;
}
int main(int argc, char **argv) {
  u32 *d;
  const int blocks = (argc > 1 ? atoi(argv[1]) : 4) * 1024;          // waves per SIMD
  CK(hipMalloc(&d, 64ull * 64 * blocks));
  CK(hipMemset(d, 0x5A, 64ull * 64 * blocks));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const double ghz = pr.clockRate / 1e6;
  const int K = 4000;
  const char *names[9] = {"v_mad_u64_u32", "v_lshrrev_b64", "v_alignbit_b32", "v_lshrrev_b32", "v_and_b32", "v_add_u32", "v_lshl_add_u64", "8 mad + 8 and", "16 mad + 16 s_nop"};
  printf("%d wave(s) per SIMD\n", blocks / 1024);
  for (int op = 0; op < 9; op++) {
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      switch (op) {
        case 0: hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, 0, d, K); break;
        case 1: hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, 0, d, K); break;
        case 2: hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(64), 0, 0, d, K); break;
        case 3: hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(64), 0, 0, d, K); break;
        case 4: hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(64), 0, 0, d, K); break;
        case 5: hipLaunchKernelGGL(k_rate<5>, dim3(blocks), dim3(64), 0, 0, d, K); break;
        case 6: hipLaunchKernelGGL(k_rate<6>, dim3(blocks), dim3(64), 0, 0, d, K); break;
        case 7: hipLaunchKernelGGL(k_rate<7>, dim3(blocks), dim3(64), 0, 0, d, K); break;
        default: hipLaunchKernelGGL(k_rate<8>, dim3(blocks), dim3(64), 0, 0, d, K); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double instr_per_wave = 16.0 * K, waves_per_simd = blocks / 1024.0;
    const double cycles = ms * 1e-3 * ghz * 1e9 / (instr_per_wave * waves_per_simd);
    printf("%-16s %8.3f ms  -> %5.2f SIMD cycles per wave-instruction at %.2f GHz (4 = full rate; the s_nop row counts its 16 multiply-adds only)\n", names[op], ms, cycles, ghz);
  }
  return 0;
}
