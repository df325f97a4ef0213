// Issue cost of the f64 vector instructions the sample body and the two-loop recursion consist of, on one SIMD with one and
// with two resident waves: streams of INDEPENDENT v_fma_f64 / v_mul_f64 / v_add_f64 (and v_fma_f32 / v_mov_b32 for scale),
// timed with the shader clock inside the kernel and with events around it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_f64_rate tools/valu_f64_rate.hip && /tmp/valu_f64_rate
// Output: cycles per wave instruction seen by one wave, and per SIMD (= that / resident waves), for each kind; one kind per
// process with an argument (`valu_f64_rate 12`), 100 = the code-size sweep.  (An EXEC-mask kind -- s_and_saveexec_b64 with an
// undefined VCC inside the loop -- hung the process on the GPU and was removed.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ACC = 16;     // independent accumulators (no dependent issue within 16 instructions)
constexpr int UNROLL = 8;   // ACC * UNROLL instructions per loop trip

template <int KIND>
__global__ void __launch_bounds__(64, 2) k_rate(int trips, double seed, double* out, long long* cycles) {
  double a[ACC];
  float f[ACC];
#pragma unroll
  for (int i = 0; i < ACC; i++) { a[i] = seed + i + threadIdx.x; f[i] = (float)a[i]; }
  const double m = 1.0000001, c = 1.0e-9;
  const float mf = 1.0000001f, cf = 1.0e-9f;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int t = 0; t < trips; t++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int i = 0; i < ACC; i++) {
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        else if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        else if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        else if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(mf), "v"(cf));
        else if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(f[i]) : "v"(mf));
        else if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(mf));
        else if (KIND == 6) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        else if (KIND == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(f[i]) : "v"(mf) : "s20", "s21");
        else if (KIND == 8) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
        else if (KIND == 9) asm volatile("v_cmp_lt_f64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(a[i]), "+v"(f[i]) : "v"(c), "v"(mf) : "vcc");
        else if (KIND == 10) asm volatile("v_mov_b64 %0, %1" : "+v"(a[i]) : "v"(c));
        else if (KIND == 11) asm volatile("v_floor_f64 %0, %0" : "+v"(a[i]));
        else if (KIND == 12) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
        else if (KIND == 13) asm volatile("v_min_i32 %0, %0, %1" : "+v"(f[i]) : "v"(mf));
        else if (KIND == 14) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(f[i]) : "v"(mf));
        else if (KIND == 15) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a[i]) : "v"(c));
        else if (KIND == 16) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
        else if (KIND == 17) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(f[i]) : "v"(a[i]));
        else if (KIND == 19) asm volatile("v_readlane_b32 s20, %0, 63" : : "v"(f[i]) : "s20");
        else if (KIND == 21) asm volatile("s_load_dwordx2 s[20:21], %3, 0x0\n s_waitcnt lgkmcnt(0)\n v_fma_f64 %0, %0, s[20:21], %2" : "+v"(a[i]) : "v"(m), "v"(c), "s"(out) : "s20", "s21");
        else if (KIND == 22) asm volatile("s_load_dwordx2 s[20:21], %3, 0x0\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n s_waitcnt lgkmcnt(0)\n v_fma_f64 %0, %0, s[20:21], %2" : "+v"(a[i]) : "v"(m), "v"(c), "s"(out) : "s20", "s21");
        else asm volatile("v_fma_f64 %0, %0, %1, %2\n s_mov_b32 s20, 0x3ff00000" : "+v"(a[i]) : "v"(m), "v"(c) : "s20");
      }
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) s += a[i] + f[i];
  if (s == 1.2345) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

// The same v_fma_f64 stream as straight-line code of BODY instructions (8 bytes each) per loop trip: does the rate hold when the
// loop body no longer fits the instruction cache (64 KB shared by two compute units)?
template <int BODY>
__global__ void __launch_bounds__(64, 2) k_code(int trips, double seed, double* out, long long* cycles) {
  double a[ACC];
#pragma unroll
  for (int i = 0; i < ACC; i++) a[i] = seed + i + threadIdx.x;
  const double m = 1.0000001, c = 1.0e-9;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  // (macro repetition: the unroller stops at 2048 instructions)
#define F16 asm volatile("v_fma_f64 %0, %0, %16, %17\n v_fma_f64 %1, %1, %16, %17\n v_fma_f64 %2, %2, %16, %17\n v_fma_f64 %3, %3, %16, %17\n" \
                         "v_fma_f64 %4, %4, %16, %17\n v_fma_f64 %5, %5, %16, %17\n v_fma_f64 %6, %6, %16, %17\n v_fma_f64 %7, %7, %16, %17\n" \
                         "v_fma_f64 %8, %8, %16, %17\n v_fma_f64 %9, %9, %16, %17\n v_fma_f64 %10, %10, %16, %17\n v_fma_f64 %11, %11, %16, %17\n" \
                         "v_fma_f64 %12, %12, %16, %17\n v_fma_f64 %13, %13, %16, %17\n v_fma_f64 %14, %14, %16, %17\n v_fma_f64 %15, %15, %16, %17" \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),  \
                           "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(m), "v"(c));
#define X4(x) x x x x
#define F64 X4(F16)
#define F256 X4(F64)
#define F1024 X4(F256)
#define F4096 X4(F1024)
  static_assert(ACC == 16, "F16 is written for 16 accumulators");
  for (int t = 0; t < trips; t++) {
    if (BODY == 1024) { F1024 }
    else if (BODY == 4096) { F4096 }
    else if (BODY == 8192) { F4096 F4096 }
    else if (BODY == 16384) { X4(F4096) }
    else { X4(F4096) X4(F4096) }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) s += a[i];
  if (s == 1.2345) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
template <int BODY>
void run_code(int waves_per_simd, double* out, long long* cyc_d) {
  const int trips = (20000 * 128) / BODY;
  hipDeviceProp_t P;
  CK(hipGetDeviceProperties(&P, 0));
  const int grid = P.multiProcessorCount * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_code<BODY>, dim3(grid), dim3(64), 0, 0, 2, 1.0, out, cyc_d);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_code<BODY>, dim3(grid), dim3(64), 0, 0, trips, 1.0, out, cyc_d);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  long long cyc = 0;
  CK(hipMemcpy(&cyc, cyc_d, 8, hipMemcpyDeviceToHost));
  const double instr = (double)trips * BODY;
  printf("v_fma_f64, loop body %6d instructions (%4d KB) %d wave(s)/SIMD: %6.2f ticks per instruction of one wave (%5.2f per SIMD); %7.3f ms -> %6.2f ns per instruction per SIMD\n",
         BODY, BODY * 8 / 1024, waves_per_simd, cyc / instr, cyc / instr / waves_per_simd, ms, ms * 1e6 / instr / waves_per_simd);
}

template <int KIND>
void run(const char* name, int waves_per_simd, double* out, long long* cyc_d) {
  const int trips = 20000;
  hipDeviceProp_t P;
  CK(hipGetDeviceProperties(&P, 0));
  const int grid = P.multiProcessorCount * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(64), 0, 0, 100, 1.0, out, cyc_d);   // warm-up
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(64), 0, 0, trips, 1.0, out, cyc_d);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  long long cyc = 0;
  CK(hipMemcpy(&cyc, cyc_d, 8, hipMemcpyDeviceToHost));
  const double instr = (double)trips * ACC * UNROLL;
  printf("%-14s %d wave(s)/SIMD: %6.2f shader-clock ticks per instruction (pair) of one wave (%5.2f per SIMD); %7.3f ms -> %6.2f ns per instruction per SIMD\n",
         name, waves_per_simd, cyc / instr, cyc / instr / waves_per_simd, ms, ms * 1e6 / instr / waves_per_simd);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;   // one kind per process (a kind that hangs then costs one timeout)
  double* out; long long* cyc;
  CK(hipMalloc(&out, 8)); CK(hipMalloc(&cyc, 8));
  for (int w = 1; w <= 2; w++) {
    if (only < 0 || only == 0) run<0>("v_fma_f64", w, out, cyc);
    if (only < 0 || only == 1) run<1>("v_mul_f64", w, out, cyc);
    if (only < 0 || only == 2) run<2>("v_add_f64", w, out, cyc);
    if (only < 0 || only == 6) run<6>("v_max_f64", w, out, cyc);
    if (only < 0 || only == 10) run<10>("v_mov_b64", w, out, cyc);
    if (only < 0 || only == 11) run<11>("v_floor_f64", w, out, cyc);
    if (only < 0 || only == 12) run<12>("v_rcp_f64", w, out, cyc);
    if (only < 0 || only == 8) run<8>("v_cmp_lt_f64", w, out, cyc);
    if (only < 0 || only == 17) run<17>("v_cvt_i32_f64", w, out, cyc);
    if (only < 0 || only == 15) run<15>("v_lshl_add_u64", w, out, cyc);
    if (only < 0 || only == 3) run<3>("v_fma_f32", w, out, cyc);
    if (only < 0 || only == 4) run<4>("v_mov_b32", w, out, cyc);
    if (only < 0 || only == 16) run<16>("v_mov_b32 dpp", w, out, cyc);
    if (only < 0 || only == 13) run<13>("v_min_i32", w, out, cyc);
    if (only < 0 || only == 14) run<14>("v_mul_lo_u32", w, out, cyc);
    if (only < 0 || only == 5) run<5>("cndmask vcc", w, out, cyc);
    if (only < 0 || only == 7) run<7>("cndmask sgpr", w, out, cyc);
    if (only < 0 || only == 9) run<9>("cmp+cndmask", w, out, cyc);
    if (only < 0 || only == 19) run<19>("v_readlane", w, out, cyc);
    if (only < 0 || only == 20) run<20>("fma64+s_mov", w, out, cyc);
    if (only < 0 || only == 21) run<21>("s_load,wait,fma", w, out, cyc);
    if (only < 0 || only == 22) run<22>("s_load,8fma,wait,fma", w, out, cyc);
  }
  if (only < 0 || only == 100)
  for (int w = 1; w <= 2; w++) {
    run_code<1024>(w, out, cyc);
    run_code<4096>(w, out, cyc);
    run_code<8192>(w, out, cyc);
    run_code<16384>(w, out, cyc);
    run_code<32768>(w, out, cyc);
  }
  int clk = 0;
  CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
  printf("device clock rate attribute: %d kHz\n", clk);
  return 0;
}
