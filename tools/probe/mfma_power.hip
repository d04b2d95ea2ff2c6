// Probe: what bf16 MFMA rate does the BOARD POWER LIMIT allow on random data?  A bare loop of independent v_mfma_f32_16x16x32_bf16 on
// register operands (no memory traffic at all), 256 blocks x 512 threads (two waves per SIMD) or x 256 (one wave per SIMD), with a DUTY
// knob: after every group of 32 MFMAs the wave idles `pad` x 64 cycles (s_sleep), to see the clock the chip holds and the power it draws
// as the MFMA pipe utilisation rises -- the shipped convolution kernels keep the pipe 56-65 % busy.
// Prints, per setting, TFLOP/s, the in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz) and the board power (hwmon power1_input of
// this GPU, sampled by a host thread while the kernels run for ~3 s).
//   hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o bin/mfma_power && bin/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <unistd.h>
#include <limits.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int NA = 4, NB_ = 8;    // 32 accumulators (128 registers), 4 + 8 operand fragments

template <int THREADS>
__global__ __launch_bounds__(THREADS) void burn(const uint32_t* seed, int iters, int pad, float* out, unsigned long long* clk) {
  const int tid = threadIdx.x + blockIdx.x * THREADS;
  bf16x8_t A[NA], B[NB_];
  uint32_t s = seed[tid & 4095] | 1u;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
  auto rbf = [&]() {   // a random bf16 in (-2, 2): random sign, exponent 120..127, random mantissa
    const uint32_t r = rnd();
    return (short)(((r & 1u) << 15) | ((120u + ((r >> 1) & 7u)) << 7) | ((r >> 8) & 0x7fu));
  };
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) A[i][e] = seed[4096] ? rbf() : (short)0;
#pragma unroll
  for (int j = 0; j < NB_; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) B[j][e] = seed[4096] ? rbf() : (short)0;
  f32x4_t acc[NA][NB_];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB_; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB_; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
    for (int p = 0; p < pad; ++p) __builtin_amdgcn_s_sleep(1);   // 64 cycles each
    // keep the accumulators bounded (and the operands changing) without leaving the MFMA pipe idle for long
    if ((it & 63) == 63) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB_; ++j) acc[i][j] *= 0.001f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB_; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[tid] = sum;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static std::string find_power_file(int dev) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) return "";
  for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
  DIR* d = opendir("/sys/class/drm");
  if (!d) return "";
  std::string found;
  while (dirent* e = readdir(d)) {
    if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
    char link[PATH_MAX], real[PATH_MAX];
    snprintf(link, sizeof link, "/sys/class/drm/%s/device", e->d_name);
    if (!realpath(link, real)) continue;
    const char* base = strrchr(real, '/');
    if (!base || strcmp(base + 1, bus) != 0) continue;
    std::string hw = std::string(link) + "/hwmon";
    if (DIR* h = opendir(hw.c_str())) {
      while (dirent* he = readdir(h))
        if (strncmp(he->d_name, "hwmon", 5) == 0) found = hw + "/" + he->d_name + "/power1_input";
      closedir(h);
    }
  }
  closedir(d);
  return found;
}

static double read_num(const std::string& p) {
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return -1;
  double v = -1;
  if (fscanf(f, "%lf", &v) != 1) v = -1;
  fclose(f);
  return v;
}

template <int THREADS>
static void run(const char* tag, const uint32_t* seed, int pad, float* out, unsigned long long* clk, const std::string& pfile, double seconds) {
  const int iters = 2000;
  std::atomic<bool> stop{false};
  std::vector<double> watts;
  std::thread sampler([&] {
    while (!stop.load()) {
      const double w = pfile.empty() ? -1 : read_num(pfile);
      if (w > 0) watts.push_back(w * 1e-6);
      usleep(20000);
    }
  });
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const auto t_begin = std::chrono::steady_clock::now();
  double ms_last = 0; int launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() < seconds) {
    hipEventRecord(e0);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(burn<THREADS>, dim3(256), dim3(THREADS), 0, 0, seed, iters, pad, out, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    ms_last = ms / 20; launches += 20;
  }
  stop = true; sampler.join();
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), clk, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int b = 0; b < 256; ++b) { cyc += (double)h[2 * b]; rt += (double)h[2 * b + 1]; }
  const double mhz = cyc / rt * 100.0;
  const double flop = 2.0 * 16 * 16 * 32 * NA * NB_ * (double)iters * (256.0 * THREADS / 64);
  double w_avg = 0; size_t n0 = watts.size() / 3;
  for (size_t i = n0; i < watts.size(); ++i) w_avg += watts[i];
  w_avg = watts.size() > n0 ? w_avg / (double)(watts.size() - n0) : -1;
  // share of the MFMA pipes' cycles in use at the clock held: 1024 SIMDs x 1024 FLOP per cycle
  const double tflops = flop / (ms_last * 1e-3) / 1e12;
  const double util = tflops * 1e12 / (1024.0 * 1024.0 * mhz * 1e6);
  printf("%-34s pad %2d: %7.1f TFLOP/s  clock %4.0f MHz  power %5.0f W  MFMA pipe %3.0f %% busy  (%.3f ms per launch)\n", tag, pad, tflops, mhz, w_avg,
         util * 100, ms_last);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  std::vector<uint32_t> hs(4097);
  srand(7);
  for (auto& v : hs) v = (uint32_t)rand() * 2654435761u + 12345u;
  uint32_t* seed; float* out; unsigned long long* clk;
  hipMalloc(&seed, hs.size() * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 512 * 8);
  const std::string pfile = find_power_file(0);
  printf("power sensor: %s\n", pfile.empty() ? "(not found)" : pfile.c_str());
  if (!pfile.empty()) { usleep(500000); printf("idle: %.0f W\n", read_num(pfile) * 1e-6); }
  for (int random = 1; random >= 0; --random) {
    hs[4096] = (uint32_t)random;
    hipMemcpy(seed, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    const char* t2 = random ? "random bf16, 2 waves per SIMD" : "zeros, 2 waves per SIMD";
    const char* t1 = random ? "random bf16, 1 wave per SIMD" : "zeros, 1 wave per SIMD";
    for (int pad : {0, 2, 4, 8}) run<512>(t2, seed, pad, out, clk, pfile, seconds);
    for (int pad : {0, 4}) run<256>(t1, seed, pad, out, clk, pfile, seconds);
  }
  return 0;
}
