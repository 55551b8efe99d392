// pcie_probe.cpp — what the box's host <-> device path can do (tools only; not part of the library).
//   hipcc -O2 -o pcie_probe pcie_probe.cpp -lpthread
// Prints GB/s of: pageable hipMemcpy D2H into fresh / touched pages, pinned D2H, pinned H2D, host memcpy pinned -> fresh
// pageable pages with 1..16 threads, hipHostRegister of fresh pages, and a pipelined pinned-staged D2H with T copy threads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static void *fresh(size_t n) { void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); return p; }
static void par_memcpy(char *d, const char *s, size_t n, int T) {
  if (T <= 1) { memcpy(d, s, n); return; }
  std::vector<std::thread> th;
  const size_t per = ((n / T) + 4095) & ~(size_t)4095;
  for (int t = 0; t < T; ++t) {
    const size_t o = per * t; if (o >= n) break;
    const size_t m = std::min(per, n - o);
    th.emplace_back([=] { memcpy(d + o, s + o, m); });
  }
  for (auto &x : th) x.join();
}
int main(int argc, char **argv) {
  const size_t GB = argc > 1 ? atol(argv[1]) : 4;
  const size_t n = GB << 30;
  char *dev; CK(hipMalloc(&dev, n)); CK(hipMemset(dev, 1, n)); CK(hipDeviceSynchronize());
  hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
  { // warm the copy path
    char *w = (char *)fresh(1 << 20); CK(hipMemcpy(w, dev, 1 << 20, hipMemcpyDeviceToHost)); munmap(w, 1 << 20); }
  { char *p = (char *)fresh(n); double t0 = now(); CK(hipMemcpy(p, dev, n, hipMemcpyDeviceToHost)); double t1 = now();
    printf("pageable D2H fresh pages   : %6.2f GB/s\n", n / (t1 - t0) / 1e9);
    t0 = now(); CK(hipMemcpy(p, dev, n, hipMemcpyDeviceToHost)); t1 = now();
    printf("pageable D2H touched pages : %6.2f GB/s\n", n / (t1 - t0) / 1e9);
    t0 = now(); CK(hipMemcpy(dev, p, n, hipMemcpyHostToDevice)); t1 = now();
    printf("pageable H2D touched pages : %6.2f GB/s\n", n / (t1 - t0) / 1e9);
    munmap(p, n); }
  const size_t PB = 256u << 20;
  char *pin[3]; for (int i = 0; i < 3; ++i) { CK(hipHostMalloc((void **)&pin[i], PB, hipHostMallocDefault)); memset(pin[i], 0, PB); }
  { double t0 = now(); for (int i = 0; i < 8; ++i) CK(hipMemcpyAsync(pin[i & 1], dev + (size_t)i * PB, PB, hipMemcpyDeviceToHost, st[0])); CK(hipStreamSynchronize(st[0])); double t1 = now();
    printf("pinned D2H                 : %6.2f GB/s\n", 8.0 * PB / (t1 - t0) / 1e9);
    t0 = now(); for (int i = 0; i < 8; ++i) CK(hipMemcpyAsync(dev + (size_t)i * PB, pin[i & 1], PB, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); t1 = now();
    printf("pinned H2D                 : %6.2f GB/s\n", 8.0 * PB / (t1 - t0) / 1e9);
    t0 = now(); for (int i = 0; i < 8; ++i) { CK(hipMemcpyAsync(pin[0], dev + (size_t)i * PB, PB, hipMemcpyDeviceToHost, st[0])); CK(hipMemcpyAsync(dev + (size_t)(i + 8) * PB % n, pin[1], PB, hipMemcpyHostToDevice, st[1])); }
    CK(hipStreamSynchronize(st[0])); CK(hipStreamSynchronize(st[1])); t1 = now();
    printf("pinned D2H + H2D together  : %6.2f GB/s each way\n", 8.0 * PB / (t1 - t0) / 1e9); }
  for (int T : {1, 2, 4, 8, 16, 32}) {
    char *p = (char *)fresh(n); double t0 = now();
    for (size_t o = 0; o < n; o += PB) par_memcpy(p + o, pin[0], PB, T);
    double t1 = now(); printf("host memcpy pinned->fresh, %2d threads: %6.2f GB/s\n", T, n / (t1 - t0) / 1e9);
    t0 = now(); for (size_t o = 0; o < n; o += PB) par_memcpy(p + o, pin[0], PB, T); t1 = now();
    printf("host memcpy pinned->touched, %2d thr  : %6.2f GB/s\n", T, n / (t1 - t0) / 1e9);
    t0 = now(); for (size_t o = 0; o < n; o += PB) par_memcpy(pin[0], p + o, PB, T); t1 = now();
    printf("host memcpy touched->pinned, %2d thr  : %6.2f GB/s\n", T, n / (t1 - t0) / 1e9);
    munmap(p, n); }
  { char *p = (char *)fresh(n); double t0 = now(); CK(hipHostRegister(p, n, hipHostRegisterDefault)); double t1 = now();
    printf("hipHostRegister fresh pages: %6.2f GB/s (%.3f s)\n", n / (t1 - t0) / 1e9, t1 - t0);
    t0 = now(); CK(hipMemcpyAsync(p, dev, n, hipMemcpyDeviceToHost, st[0])); CK(hipStreamSynchronize(st[0])); t1 = now();
    printf("registered D2H             : %6.2f GB/s\n", n / (t1 - t0) / 1e9);
    t0 = now(); CK(hipHostUnregister(p)); t1 = now(); printf("hipHostUnregister          : %.3f s\n", t1 - t0);
    t0 = now(); CK(hipHostRegister(p, n, hipHostRegisterDefault)); t1 = now();
    printf("hipHostRegister touched    : %6.2f GB/s (%.3f s)\n", n / (t1 - t0) / 1e9, t1 - t0);
    CK(hipHostUnregister(p)); munmap(p, n); }
  // pipelined staged D2H: DMA chunk k+1 into pin[(k+1)%3] while T threads copy chunk k out
  for (size_t CH : {(size_t)32 << 20, (size_t)128 << 20}) for (int T : {4, 8, 16}) for (int touched = 0; touched < 2; ++touched) {
    char *p = (char *)fresh(n); if (touched) par_memcpy(p, dev ? pin[0] : pin[0], 0, 1), memset(p, 0, n);
    hipEvent_t ev[3]; for (auto &evx : ev) CK(hipEventCreateWithFlags(&evx, hipEventDisableTiming));
    const size_t nch = n / CH; double t0 = now();
    for (size_t k = 0; k < nch + 1; ++k) {
      if (k < nch) { CK(hipMemcpyAsync(pin[k % 3], dev + k * CH, CH, hipMemcpyDeviceToHost, st[0])); CK(hipEventRecord(ev[k % 3], st[0])); }
      if (k >= 1) { CK(hipEventSynchronize(ev[(k - 1) % 3])); par_memcpy(p + (k - 1) * CH, pin[(k - 1) % 3], CH, T); }
    }
    double t1 = now(); printf("staged D2H chunk %3zu MiB, %2d threads, %s pages: %6.2f GB/s\n", CH >> 20, T, touched ? "touched" : "fresh  ", n / (t1 - t0) / 1e9);
    munmap(p, n); }
  // pipelined staged H2D
  for (int T : {4, 8}) { const size_t CH = 64u << 20; char *p = (char *)fresh(n); memset(p, 3, n);
    hipEvent_t ev[3]; for (auto &evx : ev) CK(hipEventCreateWithFlags(&evx, hipEventDisableTiming));
    const size_t nch = n / CH; double t0 = now();
    for (size_t k = 0; k < nch; ++k) {
      if (k >= 3) CK(hipEventSynchronize(ev[k % 3]));
      par_memcpy(pin[k % 3], p + k * CH, CH, T);
      CK(hipMemcpyAsync(dev + k * CH, pin[k % 3], CH, hipMemcpyHostToDevice, st[0])); CK(hipEventRecord(ev[k % 3], st[0]));
    }
    CK(hipStreamSynchronize(st[0])); double t1 = now();
    printf("staged H2D chunk 64 MiB, %2d threads: %6.2f GB/s\n", T, n / (t1 - t0) / 1e9); munmap(p, n); }
  return 0;
}
