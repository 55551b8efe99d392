// thp_probe.cpp — first-touch cost of fresh host pages with / without transparent huge pages, and hipHostMalloc's rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <chrono>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(char *p, size_t n, int T, size_t step) {
  std::vector<std::thread> th;
  const size_t per = ((n / T) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
  for (int t = 0; t < T; ++t) { const size_t o = per * t; if (o >= n) break; const size_t m = std::min(per, n - o);
    th.emplace_back([=] { for (size_t i = 0; i < m; i += step) p[o + i] = 1; }); }
  for (auto &x : th) x.join();
}
static void cat(const char *f) { FILE *fp = fopen(f, "r"); char b[256]; if (fp && fgets(b, sizeof b, fp)) printf("%s: %s", f, b); if (fp) fclose(fp); }
int main(int argc, char **argv) {
  const size_t n = (size_t)(argc > 1 ? atol(argv[1]) : 8) << 30;
  cat("/sys/kernel/mm/transparent_hugepage/enabled"); cat("/sys/kernel/mm/transparent_hugepage/defrag");
  cat("/sys/kernel/mm/transparent_hugepage/shmem_enabled"); cat("/proc/sys/vm/nr_hugepages");
  for (int huge = 0; huge < 2; ++huge) for (int T : {1, 4, 16, 64}) {
    char *p = (char *)mmap(nullptr, n + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    char *q = (char *)(((size_t)p + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1));
    if (huge) { int r = madvise(q, n, MADV_HUGEPAGE); if (r) perror("madvise HUGEPAGE"); }
    double t0 = now(); touch(q, n, T, 4096); double t1 = now();
    printf("touch fresh, %s, %2d threads: %6.2f GB/s\n", huge ? "MADV_HUGEPAGE" : "default      ", T, n / (t1 - t0) / 1e9);
    munmap(p, n + (2u << 20)); }
  for (int T : {1, 8, 32}) { char *p = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    std::vector<std::thread> th; const size_t per = n / T; double t0 = now(); int fails = 0;
    for (int t = 0; t < T; ++t) th.emplace_back([=, &fails] { if (madvise(p + per * t, per, MADV_POPULATE_WRITE)) fails++; });
    for (auto &x : th) x.join(); double t1 = now();
    printf("MADV_POPULATE_WRITE %2d threads: %6.2f GB/s (fails %d)\n", T, n / (t1 - t0) / 1e9, fails); munmap(p, n); }
  { char *p = (char *)malloc(n); double t0 = now(); touch(p, n, 16, 4096); double t1 = now();
    printf("malloc + touch 16 threads: %6.2f GB/s\n", n / (t1 - t0) / 1e9);
    double t2 = now(); free(p); printf("free: %.3f s\n", now() - t2);
    p = (char *)malloc(n); t0 = now(); touch(p, n, 16, 4096); t1 = now();
    printf("malloc again + touch 16 threads: %6.2f GB/s\n", n / (t1 - t0) / 1e9); free(p); }
  { void *h; double t0 = now(); hipError_t e = hipHostMalloc(&h, n, hipHostMallocDefault); double t1 = now();
    printf("hipHostMalloc %zu GiB: %.3f s = %6.2f GB/s (%s)\n", n >> 30, t1 - t0, n / (t1 - t0) / 1e9, hipGetErrorString(e));
    t0 = now(); (void)hipHostFree(h); printf("hipHostFree: %.3f s\n", now() - t0);
    t0 = now(); e = hipHostMalloc(&h, n, hipHostMallocDefault); t1 = now();
    printf("hipHostMalloc again: %.3f s = %6.2f GB/s\n", t1 - t0, n / (t1 - t0) / 1e9); (void)hipHostFree(h); }
  return 0;
}
