// d2h_probe.hip — what a host-pixel render can cost on this box (round 3): pinned allocation and registration
// times, device -> host copy rates (pageable, pinned, chunked through a small pinned ring) and host memcpy rates by thread count.
// Build + run: hipcc --offload-arch=gfx950 -O2 -pthread scripts/d2h_probe.hip -o /tmp/d2h_probe && /tmp/d2h_probe
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(char* dst, const char* src, size_t n, int threads) {
  if (threads <= 1) { std::memcpy(dst, src, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n / threads + 4095) & ~size_t(4095);
  for (int t = 0; t < threads; t++) {
    const size_t b = std::min(n, per * t), e = std::min(n, per * (t + 1));
    if (e > b) th.emplace_back([=] { std::memcpy(dst + b, src + b, e - b); });
  }
  for (auto& t : th) t.join();
}

// One-shot cases (a fresh process each: first-call costs are per process): argv[1] = memcpy | register | pretouch, argv[2] = MiB,
// argv[3] = 1: the destination is untouched (calloc), 0: touched.
static int oneshot(const char* mode, size_t n, bool untouched) {
  double t00 = now();
  CK(hipSetDevice(0));
  void* dev = nullptr;
  CK(hipMalloc(&dev, n));
  CK(hipMemset(dev, 1, n));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  double t0 = now();
  char* dst = (char*)std::calloc(n, 1);
  if (!untouched) std::memset(dst, 1, n);
  double t1 = now();
  double tp = 0.0, tr = 0.0;
  if (!std::strcmp(mode, "pretouch")) {
    std::vector<std::thread> th;
    const int T = 8;
    const size_t per = ((n / T) + 4095) & ~size_t(4095);
    for (int t = 0; t < T; t++) th.emplace_back([=] { for (size_t o = per * t; o < std::min(n, per * (t + 1)); o += 4096) { volatile char* p = dst + o; *p = *p; } });
    for (auto& t : th) t.join();
    tp = now() - t1;
  }
  double t2 = now();
  if (!std::strcmp(mode, "register")) { CK(hipHostRegister(dst, n, hipHostRegisterDefault)); tr = now() - t2; }
  double t3 = now();
  CK(hipMemcpyAsync(dst, dev, n, hipMemcpyDeviceToHost, st));
  CK(hipStreamSynchronize(st));
  double t4 = now();
  if (!std::strcmp(mode, "register")) CK(hipHostUnregister(dst));
  double t5 = now();
  std::printf("oneshot %-8s %4zu MiB %s: init %.1f ms | alloc%s %.2f | pretouch(8 thr) %.2f | register %.2f | D2H %.2f | unregister %.2f | total after init %.2f ms  (byte %d)\n", mode, n >> 20,
              untouched ? "untouched" : "touched  ", (t0 - t00) * 1e3, untouched ? "" : "+memset", (t1 - t0) * 1e3, tp * 1e3, tr * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t0) * 1e3, (int)dst[n - 1]);
  return 0;
}

// Steady state of a caller that hands a NEW destination to every render (Image::par_render returns a fresh Vec; numpy.empty):
// per iteration a fresh malloc (mmap'd: no pages), optionally touched by T threads — one zero byte per 4 KB page, write only (a read
// first would map the shared zero page and pay a second, copy-on-write fault) — then one pageable hipMemcpy.
static int steady(size_t n) {
  CK(hipSetDevice(0));
  void* dev = nullptr;
  CK(hipMalloc(&dev, n));
  CK(hipMemset(dev, 1, n));
  { char* w = (char*)std::malloc(n); std::memset(w, 0, n); CK(hipMemcpy(w, dev, n, hipMemcpyDeviceToHost)); std::free(w); }  // first-copy costs
  for (int T : {0, 1, 2, 4, 8, 16}) {
    double tt = 0.0, tc = 0.0;
    const int reps = 5;
    for (int r = 0; r < reps; r++) {
      char* dst = (char*)std::malloc(n);
      double t0 = now();
      if (T > 0) {
        std::vector<std::thread> th;
        const size_t per = ((n / T) + 4095) & ~size_t(4095);
        for (int t = 0; t < T; t++) th.emplace_back([=] { for (size_t o = per * t; o < std::min(n, per * (t + 1)); o += 4096) { volatile char* p = dst + o; *p = 0; } });
        for (auto& t : th) t.join();
      }
      double t1 = now();
      CK(hipMemcpy(dst, dev, n, hipMemcpyDeviceToHost));
      double t2 = now();
      tt += t1 - t0; tc += t2 - t1;
      std::free(dst);
    }
    std::printf("steady %4zu MiB, fresh destination per copy, %2d touch thread(s): touch %.2f ms + hipMemcpy %.2f ms = %.2f ms\n", n >> 20, T, tt / reps * 1e3, tc / reps * 1e3, (tt + tc) / reps * 1e3);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !std::strcmp(argv[1], "steady")) return steady((size_t)std::atoi(argv[2]) << 20);
  if (argc >= 4) return oneshot(argv[1], (size_t)std::atoi(argv[2]) << 20, argv[3][0] == '1');
  CK(hipSetDevice(0));
  const size_t sizes[] = {50ull << 20, 83ull << 20, 200ull << 20};
  void* dev = nullptr;
  CK(hipMalloc(&dev, 256ull << 20));
  CK(hipMemset(dev, 1, 256ull << 20));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (size_t n : sizes) {
    std::printf("== %zu MiB\n", n >> 20);
    char* page = (char*)std::malloc(n);
    std::memset(page, 0, n);  // touched
    double t0 = now();
    CK(hipMemcpy(page, dev, n, hipMemcpyDeviceToHost));
    double t1 = now();
    CK(hipMemcpy(page, dev, n, hipMemcpyDeviceToHost));
    double t2 = now();
    std::printf("pageable hipMemcpy D2H: first %.2f ms, second %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, n / (t2 - t1) / 1e9);
    void* pin = nullptr;
    t0 = now();
    CK(hipHostMalloc(&pin, n, hipHostMallocDefault));
    t1 = now();
    std::printf("hipHostMalloc: %.2f ms\n", (t1 - t0) * 1e3);
    for (int rep = 0; rep < 2; rep++) {
      t0 = now();
      CK(hipMemcpyAsync(pin, dev, n, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      t1 = now();
      std::printf("pinned D2H: %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    }
    for (int th : {1, 2, 4, 8}) {
      t0 = now();
      par_copy(page, (const char*)pin, n, th);
      t1 = now();
      std::printf("host memcpy pinned -> pageable, %d thread(s) (spawned per call): %.2f ms (%.1f GB/s)\n", th, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    }
    t0 = now();
    CK(hipHostFree(pin));
    t1 = now();
    std::printf("hipHostFree: %.2f ms\n", (t1 - t0) * 1e3);
    t0 = now();
    CK(hipHostRegister(page, n, hipHostRegisterDefault));
    t1 = now();
    std::printf("hipHostRegister(pageable, touched): %.2f ms\n", (t1 - t0) * 1e3);
    for (int rep = 0; rep < 2; rep++) {
      t0 = now();
      CK(hipMemcpyAsync(page, dev, n, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      t1 = now();
      std::printf("registered D2H: %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    }
    t0 = now();
    CK(hipHostUnregister(page));
    t1 = now();
    std::printf("hipHostUnregister: %.2f ms\n", (t1 - t0) * 1e3);
    // fresh (untouched) destination, as a Vec::with_capacity / numpy.empty would hand over
    char* fresh = (char*)std::malloc(n);
    t0 = now();
    CK(hipMemcpy(fresh, dev, n, hipMemcpyDeviceToHost));
    t1 = now();
    std::printf("pageable hipMemcpy D2H into untouched malloc: %.2f ms\n", (t1 - t0) * 1e3);
    std::free(fresh);
    // ring: R slots of B bytes, copy thread(s) drain
    for (size_t B : {2ull << 20, 4ull << 20, 8ull << 20}) {
      for (int T : {1, 2, 4}) {
        const int R = 8;
        void* ring = nullptr;
        double ta = now();
        CK(hipHostMalloc(&ring, R * B, hipHostMallocDefault));
        double tb = now();
        std::vector<hipEvent_t> ev(R);
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const size_t chunks = (n + B - 1) / B;
        std::vector<int> done(chunks, 0);
        // simple protocol: main issues copy c when slot free (chunk c - R drained), workers t handle chunks c % T == t
        std::vector<std::atomic<int>> issued(chunks), drained(chunks);
        for (size_t c = 0; c < chunks; c++) { issued[c] = 0; drained[c] = 0; }
        t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
          th.emplace_back([&, t] {
            for (size_t c = t; c < chunks; c += T) {
              while (!issued[c].load(std::memory_order_acquire)) std::this_thread::yield();
              (void)hipEventSynchronize(ev[c % R]);
              const size_t off = c * B, len = std::min(B, n - off);
              std::memcpy(page + off, (char*)ring + (c % R) * B, len);
              drained[c].store(1, std::memory_order_release);
            }
          });
        for (size_t c = 0; c < chunks; c++) {
          if (c >= (size_t)R) while (!drained[c - R].load(std::memory_order_acquire)) std::this_thread::yield();
          const size_t off = c * B, len = std::min(B, n - off);
          (void)hipMemcpyAsync((char*)ring + (c % R) * B, (char*)dev + off, len, hipMemcpyDeviceToHost, st);
          (void)hipEventRecord(ev[c % R], st);
          issued[c].store(1, std::memory_order_release);
        }
        for (auto& t : th) t.join();
        t1 = now();
        std::printf("ring %d x %zu MiB, %d copier thread(s): %.2f ms (%.1f GB/s)  [ring hipHostMalloc %.2f ms]\n", R, B >> 20, T, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (tb - ta) * 1e3);
        for (auto& e : ev) (void)hipEventDestroy(e);
        CK(hipHostFree(ring));
      }
    }
    std::free(page);
  }
  return 0;
}
