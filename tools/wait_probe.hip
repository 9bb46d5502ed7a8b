// wait_probe.hip -- does a host thread sleep or spin while it waits for the GPU?
// Prints CPU time / wall time of waiting for a ~20 ms kernel with different wait calls.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/wait_probe tools/wait_probe.hip ; argv[1] = 1 sets hipDeviceScheduleBlockingSync
#include <hip/hip_runtime.h>
#include <sys/resource.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void spin_kernel(unsigned long long* out, long iters) {
  unsigned long long x = threadIdx.x;
  for (long i = 0; i < iters; i++) x = x * 6364136223846793005ULL + 1442695040888963407ULL;
  out[threadIdx.x] = x;
}
static double cpu_s() {
  rusage r;
  getrusage(RUSAGE_SELF, &r);
  return r.ru_utime.tv_sec + r.ru_utime.tv_usec * 1e-6 + r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6;
}
static double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc > 1 && atoi(argv[1]) == 1) printf("hipSetDeviceFlags(hipDeviceScheduleBlockingSync) -> %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
  unsigned long long* d;
  hipMalloc(&d, 64 * 8);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t eb, en;
  hipEventCreateWithFlags(&eb, hipEventBlockingSync | hipEventDisableTiming);
  hipEventCreateWithFlags(&en, hipEventDisableTiming);
  spin_kernel<<<1, 64, 0, st>>>(d, 1000);
  hipStreamSynchronize(st);
  const long iters = 4000000;  // ~20-40 ms
  for (int mode = 0; mode < 3; mode++) {
    double c0 = cpu_s(), w0 = wall_s();
    for (int rep = 0; rep < 10; rep++) {
      spin_kernel<<<1, 64, 0, st>>>(d, iters);
      if (mode == 0) { hipEventRecord(eb, st); hipEventSynchronize(eb); }
      if (mode == 1) { hipEventRecord(en, st); hipEventSynchronize(en); }
      if (mode == 2) hipStreamSynchronize(st);
    }
    double c1 = cpu_s(), w1 = wall_s();
    const char* names[] = {"hipEventSynchronize(blocking-sync event)", "hipEventSynchronize(default event)", "hipStreamSynchronize"};
    printf("%-42s wall %.3f s  cpu %.3f s  (%.0f %% of a core)\n", names[mode], w1 - w0, c1 - c0, 100 * (c1 - c0) / (w1 - w0));
  }
  return 0;
}
