// Host <-> device copy rates of pageable memory: one call against slices copied by several host threads
// (what the host-pointer entry points of the library can expect).  hipcc -O2 -o pin_rate pin_rate.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 1ull << 30;
    uint8_t *h = nullptr;
    if (posix_memalign((void **)&h, 2 << 20, n)) return 1;
    memset(h, 1, n);
    uint8_t *d = nullptr;
    if (hipMalloc((void **)&d, n) != hipSuccess) return 1;
    (void)hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    for (int threads : {1, 2, 4, 8}) {
        for (int dir = 0; dir < 2; dir++) {
            const double t0 = now();
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; t++)
                pool.emplace_back([=] {
                    const size_t a = n / threads * t, len = n / threads;
                    if (dir == 0) (void)hipMemcpy(d + a, h + a, len, hipMemcpyHostToDevice);
                    else (void)hipMemcpy(h + a, d + a, len, hipMemcpyDeviceToHost);
                });
            for (auto &th : pool) th.join();
            printf("%d thread(s) %s %.1f GB/s\n", threads, dir ? "D2H" : "H2D", n / (now() - t0) / 1e9);
        }
    }
    return 0;
}
