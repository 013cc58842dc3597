// As bar_ping, with the 512-byte command block of the context server: 64 words, every eighth a tag.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_server(const unsigned long long *cmd, unsigned long long *resp, int n)
{
    const int lane = threadIdx.x;
    for (int seq = 1; seq <= n; seq++) {
        unsigned long long v;
        long spins = 0;
        for (;;) {
            v = __hip_atomic_load(cmd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const bool ok = (lane & 7) != 7 || v == (unsigned long long)seq;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ULL) break;
            if (++spins > 100000000L) return;
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 7)
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(resp), "v"(v) : "memory");
    }
}

static int run(const char *name, unsigned long long *cmd_host_view, const unsigned long long *cmd_dev_view, int n, int mode)
{
    unsigned long long *resp;
    CK(hipHostMalloc((void **)&resp, 64, hipHostMallocMapped));
    *resp = 0;
    alignas(64) unsigned long long stage[64];
    memset(stage, 0, sizeof stage);
    memcpy(cmd_host_view, stage, 512);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    k_server<<<1, 64, 0, st>>>(cmd_dev_view, resp, n);
    volatile unsigned long long *r = resp;
    auto t0 = std::chrono::steady_clock::now();
    for (int seq = 1; seq <= n; seq++) {
        for (int k = 0; k < 64; k++) stage[k] = (k & 7) == 7 ? (unsigned long long)seq : (unsigned long long)(seq * 64 + k);
        if (mode == 0) {
            memcpy(cmd_host_view, stage, 512);
        } else if (mode == 1) {
            for (int l = 7; l >= 0; l--) {
                _mm256_store_si256((__m256i *)(cmd_host_view + 8 * l), _mm256_load_si256((const __m256i *)(stage + 8 * l)));
                _mm256_store_si256((__m256i *)(cmd_host_view + 8 * l + 4), _mm256_load_si256((const __m256i *)(stage + 8 * l + 4)));
            }
        } else {
            for (int k = 0; k < 64; k++) ((volatile unsigned long long *)cmd_host_view)[k] = stage[k];
        }
        _mm_sfence();
        long spins = 0;
        while (*r != (unsigned long long)seq) if (++spins > 2000000000L) { printf("%s: timeout at %d\n", name, seq); return 1; }
    }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipStreamSynchronize(st));
    printf("%s (mode %d): %.2f us per round trip\n", name, mode, us / n);
    return 0;
}

int main()
{
    const int n = 20000;
    int large_bar = -1;
    hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    unsigned long long *pinned; CK(hipHostMalloc((void **)&pinned, 512, hipHostMallocMapped));
    for (int m = 0; m < 3; m++) if (run("pinned", pinned, pinned, n, m)) return 1;
    unsigned long long *fg = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&fg, 4096, hipDeviceMallocFinegrained);
    if (e == hipSuccess)
        for (int m = 0; m < 3; m++) if (run("BAR", fg, fg, n, m)) return 1;
    return 0;
}
