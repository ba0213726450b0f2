// Measures the issue rate of integer vector instructions on one CU as a function of resident
// waves per SIMD: valu_rate <waves_per_simd>.  Prints wave-instructions per cycle per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEP>
__global__ void k(uint32_t *out, unsigned long long *cyc, int iters)
{
    uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x + 7, d = 0x12345 ^ threadIdx.x;
    uint32_t e = a + 11, f = b + 13, g = c + 17, h = d + 19;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll 8
        for (int u = 0; u < 8; u++) {
            if (DEP) {   // one dependent chain of 8 typical decoder instructions
                a = __builtin_amdgcn_alignbit(a, b, c);
                a = __builtin_amdgcn_ubfe(a, 3u, d & 15);
                a = (a << 2) + b;
                a = a & 0x7FFF;
                a = a + c;
                a = a > d ? b : a;
                a = a ^ c;
                a = a >> 5;
            } else {     // eight independent instructions of the same kinds
                a = __builtin_amdgcn_alignbit(a, b, c);
                e = __builtin_amdgcn_ubfe(e, 3u, d & 15);
                f = (f << 2) + b;
                g = g & 0x7FFF1;
                h = h + c;
                b = b > d ? b : g;
                c = c ^ 0x5555;
                d = d >> 1 | 0x8000;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char **argv)
{
    int wps = argc > 1 ? atoi(argv[1]) : 1;   // waves per SIMD
    int iters = 20000;
    int threads = argc > 2 ? atoi(argv[2]) : 256;   // threads per workgroup
    int blocks_per_cu = wps * 256 / threads;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    uint32_t *out; unsigned long long *cyc;
    int nblk = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)nblk * threads * 4);
    hipMalloc(&cyc, nblk * 8);
    for (int dep = 0; dep < 2; dep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (dep) k<1><<<nblk, threads>>>(out, cyc, iters); else k<0><<<nblk, threads>>>(out, cyc, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned long long> h(nblk);
        hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= nblk;
        // s_memtime ticks at 100 MHz on gfx9? report raw and assume shader clock ticks
        double instr_per_wave = (double)iters * 64;   // 8 x 8 instructions per iteration
        printf("threads %d blocks %d waves/SIMD %d dep %d: %.0f ticks, %.3f ms (%.2f GHz if ticks are cycles), %.3f wave-instr per tick per SIMD, %.2f ns per instr per SIMD\n",
               threads, nblk, wps, dep, avg, ms, avg / (ms * 1e6), instr_per_wave * wps / avg, ms * 1e6 / (instr_per_wave * wps));
    }
    return 0;
}
