// LDS read cost on one CU by access shape: lds_rate <waves_per_simd>.  Every lane of every wave
// issues independent LDS reads in a loop; prints ns per wave-instruction per CU for
//   0 b32 aligned, consecutive lanes        1 b32 at byte stride 1 (unaligned, overlapping)
//   2 b64 aligned, consecutive              3 b64 at byte stride 1 (unaligned)
//   4 b32 aligned, random dwords            5 b32 random bytes (unaligned)
//   6 b64 aligned, random                   7 b64 random bytes (unaligned)
//   8 b128 aligned, random 48-byte rows     9 3 x b32 aligned covering an unaligned b64, stride 1
//  10 b64 store aligned, random            11 b64 store at random 2-byte offsets
//  12 4 x b16 stores at random 2-byte offsets   13 b128 store at random 2-byte offsets
//  14 u8 loads, random                     15 u16 loads, random
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((aligned(1))) uint32_t u32u;
typedef __attribute__((aligned(1))) uint64_t u64u;

template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[65536 + 64];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((uint32_t *)lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    uint32_t r = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll 4
        for (int u = 0; u < 4; u++) {
            r = r * 1664525u + 1013904223u;
            const uint32_t base = ((i * 4 + u) * 64) & 0x7FFF;
            if (MODE == 0) acc += *(uint32_t *)&lds[base + 4 * (threadIdx.x & 63) + 256 * (threadIdx.x >> 6)];
            if (MODE == 1) acc += *(u32u *)&lds[base + 1 + (threadIdx.x & 63) + 256 * (threadIdx.x >> 6)];
            if (MODE == 2) acc += (uint32_t)*(uint64_t *)&lds[base + 8 * (threadIdx.x & 63) + 512 * (threadIdx.x >> 6)];
            if (MODE == 3) acc += (uint32_t)*(u64u *)&lds[base + 1 + (threadIdx.x & 63) + 256 * (threadIdx.x >> 6)];
            if (MODE == 4) acc += *(uint32_t *)&lds[(r >> 16) & 0xFFFC];
            if (MODE == 5) acc += *(u32u *)&lds[(r >> 16) & 0xFFFF];
            if (MODE == 6) acc += (uint32_t)*(uint64_t *)&lds[(r >> 16) & 0xFFF8];
            if (MODE == 7) acc += (uint32_t)*(u64u *)&lds[(r >> 16) & 0xFFFF];
            if (MODE == 8) { uint4 v = *(uint4 *)&lds[((r >> 16) % 1360) * 48]; acc += v.x ^ v.w; }
            typedef __attribute__((aligned(2))) uint64_t u64h;
            typedef __attribute__((aligned(2))) uint4 u128h;
            if (MODE == 10) *(uint64_t *)&lds[(r >> 16) & 0xFFF8] = r;
            if (MODE == 11) *(u64h *)&lds[(r >> 16) & 0xFFFE] = r;
            if (MODE == 12) {
                uint16_t *q = (uint16_t *)&lds[(r >> 16) & 0xFFFE];
                q[0] = (uint16_t)r; q[1] = (uint16_t)(r >> 3); q[2] = (uint16_t)(r >> 5); q[3] = (uint16_t)(r >> 7);
            }
            if (MODE == 13) *(u128h *)&lds[(r >> 16) & 0xFFFE] = make_uint4(r, r + 1, r + 2, r + 3);
            if (MODE == 14) acc += lds[(r >> 16) & 0xFFFF];
            if (MODE == 15) acc += *(uint16_t *)&lds[(r >> 16) & 0xFFFE];
            if (MODE == 9) {
                const uint32_t a = base + 1 + (threadIdx.x & 63) + 256 * (threadIdx.x >> 6);
                const uint32_t *q = (const uint32_t *)&lds[a & ~3u];
                const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
                acc += __builtin_amdgcn_alignbyte(w1, w0, a) ^ __builtin_amdgcn_alignbyte(w2, w1, a);
            }
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int MODE>
void run(uint32_t *out, int wps, int iters)
{
    const int threads = wps * 256;   // one workgroup per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k<MODE><<<256, threads>>>(out, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr = (double)iters * 4 * (threads / 64);   // wave-level loads per CU
    printf("mode %d waves/SIMD %d: %.3f ms, %.2f ns per wave-load per CU (%.1f cycles at 2.4 GHz)\n", MODE, wps, ms,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}

int main(int argc, char **argv)
{
    int wps = argc > 1 ? atoi(argv[1]) : 4;
    int iters = 20000;
    uint32_t *out;
    hipMalloc(&out, 256 * 1024 * 4);
    run<0>(out, wps, iters); run<1>(out, wps, iters); run<2>(out, wps, iters); run<3>(out, wps, iters);
    run<4>(out, wps, iters); run<5>(out, wps, iters); run<6>(out, wps, iters); run<7>(out, wps, iters);
    run<8>(out, wps, iters); run<9>(out, wps, iters);
    run<10>(out, wps, iters); run<11>(out, wps, iters); run<12>(out, wps, iters); run<13>(out, wps, iters);
    run<14>(out, wps, iters); run<15>(out, wps, iters);
    return 0;
}
