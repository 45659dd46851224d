// Micro-probe: what one dependent global round trip costs in the tree kernels' launch shape (64 waves of 64 lanes,
// one per CU) and what an interleaved fire-and-forget store adds (gfx950 counts stores in vmcnt).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void chase(const int *next, int *sink, int hops, int with_store, int *scratch) {
    int i = blockIdx.x * 64 + threadIdx.x;
    int cur = i * 977 % (1 << 20);
    for (int h = 0; h < hops; h++) {
        cur = next[cur];
        if (with_store) scratch[(size_t)i * 64 + (h & 63)] = cur;
    }
    sink[i] = cur;
}
__global__ void chase_lds(int *sink, int hops) {
    __shared__ int tab[4096];
    for (int k = threadIdx.x; k < 4096; k += 64) tab[k] = (k * 61 + 17) & 4095;
    __syncthreads();
    int cur = threadIdx.x;
    for (int h = 0; h < hops; h++) cur = tab[cur];
    sink[blockIdx.x * 64 + threadIdx.x] = cur;
}
int main() {
    const int N = 1 << 22;  // 16 MB table
    std::vector<int> h(N);
    unsigned s = 12345;
    for (int i = 0; i < N; i++) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % (1 << 20); }
    int *d_next, *d_sink, *d_scr;
    CK(hipMalloc(&d_next, N * 4)); CK(hipMalloc(&d_sink, 1 << 20)); CK(hipMalloc(&d_scr, (size_t)65536 * 64 * 4));
    CK(hipMemcpy(d_next, h.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int blocks : {64, 256, 1024}) {
        for (int ws = 0; ws < 2; ws++) {
            for (int hops : {16, 64}) {
                float best = 1e9;
                for (int rep = 0; rep < 20; rep++) {
                    CK(hipEventRecord(a));
                    hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d_next, d_sink, hops, ws, d_scr);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
                }
                printf("blocks %4d store %d hops %3d : %.2f us total, %.3f us/hop\n", blocks, ws, hops, best * 1e3, best * 1e3 / hops);
            }
        }
    }
    for (int hops : {64, 1024}) {
        float best = 1e9;
        for (int rep = 0; rep < 20; rep++) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(chase_lds, dim3(64), dim3(64), 0, 0, d_sink, hops);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("lds hops %4d : %.2f us total, %.4f us/hop\n", hops, best * 1e3, best * 1e3 / hops);
    }
    // back-to-back launches to see what a warm clock gives
    CK(hipEventRecord(a));
    for (int rep = 0; rep < 200; rep++) hipLaunchKernelGGL(chase, dim3(64), dim3(64), 0, 0, d_next, d_sink, 64, 0, d_scr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("200 back-to-back 64-hop launches: %.2f us each, %.3f us/hop\n", ms * 1e3 / 200, ms * 1e3 / 200 / 64);
    return 0;
}
