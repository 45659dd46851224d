// mfma4_probe.hip -- what v_mfma_f32_4x4x1_16B_f32 computes (lane layout, rounding) and what it costs on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -o mfma4_probe tools/mfma4_probe.hip && ./mfma4_probe
// Expected layout (16 blocks of a 4x4 outer product, K = 1): lane l = 4 * block + q holds A[block][i = q] and B[block][j = q];
// D register r of lane l = D[block][i = r][j = q] = fma(A[block][r], B[block][q], C).  The probe checks that bit for bit
// with random operands, checks that a chain of them equals the k-ordered fmaf chain, and times 1 / 2 / 4 / 5 independent
// accumulator chains per wave (one wave per SIMD and two).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float *a, const float *b, const float *c, float *d) {
    const int l = threadIdx.x;
    v4f acc = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[l * 4 + r] = acc[r];
}
__global__ void k_chain(const float *a, const float *b, const float *c, float *d, int K) {   // a, b: [K][64]
    const int l = threadIdx.x;
    v4f acc = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    for (int k = 0; k < K; k++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k * 64 + l], b[k * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[l * 4 + r] = acc[r];
}
template <int NC>
__global__ void k_time(float *out, int iters, long long *cycles) {
    const int l = threadIdx.x & 63;
    v4f acc[NC];
    float a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = 1.0f + 0.001f * (l + i); b[i] = 0.5f - 0.0001f * (l * 3 + i); }
    for (int c = 0; c < NC; c++) acc[c] = v4f{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int c = 0; c < NC; c++) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], b[(i + c) & 7], acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < NC; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NC>
static int timeit(int waves_per_wg, float *dout, long long *dcyc) {
    const int iters = 2000;
    k_time<NC><<<1, 64 * waves_per_wg>>>(dout, iters, dcyc);
    CK(hipDeviceSynchronize());
    long long c;
    CK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost));
    // ticks of s_memtime per MFMA and wave
    const double per = (double)c / ((double)iters * 8 * NC);
    printf("  chains %d, waves/WG %d: %.3f memtime ticks per MFMA per wave\n", NC, waves_per_wg, per);
    return 0;
}

int main() {
    const int K = 37;
    std::vector<float> a(K * 64), b(K * 64), c(256), d(256);
    srand(1);
    for (auto &x : a) x = (float)rand() / RAND_MAX - 0.5f;
    for (auto &x : b) x = (float)rand() / RAND_MAX - 0.5f;
    for (auto &x : c) x = (float)rand() / RAND_MAX - 0.5f;
    float *da, *db, *dc, *dd; long long *dcyc; float *dout;
    CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dc, 1024)); CK(hipMalloc(&dd, 1024));
    CK(hipMalloc(&dcyc, 8)); CK(hipMalloc(&dout, 4 * 64 * 16));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(da, db, dc, dd);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int blk = l >> 2, q = l & 3;
            const float want = fmaf(a[blk * 4 + r], b[blk * 4 + q], c[l * 4 + r]);
            if (memcmp(&want, &d[l * 4 + r], 4)) { if (bad < 5) printf("layout mismatch lane %d reg %d: %g vs %g\n", l, r, want, d[l * 4 + r]); bad++; }
        }
    printf("single MFMA vs fmaf(A[blk][r], B[blk][lane&3], C): %d mismatches of 256\n", bad);
    k_chain<<<1, 64>>>(da, db, dc, dd, K);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    bad = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int blk = l >> 2, q = l & 3;
            float want = c[l * 4 + r];
            for (int k = 0; k < K; k++) want = fmaf(a[k * 64 + blk * 4 + r], b[k * 64 + blk * 4 + q], want);
            if (memcmp(&want, &d[l * 4 + r], 4)) bad++;
        }
    printf("chain of %d MFMAs vs k-ordered fmaf chain: %d mismatches of 256\n", K, bad);
    printf("timing (dependent accumulator chains per wave):\n");
    if (timeit<1>(4, dout, dcyc) || timeit<2>(4, dout, dcyc) || timeit<3>(4, dout, dcyc) || timeit<4>(4, dout, dcyc) || timeit<5>(4, dout, dcyc)) return 1;
    if (timeit<1>(8, dout, dcyc) || timeit<2>(8, dout, dcyc) || timeit<4>(8, dout, dcyc)) return 1;
    return 0;
}
