// Calibration of FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, TCC_EA0 request counters) on the access shapes of the large-batch tree
// kernel: per-LANE scattered lines (every lane of a wavefront in a different 64-byte line of a 4 GiB buffer), small pieces.
// MI355X_MICROARCH.md calibrates the counters for wide coalesced streams only ("other access widths are uncalibrated: calibrate
// on a known byte count in your own access pattern").  Build: hipcc -O2 --offload-arch=gfx950 -o tools/tcc_calibrate tools/tcc_calibrate.hip
// Run (tools/tcc_calibrate.sh): one rocprofv3 --pmc pass per counter; every kernel makes N = 2^22 accesses to distinct lines.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr uint64_t LINES = 1ull << 26;             // 64-byte lines in 4 GiB
__device__ inline uint64_t line_of(uint64_t i, uint64_t salt) { return ((i + salt) * 2654435761ull) & (LINES - 1); }   // odd multiplier: a bijection
using v4 = __attribute__((ext_vector_type(4))) uint32_t;
using v2 = __attribute__((ext_vector_type(2))) uint32_t;


__global__ void cal_read16(const uint8_t *buf, uint32_t *out, uint64_t salt) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const v4 x = *(const v4 *)(buf + line_of(i, salt) * 64);
    if (x.x == 0x12345u) out[0] = x.y;
}
__global__ void cal_read8(const uint8_t *buf, uint32_t *out, uint64_t salt) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const v2 x = *(const v2 *)(buf + line_of(i, salt) * 64 + 8);
    if (x.x == 0x12345u) out[0] = x.y;
}
__global__ void cal_read48_three_loads(const uint8_t *buf, uint32_t *out, uint64_t salt) {    // a K = 2 block as the kernels read it
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint8_t *p = buf + line_of(i, salt) * 64;
    const v4 a = *(const v4 *)p, b = *(const v4 *)(p + 16), c = *(const v4 *)(p + 32);
    if ((a.x ^ b.x ^ c.x) == 0x12345u) out[0] = a.y;
}
__global__ void cal_read64(const uint8_t *buf, uint32_t *out, uint64_t salt) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint8_t *p = buf + line_of(i, salt) * 64;
    const v4 a = *(const v4 *)p, b = *(const v4 *)(p + 16), c = *(const v4 *)(p + 32), d = *(const v4 *)(p + 48);
    if ((a.x ^ b.x ^ c.x ^ d.x) == 0x12345u) out[0] = a.y;
}
__global__ void cal_read32_half_line(const uint8_t *buf, uint32_t *out, uint64_t salt) {      // a 32-byte block, two per line
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint8_t *p = buf + line_of(i, salt) * 64 + 32 * (i & 1);
    const v4 a = *(const v4 *)p, b = *(const v4 *)(p + 16);
    if ((a.x ^ b.x) == 0x12345u) out[0] = a.y;
}
__global__ void cal_write8(uint8_t *buf, uint64_t salt) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    *(v2 *)(buf + line_of(i, salt) * 64 + 8) = v2{(uint32_t)i, 1u};
}
__global__ void cal_write16(uint8_t *buf, uint64_t salt) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    *(v4 *)(buf + line_of(i, salt) * 64 + 16) = v4{(uint32_t)i, 1u, 2u, 3u};
}
__global__ void cal_write32(uint8_t *buf, uint64_t salt) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint8_t *p = buf + line_of(i, salt) * 64;
    *(v4 *)p = v4{(uint32_t)i, 1u, 2u, 3u}; *(v4 *)(p + 16) = v4{4u, 5u, 6u, 7u};
}
__global__ void cal_write64(uint8_t *buf, uint64_t salt) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint8_t *p = buf + line_of(i, salt) * 64;
    *(v4 *)p = v4{(uint32_t)i, 1u, 2u, 3u}; *(v4 *)(p + 16) = v4{4u, 5u, 6u, 7u};
    *(v4 *)(p + 32) = v4{8u, 9u, 10u, 11u}; *(v4 *)(p + 48) = v4{12u, 13u, 14u, 15u};
}
__global__ void cal_rmw8_after_read48(uint8_t *buf, uint64_t salt) {      // descent then backup in ONE launch: read the block, store 8 bytes into it
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint8_t *p = buf + line_of(i, salt) * 64;
    const v4 a = *(const v4 *)p, b = *(const v4 *)(p + 16), c = *(const v4 *)(p + 32);
    *(v2 *)p = v2{a.x + b.x + c.x + 1u, a.y};
}
__global__ void cal_coalesced_read16(const uint8_t *buf, uint32_t *out) {                 // the guide's calibrated shape, for reference
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const v4 x = *(const v4 *)(buf + i * 16);
    if (x.x == 0x12345u) out[0] = x.y;
}
__global__ void cal_coalesced_write16(uint8_t *buf) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    *(v4 *)(buf + i * 16) = v4{(uint32_t)i, 1u, 2u, 3u};
}

// The large-batch tree kernel's shape in miniature: one tree per lane, a DEPENDENT descent over `levels` 48-byte blocks
// of the lane's own 3.3 KB tree region (the next address is computed from the loaded words), then one 8-byte store into every
// visited block -- with one block per 64-byte line (today's layout), or with the blocks of consecutive levels PAIRED in one
// line (what co-locating a node's block with its most-visited child's would give when every transition pairs: the upper bound).
template <int PAIRED>
__global__ void cal_descent(uint8_t *buf, int levels, uint64_t trees) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= trees) return;
    uint8_t *tree = buf + t * 3328;                       // 52 lines per tree, as 50 simulations of two-child blocks
    uint32_t line = 0, carry = 0;
    uint32_t at[8];
    for (int l = 0; l < levels; l++) {
        const uint32_t off = PAIRED ? (line * 64 + (l & 1) * 32) : line * 64;
        const uint8_t *p = tree + off;
        const v4 a = *(const v4 *)p, b = *(const v4 *)(p + 16);
        carry += a.x + b.y;                               // (zero-filled buffer: the value is 0, the dependency is real)
        at[l] = off;
        if (!PAIRED || (l & 1)) line = (line * 7 + 5 + carry) % 52;      // next block's line: scattered inside the tree's region
    }
    for (int l = 0; l < levels; l++) *(v2 *)(tree + at[l]) = v2{carry + (uint32_t)l, 1u};
}

int main() {
    uint8_t *buf; uint32_t *out;
    CHECK(hipMalloc(&buf, LINES * 64));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 0, LINES * 64));
    const uint64_t N = 1ull << 22;
    const dim3 g((unsigned)(N / 256)), b(256);
    // every kernel gets its own slice of the line permutation (salt): nothing it touches was touched by an earlier kernel,
    // and 4 M lines x 64 B = 256 MiB per kernel -- the buffer is 4 GiB and was just memset (far beyond L2 + Infinity Cache)
    uint64_t salt = 0;
#define RUN(k, ...) hipLaunchKernelGGL(k, g, b, 0, 0, __VA_ARGS__); CHECK(hipDeviceSynchronize()); salt += N
    for (int rep = 0; rep < 2; rep++) {
        RUN(cal_read8, buf, out, salt);
        RUN(cal_read16, buf, out, salt);
        RUN(cal_read48_three_loads, buf, out, salt);
        RUN(cal_read64, buf, out, salt);
        RUN(cal_read32_half_line, buf, out, salt);
        RUN(cal_write8, buf, salt);
        RUN(cal_write16, buf, salt);
        RUN(cal_write32, buf, salt);
        RUN(cal_write64, buf, salt);
        RUN(cal_rmw8_after_read48, buf, salt);
        hipLaunchKernelGGL(cal_coalesced_read16, g, b, 0, 0, buf + (size_t)(3ull << 30), out); CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(cal_coalesced_write16, g, b, 0, 0, buf + (size_t)(3ull << 30) + (1ull << 28)); CHECK(hipDeviceSynchronize());
    }
    {   // 1 M trees x 6 levels: one block per line vs two levels per line
        const uint64_t trees = 1ull << 20;
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(cal_descent<0>, dim3((unsigned)(trees / 64)), dim3(64), 0, 0, buf, 6, trees); CHECK(hipDeviceSynchronize());
            hipLaunchKernelGGL(cal_descent<1>, dim3((unsigned)(trees / 64)), dim3(64), 0, 0, buf, 6, trees); CHECK(hipDeviceSynchronize());
        }
    }
    printf("accesses per kernel: %llu\n", (unsigned long long)N);
    return 0;
}
