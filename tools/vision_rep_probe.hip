// Stage accounting of k_vision_initial (s_memtime ticks of thread 0, averaged over the frames' workgroups).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o vision_rep_probe.bin tools/vision_rep_probe.hip
#ifndef NO_STAMPS        // -DNO_STAMPS: the library's kernel as it ships, launch times only
#define SMZ_VISION_STAMPS
#endif
#ifndef SMZ_SRC          // (-DSMZ_SRC='"..."': another version of the kernel source, to compare output checksums)
#define SMZ_SRC "../stochastic-muzero_amd/csrc/smz_vision.hip"
#endif
#include SMZ_SRC
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    smz_vision_desc d = {}; d.A = 2; d.S = 31; d.H = 64; d.L = 1;
    if (smz_vision_layout(&d) != 0) { printf("layout failed\n"); return 1; }
    const int B = 1024;
    std::vector<float> w(d.total_floats), f((size_t)B * 3 * 98 * 98);
    for (int i = 0; i < d.total_floats; i++) w[i] = 0.02f * ((i * 37) % 19 - 9);
    for (size_t i = 0; i < f.size(); i++) f[i] = 0.001f * (float)((i * 131) % 997);
    float *dw, *df, *oh, *op, *rec;
    CK(hipMalloc(&rec, f.size() * 4));
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&df, f.size() * 4)); CK(hipMalloc(&oh, (size_t)B * 147 * 4)); CK(hipMalloc(&op, B * 2 * 4));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(df, f.data(), f.size() * 4, hipMemcpyHostToDevice));
    const char *names[9] = {"stem", "res49 x2", "widen", "res25 x2", "pool25", "res13 x3", "pool13", "res7", "head"};
    for (int rep = 0; rep < 6; rep++) {       // odd passes: with the frame copy into a record (smz_vision_initial_record)
        unsigned long long z[12] = {};
#ifdef SMZ_VISION_STAMPS
        CK(hipMemcpyToSymbol(HIP_SYMBOL(smz_rep_stamps), z, sizeof(z)));
#endif
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int k = 0; k < 50; k++)
            if (smz_vision_initial_record(&d, dw, df, rep % 2 ? rec : nullptr, oh, op, B, nullptr) != 0) { printf("launch failed\n"); return 1; }
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
#ifdef SMZ_VISION_STAMPS
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(smz_rep_stamps), sizeof(z)));
#endif
        printf("%s launch %.1f us | ticks per frame:", rep % 2 ? "record" : "plain ", ms * 1e3 / 50);
        for (int i = 0; i < 9; i++) printf(" %s %.0f |", names[i], z[i] / (50.0 * B));
        printf("\n");
    }
    {   // bit patterns of what the last launch wrote: equal checksums between two builds = bit-identical outputs
        std::vector<uint32_t> h((size_t)B * 147), q((size_t)B * 2);
        CK(hipMemcpy(h.data(), oh, h.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(q.data(), op, q.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long c1 = 0, c2 = 0;
        for (size_t i = 0; i < h.size(); i++) c1 = c1 * 1000003ull + h[i];
        for (size_t i = 0; i < q.size(); i++) c2 = c2 * 1000003ull + q[i];
        printf("checksum hidden %016llx policy %016llx\n", c1, c2);
    }
    std::vector<float> back(f.size());
    CK(hipMemcpy(back.data(), rec, f.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < f.size(); i++) bad += back[i] != f[i];
    printf("record differs from the frames in %zu floats\n", bad);
    return 0;
}
