// Phase accounting of k_vision_recurrent (s_memtime ticks summed over leaf wavefronts): reward tower | transition convs |
// prediction.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o vision_probe vision_probe.hip
#define SMZ_VISION_STAMPS
#include "../stochastic-muzero_amd/csrc/smz_vision.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    smz_vision_desc d = {}; d.A = 2; d.S = 31; d.H = 64; d.L = 1;
    if (smz_vision_layout(&d) != 0) { printf("layout failed\n"); return 1; }
    const int B = 1024;
    std::vector<float> w(d.total_floats), h((size_t)B * 147);
    for (int i = 0; i < d.total_floats; i++) w[i] = 0.02f * ((i * 37) % 19 - 9);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0.001f * (float)((i * 131) % 997);
    std::vector<int32_t> act(B); std::vector<uint8_t> br(B);
    for (int i = 0; i < B; i++) { act[i] = i & 1; br[i] = (i >> 1) & 1; }
    float *dw, *dh, *oh, *orw, *op, *ov; int32_t *da; uint8_t *db;
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dh, h.size() * 4)); CK(hipMalloc(&oh, h.size() * 4));
    CK(hipMalloc(&orw, B * 4)); CK(hipMalloc(&op, B * 2 * 4)); CK(hipMalloc(&ov, B * 4)); CK(hipMalloc(&da, B * 4)); CK(hipMalloc(&db, B));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(da, act.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, br.data(), B, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; rep++) {
        unsigned long long z[8] = {};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(smz_vision_stamps), z, sizeof(z)));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int k = 0; k < 20; k++)
            if (smz_vision_recurrent(&d, dw, dh, 147, da, db, oh, orw, op, ov, B, nullptr) != 0) { printf("launch failed\n"); return 1; }
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(smz_vision_stamps), sizeof(z)));
        const double n = 20.0 * B;
        printf("launch %.1f us | ticks per leaf: reward tower %.0f | x->plane %.0f conv_in %.0f resblocks %.0f scale+store %.0f | prediction %.0f\n",
               ms * 1e3 / 20, z[0] / n, z[3] / n, z[4] / n, z[5] / n, z[1] / n, z[2] / n);
    }
    return 0;
}
