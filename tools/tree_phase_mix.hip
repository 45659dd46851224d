// Static vector-instruction mix of the tree phases' device functions as the LDS-resident search kernel instantiates them
// (two actions, K = 2, thresholds + value terms kept, MT19937): one wrapper kernel per function, everything else opaque.
// CPU-only tool (no launch): tools/tree_phase_mix.sh compiles it to ISA and counts.  (Dynamic counts differ where a
// function loops or branches: select_block's two branch kinds, choice_noreplace's retry, the backup's level count.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../include/smz.h"
#include "../stochastic-muzero_amd/csrc/smz_device.hpp"
using namespace smz;
extern __shared__ uint32_t lds_u32[];
using R = RngT<false>;

__device__ inline Params fix(Params P) {                 // the compile-time constants of the specialised instantiation
    P.A = 2; P.K = 2; P.tpw = 2; P.S = 31; P.philox = 0; P.tree0 = 0;
    P.nodes = lds_u32; P.eb_words = 12; P.rb_words = 16; P.rp_off = 8; P.tree_words = 16 + 50 * 12 + 100;
    P.thr_off = 16 + 50 * 12; P.thr_stride = 2; P.ry_off = 12;
    return P;
}
__global__ void k_select_block(Params Pin, int b, int depth, int rv, float mn, float mx, int used, int staged, uint32_t *out) {
    const Params P = fix(Pin);
    const double *pbc = reinterpret_cast<const double *>(lds_u32 + 8192);
    out[threadIdx.x] = select_block<2, true, R>(P, lds_u32 + threadIdx.x * 4, b + threadIdx.x, depth, rv, mn, mx, lds_u32 + 4096, used, staged, pbc);
}
__global__ void k_expand(Params Pin, float *pol, float rew, float val, uint32_t *out) {
    const Params P = fix(Pin);
    R rng; rng.bind(P, threadIdx.x, true);
    rng.load(P.mt, P.rng_pos[threadIdx.x], lds_u32 + 4096 + threadIdx.x * kRngStride, kRngStage);
    TreeHdr h = P.hdr[threadIdx.x];
    float lr = 0.f;
    const uint4 *rec = reinterpret_cast<const uint4 *>(lds_u32 + 6000);
    expand_backup_tree<2, 2, true, true, true, R>(P, threadIdx.x, rng, h, pol + threadIdx.x * 4, rew, val, rec, &lr);
    P.hdr[threadIdx.x] = h; out[threadIdx.x] = __float_as_uint(lr) + rng.pack();
}
__global__ void k_backup(Params Pin, int len, float val, float lrw, float *out) {
    const Params P = fix(Pin);
    float mn = out[0], mx = out[1], vr = 0.f;
    const uint4 *rec = reinterpret_cast<const uint4 *>(lds_u32 + 6000);
    backup_levels_lanes<2, true>(P, threadIdx.x & 1, threadIdx.x / 2, len, val, lrw, rec, mn, mx, vr);
    out[threadIdx.x] = mn + mx + vr;
}
__global__ void k_chase_records_leaf(Params Pin, uint32_t *out) {
    const Params P = fix(Pin);
    uint16_t *sel = reinterpret_cast<uint16_t *>(lds_u32 + 7000), *path = sel + 128;
    const int len = select_chase(sel + (threadIdx.x & 1) * 64, path + (threadIdx.x & 1) * 64);
    uint4 *rec = reinterpret_cast<uint4 *>(lds_u32 + 6000);
    for (int d = threadIdx.x >> 1; d < len; d += 32) select_record(P, lds_u32, path, d, rec);
    const Leaf L = select_leaf(P, lds_u32, path, len);
    out[threadIdx.x] = L.leaf_id + L.parent_id + L.action + L.branch + select_words(len, 2);
}
