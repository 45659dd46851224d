// smz_vision_search.hip -- the whole Monte_carlo_tree_search.run (mcts:311-349) of every tree in ONE launch for the
// `vision_model` family (neural_network_vision_model.py:41-515): smz_search_vision / smz_search_vision_act.
//
// Why a second single-launch kernel: step-wise, a vision simulation round is two launches (k_vision_recurrent 20 us +
// k_expand_backup 14 us at 1024 trees) and every leaf wavefront streams ~180 KB of tower weights out of L2 -- 180 MB per
// round, the L2 running at 8.5 TB/s for weights that never change.  Here a workgroup of 8 wavefronts owns 16 trees for
// the whole search:
//   * tree phases (expand + backup, select) as in k_search_mlp: the tree's lane, path records and staged random words in LDS;
//   * the convolutional part of a leaf's networks (3x7x7 hidden state, one pixel per lane) by the leaf's own wavefront,
//     the code of the wave-per-leaf kernel (smz_vision_device.hpp);
//   * the five 147 -> H -> [H ->] S/A towers (dynamics reward | prediction value, policy | afterstate-prediction value,
//     policy) for all 16 leaves AT ONCE on the matrix cores: v_mfma_f32_16x16x4_f32 with A = 16 output neurons x 4 inputs
//     of a weight matrix, B = 4 inputs x 16 leaves from a k-major LDS tile, C = bias.  The first-layer matrices (740 of
//     the 1188 fragments: 111 VGPRs per wave) stay in REGISTERS for the whole search; the small second / output matrices
//     are re-read from L2 once per round and workgroup (112 KB instead of 16 x 180 KB).  A tower no leaf of the workgroup
//     needs this round (all leaves on one branch) is skipped.
// f32 in, f32 accumulate: an f32-input MFMA is a k-ordered fma chain, exactly what dense_stream() of the wave-per-leaf
// kernel computes -- the two paths give bit-identical searches (tests/test_gpu_end_to_end.py).
// Limits of this kernel (anything else runs step-wise): maxium_action_sample == 2, A <= 4, S <= 32, H <= 64,
// SMZ_RNG_MT19937_NUMPY, LDS working set <= 160 KB (about 250 simulations).
#define SMZ_PART 5
#include "smz_kernels.hip"
#include "smz_vision_device.hpp"

using smz_mlp::decode_lanes;
using smz_mlp::lds_sync;
using smz_mlp::softmax_lanes;
using smz_mlp::up4;
using smz_vision::conv3x3;
using smz_vision::kC;
using smz_vision::kFlat;
using smz_vision::kN;
using smz_vision::kPad;
using smz_vision::kPix;
using smz_vision::kSmallMax;
using smz_vision::residual_block;
using smz_vision::scale_channels;
using smz_vision::uniform_ptr;

namespace {

constexpr int kVL = 16;                              // columns of an MFMA tile = leaf slots of a workgroup
constexpr int kTowers = 5;                           // 0 dyn reward | 1 pre value | 2 pre policy | 3 apr value | 4 apr policy
constexpr int kFk = 148, kYS = 36;                   // flat inputs per tower (147 + pad); raw-output row stride (floats)
constexpr int kJobs = 4 * kTowers;                   // (tower, output tile of 16 neurons) pairs of a 64-wide layer
constexpr int kOutJobs = 8;                          // output-layer (tower, tile) pairs: (0,0) (0,1) (1,0) (1,1) (2,0) (3,0) (3,1) (4,0)
#ifndef SMZ_VISION_ROLL
#define SMZ_VISION_ROLL 1
#endif
constexpr bool kRollConv = SMZ_VISION_ROLL != 0;     // nine convolution taps as a rolled loop (register relief) or unrolled
// Geometry: VW wavefronts x VT trees per workgroup (VW * VT <= 16 leaves; the other tile columns stay zero).
//   <4, 1>: one wavefront per SIMD -> 512 VGPRs each: ALL tower fragments of the wave's jobs stay in registers for the whole
//           search (5 first-layer + 5 hidden-layer + 2 output jobs = 297 VGPRs); 4 trees per workgroup, so 1024 trees fill
//           the 256 CUs.  Three quarters of every tile's columns are empty -- the matrix pipe has nothing else to do.
//   <8, 2>: 16 trees per workgroup, full tiles, 256 VGPRs per wave: one resident first-layer job, the rest re-read from L2.
template <int VW> struct VGeo {
    static constexpr int r1 = VW == 4 ? 5 : 1;       // resident first-layer jobs per wave
    static constexpr int j1 = (kJobs + VW - 1) / VW; // first-layer (and hidden-layer) jobs per wave
    static constexpr int rm = VW == 4 ? 5 : 0;       // resident hidden-layer jobs
    static constexpr int jo = kOutJobs / VW;         // output jobs per wave
    static constexpr int ro = VW == 4 ? 2 : 0;       // resident output jobs
};
typedef float v4f __attribute__((ext_vector_type(4)));

struct VisLds {                                      // float offsets from the dynamic LDS base
    int small, pbc, wave, per_wave, plane, pv, rng, outs, F, H1, H2, Y, br, total;
};
__host__ __device__ inline VisLds vis_lds(const Params &P, int A, int VW, int VT) {
    VisLds m;
    m.small = 0;
    m.pbc = kSmallMax + 10 * 108;                                 // (+ tap-major copies of the ten 3x3 convolution pieces)
    m.wave = m.pbc + r4(2 * 2 * (P.sims + 2));
    m.plane = 0;                                                  // float4 plane[81] -> 324 floats
    m.pv = r4(kPad * kPad * 4);
    m.rng = m.pv + VT * P.P * 4;
    m.outs = m.rng + r4(VT * kRngStride);
    m.per_wave = m.outs + r4(VT * (A + 2));
    m.F = m.wave + VW * m.per_wave;
    m.H1 = m.F + 3 * kFk * kVL;
    m.H2 = m.H1 + kTowers * 64 * kVL;
    m.Y = m.H2 + kTowers * 64 * kVL;
    m.br = m.Y + kTowers * kVL * kYS + 64;                        // (+64: the tails read a full wave width of a row)
    m.total = m.br + kVL;
    return m;
}

__device__ inline int tower_off(int t) {             // index of a tower's six offsets (W1,b1,Wm,bm,Wo,bo) in smz_vision_desc::off
    return t == 0 ? SMZ_V_TRANS_BASE + SMZ_VT_TOWER
                  : SMZ_V_PRED_BASE + (t >= 3 ? SMZ_V_PRED_STRIDE : 0) + ((t == 1 || t == 3) ? SMZ_VP_VTOWER : SMZ_VP_PTOWER);
}
__device__ inline int tower_input(int t) { return t == 0 ? 0 : ((t == 1 || t == 3) ? 1 : 2); }   // which flat tile feeds it
__device__ inline bool tower_dyn(int t) { return t < 3; }

// element (k, o) of a packed tower matrix (4-way interleaved input-major, OP = 64); zero beyond its K rows
__device__ inline float tw(const float *base, int k, int o, int K4) { return k < K4 ? base[((k >> 2) * kWave + o) * 4 + (k & 3)] : 0.f; }

// NJ (1 or 2) (tower, tile) jobs of a layer at once: KSTEPS MFMA steps each over the k-major tiles bt[j] with A fragments
// wf[j].  Two jobs give two independent accumulator chains (an f32 MFMA has a 40-cycle dependent latency against a 32-cycle
// issue interval), and the B operands of chunk c + 1 are read from LDS before the products of chunk c are issued.
template <int KSTEPS, int NJ>
__device__ inline void tower_layers(const float *const (&wf)[NJ], const float *const (&bt)[NJ], v4f (&acc)[NJ], int g, int i) {
    constexpr int CH = 8, NCH = (KSTEPS + CH - 1) / CH;
    float b[2][NJ][CH];
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int kk = 0; kk < CH; kk++) if (kk < KSTEPS) b[0][j][kk] = bt[j][(4 * kk + g) * kVL + i];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int kk = 0; kk < CH; kk++)
                    if ((c + 1) * CH + kk < KSTEPS) b[(c + 1) & 1][j][kk] = bt[j][(4 * ((c + 1) * CH + kk) + g) * kVL + i];
        }
#pragma unroll
        for (int kk = 0; kk < CH; kk++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
                if (c * CH + kk < KSTEPS) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][c * CH + kk], b[c & 1][j][kk], acc[j], 0, 0, 0);
    }
}
template <int KSTEPS>
__device__ inline v4f tower_layer(const float (&wf)[KSTEPS], const float *bt, v4f acc, int g, int i) {
    const float *const w[1] = {wf}, *const t[1] = {bt};
    v4f a[1] = {acc};
    tower_layers<KSTEPS, 1>(w, t, a, g, i);
    return a[0];
}
// A fragments of one (tower, tile) job straight from the packed buffer in L2
template <int KSTEPS>
__device__ inline void load_frags(float (&wf)[KSTEPS], const float *W, int o, int g, int K4) {
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) wf[kk] = tw(W, 4 * kk + g, o, K4);
}

__device__ inline void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // LDS hand-offs only

// 1x1 convolution (with bias) of CIN channels -> column `leaf` of a k-major flat tile (torch's Flatten of [3,7,7])
template <int CIN>
__device__ inline void mix_to_tile(float *tile, int leaf, int p, bool active, const float (&x)[4], const float *__restrict__ w,
                                   const float *__restrict__ b) {
#pragma unroll
    for (int oc = 0; oc < kC; oc++) {
        float s = 0.f;
#pragma unroll
        for (int ic = 0; ic < CIN; ic++) s = fmaf(w[oc * CIN + ic], x[ic], s);
        s += b[oc];
        if (active) tile[(oc * kPix + p) * kVL + leaf] = s;
    }
}

extern __shared__ float4 smz_vsearch_lds4[];

__device__ inline void out_job(int j, int &t, int &mt) {   // (tower, tile) of output job j
    t = j < 2 ? 0 : (j < 4 ? 1 : (j == 4 ? 2 : (j < 7 ? 3 : 4)));
    mt = (j == 1 || j == 3 || j == 6) ? 1 : 0;
}

template <int MAXA, int VW, int VT>
__global__ void __launch_bounds__(VW *kWave) k_search_vision(Params Pin, smz_vision_desc d, const float *__restrict__ weights,
                                                             const float *__restrict__ hidden0, const float *__restrict__ policy0,
                                                             int train, ActOut act) {
    constexpr int KS = 2;
    using G = VGeo<VW>;
    static_assert(VW * VT <= kVL && (VT == 1 || VT == 2), "workgroup geometry");
    Params P = Pin;
    P.K = KS; P.tpw = VT;
    fix_layout(P, false, true);
    float *lds = reinterpret_cast<float *>(smz_vsearch_lds4);
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int A = P.A, S = d.S, K4h = up4(d.H);
    const VisLds ml = vis_lds(P, A, VW, VT);
    // ---- one-time staging: small weights, pb_c table (+ reciprocals), zeroed tiles ---------------------------------------
    for (int i = threadIdx.x; i < d.small_floats / 4; i += blockDim.x)
        reinterpret_cast<float4 *>(lds + ml.small)[i] = reinterpret_cast<const float4 *>(weights)[i];
    double *pbc_lds = reinterpret_cast<double *>(lds + ml.pbc);
    const int n_pbc = P.sims + 2;
    for (int i = threadIdx.x; i < n_pbc; i += blockDim.x) {
        pbc_lds[i] = P.pbc_sqrt[i];
        pbc_lds[n_pbc + i] = i > 0 ? 1.0 / (double)i : 0.0;
    }
    for (int i = threadIdx.x + ml.wave; i < ml.total; i += blockDim.x) lds[i] = 0.f;     // planes' borders, pad rows / empty columns
    // tap-major copies of the 3x3 convolutions: [net 0..1 transition: conv_in, res_a, res_b | net 0..1 prediction: res_a, res_b]
    float *tm = lds + kSmallMax;
    for (int n = 0; n < 2; n++) {
        const int32_t *o = d.off + SMZ_V_TRANS_BASE + n * SMZ_V_TRANS_STRIDE, *q = d.off + SMZ_V_PRED_BASE + n * SMZ_V_PRED_STRIDE;
        smz_vision::tap_major<4>(tm + (n * 3 + 0) * 108, weights + o[SMZ_VT_CONV_IN], threadIdx.x, blockDim.x);
        smz_vision::tap_major<3>(tm + (n * 3 + 1) * 108, weights + o[SMZ_VT_RES_A], threadIdx.x, blockDim.x);
        smz_vision::tap_major<3>(tm + (n * 3 + 2) * 108, weights + o[SMZ_VT_RES_B], threadIdx.x, blockDim.x);
        smz_vision::tap_major<3>(tm + (6 + n * 2 + 0) * 108, weights + q[SMZ_VP_RES_A], threadIdx.x, blockDim.x);
        smz_vision::tap_major<3>(tm + (6 + n * 2 + 1) * 108, weights + q[SMZ_VP_RES_B], threadIdx.x, blockDim.x);
    }
    // convolution / batch-norm / 1x1 weights from the workgroup's LDS copy (as the wave-per-leaf kernel: through the scalar
    // cache a convolution measured 2x slower here too)
    const float *small = lds + ml.small;
    float *wl = lds + ml.wave + wave * ml.per_wave;
    float4 *plane = reinterpret_cast<float4 *>(wl + ml.plane);
    uint4 *pvals = reinterpret_cast<uint4 *>(wl + ml.pv);
    uint32_t *rng_tile = reinterpret_cast<uint32_t *>(wl + ml.rng);
    float *outs = wl + ml.outs;
    float *F = lds + ml.F, *H1 = lds + ml.H1, *H2 = lds + ml.H2, *Y = lds + ml.Y;
    int *br = reinterpret_cast<int *>(lds + ml.br);
    const int slot = A + 2;
    const int g = lane >> 4, li = lane & 15;
    // ---- register-resident fragments of this wave's jobs (job q of a layer = pair number wave + VW * q) -------------------
    float w1[G::r1][37], wmr[G::rm > 0 ? G::rm : 1][16], wor[G::ro > 0 ? G::ro : 1][16];
#pragma unroll
    for (int q = 0; q < G::r1; q++) {
        const int j = wave + VW * q;
        load_frags<37>(w1[q], weights + d.off[tower_off(j >> 2)], 16 * (j & 3) + li, g, kFk);
    }
#pragma unroll
    for (int q = 0; q < G::rm; q++) {
        const int j = wave + VW * q;
        load_frags<16>(wmr[q], weights + d.off[tower_off(j >> 2) + 2], 16 * (j & 3) + li, g, d.L > 0 ? K4h : 0);
    }
#pragma unroll
    for (int q = 0; q < G::ro; q++) {
        int t, mt;
        out_job(wave + VW * q, t, mt);
        load_frags<16>(wor[q], weights + d.off[tower_off(t) + 4], 16 * mt + li, g, K4h);
    }
    __syncthreads();

    const int tree0 = (blockIdx.x * VW + wave) * VT;
    const int tree = tree0 + lane;
    const bool valid = lane < VT && tree < P.B && tree_active(P, tree);
    const bool live0 = __shfl((int)valid, 0) != 0, live1 = VT > 1 && __shfl((int)valid, 1) != 0;
    const bool active = lane < kPix;
    const int p = active ? lane : kPix - 1, pp = (p / kN + 1) * kPad + (p % kN + 1);

    // ---- root: hidden state and policy come from smz_vision_initial (one workgroup per frame) ---------------------------
#pragma unroll
    for (int r = 0; r < VT; r++) {
        if (!(r ? live1 : live0)) continue;                              // wave-uniform
        const int row = tree0 + r;
        for (int k = lane; k < kFlat; k += kWave) P.hidden[(size_t)row * P.N * P.hs + k] = hidden0[(size_t)row * kFlat + k];
        if (lane < A) outs[r * slot + lane] = policy0[(size_t)row * A + lane];
    }
    lds_sync();
    int packed = wave_stage_rng<false>(P, tree, valid, rng_tile);
    RngMt rng;
    rng.bind(P, tree, valid);
    TreeHdr h = {0, 0, 0.f, 0.f, 0, 0.f, 0, 0};
    if (valid) {
        rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
        root_init_tree<MAXA>(P, tree, rng, outs + lane * slot, nullptr, train != 0);
        h = P.hdr[tree];
        packed = rng.pack();
    }
    unsigned n_dec = 0, n_chance = 0, n_children = 0;
    if (P.sims > 0) packed = wave_stage_rng_from<4, false>(P, tree, valid, rng_tile, packed);

    // (SMZ_DEBUG_SKIP=16 with statistics on: s_memtime phase accounting -> stats[4..7] = tree | conv | wait | towers + tails)
    const bool prof = (P.dbg & 16) && P.stats;
    unsigned long long t_tree = 0, t_conv = 0, t_wait = 0, t_tow = 0, t0 = 0, t1 = 0;
#define SMZ_VSTAMP(acc) if (prof) { t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
    // ---- simulations -----------------------------------------------------------------------------------------------------
    for (int s = 0; s < P.sims; s++) {
        if (prof) t0 = __builtin_amdgcn_s_memtime();
        Leaf L = {0, 0, 0, 0};
        if (valid) {
            rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
            if (s > 0) expand_backup_tree<MAXA, KS>(P, tree, rng, h, outs + lane * slot, outs[lane * slot + A + 1],
                                                    outs[lane * slot + A], pvals + lane * P.P);
            int len = 0;
            L = select_tree<MAXA, KS, false, true>(P, tree, rng, h, pbc_lds, len, n_dec, n_chance, n_children, pvals + lane * P.P);
            h.path_len = len;
            packed = rng.pack();
        }
        SMZ_VSTAMP(t_tree)
        // hidden rows written in earlier rounds (by this wave) may be this round's parent rows
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        StagePre<VT> pre;
        stage_issue<VT, false>(P, tree, valid, packed, pre);
        // ---- convolutional part, one leaf after the other: flat inputs of the leaf's towers into the k-major tiles ---------
        const bool dyn0 = __builtin_amdgcn_readlane(L.branch, 0) != 0, dyn1 = VT > 1 && __builtin_amdgcn_readlane(L.branch, 1) != 0;
#pragma unroll 1
        for (int r = 0; r < VT; r++) {                                   // (rolled: one copy of the convolution code)
            const int leaf = VT * wave + r;
            const int parent = __shfl(L.parent_id, r), actn = __shfl(L.action, r);
            const int leaf_id = __shfl(L.leaf_id, r);
            const bool dyn = r ? dyn1 : dyn0, liv = r ? live1 : live0;
            if (lane == 0) br[leaf] = liv ? (dyn ? 1 : 0) : -1;
            if (!liv) continue;                                          // wave-uniform: the column stays zero
            const int row = tree0 + r;
            const float *hrow = P.hidden + ((size_t)row * P.N + parent) * P.hs;
            const float a_plane = (float)(actn + 1) / (float)d.A;        // muzero_model.py:511-522
            float x[4] = {hrow[p], hrow[kPix + p], hrow[2 * kPix + p], a_plane};
            const int32_t *o = d.off + SMZ_V_TRANS_BASE + (dyn ? 0 : SMZ_V_TRANS_STRIDE);
            if (dyn) mix_to_tile<4>(F, leaf, p, active, x, uniform_ptr(small, o[SMZ_VT_MIX_W]), uniform_ptr(small, o[SMZ_VT_MIX_B]));
            if (active) plane[pp] = make_float4(x[0], x[1], x[2], x[3]);
            lds_sync();
            float t[kC];
            const int ni = dyn ? 0 : 1;
            smz_vision::conv3x3_t<4>(plane, pp, reinterpret_cast<const float4 *>(tm + (ni * 3 + 0) * 108), t);
            lds_sync();
            {
                const float *bn = uniform_ptr(small, o[SMZ_VT_BN_IN]);
#pragma unroll
                for (int c = 0; c < kC; c++) t[c] = fmaxf(t[c] * bn[c] + bn[kC + c], 0.f);
            }
            {
                const float4 *wa = reinterpret_cast<const float4 *>(tm + (ni * 3 + 1) * 108), *wb = reinterpret_cast<const float4 *>(tm + (ni * 3 + 2) * 108);
                const float *bn = uniform_ptr(small, o[SMZ_VT_RES_BN]);
                for (int i = 0; i < d.L; i++) smz_vision::residual_block_t(plane, pp, active, wa, wb, bn, t);
            }
#pragma unroll
            for (int c = 0; c < kC; c++) t[c] = fmaxf(t[c], 0.f);
            scale_channels(t);
            if (active) {
                float *ho = P.hidden + ((size_t)row * P.N + leaf_id) * P.hs;
                ho[p] = t[0]; ho[kPix + p] = t[1]; ho[2 * kPix + p] = t[2];
            }
            const int32_t *q = d.off + SMZ_V_PRED_BASE + (dyn ? 0 : SMZ_V_PRED_STRIDE);
            {
                const float4 *wa = reinterpret_cast<const float4 *>(tm + (6 + ni * 2 + 0) * 108), *wb = reinterpret_cast<const float4 *>(tm + (6 + ni * 2 + 1) * 108);
                const float *bn = uniform_ptr(small, q[SMZ_VP_RES_BN]);
                for (int i = 0; i < d.L; i++) smz_vision::residual_block_t(plane, pp, active, wa, wb, bn, t);
            }
            const float xs[4] = {t[0], t[1], t[2], 0.f};
            mix_to_tile<kC>(F + kFk * kVL, leaf, p, active, xs, uniform_ptr(small, q[SMZ_VP_VMIX_W]), uniform_ptr(small, q[SMZ_VP_VMIX_B]));
            mix_to_tile<kC>(F + 2 * kFk * kVL, leaf, p, active, xs, uniform_ptr(small, q[SMZ_VP_PMIX_W]), uniform_ptr(small, q[SMZ_VP_PMIX_B]));
        }
        SMZ_VSTAMP(t_conv)
        wg_barrier();                                                    // flat tiles and branch flags complete
        SMZ_VSTAMP(t_wait)
        const int mine = br[li];
        const bool need_dyn = __ballot(mine == 1) != 0ull, need_ady = __ballot(mine == 0) != 0ull;
        // ---- towers, layer 1 ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < G::r1; q += 2) {                             // register-resident jobs, two chains at a time
            const int j0 = wave + VW * q, j1 = wave + VW * (q + 1);
            const bool two = q + 1 < G::r1;
            const int t0 = j0 >> 2, t1 = two ? j1 >> 2 : t0;
            const bool n0 = tower_dyn(t0) ? need_dyn : need_ady, n1 = two && (tower_dyn(t1) ? need_dyn : need_ady);   // wave-uniform
            const float *bv0 = weights + d.off[tower_off(t0) + 1] + 16 * (j0 & 3) + 4 * g;
            const float *bv1 = weights + d.off[tower_off(t1) + 1] + 16 * (j1 & 3) + 4 * g;
            v4f acc[2] = {v4f{bv0[0], bv0[1], bv0[2], bv0[3]}, v4f{bv1[0], bv1[1], bv1[2], bv1[3]}};
            if (n0 && n1) {
                const float *const w[2] = {w1[q], w1[two ? q + 1 : q]};
                const float *const t[2] = {F + tower_input(t0) * kFk * kVL, F + tower_input(t1) * kFk * kVL};
                tower_layers<37, 2>(w, t, acc, g, li);
            } else if (n0) {
                acc[0] = tower_layer<37>(w1[q], F + tower_input(t0) * kFk * kVL, acc[0], g, li);
            } else if (n1) {
                acc[1] = tower_layer<37>(w1[two ? q + 1 : q], F + tower_input(t1) * kFk * kVL, acc[1], g, li);
            }
            if (n0) {
                float *dst = H1 + t0 * 64 * kVL + (16 * (j0 & 3) + 4 * g) * kVL + li;
#pragma unroll
                for (int r = 0; r < 4; r++) dst[r * kVL] = fmaxf(acc[0][r], 0.f);
            }
            if (n1) {
                float *dst = H1 + t1 * 64 * kVL + (16 * (j1 & 3) + 4 * g) * kVL + li;
#pragma unroll
                for (int r = 0; r < 4; r++) dst[r * kVL] = fmaxf(acc[1][r], 0.f);
            }
        }
#pragma unroll 1
        for (int q = G::r1; q < G::j1; q++) {                            // jobs whose fragments are re-read from L2
            const int j = wave + VW * q, t = j >> 2, mt = j & 3;
            if (j < kJobs && (tower_dyn(t) ? need_dyn : need_ady)) {
                const int32_t *o = d.off + tower_off(t);
                const float *bv = weights + o[1] + 16 * mt + 4 * g;
                float wt[37];
                load_frags<37>(wt, weights + o[0], 16 * mt + li, g, kFk);
                const v4f acc = tower_layer<37>(wt, F + tower_input(t) * kFk * kVL, v4f{bv[0], bv[1], bv[2], bv[3]}, g, li);
                float *dst = H1 + t * 64 * kVL + (16 * mt + 4 * g) * kVL + li;
#pragma unroll
                for (int r = 0; r < 4; r++) dst[r * kVL] = fmaxf(acc[r], 0.f);
            }
        }
        wg_barrier();
        // ---- hidden layers (the SAME Linear(H,H) applied L times) and the output layer ------------------------------------------
        float *hin = H1, *hout = H2;
        for (int l = 0; l < d.L; l++) {
#pragma unroll
            for (int q = 0; q < G::rm; q += 2) {
                const int j0 = wave + VW * q, j1 = wave + VW * (q + 1);
                const bool two = q + 1 < G::rm;
                const int t0 = j0 >> 2, t1 = two ? j1 >> 2 : t0;
                const bool n0 = tower_dyn(t0) ? need_dyn : need_ady, n1 = two && (tower_dyn(t1) ? need_dyn : need_ady);
                const float *bv0 = weights + d.off[tower_off(t0) + 3] + 16 * (j0 & 3) + 4 * g;
                const float *bv1 = weights + d.off[tower_off(t1) + 3] + 16 * (j1 & 3) + 4 * g;
                v4f acc[2] = {v4f{bv0[0], bv0[1], bv0[2], bv0[3]}, v4f{bv1[0], bv1[1], bv1[2], bv1[3]}};
                if (n0 && n1) {
                    const float *const w[2] = {wmr[q], wmr[two ? q + 1 : q]};
                    const float *const t[2] = {hin + t0 * 64 * kVL, hin + t1 * 64 * kVL};
                    tower_layers<16, 2>(w, t, acc, g, li);
                } else if (n0) {
                    acc[0] = tower_layer<16>(wmr[q], hin + t0 * 64 * kVL, acc[0], g, li);
                } else if (n1) {
                    acc[1] = tower_layer<16>(wmr[two ? q + 1 : q], hin + t1 * 64 * kVL, acc[1], g, li);
                }
                if (n0) {
                    float *dst = hout + t0 * 64 * kVL + (16 * (j0 & 3) + 4 * g) * kVL + li;
#pragma unroll
                    for (int r = 0; r < 4; r++) dst[r * kVL] = fmaxf(acc[0][r], 0.f);
                }
                if (n1) {
                    float *dst = hout + t1 * 64 * kVL + (16 * (j1 & 3) + 4 * g) * kVL + li;
#pragma unroll
                    for (int r = 0; r < 4; r++) dst[r * kVL] = fmaxf(acc[1][r], 0.f);
                }
            }
#pragma unroll 1
            for (int q = G::rm; q < G::j1; q++) {
                const int j = wave + VW * q, t = j >> 2, mt = j & 3;
                if (j < kJobs && (tower_dyn(t) ? need_dyn : need_ady)) {
                    const int32_t *o = d.off + tower_off(t);
                    const float *bv = weights + o[3] + 16 * mt + 4 * g;
                    float wm[16];
                    load_frags<16>(wm, weights + o[2], 16 * mt + li, g, K4h);
                    const v4f acc = tower_layer<16>(wm, hin + t * 64 * kVL, v4f{bv[0], bv[1], bv[2], bv[3]}, g, li);
                    float *dst = hout + t * 64 * kVL + (16 * mt + 4 * g) * kVL + li;
#pragma unroll
                    for (int r = 0; r < 4; r++) dst[r * kVL] = fmaxf(acc[r], 0.f);
                }
            }
            wg_barrier();
            float *tmp = hin; hin = hout; hout = tmp;
        }
#pragma unroll
        for (int q = 0; q < G::jo; q++) {
            int t, mt;
            out_job(wave + VW * q, t, mt);
            if (tower_dyn(t) ? need_dyn : need_ady) {
                const int32_t *o = d.off + tower_off(t);
                const float *bv = weights + o[5] + 16 * mt + 4 * g;
                v4f acc = v4f{bv[0], bv[1], bv[2], bv[3]};
                if (q < G::ro) {
                    acc = tower_layer<16>(wor[q < G::ro ? q : 0], hin + t * 64 * kVL, acc, g, li);
                } else {
                    float wo[16];
                    load_frags<16>(wo, weights + o[4], 16 * mt + li, g, K4h);
                    acc = tower_layer<16>(wo, hin + t * 64 * kVL, acc, g, li);
                }
                *reinterpret_cast<v4f *>(Y + t * kVL * kYS + li * kYS + 16 * mt + 4 * g) = acc;
            }
        }
        wg_barrier();
        // ---- tails: this wave's own leaves, lane = output --------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < VT; r++) {
            if (!(r ? live1 : live0)) continue;
            const int leaf = VT * wave + r;
            const bool dyn = r ? dyn1 : dyn0;
            float reward = 0.f;
            if (dyn) {
                const float v[1] = {lane < 32 ? Y[0 * kVL * kYS + leaf * kYS + lane] : 0.f};
                reward = decode_lanes<1>(v, 0, S, lane);
            }
            const float vv[1] = {lane < 32 ? Y[(dyn ? 1 : 3) * kVL * kYS + leaf * kYS + lane] : 0.f};
            const float value = decode_lanes<1>(vv, 0, S, lane);
            const float vp[1] = {lane < 32 ? Y[(dyn ? 2 : 4) * kVL * kYS + leaf * kYS + lane] : 0.f};
            softmax_lanes<1>(vp, A, lane, outs + r * slot);
            if (lane == 0) { outs[r * slot + A] = value; outs[r * slot + A + 1] = reward; }
        }
        lds_sync();
        packed = stage_finish<VT, false>(P, tree, valid, rng_tile, packed, pre);
        SMZ_VSTAMP(t_tow)
    }
#undef SMZ_VSTAMP
    if (prof && lane == 0) {
        atomicAdd(&P.stats[4], t_tree); atomicAdd(&P.stats[5], t_conv);
        atomicAdd(&P.stats[6], t_wait); atomicAdd(&P.stats[7], t_tow);
    }
    if (valid) {
        if (P.sims > 0) {
            rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
            expand_backup_tree<MAXA, KS>(P, tree, rng, h, outs + lane * slot, outs[lane * slot + A + 1], outs[lane * slot + A],
                                         pvals + lane * P.P);
            for (int i = 0; i < h.path_len; i++) P.path[(size_t)tree * P.P + i] = pvals[lane * P.P + i];
            packed = rng.pack();
        }
        P.hdr[tree] = h;
        if (act.action) {
            act_tree<MAXA>(P, tree, rng, act.temperature, act.action, act.policy, act.child_visits, act.root_value);
            packed = rng.pack();
        }
        P.rng_pos[tree] = packed;
    }
}

int search_vision_launch(smz_handle *h, const smz_vision_desc *desc, const float *weights_dev, const float *hidden0_dev,
                         const float *policy0_dev, int train, ActOut act, const double *pow_table_host, smz_stream stream) {
    if (!h || !desc || !weights_dev || !hidden0_dev || !policy0_dev) return fail(SMZ_ERR_INVALID, "smz_search_vision: null argument%s");
    smz_vision_desc t = *desc;
    if (smz_vision_layout(&t) != SMZ_OK || t.total_floats != desc->total_floats)
        return fail(SMZ_ERR_INVALID, "smz_search_vision: descriptor does not describe a vision_model weight buffer%s");
    if (desc->A != h->P.A || h->P.S != kFlat)
        return fail(SMZ_ERR_INVALID, "smz_search_vision: network dimensions differ from the handle's (hidden_size must be 147)%s");
    if (h->K != 2 || h->P.A > 4 || desc->S > 32 || desc->H > 64 || h->P.philox)
        return fail(SMZ_ERR_TOO_LARGE, "smz_search_vision: outside the single-launch kernel's limits (K = 2, A <= 4, S <= 32, H <= 64, "
                                       "MT19937 streams): use the step-wise entry points%s");
    if (train && h->cfg.num_simulations > 0 && !(h->cfg.root_dirichlet_alpha > 0))
        return fail(SMZ_ERR_INVALID, "root_dirichlet_alpha must be > 0 to draw noise (numpy raises ValueError)%s");
    DeviceGuard guard(h->cfg.device);
    Params P = h->P;
    if (act.action && pow_table_host && act.temperature >= 0.3) {
        if (!h->pow_valid || h->pow_T != act.temperature) {
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            HIP_TRY(hipMemcpy(h->d_pow, pow_table_host, ((size_t)h->cfg.num_simulations + 1) * sizeof(double), hipMemcpyHostToDevice));
            h->pow_T = act.temperature;
            h->pow_valid = true;
        }
        P.pow_table = h->d_pow;
    }
    // geometry: 4 waves x 1 tree (everything register-resident, 4 trees per workgroup) unless SMZ_VISION_GEOMETRY=8x2
    bool wide = false;
    if (const char *e = getenv("SMZ_VISION_GEOMETRY")) wide = strcmp(e, "8x2") == 0;
    const int VW = wide ? 8 : 4, VT = wide ? 2 : 1;
    P.tpw = VT;
    const VisLds ml = vis_lds(P, P.A, VW, VT);
    const size_t lds = (size_t)ml.total * sizeof(float);
    if (lds > 160 * 1024) return fail(SMZ_ERR_TOO_LARGE, "smz_search_vision: working set exceeds the 160 KB LDS of a CU%s");
    const int blocks = (P.B + VW * VT - 1) / (VW * VT);
#define SMZ_LAUNCH_VS(MA, W, T)                                                                                        \
    {                                                                                                                  \
        static size_t granted_dev[64] = {};                                                                            \
        size_t &granted = granted_dev[h->cfg.device & 63];                                                             \
        if (lds > granted) {                                                                                           \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_vision<MA, W, T>),                         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)               \
                return fail(SMZ_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed%s");                             \
            granted = lds;                                                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL((k_search_vision<MA, W, T>), dim3(blocks), dim3(W * kWave), lds, (hipStream_t)stream, P,    \
                           *desc, weights_dev, hidden0_dev, policy0_dev, train, act);                                  \
    }
    if (wide) { if (h->maxa == 2) SMZ_LAUNCH_VS(2, 8, 2) else SMZ_LAUNCH_VS(4, 8, 2) }
    else { if (h->maxa == 2) SMZ_LAUNCH_VS(2, 4, 1) else SMZ_LAUNCH_VS(4, 4, 1) }
#undef SMZ_LAUNCH_VS
    h->root_ready = true;
    h->selected = false;
    return launch_check();
}

}  // namespace

extern "C" {

int smz_search_vision(smz_handle *h, const smz_vision_desc *desc, const float *weights_dev, const float *hidden0_dev,
                      const float *policy0_dev, int train, smz_stream stream) {
    return search_vision_launch(h, desc, weights_dev, hidden0_dev, policy0_dev, train, ActOut{0.0, nullptr, nullptr, nullptr, nullptr},
                                nullptr, stream);
}

int smz_search_vision_act(smz_handle *h, const smz_vision_desc *desc, const float *weights_dev, const float *hidden0_dev,
                          const float *policy0_dev, int train, double temperature, const double *pow_table_host,
                          int32_t *action_dev, double *policy_dev, double *child_visits_dev, float *root_value_dev,
                          smz_stream stream) {
    if (!action_dev || !policy_dev || !child_visits_dev) return fail(SMZ_ERR_INVALID, "smz_search_vision_act: null output%s");
    return search_vision_launch(h, desc, weights_dev, hidden0_dev, policy0_dev, train,
                                ActOut{temperature, action_dev, policy_dev, child_visits_dev, root_value_dev}, pow_table_host, stream);
}

}  // extern "C"
