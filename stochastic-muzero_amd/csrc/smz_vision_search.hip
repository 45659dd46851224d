// smz_vision_search.hip -- the whole Monte_carlo_tree_search.run (mcts:311-349) of every tree in ONE launch for the
// `vision_model` family (neural_network_vision_model.py:41-515): smz_search_vision / smz_search_vision_act.
//
// Why a second single-launch kernel: step-wise, a vision simulation round is two launches (k_vision_recurrent 20 us +
// k_expand_backup 14 us at 1024 trees) and every leaf wavefront streams ~180 KB of tower weights out of L2 -- 180 MB per
// round, the L2 running at 8.5 TB/s for weights that never change.  Here a workgroup of FOUR wavefronts owns four trees for
// the whole search (1024 trees = one workgroup per CU, one wavefront per SIMD, 512 registers per lane):
//   * tree phases (expand + backup, select) as in k_search_mlp: the tree's lane, path records and staged random words in LDS;
//   * the convolutional part of a leaf's networks (3x7x7 hidden state, one pixel per lane) by the leaf's own wavefront,
//     the code of the wave-per-leaf kernel (smz_vision_device.hpp);
//   * the five 147 -> H -> [H ->] S/A towers (dynamics reward | prediction value, policy | afterstate-prediction value,
//     policy) for the four leaves AT ONCE on the matrix cores with v_mfma_f32_4x4x1_16B_f32: sixteen 4x4 outer products per
//     instruction, block = 4 output neurons (A operand: one weight per lane) x 4 leaves (B operand: one activation per
//     lane), one input per instruction, accumulators = 4 registers per lane.  Every lane carries useful work (a
//     16x16x4 tile would be three quarters empty with four leaves: measured 0.96 ms per search against 0.8 ms).
//     A layer's inputs are cut into four contiguous quarters summed separately -- dense_stream() of the wave-per-leaf
//     kernel does the same -- which turns a 64-neuron layer into four 64-lane rows; wave w owns tower w (rows = quarters,
//     combined in-lane) and a quarter of tower 4's neurons (one row, quarters in the four 16-lane groups, combined with two
//     cross-lane exchanges).  ALL weights of the wave's rows stay in REGISTERS for the whole search: 200 + 80 + 32 per lane.
//     A tower no leaf of the workgroup needs this round (all leaves on one branch) is skipped.
// f32 in, f32 accumulate, one rounding per multiply-add: the two paths give bit-identical searches
// (tests/test_gpu_end_to_end.py).
// Limits of this kernel (anything else runs step-wise): maxium_action_sample == 2, A <= 4, S <= 32, H <= 64,
// SMZ_RNG_MT19937_NUMPY, LDS working set <= 160 KB.
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
using smz_vision::kFlat4;
using smz_vision::kN;
using smz_vision::kPad;
using smz_vision::kPix;
using smz_vision::kSmallMax;
using smz_vision::residual_block;
using smz_vision::scale_channels;
using smz_vision::uniform_ptr;

namespace {

#ifndef SMZ_VISION_CONV_MFMA
// SMZ_VISION_BIAS_LDS (round 5): the towers' biases, exactly the four floats every lane starts its accumulators from, are
// staged in LDS once per launch.  Before, every tower layer of every round began with a scalar load of the bias offset, a
// global load of the bias and a wait for it (an L2 round trip at one wavefront per SIMD: nothing hides it) -- four to five such
// trips per round (profiles/r05_s_stage_waits.txt).  =0 builds keep the global loads.
// SMZ_VISION_BPS (round 5): the two-action instantiation selects with k_search_mlp's block-parallel selection -- one tree per
// wavefront = at most 63 blocks for 62 simulations: the whole tree's picks in ONE pass of select_block (a lane per block), a
// pointer chase, a lane per level for the path records -- and requests the leaf's parent planes (an L2 round trip) and the next
// round's source words right behind the chase.  Needs the chance thresholds beside the blocks (expand_backup_tree<THR>: in the
// padding of the 64-byte blocks).  A level whose words lie beyond the staged window falls back to the paired descent.
#ifndef SMZ_VISION_BPS
#define SMZ_VISION_BPS 1
#endif
#ifndef SMZ_VISION_BPS_LF
#define SMZ_VISION_BPS_LF 1
#endif
#ifndef SMZ_VISION_BIAS_LDS
#define SMZ_VISION_BIAS_LDS 1
#endif
#define SMZ_VISION_CONV_MFMA 1                       // 3x3 convolutions as v_mfma_f32_4x4x1 chains (smz_vision_device.hpp conv3x3_m)
#endif
constexpr bool kConvMfma = SMZ_VISION_CONV_MFMA != 0;
constexpr int kConvTab = kConvMfma ? smz_vision::kMmFloats : smz_vision::kTapFloats;     // floats per convolution piece in LDS
constexpr int kVW = 4;                               // wavefronts = trees = leaf slots of a workgroup
constexpr int kTowers = 5;                           // 0 dyn reward | 1 pre value | 2 pre policy | 3 apr value | 4 apr policy
constexpr int kP1 = 40;                              // inputs per quarter of the 147-input layer (10 groups of four)
constexpr int kFS = 164;                             // floats per (input kind, leaf) row of the flat tile: 4 x 40 + bank spread
constexpr int kHS = 68;                              // floats per (tower, leaf) row of a hidden tile: 64 + bank spread
constexpr int kBiasUses = 5;                         // layer 1 (own tower | tower 4), hidden layer (own | tower 4), output layer
constexpr int kYS = 36;                              // floats per (tower, leaf) row of the raw outputs
typedef float v4f __attribute__((ext_vector_type(4)));

struct VisLds {                                      // float offsets from the dynamic LDS base
    int small, pbc, wave, per_wave, plane, pv, rng, outs, prof, sel, sel_n, F, H1, H2, Y, br, bias, trees, total;
};
__host__ __device__ inline VisLds vis_lds(const Params &P, int A) {
    VisLds m;
    m.small = 0;
    m.pbc = kSmallMax + 10 * kConvTab;                                               // (+ the ten 3x3 convolution pieces in the layout their kernel code wants)
    m.wave = m.pbc + r4(2 * 2 * (P.sims + 2));
    m.plane = 0;                                                  // float4 plane[81] -> 324 floats
    m.pv = r4(kPad * kPad * 4);
    m.rng = m.pv + P.P * 4;
    m.outs = m.rng + r4(kRngStride);
    m.prof = m.outs + r4(A + 2);                                  // eight 64-bit phase counters (SMZ_DEBUG_SKIP=16)
    m.sel = m.prof + 16;                                          // [2][sel_n] 16-bit words: picks per block | the path (select_block / select_chase)
    m.sel_n = (P.sims + 2 + 1) & ~1;
    m.per_wave = m.sel + (SMZ_VISION_BPS ? r4(m.sel_n) : 0);
    m.F = m.wave + kVW * m.per_wave;
    m.H1 = m.F + 3 * kVW * kFS;
    m.H2 = m.H1 + kTowers * kVW * kHS;
    m.Y = m.H2 + kTowers * kVW * kHS;
    m.br = m.Y + kTowers * kVW * kYS + 64;                        // (+64: the tails read a full wave width of a row)
    m.bias = (m.br + kVW + 15) & ~15;                                        // [wave][kBiasUses][lane] four floats: tower biases as each lane adds them
    m.trees = m.bias + (SMZ_VISION_BIAS_LDS ? kVW * kBiasUses * kWave * 4 : 0);   // the four trees' child blocks (written back at the end)
    m.total = m.trees + kVW * (int)P.tree_words;
    return m;
}

__device__ inline int tower_off(int t) {             // index of a tower's six offsets (W1,b1,Wm,bm,Wo,bo) in smz_vision_desc::off
    return t == 0 ? SMZ_V_TRANS_BASE + SMZ_VT_TOWER
                  : SMZ_V_PRED_BASE + (t >= 3 ? SMZ_V_PRED_STRIDE : 0) + ((t == 1 || t == 3) ? SMZ_VP_VTOWER : SMZ_VP_PTOWER);
}
__device__ inline int tower_input(int t) { return t == 0 ? 0 : ((t == 1 || t == 3) ? 1 : 2); }   // which flat tile feeds it
__device__ inline bool tower_dyn(int t) { return t < 3; }

// Weights of one 64-lane row: inputs part * pl4 + (0 .. NS-1) of output neuron o of a packed tower matrix (4-way interleaved
// input-major, 64 outputs wide, K4 input rows); zero beyond the quarter, beyond the matrix, and for a lane without work.
template <int NS>
__device__ inline void load_row(float (&w)[NS], const float *W, int o, int part, int pl4, int K4, bool on) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int k = part * pl4 + s;
        w[s] = (on && s < pl4 && k < K4) ? W[((k >> 2) * kWave + o) * 4 + (k & 3)] : 0.f;
    }
}

// NR rows at once, rows R0 .. R0 + NR - 1 of w / bx / acc: NS inputs each.  bx[r] = this lane's activations of row r in LDS
// (its leaf's row of the tile, at the quarter's first input; 16-byte aligned).  One v_mfma_f32_4x4x1 per row and input; the
// NR accumulator chains are independent (one wave issues one such MFMA per ~8.5 cycles from three chains on: mfma4_probe),
// and the activations of the next four inputs are read before the products of the current four are issued.
template <int NS, int NR, int R0, int NW>
__device__ inline void rows_mfma(const float (&w)[NW][NS], const float *const (&bx)[NW], v4f (&acc)[NW]) {
    constexpr int NC = NS / 4;
    float4 b[2][NR];
#pragma unroll
    for (int r = 0; r < NR; r++) b[0][r] = *reinterpret_cast<const float4 *>(bx[R0 + r]);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (c + 1 < NC) {
#pragma unroll
            for (int r = 0; r < NR; r++) b[(c + 1) & 1][r] = *reinterpret_cast<const float4 *>(bx[R0 + r] + 4 * (c + 1));
        }
#pragma unroll
        for (int r = 0; r < NR; r++) acc[R0 + r] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[R0 + r][4 * c + 0], b[c & 1][r].x, acc[R0 + r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; r++) acc[R0 + r] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[R0 + r][4 * c + 1], b[c & 1][r].y, acc[R0 + r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; r++) acc[R0 + r] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[R0 + r][4 * c + 2], b[c & 1][r].z, acc[R0 + r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; r++) acc[R0 + r] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[R0 + r][4 * c + 3], b[c & 1][r].w, acc[R0 + r], 0, 0, 0);
    }
}
// value of lane ^ MASK without an LDS-crossbar round trip: rows (^16) and halves (^32) through gfx950's permlane swaps,
// 4-lane groups (^4, ^8) through DPP row shifts selected per group with bank masks
template <int MASK>
__device__ inline float lane_xor(float v) {
    const unsigned u = __float_as_uint(v);
    const int lane = threadIdx.x & 63;
    if constexpr (MASK == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);     // r[0] = rows (0,0,2,2), r[1] = rows (1,1,3,3)
        return __uint_as_float((lane & 16) ? r[0] : r[1]);
    } else if constexpr (MASK == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0] = halves (lo,lo), r[1] = halves (hi,hi)
        return __uint_as_float((lane & 32) ? r[0] : r[1]);
    } else {
        static_assert(MASK == 4 || MASK == 8, "lane_xor");
        // row_shr:MASK reaches the upper group of each pair, row_shl:MASK the lower one
        const int up = __builtin_amdgcn_update_dpp((int)u, (int)u, 0x110 + MASK, 0xf, MASK == 4 ? 0xa : 0xc, false);
        return __int_as_float(__builtin_amdgcn_update_dpp(up, (int)u, 0x100 + MASK, 0xf, MASK == 4 ? 0x5 : 0x3, false));
    }
}
template <int MASK>
__device__ inline v4f xadd(v4f v) {                   // v + (v of lane ^ MASK), the four registers
    return v4f{v[0] + lane_xor<MASK>(v[0]), v[1] + lane_xor<MASK>(v[1]), v[2] + lane_xor<MASK>(v[2]), v[3] + lane_xor<MASK>(v[3])};
}
__device__ inline v4f relu4(v4f v) { return v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
__device__ inline v4f bias4(const float *b, bool on) { return on ? v4f{b[0], b[1], b[2], b[3]} : v4f{0.f, 0.f, 0.f, 0.f}; }

// A 64-neuron layer of the five towers for the workgroup's four leaves: this wave's five rows (see the file header).
//   rows 0..3: tower `wave`, quarter = row, neuron = lane  -> combined in-lane, every lane stores four neurons of its leaf
//   row 4:     tower 4, quarter = lane / 16, neuron = 16 * wave + lane % 16 -> combined across the 16-lane groups
// in: per-(tower | input kind, leaf) rows of `stride` floats; out: rows of kHS floats, relu applied.
template <int NS>
__device__ inline void tower_layer64(const float (&w)[5][NS], const float *in, int stride, bool by_kind, int pl4, const float *bias_a,
                                     const float *bias_b, const v4f *bias_lds, float *out, int wave, int lane, bool need_a, bool need_b) {
    const int blk = lane >> 2, lf = lane & 3, grp = lane >> 4;
    const int ia = by_kind ? tower_input(wave) : wave, ib = by_kind ? tower_input(4) : 4;
    const float *xa = in + (ia * kVW + lf) * stride, *xb = in + (ib * kVW + lf) * stride + grp * pl4;
    const float *const bx[5] = {xa, xa + pl4, xa + 2 * pl4, xa + 3 * pl4, xb};
    v4f acc[5];
    acc[1] = acc[2] = acc[3] = v4f{0.f, 0.f, 0.f, 0.f};
    if (SMZ_VISION_BIAS_LDS) {
        acc[0] = bias_lds[lane];
        acc[4] = bias_lds[kWave + lane];
    } else {
        acc[0] = bias4(bias_a + 4 * blk, true);
        acc[4] = bias4(bias_b + 16 * wave + 4 * (blk & 3), grp == 0);
    }
    if (need_a && need_b) rows_mfma<NS, 5, 0>(w, bx, acc);            // (wave-uniform)
    else if (need_a) rows_mfma<NS, 4, 0>(w, bx, acc);
    else if (need_b) rows_mfma<NS, 1, 4>(w, bx, acc);
    if (need_a) {
        const v4f y = relu4((acc[0] + acc[1]) + (acc[2] + acc[3]));
        *reinterpret_cast<v4f *>(out + (wave * kVW + lf) * kHS + 4 * blk) = y;
    }
    if (need_b) {
        const v4f y = relu4(xadd<32>(xadd<16>(acc[4])));
        if (grp == 0) *reinterpret_cast<v4f *>(out + (4 * kVW + lf) * kHS + 16 * wave + 4 * blk) = y;
    }
}

__device__ inline void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // LDS hand-offs only

// 1x1 convolution (with bias) of CIN channels -> row `leaf` of a flat tile (torch's Flatten of [3,7,7])
template <int CIN>
__device__ inline void mix_to_tile(float *tile, int leaf, int p, bool active, const float (&x)[4], const float *__restrict__ w,
                                   const float *__restrict__ b) {
#pragma unroll
    for (int oc = 0; oc < kC; oc++) {
        float s = 0.f;
#pragma unroll
        for (int ic = 0; ic < CIN; ic++) s = fmaf(w[oc * CIN + ic], x[ic], s);
        s += b[oc];
        if (active) tile[leaf * kFS + oc * kPix + p] = s;
    }
}

extern __shared__ float4 smz_vsearch_lds4[];

// AEQ: the action count equals its bucket MAXA -- a compile-time constant then for everything inlined below (the per-action
// arrays of the root level stay in registers instead of scratch memory)
// PHX (round 6): the handle draws from Philox4x32-10 counter streams (SMZ_RNG_PHILOX) -- before, such a handle was refused here and
// ran the step-wise kernels.  The MT19937 instantiations are unchanged (RngT<false> carries no Philox state).
template <int MAXA, bool AEQ, bool PHX = false>
// -DSMZ_VISION_WPE=2 (A/B builds only, tools/vision_wpe_ab.sh): register-allocate for TWO wavefronts per SIMD (256 registers a
// lane instead of 512; the tower weights no longer fit and spill) -- the measurement behind DESIGN 9.3's occupancy argument
#ifdef SMZ_VISION_WPE
#define SMZ_VISION_OCC __attribute__((amdgpu_waves_per_eu(SMZ_VISION_WPE, SMZ_VISION_WPE)))
#else
#define SMZ_VISION_OCC
#endif
__global__ void __launch_bounds__(kVW *kWave) SMZ_VISION_OCC k_search_vision(Params Pin, smz_vision_desc d, const float *__restrict__ weights,
                                                              const float *__restrict__ hidden0, const float *__restrict__ policy0,
                                                              int train, ActOut act) {
    constexpr int KS = 2, VT = 1;
    Params P = Pin;
    P.K = KS; P.tpw = VT;
    P.philox = PHX ? 1 : 0;         // (a constant in everything inlined below)
    if (AEQ) P.A = MAXA;            // (not d.A: a modified copy of the descriptor, indexed at run time, would live in scratch)
    fix_layout(P, AEQ, true);
    // block-parallel selection (two actions, two children, at most 126 simulations: 7-bit block indices): chance thresholds
    // in the padding of the 64-byte blocks, as the global-memory-tree instantiations of k_search_mlp keep them
    constexpr bool VBPS = SMZ_VISION_BPS && MAXA == 2 && AEQ;
    const bool vbps = VBPS && P.sims <= 126 && P.eb_words >= 14;           // (wave-uniform)
    // (the children's stored value terms -- YV, as the LDS-resident k_search_mlp keeps them -- measured equal here: 80.1 against
    //  80.4 M, profiles/r05_ah_vision_bps_ab.txt)
    if (VBPS) { P.thr_off = P.rb_words + 12; P.thr_stride = P.eb_words; P.ry_off = -1; }
    float *lds = reinterpret_cast<float *>(smz_vsearch_lds4);
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int A = P.A, S = d.S, K4h = up4(d.H);
    const int plh4 = 4 * (((K4h >> 2) + 3) >> 2);                     // inputs per quarter of an H-input layer (16 for H = 64)
    const VisLds ml = vis_lds(P, A);
    // ---- one-time staging: small weights, pb_c table (+ reciprocals), zeroed tiles ---------------------------------------
    for (int i = threadIdx.x; i < d.small_floats / 4; i += blockDim.x)
        reinterpret_cast<float4 *>(lds + ml.small)[i] = reinterpret_cast<const float4 *>(weights)[i];
    double *pbc_lds = reinterpret_cast<double *>(lds + ml.pbc);
    const int n_pbc = P.sims + 2;
    for (int i = threadIdx.x; i < n_pbc; i += blockDim.x) {
        pbc_lds[i] = P.pbc_sqrt[i];
        pbc_lds[n_pbc + i] = i > 0 ? 1.0 / (double)i : 0.0;
    }
    for (int i = threadIdx.x + ml.wave; i < ml.total; i += blockDim.x)
        if (i < ml.bias || i >= ml.trees) lds[i] = 0.f;            // planes' borders, pad inputs of the tiles (not the staged biases: no barrier in between)
    // tap-major copies of the 3x3 convolutions: [net 0..1 transition: conv_in, res_a, res_b | net 0..1 prediction: res_a, res_b]
    float *tm = lds + kSmallMax;
    for (int n = 0; n < 2; n++) {
        const int32_t *o = d.off + SMZ_V_TRANS_BASE + n * SMZ_V_TRANS_STRIDE, *q = d.off + SMZ_V_PRED_BASE + n * SMZ_V_PRED_STRIDE;
        if (kConvMfma) {
            smz_vision::mfma_table<4>(tm + (n * 3 + 0) * kConvTab, weights + o[SMZ_VT_CONV_IN], threadIdx.x, blockDim.x);
            smz_vision::mfma_table<3>(tm + (n * 3 + 1) * kConvTab, weights + o[SMZ_VT_RES_A], threadIdx.x, blockDim.x);
            smz_vision::mfma_table<3>(tm + (n * 3 + 2) * kConvTab, weights + o[SMZ_VT_RES_B], threadIdx.x, blockDim.x);
            smz_vision::mfma_table<3>(tm + (6 + n * 2 + 0) * kConvTab, weights + q[SMZ_VP_RES_A], threadIdx.x, blockDim.x);
            smz_vision::mfma_table<3>(tm + (6 + n * 2 + 1) * kConvTab, weights + q[SMZ_VP_RES_B], threadIdx.x, blockDim.x);
        } else {
            smz_vision::tap_major<4>(tm + (n * 3 + 0) * kConvTab, weights + o[SMZ_VT_CONV_IN], threadIdx.x, blockDim.x);
            smz_vision::tap_major<3>(tm + (n * 3 + 1) * kConvTab, weights + o[SMZ_VT_RES_A], threadIdx.x, blockDim.x);
            smz_vision::tap_major<3>(tm + (n * 3 + 2) * kConvTab, weights + o[SMZ_VT_RES_B], threadIdx.x, blockDim.x);
            smz_vision::tap_major<3>(tm + (6 + n * 2 + 0) * kConvTab, weights + q[SMZ_VP_RES_A], threadIdx.x, blockDim.x);
            smz_vision::tap_major<3>(tm + (6 + n * 2 + 1) * kConvTab, weights + q[SMZ_VP_RES_B], threadIdx.x, blockDim.x);
        }
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
    const int blk = lane >> 2, lf = lane & 3;
    // the workgroup's four trees live in LDS for the search: a tree level is an LDS round trip instead of an L2 one
    uint32_t *const nodes_global = P.nodes;
    uint32_t *const nodes_lds = reinterpret_cast<uint32_t *>(lds + ml.trees);
    P.nodes = nodes_lds;
    P.tree0 = blockIdx.x * kVW;
    // ---- register-resident weights of this wave's rows ---------------------------------------------------------------------
    const int32_t *offa = d.off + tower_off(wave), *offb = d.off + tower_off(4);
    float w1[5][kP1], wm[5][16], wo[2][16];
#pragma unroll
    for (int r = 0; r < 4; r++) load_row<kP1>(w1[r], weights + offa[0], lane, r, kP1, kFlat4, true);
    load_row<kP1>(w1[4], weights + offb[0], 16 * wave + (lane & 15), lane >> 4, kP1, kFlat4, true);
#pragma unroll
    for (int r = 0; r < 4; r++) load_row<16>(wm[r], weights + offa[2], lane, r, plh4, d.L > 0 ? K4h : 0, true);
    load_row<16>(wm[4], weights + offb[2], 16 * wave + (lane & 15), lane >> 4, plh4, d.L > 0 ? K4h : 0, true);
    // output layer: waves 0..2 = the 32-wide outputs of towers 0 | 1 | 3 (row = quarters 0,1 | 2,3 in the two half-waves,
    // neuron = lane % 32); wave 3 = the A-wide policy outputs of towers 2 (lanes 0..15) and 4 (lanes 16..31), quarter =
    // (lane / 4) % 4, neuron = lane % 4
    const int t_out = wave == 0 ? 0 : (wave == 1 ? 1 : 3);
    const int32_t *offo = d.off + tower_off(wave < 3 ? t_out : ((lane & 16) ? 4 : 2));
    if (wave < 3) {
        load_row<16>(wo[0], weights + offo[4], lane & 31, lane >> 5, plh4, K4h, true);
        load_row<16>(wo[1], weights + offo[4], lane & 31, 2 + (lane >> 5), plh4, K4h, true);
    } else {
        load_row<16>(wo[0], weights + offo[4], lane & 3, (lane >> 2) & 3, plh4, K4h, lane < 32);
        load_row<16>(wo[1], weights + offo[4], 0, 0, plh4, K4h, false);
    }
    v4f *const bias_lds = reinterpret_cast<v4f *>(lds + ml.bias) + wave * kBiasUses * kWave;
    if (SMZ_VISION_BIAS_LDS) {
        const int grp = lane >> 4;
        bias_lds[0 * kWave + lane] = bias4(weights + offa[1] + 4 * blk, true);
        bias_lds[1 * kWave + lane] = bias4(weights + offb[1] + 16 * wave + 4 * (blk & 3), grp == 0);
        bias_lds[2 * kWave + lane] = bias4(weights + offa[3] + 4 * blk, d.L > 0);
        bias_lds[3 * kWave + lane] = bias4(weights + offb[3] + 16 * wave + 4 * (blk & 3), d.L > 0 && grp == 0);
        bias_lds[4 * kWave + lane] = wave < 3 ? bias4(weights + offo[5] + 4 * (blk & 7), lane < 32)
                                              : bias4(weights + offo[5], (lane & 12) == 0 && lane < 32);
    }
    __syncthreads();

    const int tree0 = blockIdx.x * kVW + wave;
    const int tree = tree0 + lane;
    const bool valid = lane < VT && tree < P.B && tree_active(P, tree);
    const bool live0 = __builtin_amdgcn_readlane((int)valid, 0) != 0;
    const bool active = lane < kPix;
    const int p = active ? lane : kPix - 1, pp = (p / kN + 1) * kPad + (p % kN + 1);

    // ---- root: hidden state and policy come from smz_vision_initial (one workgroup per frame) ---------------------------
    if (live0) {                                                         // wave-uniform
        for (int k = lane; k < kFlat; k += kWave) P.hidden[(size_t)tree0 * P.N * P.hs + k] = hidden0[(size_t)tree0 * kFlat + k];
        if (lane < A) outs[lane] = policy0[(size_t)tree0 * A + lane];
    }
    lds_sync();
    int packed = wave_stage_rng<PHX>(P, tree, valid, rng_tile);
    RngT<PHX> rng;
    rng.bind(P, tree, valid);
    TreeHdr h = {0, 0, 0.f, 0.f, 0, 0.f, 0, 0};
    if (valid) {
        rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
        root_init_tree<MAXA>(P, tree, rng, outs + lane * slot, nullptr, train != 0);
        h = P.hdr[tree];
        packed = rng.pack();
    }
    unsigned n_dec = 0, n_chance = 0, n_children = 0;
    if (P.sims > 0) packed = wave_stage_rng_from<4, PHX>(P, tree, valid, rng_tile, packed, rng.block());

    // (SMZ_DEBUG_SKIP=16 with statistics on: s_memtime phase accounting per wave in LDS -> stats[8..15] = tree | parent-row
    //  load | transition convolutions | prediction convolutions | barrier wait | layer 1 | hidden + output layers | tails +
    //  word staging; stats[4..7] = tree | conv | wait | towers + tails as sums of those)
    const bool prof = (P.dbg & 16) && P.stats;
    unsigned long long *pc = reinterpret_cast<unsigned long long *>(wl + ml.prof);
    unsigned long long t0 = 0;
#define SMZ_VSTAMP(i) if (prof) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) pc[i] += t1 - t0; t0 = t1; }
#define SMZ_VDRAIN() if (prof) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // ---- simulations -----------------------------------------------------------------------------------------------------
    for (int s = 0; s < P.sims; s++) {
        if (prof) t0 = __builtin_amdgcn_s_memtime();
        Leaf L = {0, 0, 0, 0};
        // The tree phases use the spare lanes as k_search_mlp's specialised instantiation does: the backup runs one lane per
        // path level (lanes 0..7, backup_levels_lanes), the descent of a two-action tree scores one child per lane (lanes 0
        // and 2, pick_decision_pair; the helper works on a copy of the stream position and of the MinMax bounds).
        float leaf_rw = 0.f;
        if (valid) {
            rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
            if (s > 0) expand_backup_tree<MAXA, KS, true, VBPS>(P, tree, rng, h, outs + lane * slot, outs[lane * slot + A + 1],
                                                                outs[lane * slot + A], pvals + lane * P.P, &leaf_rw);
        }
        if (s > 0 && live0) {
            const int len = __builtin_amdgcn_readlane(h.path_len, 0);
            const float lrw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(leaf_rw), 0));
            if (lane < 8) {
                float mn = lane == 0 ? h.mn : __builtin_inff(), mx = lane == 0 ? h.mx : -__builtin_inff(), v_root = 0.f;
                backup_levels_lanes<1>(P, tree0, lane, len, outs[A], lrw, pvals, mn, mx, v_root);
                if (lane == 0) {   // the root itself (reward 0)
                    const float nvs = h.root_value_sum + v_root;
                    const int nvc = h.root_visit + 1;
                    h.root_value_sum = nvs;
                    h.root_visit = nvc;
                    const float qv = nvs / (float)nvc;
                    if (qv > mx) mx = qv;
                    if (qv < mn) mn = qv;
                    h.mn = mn;
                    h.mx = mx;
                }
            }
        }
        bool bps_done = false;                 // (wave-uniform) this round's leaf came out of the block-parallel selection
        float early_x[3] = {0.f, 0.f, 0.f};    // ... and its parent planes were requested there
        StagePre<VT> pre;
        if constexpr (VBPS) if (vbps && live0) {
            uint16_t *selw = reinterpret_cast<uint16_t *>(wl + ml.sel), *pathw = selw + ml.sel_n;
            if (valid && s > 0) selw[h.n_exp] = (uint16_t)(h.path_len << 9);          // depth of the node the expansion created
            lds_sync();
            const int nexp = __builtin_amdgcn_readlane(h.n_exp, 0), rvis = __builtin_amdgcn_readlane(h.root_visit, 0);
            const float bmn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.mn), 0));
            const float bmx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.mx), 0));
            const int bused = __builtin_amdgcn_readlane(valid ? rng.used : 0, 0), bstaged = __builtin_amdgcn_readlane(valid ? rng.staged : 0, 0);
            const int bstage = __builtin_amdgcn_readlane(valid ? (int)(rng.stage - rng_tile) : 0, 0);
            const uint32_t *stb = tree_base(P, tree0);
            for (int base = 0; base <= nexp; base += kWave) {
                const int b = base + lane;
                if (b <= nexp) {
                    const int depth = b == 0 ? 0 : (int)(selw[b] >> 9);
                    // (LF: a block's auxiliary words requested with its children's fields -- one wavefront per SIMD: every LDS round trip counts)
                    const uint32_t r = select_block<MAXA, false, RngT<PHX>, SMZ_VISION_BPS_LF>(P, stb, b, depth, rvis, bmn, bmx, rng_tile + bstage, bused, bstaged, pbc_lds);
                    selw[b] = (uint16_t)((depth << 9) | r);
                }
            }
            lds_sync();
            int len = 0;
            if (valid) len = select_chase(selw, pathw);
            lds_sync();
            const int blen = __builtin_amdgcn_readlane(len, 0);
            if (blen > 0) {                                                   // (wave-uniform; 0: a level's words beyond the staged window)
                if (valid) {                                                  // the words the descent's levels drew
                    const int nw = select_words(len, A);
                    rng.used += nw; rng.ready -= nw; rng.idx += nw;
                    if (rng.idx >= kMtN) { rng.idx -= kMtN; rng.wrapped(); }
                    packed = rng.pack();
                }
                int par = 0;
                if (valid && len > 1) { const int loc = pathw[len - 2], pb = loc >> 8; par = pb == 0 ? 1 + (loc & 3) : 1 + A + (pb - 1) * 2 + (loc & 3); }
                const int parent = __builtin_amdgcn_readlane(par, 0);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // (planes stored in earlier rounds may be this round's parents)
                const float *hrow = P.hidden + ((size_t)tree0 * P.N + parent) * P.hs;
                early_x[0] = hrow[p]; early_x[1] = hrow[kPix + p]; early_x[2] = hrow[2 * kPix + p];
                stage_issue<VT, PHX>(P, tree, valid, packed, pre);
                if (SMZ_VISION_BPS_LF && valid) L = select_leaf(P, stb, pathw, len);      // (its word requested before the records')
                for (int dd = lane; dd < blen; dd += kWave) select_record(P, stb, pathw, dd, pvals);
                if (valid) { if (!SMZ_VISION_BPS_LF) L = select_leaf(P, stb, pathw, len); h.path_len = len; }
                bps_done = true;
            }
        }
        if (bps_done) {
        } else if constexpr (MAXA == 2) {
            const int pk = __builtin_amdgcn_readlane(valid ? rng.pack() : 0, 0), us = __builtin_amdgcn_readlane(valid ? rng.used : 0, 0);
            const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)rng.block(), 0), pk0 = (uint32_t)__builtin_amdgcn_readlane((int)rng.key0(), 0),
                           pk1 = (uint32_t)__builtin_amdgcn_readlane((int)rng.key1(), 0);
            const float hmn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.mn), 0));
            const float hmx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.mx), 0));
            const int hrv = __builtin_amdgcn_readlane(h.root_visit, 0);
            if ((lane == 0 || lane == 2) && live0) {
                TreeHdr hs = h;
                if (lane == 2) {
                    rng.load(P.mt + (size_t)tree0 * kMtN, pk, rng_tile, kRngStage);
                    rng.used = us;
                    rng.follow(pb, pk0, pk1);      // (Philox: the helper draws beyond the staged window from the TREE's stream)
                    hs.mn = hmn; hs.mx = hmx; hs.root_visit = hrv;
                }
                int len = 0;
                const Leaf Lp = select_tree<MAXA, KS, false, true, true, VBPS>(P, tree0, rng, hs, pbc_lds, len, n_dec, n_chance, n_children,
                                                                              pvals, lane >> 1);
                if (lane == 0) {
                    L = Lp;
                    h.path_len = len;
                    packed = rng.pack();
                }
            }
        } else if (valid) {
            int len = 0;
            L = select_tree<MAXA, KS, false, true>(P, tree, rng, h, pbc_lds, len, n_dec, n_chance, n_children, pvals + lane * P.P);
            h.path_len = len;
            packed = rng.pack();
        }
        SMZ_VSTAMP(0)
        // hidden rows written in earlier rounds (by this wave) may be this round's parent rows
        if (!bps_done) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            stage_issue<VT, PHX>(P, tree, valid, packed, pre);
        }
        // ---- convolutional part of this wave's leaf: flat inputs of its towers into the tiles ------------------------------
        const bool dyn = __builtin_amdgcn_readlane(L.branch, 0) != 0;
        const int leaf = wave;
        if (lane == 0) br[leaf] = live0 ? (dyn ? 1 : 0) : -1;
        if (live0) {                                                     // wave-uniform: a dead leaf's rows stay zero
            const int parent = __builtin_amdgcn_readlane(L.parent_id, 0), actn = __builtin_amdgcn_readlane(L.action, 0);
            const int leaf_id = __builtin_amdgcn_readlane(L.leaf_id, 0);
            const float *hrow = P.hidden + ((size_t)tree0 * P.N + parent) * P.hs;
            const float a_plane = (float)(actn + 1) / (float)d.A;        // muzero_model.py:511-522
            float x[4] = {early_x[0], early_x[1], early_x[2], a_plane};
            if (!bps_done) { x[0] = hrow[p]; x[1] = hrow[kPix + p]; x[2] = hrow[2 * kPix + p]; }
            SMZ_VDRAIN()
            SMZ_VSTAMP(1)
            const int32_t *o = d.off + SMZ_V_TRANS_BASE + (dyn ? 0 : SMZ_V_TRANS_STRIDE);
            if (dyn) mix_to_tile<4>(F, leaf, p, active, x, uniform_ptr(small, o[SMZ_VT_MIX_W]), uniform_ptr(small, o[SMZ_VT_MIX_B]));
            if (active) plane[pp] = make_float4(x[0], x[1], x[2], x[3]);
            lds_sync();
            float t[kC];
            const int ni = dyn ? 0 : 1;
            if (kConvMfma) smz_vision::conv3x3_m<4>(plane, pp, tm + (ni * 3 + 0) * kConvTab, lane, t);
            else smz_vision::conv3x3_t<4>(plane, pp, reinterpret_cast<const float4 *>(tm + (ni * 3 + 0) * kConvTab), t);
            lds_sync();
            {
                const float *bn = uniform_ptr(small, o[SMZ_VT_BN_IN]);
#pragma unroll
                for (int c = 0; c < kC; c++) t[c] = fmaxf(t[c] * bn[c] + bn[kC + c], 0.f);
            }
            {
                const float *ta = tm + (ni * 3 + 1) * kConvTab, *tb = tm + (ni * 3 + 2) * kConvTab;
                const float4 *wa = reinterpret_cast<const float4 *>(ta), *wb = reinterpret_cast<const float4 *>(tb);
                const float *bn = uniform_ptr(small, o[SMZ_VT_RES_BN]);
                for (int i = 0; i < d.L; i++) {
                    if (kConvMfma) smz_vision::residual_block_m(plane, pp, active, ta, tb, bn, lane, t);
                    else smz_vision::residual_block_t(plane, pp, active, wa, wb, bn, t);
                }
            }
#pragma unroll
            for (int c = 0; c < kC; c++) t[c] = fmaxf(t[c], 0.f);
            scale_channels(t);
            if (active) {
                float *ho = P.hidden + ((size_t)tree0 * P.N + leaf_id) * P.hs;
                ho[p] = t[0]; ho[kPix + p] = t[1]; ho[2 * kPix + p] = t[2];
            }
            SMZ_VDRAIN()
            SMZ_VSTAMP(2)
            const int32_t *q = d.off + SMZ_V_PRED_BASE + (dyn ? 0 : SMZ_V_PRED_STRIDE);
            {
                const float *ta = tm + (6 + ni * 2 + 0) * kConvTab, *tb = tm + (6 + ni * 2 + 1) * kConvTab;
                const float4 *wa = reinterpret_cast<const float4 *>(ta), *wb = reinterpret_cast<const float4 *>(tb);
                const float *bn = uniform_ptr(small, q[SMZ_VP_RES_BN]);
                for (int i = 0; i < d.L; i++) {
                    if (kConvMfma) smz_vision::residual_block_m(plane, pp, active, ta, tb, bn, lane, t);
                    else smz_vision::residual_block_t(plane, pp, active, wa, wb, bn, t);
                }
            }
            const float xs[4] = {t[0], t[1], t[2], 0.f};
            mix_to_tile<kC>(F + kVW * kFS, leaf, p, active, xs, uniform_ptr(small, q[SMZ_VP_VMIX_W]), uniform_ptr(small, q[SMZ_VP_VMIX_B]));
            mix_to_tile<kC>(F + 2 * kVW * kFS, leaf, p, active, xs, uniform_ptr(small, q[SMZ_VP_PMIX_W]), uniform_ptr(small, q[SMZ_VP_PMIX_B]));
        }
        SMZ_VSTAMP(3)
        wg_barrier();                                                    // flat tiles and branch flags complete
        SMZ_VSTAMP(4)
        const int mine = br[lf];
        const bool need_dyn = __ballot(mine == 1) != 0ull, need_ady = __ballot(mine == 0) != 0ull;
        const bool need_a = wave < 3 ? need_dyn : need_ady, need_b = need_ady;          // tower `wave` | tower 4
        // ---- towers: 147 -> H, [H -> H] x L (the SAME Linear applied L times), H -> S / A ---------------------------------------
        tower_layer64<kP1>(w1, F, kFS, true, kP1, weights + offa[1], weights + offb[1], bias_lds, H1, wave, lane, need_a, need_b);
        wg_barrier();
        SMZ_VSTAMP(5)
        float *hin = H1, *hout = H2;
        for (int l = 0; l < d.L; l++) {
            tower_layer64<16>(wm, hin, kHS, false, plh4, weights + offa[3], weights + offb[3], bias_lds + 2 * kWave, hout, wave, lane, need_a, need_b);
            wg_barrier();
            float *tmp = hin; hin = hout; hout = tmp;
        }
        if (wave < 3) {
            if (wave < 2 ? need_dyn : need_ady) {
                const float *x = hin + (t_out * kVW + lf) * kHS + (lane >> 5) * plh4;
                const float *const bx[2] = {x, x + 2 * plh4};
                v4f acc[2] = {SMZ_VISION_BIAS_LDS ? bias_lds[4 * kWave + lane] : bias4(weights + offo[5] + 4 * (blk & 7), lane < 32), v4f{0.f, 0.f, 0.f, 0.f}};
                rows_mfma<16, 2, 0>(wo, bx, acc);
                const v4f y = xadd<32>(acc[0]) + xadd<32>(acc[1]);
                if (lane < 32) *reinterpret_cast<v4f *>(Y + (t_out * kVW + lf) * kYS + 4 * blk) = y;
            }
        } else {
            const int t = (lane & 16) ? 4 : 2;
            const float *x = hin + (t * kVW + lf) * kHS + ((lane >> 2) & 3) * plh4;
            const float *const bx[2] = {x, x};
            v4f acc[2] = {SMZ_VISION_BIAS_LDS ? bias_lds[4 * kWave + lane] : bias4(weights + offo[5], (lane & 12) == 0 && lane < 32), v4f{0.f, 0.f, 0.f, 0.f}};
            rows_mfma<16, 1, 0>(wo, bx, acc);
            const v4f y = xadd<8>(xadd<4>(acc[0]));
            if ((lane & 12) == 0 && lane < 32) *reinterpret_cast<v4f *>(Y + (t * kVW + lf) * kYS) = y;
        }
        wg_barrier();
        SMZ_VSTAMP(6)
        // ---- tails: this wave's own leaf, lane = output --------------------------------------------------------------------------
        if (live0) {
            float reward = 0.f;
            if (dyn) {
                const float v[1] = {lane < 32 ? Y[(0 * kVW + leaf) * kYS + lane] : 0.f};
                reward = decode_lanes<1>(v, 0, S, lane);
            }
            const float vv[1] = {lane < 32 ? Y[((dyn ? 1 : 3) * kVW + leaf) * kYS + lane] : 0.f};
            const float value = decode_lanes<1>(vv, 0, S, lane);
            const float vp[1] = {lane < 32 ? Y[((dyn ? 2 : 4) * kVW + leaf) * kYS + lane] : 0.f};
            softmax_lanes<1>(vp, A, lane, outs);
            if (lane == 0) { outs[A] = value; outs[A + 1] = reward; }
        }
        lds_sync();
        packed = stage_finish<VT, PHX>(P, tree, valid, rng_tile, packed, pre, rng.block());
        SMZ_VSTAMP(7)
    }
#undef SMZ_VSTAMP
#undef SMZ_VDRAIN
    if (prof && lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&P.stats[8 + i], pc[i]);
        atomicAdd(&P.stats[4], pc[0]); atomicAdd(&P.stats[5], pc[1] + pc[2] + pc[3]);
        atomicAdd(&P.stats[6], pc[4]); atomicAdd(&P.stats[7], pc[5] + pc[6] + pc[7]);
    }
    if (valid) {
        if (P.sims > 0) {
            rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
            expand_backup_tree<MAXA, KS, false, VBPS>(P, tree, rng, h, outs + lane * slot, outs[lane * slot + A + 1], outs[lane * slot + A],
                                         pvals + lane * P.P);
            for (int i = 0; i < h.path_len; i++) P.path[(size_t)i * P.B + tree] = pvals[lane * P.P + i];
            packed = rng.pack();
        }
        P.hdr[tree] = h;
        if (act.action) {
            act_tree<MAXA>(P, tree, rng, act.temperature, act.action, act.policy, act.child_visits, act.root_value);
            packed = rng.pack();
        }
        P.rng_pos[tree] = packed;
        rng.save(P, tree);
    }
    lds_sync();
    if (live0) {                                                         // the finished tree back to its place in global memory
        const uint32_t *src = nodes_lds + (size_t)wave * P.tree_words;
        uint32_t *dst = nodes_global + (size_t)tree0 * P.tree_words;
        for (int i = lane; i < (int)P.tree_words; i += kWave) dst[i] = src[i];
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
    if (h->K != 2 || h->P.A > 4 || desc->S > 32 || desc->H > 64)
        return fail(SMZ_ERR_TOO_LARGE, "smz_search_vision: outside the single-launch kernel's limits (K = 2, A <= 4, S <= 32, H <= 64): "
                                       "use the step-wise entry points%s");
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
    P.tpw = 1;
    const VisLds ml = vis_lds(P, P.A);
    const size_t lds = (size_t)ml.total * sizeof(float);
    if (lds > 160 * 1024) return fail(SMZ_ERR_TOO_LARGE, "smz_search_vision: working set exceeds the 160 KB LDS of a CU%s");
    const int blocks = (P.B + kVW - 1) / kVW;
#define SMZ_LAUNCH_VS(MA, EQ) { if (P.philox) SMZ_LAUNCH_VS2(MA, EQ, true) else SMZ_LAUNCH_VS2(MA, EQ, false) }
#define SMZ_LAUNCH_VS2(MA, EQ, PX)                                                                                     \
    {                                                                                                                  \
        static size_t granted_dev[64] = {};                                                                            \
        size_t &granted = granted_dev[h->cfg.device & 63];                                                             \
        if (lds > granted) {                                                                                           \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_vision<MA, EQ, PX>),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)               \
                return fail(SMZ_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed%s");                             \
            granted = lds;                                                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL((k_search_vision<MA, EQ, PX>), dim3(blocks), dim3(kVW * kWave), lds, (hipStream_t)stream, P, \
                           *desc, weights_dev, hidden0_dev, policy0_dev, train, act);                                  \
        snprintf(h->last_kernel, sizeof(h->last_kernel), PX ? "k_search_vision<%d, %s, true>" : "k_search_vision<%d, %s>", MA, \
                 EQ ? "true" : "false");                                                                               \
    }
    if (h->maxa == 2 && P.A == 2) SMZ_LAUNCH_VS(2, true)
    else if (h->maxa == 2) SMZ_LAUNCH_VS(2, false)
    else if (P.A == 4) SMZ_LAUNCH_VS(4, true)
    else SMZ_LAUNCH_VS(4, false)
#undef SMZ_LAUNCH_VS
#undef SMZ_LAUNCH_VS2
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
