// smz_search_reg.hip -- the whole Monte_carlo_tree_search.run (mcts:311-349) of every tree in ONE launch for the network
// shape the reference ships (state_space_dimensions 31, hidden_layer_dimensions 64, number_of_hidden_layer 0, 2 or 4
// actions, maxium_action_sample 2), with the networks' weights in REGISTERS and the layers on the matrix cores.
//
// k_search_mlp (smz_kernels.hip) keeps one LDS copy of the weights per workgroup and evaluates a wave's two leaves on the
// vector units: every round each of the 8 waves re-reads ~80 KB of weights from LDS (the LDS pipe is ~90 % busy in that
// phase) and the tails of a leaf (softmax, support decode, min-max scaling: reductions over 31 outputs) occupy a whole
// wavefront per leaf.  Here a wavefront owns FOUR trees and runs alone on its SIMD (4096 trees = 1024 wavefronts = one per
// SIMD of the chip), so it has 512 registers per lane:
//   * the eight matrices of the four recurrent networks (392 registers per lane) stay resident for the whole search;
//   * a layer is a chain of v_mfma_f32_4x4x1_16B_f32: 16 blocks of (4 output neurons x 4 leaves), one input per
//     instruction -- A operand = the lane's weight of that input, B operand = that input of the lane's leaf (16-byte LDS
//     reads serve four inputs), accumulators = 4 registers.  Even and odd inputs accumulate separately and are added at
//     the end, exactly as smz_mlp::dense() does on the vector units, and an f32 MFMA is an fma with one rounding
//     (tools/mfma4_probe.hip), so the outputs are bit-identical to the vector-unit heads;
//   * the tails work on that layout directly -- lane = (block, leaf), register = neuron within the block -- for the four
//     leaves at once.  The sums follow the association of smz_mlp::wave_sum (a balanced tree over the output index:
//     registers, then lanes ^ 4, ^ 8 on the DPP crossbar, then rows ^ 16, ^ 32 with v_permlane16/32_swap), so they
//     round identically too;
//   * no workgroup barrier, no LDS weight traffic; the tree phases (expand + backup, select) run in lanes 0..3.
// Same results as k_search_mlp and as the step-wise kernels, bit for bit (tests/test_gpu_end_to_end.py,
// tests/test_gpu_fullsize_parity.py).
#define SMZ_PART 5
#include "smz_kernels.hip"

using smz_mlp::elu;
using smz_mlp::lds_sync;
using smz_mlp::smz_exp;
using smz_mlp::support_to_scalar;
using smz_mlp::up4;

namespace {

constexpr int kRW = 4, kRT = 4;                      // wavefronts per workgroup, trees per wavefront
constexpr int kS = kFastS, kH = kFastH;              // 31, 64
constexpr int kXS = 36, kTS = 68, kHS2 = 36;         // floats per leaf row of the input / trunk / hidden tiles (bank spread)
typedef float v4f __attribute__((ext_vector_type(4)));

struct RegLds {                                      // float offsets from the dynamic LDS base
    int pbc, bias, pin, wave, per_wave, scratch, x, ta, hb, outs, pv, rng, prof, total;
};
__host__ __device__ inline RegLds reg_lds(const Params &P, int A) {
    RegLds m;
    m.pbc = 0;
    m.bias = r4(2 * 2 * (P.sims + 2));
    m.pin = m.bias + 8 * kWave;                      // the two prediction input matrices (32 x 64 each): read as A operands
    m.wave = m.pin + 2 * 32 * kWave;
    m.scratch = 0;                                   // root evaluation (smz_mlp::initial_row): 36 + 64 + 32 floats
    m.x = 132;
    m.ta = m.x + kRT * kXS;
    m.hb = m.ta + 2 * kRT * kTS;                     // trunk and hidden tiles: one set per branch
    m.outs = m.hb + 2 * kRT * kHS2;
    m.pv = r4(m.outs + kRT * (A + 2));
    m.rng = m.pv + kRT * P.P * 4;
    m.prof = m.rng + r4(kRT * kRngStride);
    m.per_wave = m.prof + 16;
    m.total = m.wave + kRW * m.per_wave;
    return m;
}

// ---- cross-lane pieces of the tails ------------------------------------------------------------------------------------
// v + (v of lane ^ 4) / (lane ^ 8): two DPP instructions, one per half of the lanes (bank_mask selects the 4-lane groups);
// two values per call, so that each instruction fills the other's DPP wait state
#define SMZ_XOR_DPP2(NAME, INSN, SH, MHI, MLO)                                                             \
    __device__ inline void NAME(float &a, float &b) {                                                      \
        float t, u;                                                                                        \
        asm("s_nop 1\n\t"                                                                                  \
            INSN " %0, %2, %2 row_shr:" SH " row_mask:0xf bank_mask:" MHI "\n\t"                           \
            INSN " %1, %3, %3 row_shr:" SH " row_mask:0xf bank_mask:" MHI "\n\t"                           \
            INSN " %0, %2, %2 row_shl:" SH " row_mask:0xf bank_mask:" MLO "\n\t"                           \
            INSN " %1, %3, %3 row_shl:" SH " row_mask:0xf bank_mask:" MLO                                  \
            : "=&v"(t), "=&v"(u) : "v"(a), "v"(b));                                                        \
        a = t; b = u;                                                                                      \
    }
SMZ_XOR_DPP2(add2_x4, "v_add_f32_dpp", "4", "0xa", "0x5")
SMZ_XOR_DPP2(add2_x8, "v_add_f32_dpp", "8", "0xc", "0x3")
SMZ_XOR_DPP2(max2_x4, "v_max_f32_dpp", "4", "0xa", "0x5")
SMZ_XOR_DPP2(max2_x8, "v_max_f32_dpp", "8", "0xc", "0x3")
SMZ_XOR_DPP2(min2_x4, "v_min_f32_dpp", "4", "0xa", "0x5")
SMZ_XOR_DPP2(min2_x8, "v_min_f32_dpp", "8", "0xc", "0x3")
#undef SMZ_XOR_DPP2
// (v of lane ^ 16, v of lane ^ 32) partners through the row swaps of gfx950
__device__ inline float other16(float v) {           // value of lane ^ 16
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);     // r[0] = rows (0,0,2,2), r[1] = rows (1,1,3,3)
    const int lane = threadIdx.x & 63;
    return __uint_as_float((lane & 16) ? r[0] : r[1]);
}
__device__ inline float other32(float v) {           // value of lane ^ 32
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0] = halves (lo,lo), r[1] = halves (hi,hi)
    const int lane = threadIdx.x & 63;
    return __uint_as_float((lane & 32) ? r[0] : r[1]);
}
// Reductions over the 64 output positions of a leaf, N independent ones at a time (level by level, so that their latencies
// overlap).  Sums follow the association of smz_mlp::wave_sum: s[i] enters as the lane's ((p0 + p1) + (p2 + p3)) of its four
// positions 4 * block + r (exact zeros where a position is not a member); then lanes ^ 4, ^ 8, ^ 16, ^ 32.  Every lane of
// the leaf gets the total.
template <int N>
__device__ inline void tree_sum(float (&s)[N]) {
    float pad = 0.f;
#pragma unroll
    for (int i = 0; i < N; i += 2) { if (i + 1 < N) add2_x4(s[i], s[i + 1]); else add2_x4(s[i], pad); }
#pragma unroll
    for (int i = 0; i < N; i += 2) { if (i + 1 < N) add2_x8(s[i], s[i + 1]); else add2_x8(s[i], pad); }
    float o[N];
#pragma unroll
    for (int i = 0; i < N; i++) o[i] = other16(s[i]);
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = s[i] + o[i];
#pragma unroll
    for (int i = 0; i < N; i++) o[i] = other32(s[i]);
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = s[i] + o[i];
}
template <int N>
__device__ inline void tree_max(float (&s)[N]) {
    float pad = 0.f;
#pragma unroll
    for (int i = 0; i < N; i += 2) { if (i + 1 < N) max2_x4(s[i], s[i + 1]); else max2_x4(s[i], pad); }
#pragma unroll
    for (int i = 0; i < N; i += 2) { if (i + 1 < N) max2_x8(s[i], s[i + 1]); else max2_x8(s[i], pad); }
    float o[N];
#pragma unroll
    for (int i = 0; i < N; i++) o[i] = other16(s[i]);
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = fmaxf(s[i], o[i]);
#pragma unroll
    for (int i = 0; i < N; i++) o[i] = other32(s[i]);
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = fmaxf(s[i], o[i]);
}
template <int N>
__device__ inline void tree_min(float (&s)[N]) {
    float pad = 0.f;
#pragma unroll
    for (int i = 0; i < N; i += 2) { if (i + 1 < N) min2_x4(s[i], s[i + 1]); else min2_x4(s[i], pad); }
#pragma unroll
    for (int i = 0; i < N; i += 2) { if (i + 1 < N) min2_x8(s[i], s[i + 1]); else min2_x8(s[i], pad); }
    float o[N];
#pragma unroll
    for (int i = 0; i < N; i++) o[i] = other16(s[i]);
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = fminf(s[i], o[i]);
#pragma unroll
    for (int i = 0; i < N; i++) o[i] = other32(s[i]);
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = fminf(s[i], o[i]);
}

// ---- layers -------------------------------------------------------------------------------------------------------------
// weights of the lane's output neuron (= lane) of a packed matrix (4-way interleaved input-major, 64 outputs wide)
template <int K4>
__device__ inline void load_w(float (&w)[K4], const float *W, int lane) {
#pragma unroll
    for (int k = 0; k < K4; k++) w[k] = W[((k >> 2) * kWave + lane) * 4 + (k & 3)];
}
struct NetRegs {                                     // the two networks of one branch: (afterstate) dynamics + (afterstate) prediction
    float din[kXS], dout[kH], pout[kH];         // (the 31-input prediction trunk matrices are read from LDS: registers)
};
__device__ inline void load_net(NetRegs &n, const float *weights, const smz_mlp_desc &d, bool dyn, int lane) {
    load_w<kXS>(n.din, weights + (dyn ? d.off[smz_mlp::M_DYN_IN] : d.off[smz_mlp::M_ADY_IN]), lane);
    load_w<kH>(n.dout, weights + (dyn ? d.off[smz_mlp::M_DYN_OUT] : d.off[smz_mlp::M_ADY_OUT]), lane);
    load_w<kH>(n.pout, weights + (dyn ? d.off[smz_mlp::M_PRE_OUT] : d.off[smz_mlp::M_APR_OUT]), lane);
}
// The same layer of the branches b with ON[b]: y[b][r] = bias_b[4 * block + r] + sum_k W_b[k][4 * block + r] * x_b[leaf][k] for
// the lane's (block, leaf).  Even inputs in one accumulator chain (from the bias), odd inputs in the other (from zero), added
// at the end -- smz_mlp::dense()'s order; two branches give four independent chains (one wave issues a 4x4x1 MFMA per ~8.5
// cycles from three chains on, per ~9.5 from two: tools/mfma4_probe.hip).
template <int K4, bool D, bool Y>
__device__ inline void layers(const float (&wd)[K4], const float (&wy)[K4], const float *bias_d, const float *bias_y,
                              const float *xd, const float *xy, int blk, v4f (&y)[2]) {
    constexpr int NC = K4 / 4;
    v4f e[2], o[2];
    float4 x[2][2];
    if (D) { const float4 b = *reinterpret_cast<const float4 *>(bias_d + 4 * blk); e[0] = v4f{b.x, b.y, b.z, b.w}; o[0] = v4f{0.f, 0.f, 0.f, 0.f}; x[0][0] = *reinterpret_cast<const float4 *>(xd); }
    if (Y) { const float4 b = *reinterpret_cast<const float4 *>(bias_y + 4 * blk); e[1] = v4f{b.x, b.y, b.z, b.w}; o[1] = v4f{0.f, 0.f, 0.f, 0.f}; x[1][0] = *reinterpret_cast<const float4 *>(xy); }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (c + 1 < NC) {
            if (D) x[0][(c + 1) & 1] = *reinterpret_cast<const float4 *>(xd + 4 * (c + 1));
            if (Y) x[1][(c + 1) & 1] = *reinterpret_cast<const float4 *>(xy + 4 * (c + 1));
        }
        if (D) e[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[4 * c + 0], x[0][c & 1].x, e[0], 0, 0, 0);
        if (Y) e[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wy[4 * c + 0], x[1][c & 1].x, e[1], 0, 0, 0);
        if (D) o[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[4 * c + 1], x[0][c & 1].y, o[0], 0, 0, 0);
        if (Y) o[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wy[4 * c + 1], x[1][c & 1].y, o[1], 0, 0, 0);
        if (D) e[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[4 * c + 2], x[0][c & 1].z, e[0], 0, 0, 0);
        if (Y) e[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wy[4 * c + 2], x[1][c & 1].z, e[1], 0, 0, 0);
        if (D) o[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[4 * c + 3], x[0][c & 1].w, o[0], 0, 0, 0);
        if (Y) o[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wy[4 * c + 3], x[1][c & 1].w, o[1], 0, 0, 0);
    }
    if (D) y[0] = e[0] + o[0];
    if (Y) y[1] = e[1] + o[1];
}

// the same with the A operands read from an LDS copy of the packed matrices (16-byte read = the lane's weights of four inputs)
template <int K4, bool D, bool Y>
__device__ inline void layers_lds(const float *wd, const float *wy, const float *bias_d, const float *bias_y, const float *xd,
                                  const float *xy, int blk, int lane, v4f (&y)[2]) {
    constexpr int NC = K4 / 4;
    v4f e[2], o[2];
    float4 x[2][2], w[2][2];
    if (D) { const float4 b = *reinterpret_cast<const float4 *>(bias_d + 4 * blk); e[0] = v4f{b.x, b.y, b.z, b.w}; o[0] = v4f{0.f, 0.f, 0.f, 0.f};
             x[0][0] = *reinterpret_cast<const float4 *>(xd); w[0][0] = *reinterpret_cast<const float4 *>(wd + 4 * lane); }
    if (Y) { const float4 b = *reinterpret_cast<const float4 *>(bias_y + 4 * blk); e[1] = v4f{b.x, b.y, b.z, b.w}; o[1] = v4f{0.f, 0.f, 0.f, 0.f};
             x[1][0] = *reinterpret_cast<const float4 *>(xy); w[1][0] = *reinterpret_cast<const float4 *>(wy + 4 * lane); }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (c + 1 < NC) {
            if (D) { x[0][(c + 1) & 1] = *reinterpret_cast<const float4 *>(xd + 4 * (c + 1)); w[0][(c + 1) & 1] = *reinterpret_cast<const float4 *>(wd + ((c + 1) * kWave + lane) * 4); }
            if (Y) { x[1][(c + 1) & 1] = *reinterpret_cast<const float4 *>(xy + 4 * (c + 1)); w[1][(c + 1) & 1] = *reinterpret_cast<const float4 *>(wy + ((c + 1) * kWave + lane) * 4); }
        }
        if (D) e[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0][c & 1].x, x[0][c & 1].x, e[0], 0, 0, 0);
        if (Y) e[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[1][c & 1].x, x[1][c & 1].x, e[1], 0, 0, 0);
        if (D) o[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0][c & 1].y, x[0][c & 1].y, o[0], 0, 0, 0);
        if (Y) o[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[1][c & 1].y, x[1][c & 1].y, o[1], 0, 0, 0);
        if (D) e[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0][c & 1].z, x[0][c & 1].z, e[0], 0, 0, 0);
        if (Y) e[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[1][c & 1].z, x[1][c & 1].z, e[1], 0, 0, 0);
        if (D) o[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0][c & 1].w, x[0][c & 1].w, o[0], 0, 0, 0);
        if (Y) o[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[1][c & 1].w, x[1][c & 1].w, o[1], 0, 0, 0);
    }
    if (D) y[0] = e[0] + o[0];
    if (Y) y[1] = e[1] + o[1];
}

// The networks of the branches the wave's four leaves are on (monte_carlo_tree_search.py:333-342, smz_mlp::recurrent_rows):
// D = some leaf took the dynamics + prediction branch, Y = some leaf the afterstate pair; both together run interleaved,
// layer by layer.  Results are committed -- hidden row to the tree, policy / value / reward to `outs` -- per leaf for its
// own branch: cd / cy = the lane's leaf is live and on that branch; dst = the leaf's new hidden row.
// ta, hb: trunk / hidden-state tiles of the two branches, [2][kRT][kTS] and [2][kRT][kHS2].
template <int A, bool D, bool Y>
__device__ inline void eval_branches(const NetRegs &nd, const NetRegs &ny, const float *bias, const float *pin, const float *xt, float *ta, float *hb,
                                     float *outs, bool cd, bool cy, float *dst, int lane) {
    constexpr int S = kS, half = S / 2, slot = A + 2;
    const int blk = lane >> 2, q = lane & 3;
    const float *bd = bias, *by = bias + 4 * kWave;
    float *tad = ta + q * kTS, *tay = ta + (kRT + q) * kTS, *hbd = hb + q * kHS2, *hby = hb + (kRT + q) * kHS2;
    v4f y[2];
    // (afterstate) dynamics: trunk
    layers<kXS, D, Y>(nd.din, ny.din, bd, by, xt + q * kXS, xt + q * kXS, blk, y);
    if (D) *reinterpret_cast<float4 *>(tad + 4 * blk) = make_float4(elu(y[0][0]), elu(y[0][1]), elu(y[0][2]), elu(y[0][3]));
    if (Y) *reinterpret_cast<float4 *>(tay + 4 * blk) = make_float4(elu(y[1][0]), elu(y[1][1]), elu(y[1][2]), elu(y[1][3]));
    lds_sync();
    layers<kH, D, Y>(nd.dout, ny.dout, bd + kWave, by + kWave, tad, tay, blk, y);
    // dynamics: [reward logits 0..S-1 | next state S..2S-1] (smz_mlp::decode_scale_lanes); afterstate dynamics: next state
    // 0..S-1 (smz_mlp::scale_lanes)
    float reward = 0.f;
    {
        float mn[2] = {__builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int o = 4 * blk + r;
            if (D) {
                if (o < S) mx[2] = fmaxf(mx[2], y[0][r]);
                else if (o < 2 * S) { mn[0] = fminf(mn[0], y[0][r]); mx[0] = fmaxf(mx[0], y[0][r]); }
            }
            if (Y && o < S) { mn[1] = fminf(mn[1], y[1][r]); mx[1] = fmaxf(mx[1], y[1][r]); }
        }
        tree_min<2>(mn);
        tree_max<3>(mx);
        if (D) {
            float dn[2] = {0.f, 0.f}, de[4], nu[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = 4 * blk + r;
                const float e = o < S ? smz_exp(y[0][r] - mx[2]) : 0.f;
                de[r] = e;
                nu[r] = o < S ? 0.f + (float)(o - half) * e : 0.f;
            }
            dn[0] = (de[0] + de[1]) + (de[2] + de[3]);
            dn[1] = (nu[0] + nu[1]) + (nu[2] + nu[3]);
            tree_sum<2>(dn);
            reward = support_to_scalar(dn[1], dn[0]);
        }
        float scd = mx[0] - mn[0], scy = mx[1] - mn[1];
        if (scd < 1e-5f) scd += 1e-5f;
        if (scy < 1e-5f) scy += 1e-5f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int o = 4 * blk + r;
            if (D && o >= S && o < 2 * S && cd) {
                const float hv = __fdividef(y[0][r] - mn[0], scd);
                hbd[o - S] = hv;
                dst[o - S] = hv;
            }
            if (Y && o < S && cy) {
                const float hv = __fdividef(y[1][r] - mn[1], scy);
                hby[o] = hv;
                dst[o] = hv;
            }
        }
    }
    lds_sync();
    // (afterstate) prediction on the new hidden state
    layers_lds<32, D, Y>(pin, pin + 32 * kWave, bd + 2 * kWave, by + 2 * kWave, hbd, hby, blk, lane, y);
    if (D) *reinterpret_cast<float4 *>(tad + 4 * blk) = make_float4(elu(y[0][0]), elu(y[0][1]), elu(y[0][2]), elu(y[0][3]));
    if (Y) *reinterpret_cast<float4 *>(tay + 4 * blk) = make_float4(elu(y[1][0]), elu(y[1][1]), elu(y[1][2]), elu(y[1][3]));
    lds_sync();
    layers<kH, D, Y>(nd.pout, ny.pout, bd + 3 * kWave, by + 3 * kWave, tad, tay, blk, y);
    {   // [policy logits 0..A-1 | value logits A..A+S-1] (smz_mlp::softmax_decode_lanes)
        float m[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};     // policy d, value d, policy y, value y
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = 4 * blk + r;
                if (!(b ? Y : D)) continue;
                if (o < A) m[2 * b] = fmaxf(m[2 * b], y[b][r]);
                else if (o < A + S) m[2 * b + 1] = fmaxf(m[2 * b + 1], y[b][r]);
            }
        tree_max<4>(m);
        float e[2][4], sums[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};             // dp, dv, nv of d | of y
#pragma unroll
        for (int b = 0; b < 2; b++) {
            if (!(b ? Y : D)) continue;
            float ep[4], ev[4], nv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = 4 * blk + r;
                const bool pol = o < A, val = !pol && o < A + S;
                e[b][r] = (pol || val) ? smz_exp(y[b][r] - (pol ? m[2 * b] : m[2 * b + 1])) : 0.f;
                ep[r] = pol ? e[b][r] : 0.f;
                ev[r] = val ? e[b][r] : 0.f;
                nv[r] = val ? 0.f + (float)(o - A - half) * e[b][r] : 0.f;
            }
            sums[3 * b + 0] = (ep[0] + ep[1]) + (ep[2] + ep[3]);
            sums[3 * b + 1] = (ev[0] + ev[1]) + (ev[2] + ev[3]);
            sums[3 * b + 2] = (nv[0] + nv[1]) + (nv[2] + nv[3]);
        }
        if (D && Y) tree_sum<6>(sums);
        else if (D) { float t[3] = {sums[0], sums[1], sums[2]}; tree_sum<3>(t); sums[0] = t[0]; sums[1] = t[1]; sums[2] = t[2]; }
        else { float t[3] = {sums[3], sums[4], sums[5]}; tree_sum<3>(t); sums[3] = t[0]; sums[4] = t[1]; sums[5] = t[2]; }
        if (D && cd && blk == 0) {
            const float value = support_to_scalar(sums[2], sums[1]);
#pragma unroll
            for (int r = 0; r < 4; r++) if (r < A) outs[q * slot + r] = __fdividef(e[0][r], sums[0]);
            outs[q * slot + A] = value;
            outs[q * slot + A + 1] = reward;
        }
        if (Y && cy && blk == 0) {
            const float value = support_to_scalar(sums[5], sums[4]);
#pragma unroll
            for (int r = 0; r < 4; r++) if (r < A) outs[q * slot + r] = __fdividef(e[1][r], sums[3]);
            outs[q * slot + A] = value;
            outs[q * slot + A + 1] = 0.f;
        }
    }
    lds_sync();
}

extern __shared__ float4 smz_reg_lds4[];

template <int A>
__global__ void __launch_bounds__(kRW *kWave) k_search_mlp_reg(Params Pin, smz_mlp_desc d, const float *__restrict__ weights,
                                                               const float *__restrict__ obs, int train, ActOut act) {
    constexpr int KS = 2, MAXA = A, slot = A + 2;
    Params P = Pin;
    P.tree0 = 0;
    P.A = A; P.tpw = kRT; P.K = KS; P.S = kS;
    d.A = A; d.S = kS; d.H = kH; d.L = 0; d.OP = kWave;
    fix_layout(P, true, true);
    P.hs = (kS + 15) & ~15;
    float *lds = reinterpret_cast<float *>(smz_reg_lds4);
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const RegLds ml = reg_lds(P, A);
    double *pbc_lds = reinterpret_cast<double *>(lds + ml.pbc);
    const int n_pbc = P.sims + 2;
    for (int i = threadIdx.x; i < n_pbc; i += blockDim.x) {
        pbc_lds[i] = P.pbc_sqrt[i];
        pbc_lds[n_pbc + i] = i > 0 ? 1.0 / (double)i : 0.0;      // IEEE division: correctly rounded reciprocals
    }
    {   // biases of the eight matrices: [dyn in, dyn out, pre in, pre out | ady in, ady out, apr in, apr out] x 64
        const int mats[8] = {smz_mlp::M_DYN_IN, smz_mlp::M_DYN_OUT, smz_mlp::M_PRE_IN, smz_mlp::M_PRE_OUT,
                             smz_mlp::M_ADY_IN, smz_mlp::M_ADY_OUT, smz_mlp::M_APR_IN, smz_mlp::M_APR_OUT};
        for (int i = threadIdx.x; i < 8 * kWave; i += blockDim.x) {
            int off = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) if ((i >> 6) == m) off = d.off[smz_mlp::M_COUNT + mats[m]];
            lds[ml.bias + i] = weights[off + (i & 63)];
        }
    }
    for (int i = threadIdx.x; i < 32 * kWave / 4; i += blockDim.x) {
        reinterpret_cast<float4 *>(lds + ml.pin)[i] = reinterpret_cast<const float4 *>(weights + d.off[smz_mlp::M_PRE_IN])[i];
        reinterpret_cast<float4 *>(lds + ml.pin + 32 * kWave)[i] = reinterpret_cast<const float4 *>(weights + d.off[smz_mlp::M_APR_IN])[i];
    }
    float *wl = lds + ml.wave + wave * ml.per_wave;
    for (int i = lane; i < ml.per_wave; i += kWave) wl[i] = 0.f;
    float *scratch = wl + ml.scratch, *xt = wl + ml.x, *ta = wl + ml.ta, *hb = wl + ml.hb, *outs = wl + ml.outs;
    uint4 *pvals = reinterpret_cast<uint4 *>(wl + ml.pv);
    uint32_t *rng_tile = reinterpret_cast<uint32_t *>(wl + ml.rng);
    __syncthreads();

    const int tree0 = (blockIdx.x * kRW + wave) * kRT;
    const int tree = tree0 + lane;
    const bool valid = lane < kRT && tree < P.B && tree_active(P, tree);
    if (__ballot(valid) == 0ull) return;                 // (no workgroup barrier below)
    const int q = lane & 3;
    const bool live_q = __shfl((int)valid, q) != 0;      // the lane's leaf column belongs to a searched tree

    // ---- root: representation + prediction per tree on the vector units, weights from global memory (once per search) ----
    for (int t = 0; t < kRT; t++) {
        if (!__shfl((int)valid, t)) continue;            // wave-uniform
        const int row = tree0 + t;
        smz_mlp::initial_row<1>(weights, d, weights, d, scratch, obs + (size_t)row * d.obs, P.hidden + (size_t)row * P.N * P.hs,
                                nullptr, outs + t * slot);
    }
    // ---- the recurrent networks into registers ----------------------------------------------------------------------------
    NetRegs nd, na;
    load_net(nd, weights, d, true, lane);
    load_net(na, weights, d, false, lane);

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

    // (SMZ_DEBUG_SKIP=64 with statistics on: s_memtime phase accounting -> stats[8..11] = tree | inputs | networks | staging)
    const bool prof = (P.dbg & 64) && P.stats;
    unsigned long long *pc = reinterpret_cast<unsigned long long *>(wl + ml.prof);
    unsigned long long t0 = 0;
#define SMZ_RSTAMP(i) if (prof) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) pc[i] += t1 - t0; t0 = t1; }
    const float *bias = lds + ml.bias, *pin = lds + ml.pin;
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
        SMZ_RSTAMP(0)
        // hidden rows written in earlier rounds (by this wave) may be this round's parent rows
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        StagePre<kRT> pre;
        stage_issue<kRT, false>(P, tree, valid, packed, pre);
        // network inputs [hidden | one-hot action] of the four leaves (muzero_model.py:496-523)
        {
            const int parent = __shfl(L.parent_id, lane >> 4), actn = __shfl(L.action, lane >> 4);   // 16 lanes per leaf
            const float *src = P.hidden + ((size_t)(tree0 + (lane >> 4)) * P.N + parent) * P.hs;
            const bool lv = __shfl((int)valid, lane >> 4) != 0;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int k = (lane & 15) + 16 * j;
                if (k < kXS) xt[(lane >> 4) * kXS + k] = (k < kS) ? (lv ? src[k] : 0.f) : ((k < kS + A && (k - kS) == actn) ? 1.f : 0.f);
            }
        }
        const int br = __shfl(L.branch, q);
        const int leaf_id = __shfl(L.leaf_id, q);
        float *dst = P.hidden + ((size_t)(tree0 + q) * P.N + leaf_id) * P.hs;
        const bool need_dyn = __ballot(live_q && br != 0) != 0ull, need_ady = __ballot(live_q && br == 0) != 0ull;
        lds_sync();
        SMZ_RSTAMP(1)
        const bool cd = live_q && br != 0, cy = live_q && br == 0;
        if (need_dyn && need_ady) eval_branches<A, true, true>(nd, na, bias, pin, xt, ta, hb, outs, cd, cy, dst, lane);
        else if (need_dyn) eval_branches<A, true, false>(nd, na, bias, pin, xt, ta, hb, outs, cd, cy, dst, lane);
        else if (need_ady) eval_branches<A, false, true>(nd, na, bias, pin, xt, ta, hb, outs, cd, cy, dst, lane);
        SMZ_RSTAMP(2)
        packed = stage_finish<kRT, false>(P, tree, valid, rng_tile, packed, pre);
        SMZ_RSTAMP(3)
    }
#undef SMZ_RSTAMP
    if (prof && lane == 0)
        for (int i = 0; i < 4; i++) atomicAdd(&P.stats[8 + i], pc[i]);
    if (valid) {
        if (P.sims > 0) {
            rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
            expand_backup_tree<MAXA, KS>(P, tree, rng, h, outs + lane * slot, outs[lane * slot + A + 1], outs[lane * slot + A],
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
}

}  // namespace

// Launcher for smz_search_mlp(_act) (smz_kernels.hip decides when this kernel applies).
int smz_internal_search_launch_reg(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev,
                                   int train, double temperature, int32_t *action, double *policy, double *child_visits,
                                   float *root_value, const double *pow_table_host, smz_stream stream) {
    const ActOut act = {temperature, action, policy, child_visits, root_value};
    Params P = h->P;
    if (act.action && pow_table_host && act.temperature >= 0.3) {       // as smz_act: the power table of this temperature
        if (!h->pow_valid || h->pow_T != act.temperature) {
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            HIP_TRY(hipMemcpy(h->d_pow, pow_table_host, ((size_t)h->cfg.num_simulations + 1) * sizeof(double), hipMemcpyHostToDevice));
            h->pow_T = act.temperature;
            h->pow_valid = true;
        }
        P.pow_table = h->d_pow;
    }
    P.tpw = kRT;
    const RegLds ml = reg_lds(P, P.A);
    const size_t lds = (size_t)ml.total * sizeof(float);
    if (lds > 160 * 1024) return fail(SMZ_ERR_TOO_LARGE, "smz_search_mlp: working set exceeds the 160 KB LDS of a CU%s");
    const int blocks = (P.B + kRW * kRT - 1) / (kRW * kRT);
#define SMZ_LAUNCH_REG(AA)                                                                                             \
    {                                                                                                                  \
        static size_t granted_dev[64] = {};                                                                            \
        size_t &granted = granted_dev[h->cfg.device & 63];                                                             \
        if (lds > granted) {                                                                                           \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_mlp_reg<AA>),                              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)               \
                return fail(SMZ_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed%s");                             \
            granted = lds;                                                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL((k_search_mlp_reg<AA>), dim3(blocks), dim3(kRW * kWave), lds, (hipStream_t)stream, P,       \
                           *desc, weights_dev, obs_dev, train, act);                                                   \
    }
    if (P.A == 2) SMZ_LAUNCH_REG(2) else SMZ_LAUNCH_REG(4)
#undef SMZ_LAUNCH_REG
    h->root_ready = true;
    h->selected = false;
    return launch_check();
}
