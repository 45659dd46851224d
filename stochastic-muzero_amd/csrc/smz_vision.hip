// smz_vision.hip -- fused head kernels of the `vision_model` family (C ABI: smz_vision_layout / _initial / _recurrent).
//
// What they replace: the five *_inference calls of the reference on its ResNet-v2 family
// (neural_network_vision_model.py:41-515 through muzero_model.py:802-909), which per leaf is a chain of ~60 tiny
// convolutions / batch-norms / linears on a 3x7x7 hidden state -- hopelessly launch-bound as library calls (measured:
// 4.4 ms per simulation round at 1024 trees through torch-ROCm modules).  Here one wavefront evaluates one leaf:
//   lane p < 49 <-> pixel (p / 7, p % 7); the 3 channels of a pixel live in that lane's registers;
//   3x3 convolutions read their 9 neighbours as 16-byte LDS loads from a zero-bordered 9x9 plane of float4
//   (c0,c1,c2,action plane); the 81/108 convolution weights of a layer are wave-uniform reads of the workgroup's LDS
//   copy of the small pieces (the towers' matrices are far too large for that);
//   the 147 -> H -> S/A towers reuse the MLP family's dense() (one output neuron per lane, weights streamed from
//   L2 in the 4-way interleaved layout), ReLU instead of ELU.
// The representation network (98x98x3 frame -> 3x7x7) runs once per search: one 256-thread workgroup per frame with
// all feature maps in LDS.
//
// Eval-mode batch-norm is folded on the host to y = x * scale + shift exactly as ATen does on the CPU
// (scale = weight / sqrt(var + eps), shift = bias - mean * scale); multiply and add stay separate instructions.
// Floating-point parity class: 2e-5 on hidden planes / policies against the reference's recorded outputs
// (tests/test_gpu_end_to_end.py::test_vision_*), like the MLP family.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/smz.h"
#include "smz_vision_device.hpp"

using namespace smz_mlp;
using namespace smz_vision;

namespace {

constexpr int kRecWaves = 4;

#ifdef SMZ_VISION_STAMPS        // tools/vision_probe.hip: s_memtime phase accounting, not part of the library build
__device__ unsigned long long smz_vision_stamps[8];
#define SMZ_STAMP(i)                                                                                  \
    do {                                                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                 \
        if (lane == 0) atomicAdd(&smz_vision_stamps[i], now_ - last_);                                \
        last_ = now_;                                                                                 \
    } while (0)
#define SMZ_STAMP_INIT() unsigned long long last_ = __builtin_amdgcn_s_memtime()
__device__ unsigned long long smz_rep_stamps[12];      // k_vision_initial: per stage, thread 0 of every workgroup
#define SMZ_RSTAMP(i)                                                                                 \
    do {                                                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                 \
        if (threadIdx.x == 0) atomicAdd(&smz_rep_stamps[i], now_ - last_);                            \
        last_ = now_;                                                                                 \
    } while (0)
#else
#define SMZ_STAMP(i) do {} while (0)
#define SMZ_RSTAMP(i) do {} while (0)
#define SMZ_STAMP_INIT() do {} while (0)
#endif

// One wavefront per leaf: (afterstate) dynamics + (afterstate) prediction, selected by the leaf's branch flag
// (monte_carlo_tree_search.py:333-342).  parent_hidden [B, ld] (first 147 floats used), last_action [B], branch [B].
__global__ void __launch_bounds__(kRecWaves *kWave) k_vision_recurrent(
    smz_vision_desc d, const float *__restrict__ weights, const float *__restrict__ parent_hidden, int ld,
    const int32_t *__restrict__ last_action, const uint8_t *__restrict__ branch, float *__restrict__ hidden_out,
    float *__restrict__ reward_out, float *__restrict__ policy_out, float *__restrict__ value_out, int B) {
    __shared__ WaveLds lds[kRecWaves];
    __shared__ float4 small4[kSmallMax / 4];
    // convolution / batch-norm / 1x1 weights of the four nets: one coalesced copy per workgroup.  (Read through the
    // scalar cache instead, every layer of every leaf misses it once per launch -- the cache starts cold and all leaf
    // wavefronts reach a layer together -- and each miss queues behind the tower weight stream in L2.)
    for (int i = threadIdx.x; i < d.small_floats / 4; i += blockDim.x)
        small4[i] = reinterpret_cast<const float4 *>(weights)[i];
    __syncthreads();
    const float *small = reinterpret_cast<const float *>(small4);
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int row = blockIdx.x * kRecWaves + wave;
    if (row >= B) return;                                   // wave-uniform (after the only workgroup barrier)
    WaveLds &l = lds[wave];
    SMZ_STAMP_INIT();
    zero_wave_lds(l, lane);
    const bool active = lane < kPix;
    const int p = active ? lane : kPix - 1, pp = (p / kN + 1) * kPad + (p % kN + 1);
    const bool dyn = __builtin_amdgcn_readfirstlane((int)branch[row]) != 0;     // uniform: picks SGPR weight pointers
    const float a_plane = (float)(last_action[row] + 1) / (float)d.A;         // muzero_model.py:511-522
    const float *hrow = parent_hidden + (size_t)row * ld;
    float x[4] = {hrow[p], hrow[kPix + p], hrow[2 * kPix + p], a_plane};
    const int32_t *o = d.off + SMZ_V_TRANS_BASE + (dyn ? 0 : SMZ_V_TRANS_STRIDE);

    // reward branch first (it reads x only): conv1x1(4->3) -> flatten -> tower -> support decode
    float reward = 0.f;
    if (dyn) {
        float acc[1][1];
        mix_to_flat<4>(l.flat, p, active, x, uniform_ptr(small, o[SMZ_VT_MIX_W]), uniform_ptr(small, o[SMZ_VT_MIX_B]));
        tower(weights, o + SMZ_VT_TOWER, l, d, lane, acc);
        reward = decode_lanes<1>(acc[0], 0, d.S, lane);
        lds_sync();
    }
    SMZ_STAMP(0);
    // next state: conv3x3(4->3) bn relu [block] x L relu, scaled per pixel
    if (active) l.plane[pp] = make_float4(x[0], x[1], x[2], x[3]);
    lds_sync();
    float t[kC];
    SMZ_STAMP(3);
    conv3x3<4>(l.plane, pp, uniform_ptr(small, o[SMZ_VT_CONV_IN]), t);
    lds_sync();
    SMZ_STAMP(4);
    {
        const float *bn = uniform_ptr(small, o[SMZ_VT_BN_IN]);
#pragma unroll
        for (int c = 0; c < kC; c++) t[c] = fmaxf(t[c] * bn[c] + bn[kC + c], 0.f);
    }
    {
        const float *wa = uniform_ptr(small, o[SMZ_VT_RES_A]), *wb = uniform_ptr(small, o[SMZ_VT_RES_B]);
        const float *bn = uniform_ptr(small, o[SMZ_VT_RES_BN]);
        for (int i = 0; i < d.L; i++) residual_block(l.plane, pp, active, wa, wb, bn, t);
    }
    SMZ_STAMP(5);
#pragma unroll
    for (int c = 0; c < kC; c++) t[c] = fmaxf(t[c], 0.f);
    scale_channels(t);
    if (active) {
        float *ho = hidden_out + (size_t)row * kFlat;
        ho[p] = t[0]; ho[kPix + p] = t[1]; ho[2 * kPix + p] = t[2];
    }
    SMZ_STAMP(1);
    const float value = predict(weights, small, d, dyn ? SMZ_V_PRE : SMZ_V_APR, l, lane, p, pp, active, t,
                                policy_out + (size_t)row * d.A, true);
    SMZ_STAMP(2);
    if (lane == 0) {
        reward_out[row] = reward;
        value_out[row] = value;
    }
}

// -------------------------------------------------------------------------------------------------------------------
// representation: one workgroup per frame, feature maps in LDS
// -------------------------------------------------------------------------------------------------------------------
constexpr int kRepThreads = 256;
constexpr int kMaxMap = 49 * 49;                 // largest feature map (1 channel, 49x49)
constexpr int kMaxPad = 51 * 51;                 // same, zero bordered (3 x 27 x 27 = 2187 is smaller)

struct RepLds {
    float t[kMaxMap];      // residual stream  [C][N][N]
    float u[kMaxPad];      // zero-bordered input of a convolution [C][N+2][N+2]
    float v[kMaxPad];      // second zero-bordered buffer: a convolution writes relu(bn(.)) of its output straight into the
                           // interior of the next convolution's input (no separate padding pass, one barrier less)
    WaveLds head;          // root prediction (wave 0)
};

// Barrier between the representation's LDS stages: waits for the wave's LDS traffic only, so the frame-copy loads and stores
// below stay in flight across it (__syncthreads also drains the vector-memory counter).
__device__ inline void rep_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The trajectory record of the frame (smz_vision_initial_record), spread over the launch: all workgroups read their frames at
// the same moment -- the stem runs at HBM speed -- so a copy issued there only queues behind it.  Each residual block moves a
// few 16-byte pieces per thread instead: loaded before a convolution, stored after it, nothing waits in between.
struct FrameCopy {
    typedef float v4f __attribute__((ext_vector_type(4)));
    static constexpr int kPieces = 3 * kFrame * kFrame / 4;
    const v4f *src;
    v4f *dst;            // nullptr: no record
    int at;              // next piece of thread 0
    template <int N> __device__ inline void load(v4f (&r)[N]) const {
        if (!dst) return;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const int i = at + k * kRepThreads + (int)threadIdx.x;
            if (i < kPieces) r[k] = src[i];        // (the stem has just read the frame: an ordinary load finds it in the caches;
                                                   //  a non-temporal one goes to HBM again: +15 us per launch instead of +4)
        }
    }
    template <int N> __device__ inline void store(const v4f (&r)[N]) {
        if (!dst) return;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const int i = at + k * kRepThreads + (int)threadIdx.x;
            if (i < kPieces) __builtin_nontemporal_store(r[k], dst + i);
        }
        at += N * kRepThreads;
    }
};

// (map sizes are template parameters: the pixel -> (row, column) divisions and the tap offsets become constants)
// u <- border-padded f(src) for a [C][N][N] map; f = relu(bn(.)) when bn != nullptr, identity otherwise
template <int C, int N>
__device__ inline void pad_store(float *u, const float *src, const float *bn) {
    constexpr int P = N + 2;
    for (int i = threadIdx.x; i < C * P * P; i += kRepThreads) {
        const int c = i / (P * P), r = i % (P * P), y = r / P - 1, x = r % P - 1;
        float val = 0.f;
        if (y >= 0 && y < N && x >= 0 && x < N) {
            val = src[(c * N + y) * N + x];
            if (bn) val = fmaxf(val * bn[c] + bn[C + c], 0.f);
        }
        u[i] = val;
    }
    rep_barrier();
}
// zero border of a [C][N+2][N+2] buffer whose interior a convolution is about to fill
template <int C, int N>
__device__ inline void zero_border(float *u) {
    constexpr int P = N + 2;
    for (int i = threadIdx.x; i < C * 4 * P; i += kRepThreads) {
        const int c = i / (4 * P), r = i % (4 * P), side = r / P, k = r % P;
        const int y = side == 0 ? 0 : (side == 1 ? P - 1 : k), x = side < 2 ? k : (side == 2 ? 0 : P - 1);
        u[(c * P + y) * P + x] = 0.f;
    }
}

// acc[oc] = sum w[oc][ic][ky][kx] * u[ic][y*stride + ky][x*stride + kx] for one output pixel of a [CIN][P][P] bordered map
// (input channels outermost, taps inside: the order of the wave-per-leaf convolutions is per tap -- the two kernels never meet
// on the same tensor)
template <int CIN, int COUT, int P, int STRIDE>
__device__ inline void conv_pixel(const float *u, int y, int x, const float *w, float (&acc)[COUT]) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float *up = u + y * STRIDE * P + x * STRIDE;
    if constexpr (COUT == 3) {      // outputs 0 and 1 advance in one v_pk_fma_f32, output 2 in a v_fma_f32
        v2f a01 = {0.f, 0.f};
        float a2 = 0.f;
#pragma unroll 1            // (rolled: fully unrolled, the 27 taps of a 3-channel pixel need ~200 registers and the kernel, capped at
                            //  128 for four workgroups per CU, spilled 88 of them to scratch memory: 153 -> 117 us)
        for (int ic = 0; ic < CIN; ic++)
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const float val = up[ic * P * P + (t / 3) * P + t % 3];
                const v2f wp = {w[(0 * CIN + ic) * 9 + t], w[(1 * CIN + ic) * 9 + t]}, vv = {val, val};
                a01 = __builtin_elementwise_fma(wp, vv, a01);
                a2 = fmaf(w[(2 * CIN + ic) * 9 + t], val, a2);
            }
        acc[0] = a01.x; acc[1] = a01.y; acc[2] = a2;
    } else {
#pragma unroll
        for (int oc = 0; oc < COUT; oc++) acc[oc] = 0.f;
#pragma unroll
        for (int ic = 0; ic < CIN; ic++)
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const float val = up[ic * P * P + (t / 3) * P + t % 3];
#pragma unroll
                for (int oc = 0; oc < COUT; oc++) acc[oc] = fmaf(w[(oc * CIN + ic) * 9 + t], val, acc[oc]);
            }
    }
}
// dst[oc][y][x] = conv (+ res[oc][y][x] if res)
template <int CIN, int COUT, int NIN, int NOUT, int STRIDE>
__device__ inline void conv_map(float *dst, const float *u, const float *w, const float *res) {
    for (int i = threadIdx.x; i < NOUT * NOUT; i += kRepThreads) {
        const int y = i / NOUT, x = i % NOUT;
        float acc[COUT];
        conv_pixel<CIN, COUT, NIN + 2, STRIDE>(u, y, x, w, acc);
#pragma unroll
        for (int oc = 0; oc < COUT; oc++) {
            const int j = oc * NOUT * NOUT + i;
            dst[j] = res ? acc[oc] + res[j] : acc[oc];
        }
    }
    rep_barrier();
}
// the same convolution writing relu(bn(conv)) into the interior of the zero-bordered buffer `un` (same size, stride 1)
template <int C, int N>
__device__ inline void conv_bn_pad(float *un, const float *u, const float *w, const float *bn) {
    constexpr int P = N + 2;
    for (int i = threadIdx.x; i < N * N; i += kRepThreads) {
        const int y = i / N, x = i % N;
        float acc[C];
        conv_pixel<C, C, P, 1>(u, y, x, w, acc);
#pragma unroll
        for (int oc = 0; oc < C; oc++) un[(oc * P + y + 1) * P + x + 1] = fmaxf(acc[oc] * bn[oc] + bn[C + oc], 0.f);
    }
    rep_barrier();
}

// v2 residual block on a map: t += convA(f(convB(f(convA(f(t)))))), f = relu(bn(.)) with ONE batch-norm
// (neural_network_vision_model.py:41-79).  `fresh`: the borders of l.v are not known to be zero for this N yet.
// `fc`: NP pieces of the frame copy per thread ride along with each of the three convolutions.
template <int C, int N, int NP>
__device__ inline void residual_map(RepLds &l, const float *wa, const float *wb, const float *bn, bool fresh, FrameCopy &fc) {
    FrameCopy::v4f r[NP];
    if (fresh) zero_border<C, N>(l.v);
    fc.load(r);
    pad_store<C, N>(l.u, l.t, bn);                     // (its barrier also covers the border zeroing)
    conv_bn_pad<C, N>(l.v, l.u, wa, bn);
    fc.store(r); fc.load(r);
    conv_bn_pad<C, N>(l.u, l.v, wb, bn);               // l.u's borders are zero from pad_store
    fc.store(r); fc.load(r);
    conv_map<C, C, N, N, 1>(l.t, l.u, wa, l.t);        // each thread reads and writes only its own pixels of t
    fc.store(r);
}

// AvgPool2d(3, stride 2, padding 1), count_include_pad: sum of the zero-padded window / 9
template <int C, int NIN, int NOUT>
__device__ inline void pool_map(RepLds &l) {
    pad_store<C, NIN>(l.u, l.t, nullptr);
    constexpr int P = NIN + 2;
    for (int i = threadIdx.x; i < C * NOUT * NOUT; i += kRepThreads) {
        const int c = i / (NOUT * NOUT), r = i % (NOUT * NOUT), y = r / NOUT, x = r % NOUT;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; t++) s += l.u[(c * P + 2 * y + t / 3) * P + 2 * x + t % 3];
        l.t[i] = s / 9.0f;
    }
    rep_barrier();
}

// (4 workgroups per CU: the register cap costs a few spills but hides the barrier chain better: 223 -> 204 us)
#ifndef SMZ_REP_WGS
#define SMZ_REP_WGS 4
#endif
__global__ void __launch_bounds__(kRepThreads, SMZ_REP_WGS) k_vision_initial(smz_vision_desc d, const float *__restrict__ weights,
                                                                const float *__restrict__ frames,
                                                                float *__restrict__ frames_copy,
                                                                float *__restrict__ hidden_out,
                                                                float *__restrict__ policy_out) {
    __shared__ RepLds l;
    const int row = blockIdx.x;
    const float *f = frames + (size_t)row * 3 * kFrame * kFrame;
    const int32_t *o = d.off + SMZ_V_REP_BASE;
    SMZ_STAMP_INIT();
    // stem: conv3x3 stride 2 pad 1, 3 -> 1 channels, 98 -> 49, straight from global memory (it runs at HBM speed: 118 MB of
    // frames in ~16 us at 1024 frames).  Output pixel (y, x) reads its taps kx = 1, 2 as ONE aligned 8-byte load per (channel,
    // row) -- consecutive lanes read consecutive 8 bytes -- and kx = 0 as a single float.
    {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const float *w = weights + o[SMZ_VR_STEM];
        for (int i = threadIdx.x; i < 49 * 49; i += kRepThreads) {       // (two pixels per trip measured slower: 60.7 k vs 54.8 k cycles)
            const int y = i / 49, x = i % 49;
            float acc = 0.f;
#pragma unroll
            for (int ic = 0; ic < 3; ic++)
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                    const int yy = 2 * y - 1 + ky, at = (ic * kFrame + yy) * kFrame + 2 * x;      // (yy <= 97, 2x + 1 <= 97)
                    const bool in = ky > 0 || y > 0;
                    const v2f c = in ? *reinterpret_cast<const v2f *>(f + at) : v2f{0.f, 0.f};
                    const float lft = (in && x > 0) ? f[at - 1] : 0.f;
                    acc = fmaf(w[ic * 9 + ky * 3 + 0], lft, acc);
                    acc = fmaf(w[ic * 9 + ky * 3 + 1], c.x, acc);
                    acc = fmaf(w[ic * 9 + ky * 3 + 2], c.y, acc);
                }
            l.t[i] = acc;
        }
        rep_barrier();
    }
    // the frame's copy into the trajectory record: 30 pieces per thread over the residual blocks (6 + 12 + 9 + 3 >= 28.2)
    FrameCopy fc{reinterpret_cast<const FrameCopy::v4f *>(f),
                 frames_copy ? reinterpret_cast<FrameCopy::v4f *>(frames_copy + (size_t)row * 3 * kFrame * kFrame) : nullptr, 0};
    SMZ_RSTAMP(0);
    {
        const float *wa = weights + o[SMZ_VR_NARROW_A], *wb = weights + o[SMZ_VR_NARROW_B], *bn = weights + o[SMZ_VR_NARROW_BN];
        residual_map<1, 49, 1>(l, wa, wb, bn, true, fc);
        residual_map<1, 49, 1>(l, wa, wb, bn, false, fc);
    }
    SMZ_RSTAMP(1);
    // widen: conv3x3 stride 2, 1 -> 3 channels, 49 -> 25
    pad_store<1, 49>(l.u, l.t, nullptr);
    conv_map<1, 3, 49, 25, 2>(l.t, l.u, weights + o[SMZ_VR_WIDEN], nullptr);     // (reads l.u only: l.t can take the result)
    SMZ_RSTAMP(2);
    {
        const float *wa = weights + o[SMZ_VR_WIDE_A], *wb = weights + o[SMZ_VR_WIDE_B], *bn = weights + o[SMZ_VR_WIDE_BN];
        residual_map<3, 25, 2>(l, wa, wb, bn, true, fc);
        residual_map<3, 25, 2>(l, wa, wb, bn, false, fc);
        SMZ_RSTAMP(3);
        pool_map<3, 25, 13>(l);
        SMZ_RSTAMP(4);
        residual_map<3, 13, 1>(l, wa, wb, bn, true, fc);
        residual_map<3, 13, 1>(l, wa, wb, bn, false, fc);
        residual_map<3, 13, 1>(l, wa, wb, bn, false, fc);
        SMZ_RSTAMP(5);
        pool_map<3, 13, 7>(l);
        SMZ_RSTAMP(6);
    }
    residual_map<3, 7, 1>(l, weights + o[SMZ_VR_LAST_A], weights + o[SMZ_VR_LAST_B], weights + o[SMZ_VR_LAST_BN], true, fc);
    SMZ_RSTAMP(7);
    // wave 0: per-pixel scaling, hidden state out, root policy (the root value is discarded, mcts:319-321)
    if (threadIdx.x < kWave) {
        const int lane = threadIdx.x;
        const bool active = lane < kPix;
        const int p = active ? lane : kPix - 1, pp = (p / kN + 1) * kPad + (p % kN + 1);
        float t[kC] = {l.t[p], l.t[kPix + p], l.t[2 * kPix + p]};
        scale_channels(t);
        if (active) {
            float *ho = hidden_out + (size_t)row * kFlat;
            ho[p] = t[0]; ho[kPix + p] = t[1]; ho[2 * kPix + p] = t[2];
        }
        zero_wave_lds(l.head, lane);
        predict(weights, weights, d, SMZ_V_PRE, l.head, lane, p, pp, active, t, policy_out + (size_t)row * d.A, false);
        SMZ_RSTAMP(8);
    }
}

int fill_layout(smz_vision_desc *d) {
    const int OP = kWave, K4h = up4(d->H);
    int off = 0;
    auto take = [&](int idx, int floats) { d->off[idx] = off; off += up4(floats); };
    auto take_tower = [&](int base) {
        take(base + 0, kFlat4 * OP); take(base + 1, OP);
        take(base + 2, K4h * OP);    take(base + 3, OP);
        take(base + 4, K4h * OP);    take(base + 5, OP);
    };
    for (int i = 0; i < SMZ_V_OFFSETS; i++) d->off[i] = 0;
    // (1) the small pieces of the four recurrent nets, contiguous from offset 0: the recurrent kernel stages
    //     [0, small_floats) into LDS once per workgroup
    for (int n = 0; n < 2; n++) {            // dynamics, afterstate dynamics
        const int b = SMZ_V_TRANS_BASE + n * SMZ_V_TRANS_STRIDE;
        take(b + SMZ_VT_CONV_IN, 3 * 4 * 9); take(b + SMZ_VT_BN_IN, 6);
        take(b + SMZ_VT_RES_A, 81); take(b + SMZ_VT_RES_B, 81); take(b + SMZ_VT_RES_BN, 6);
        take(b + SMZ_VT_MIX_W, 12); take(b + SMZ_VT_MIX_B, 3);
    }
    for (int n = 0; n < 2; n++) {            // prediction, afterstate prediction
        const int b = SMZ_V_PRED_BASE + n * SMZ_V_PRED_STRIDE;
        take(b + SMZ_VP_RES_A, 81); take(b + SMZ_VP_RES_B, 81); take(b + SMZ_VP_RES_BN, 6);
        take(b + SMZ_VP_VMIX_W, 9); take(b + SMZ_VP_VMIX_B, 3);
        take(b + SMZ_VP_PMIX_W, 9); take(b + SMZ_VP_PMIX_B, 3);
    }
    d->small_floats = off;
    // (2) the towers (streamed from L2) and the representation network
    for (int n = 0; n < 2; n++) take_tower(SMZ_V_TRANS_BASE + n * SMZ_V_TRANS_STRIDE + SMZ_VT_TOWER);
    for (int n = 0; n < 2; n++) {
        take_tower(SMZ_V_PRED_BASE + n * SMZ_V_PRED_STRIDE + SMZ_VP_VTOWER);
        take_tower(SMZ_V_PRED_BASE + n * SMZ_V_PRED_STRIDE + SMZ_VP_PTOWER);
    }
    const int r = SMZ_V_REP_BASE;
    take(r + SMZ_VR_STEM, 27);
    take(r + SMZ_VR_NARROW_A, 9); take(r + SMZ_VR_NARROW_B, 9); take(r + SMZ_VR_NARROW_BN, 2);
    take(r + SMZ_VR_WIDEN, 27);
    take(r + SMZ_VR_WIDE_A, 81); take(r + SMZ_VR_WIDE_B, 81); take(r + SMZ_VR_WIDE_BN, 6);
    take(r + SMZ_VR_LAST_A, 81); take(r + SMZ_VR_LAST_B, 81); take(r + SMZ_VR_LAST_BN, 6);
    d->OP = OP;
    d->total_floats = off;
    return off > 0 && d->small_floats <= kSmallMax ? SMZ_OK : SMZ_ERR_INVALID;
}

int vision_check(const smz_vision_desc *d, const void *w) {
    if (!d || !w) return SMZ_ERR_INVALID;
    smz_vision_desc t = *d;
    if (smz_vision_layout(&t) != SMZ_OK || t.total_floats != d->total_floats || t.OP != d->OP || t.small_floats != d->small_floats) return SMZ_ERR_INVALID;
    for (int i = 0; i < SMZ_V_OFFSETS; i++)
        if (t.off[i] != d->off[i]) return SMZ_ERR_INVALID;
    return SMZ_OK;
}

}  // namespace

extern thread_local char smz_g_err[512];     // the library's last-error text (smz_kernels.hip)
namespace {
int verr(int code, const char *text) {
    snprintf(smz_g_err, sizeof(smz_g_err), "%s", text);
    return code;
}
}  // namespace

extern "C" {

int smz_vision_layout(smz_vision_desc *d) {
    // one output neuron per lane in the towers: widths up to 64; the reference's default is H = 64
    if (!d || d->A < 1 || d->A > kWave || d->S < 1 || d->S > kWave || d->H < 1 || d->H > kWave || d->L < 0) return SMZ_ERR_INVALID;
    return fill_layout(d);
}

int smz_vision_initial_record(const smz_vision_desc *d, const float *weights_dev, const float *frames_dev, float *frames_copy_dev,
                              float *hidden_out_dev, float *policy_out_dev, int B, smz_stream stream) {
    if (vision_check(d, weights_dev) != SMZ_OK || !frames_dev || !hidden_out_dev || !policy_out_dev || B < 1)
        return verr(SMZ_ERR_INVALID, "smz_vision_initial: bad argument (descriptor / weights / null pointer / B < 1)");
    // (8-byte loads of pixel pairs; 16-byte pieces of the frame copy; a frame is 7203 of them)
    if (((uintptr_t)frames_dev & 7) || (frames_copy_dev && (((uintptr_t)frames_dev | (uintptr_t)frames_copy_dev) & 15)))
        return verr(SMZ_ERR_INVALID, "smz_vision_initial: frames must be 8-byte aligned (16-byte aligned, both pointers, with a record copy)");
    hipLaunchKernelGGL(k_vision_initial, dim3(B), dim3(kRepThreads), 0, (hipStream_t)stream, *d, weights_dev, frames_dev,
                       frames_copy_dev, hidden_out_dev, policy_out_dev);
    return hipGetLastError() == hipSuccess ? SMZ_OK : SMZ_ERR_HIP;
}
int smz_vision_initial(const smz_vision_desc *d, const float *weights_dev, const float *frames_dev, float *hidden_out_dev,
                       float *policy_out_dev, int B, smz_stream stream) {
    return smz_vision_initial_record(d, weights_dev, frames_dev, nullptr, hidden_out_dev, policy_out_dev, B, stream);
}

int smz_vision_recurrent(const smz_vision_desc *d, const float *weights_dev, const float *parent_hidden_dev, int ld,
                         const int32_t *last_action_dev, const uint8_t *branch_dev, float *hidden_out_dev,
                         float *reward_out_dev, float *policy_out_dev, float *value_out_dev, int B, smz_stream stream) {
    if (vision_check(d, weights_dev) != SMZ_OK || !parent_hidden_dev || ld < kFlat || !last_action_dev || !branch_dev ||
        !hidden_out_dev || !reward_out_dev || !policy_out_dev || !value_out_dev || B < 1)
        return SMZ_ERR_INVALID;
    hipLaunchKernelGGL(k_vision_recurrent, dim3((B + kRecWaves - 1) / kRecWaves), dim3(kRecWaves * kWave), 0,
                       (hipStream_t)stream, *d, weights_dev, parent_hidden_dev, ld, last_action_dev, branch_dev,
                       hidden_out_dev, reward_out_dev, policy_out_dev, value_out_dev, B);
    return hipGetLastError() == hipSuccess ? SMZ_OK : SMZ_ERR_HIP;
}

}  // extern "C"
