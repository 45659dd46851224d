// smz_kernels.hip -- gfx950 kernels and the C ABI of libsmz.so (see include/smz.h for the contract).
//
// Execution shape: one search tree per lane, one 64-lane wavefront per workgroup, ceil(B/64) workgroups.  The
// per-tree control flow (descent depth, rejection loops of the numpy samplers) is data dependent, so trees are
// kept on separate lanes and diverge freely; the only wave-cooperative parts are the row moves of hidden states
// (parent -> network input, network output -> leaf), where the lanes of a wave walk its 64 trees together so that
// each row is read and written as one contiguous segment instead of 64 strided dwords.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no FMA contraction: parity with the reference's
// separately rounded multiplies and adds).
// SMZ_PART: 0 / undefined = the whole library part in one translation unit; the Makefile compiles this file five times --
// 1 = everything but the single-launch search kernel and the fused expand+backup+select entry point, 2 = the search
// kernel for the action buckets 2 and 4 with smz_search_mlp(_act), 4 = the search kernel for the buckets 8-32,
// 3 = only smz_expand_backup_select, 6 = the masked / Philox instantiations of the search kernel that keep the trees in LDS,
// 5 = nothing but the shared helpers and the handle (included by smz_vision_search.hip)
// -- the template instantiations behind those are most of the compile time, and the parts build in parallel.
#ifndef SMZ_SEARCH_THREADS
#define SMZ_SEARCH_THREADS 512   // threads per workgroup the single-launch search is register-allocated for (8 waves: 2 per SIMD, 256 VGPRs)
#endif
#ifndef SMZ_EB_WAVES
#define SMZ_EB_WAVES 4   // waves per SIMD the step-wise tree kernels are register-allocated for
#endif
#ifndef SMZ_PAIR_K4
#define SMZ_PAIR_K4 1              // KS = 4: the paired descent on every decision level (0: one lane scores all four children)
#endif
#ifndef SMZ_KS4
#define SMZ_KS4 1                  // the compile-time K = 4 instantiation for the A = 4 bucket (0: run-time K as in round 5)
#endif
#ifndef SMZ_PART
#define SMZ_PART 0
#endif
// Wave priorities inside the search kernel (s_setprio): the network evaluation is throughput work, the tree phases are
// dependent chains that mostly wait; letting the evaluating wave of a SIMD issue first measured +4 % (392 -> 408 M
// simulations/s, same box, back to back; every non-uniform assignment tried beat uniform priorities).
#ifndef SMZ_PRIO_TREE
#define SMZ_PRIO_TREE 0
#define SMZ_PRIO_HEADS 3
#endif
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <algorithm>
#include <vector>

#include "../../include/smz.h"
#include "smz_device.hpp"
#include "smz_mlp_device.hpp"

using namespace smz;

// (file scope, not the anonymous namespace: it crosses the translation units this file is compiled into)
struct EnvStep {
    double *state;                 // [B][4] f64 in/out; nullptr: no env step in this launch
    float *obs_out, *reward_out;
    uint8_t *flag_out;
    int32_t *step_count, *episode; // nullptr: no bookkeeping (plain step: flag = terminated)
    uint8_t *active;
    int limit, on_end;
    uint64_t reset_seed;
    long long first_env;
    double *traj;                  // [T][B][13] f64 or nullptr
    int t;
};

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct RowGeom {  // how the 64 lanes of a wave are split over rows of `width` floats
    int width, lpr, rows_per_iter;
};

__host__ __device__ inline RowGeom row_geom(int width) {
    int lpr = 1;
    while (lpr < width && lpr < kWave) lpr <<= 1;
    RowGeom g;
    g.width = width;
    g.lpr = lpr;
    g.rows_per_iter = kWave / lpr;
    return g;
}

// numpy `seed(int)`: init_genrand (numpy/random/src/mt19937/mt19937.c mt19937_seed); pos = 624.
#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(kWave) k_seed(Params P, const uint64_t *seeds) {
    const int tree = blockIdx.x * kWave + threadIdx.x;
    if (tree >= P.B) return;
    if (P.philox) {          // counter-based stream: the seed is the key, nothing else to initialise
        const_cast<uint32_t *>(P.rng_key)[2 * tree] = (uint32_t)(seeds[tree] & 0xffffffffull);
        const_cast<uint32_t *>(P.rng_key)[2 * tree + 1] = (uint32_t)(seeds[tree] >> 32);
        P.rng_block[tree] = 0u;
        P.rng_pos[tree] = 0;
        return;
    }
    uint32_t s = (uint32_t)(seeds[tree] & 0xffffffffull);
    uint32_t *mt = P.mt + (size_t)tree * kMtN;
    for (int i = 0; i < kMtN; i++) {
        mt[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    P.rng_pos[tree] = 0;  // idx 0, nothing pre-twisted: the first draw twists word 0 (== numpy pos 624)
}
#endif

// Value of lane 0 or lane 1 (src = 0 / 1, may differ per lane): two v_readlane and a select instead of a ds_bpermute round
// trip (~120 cycles in the middle of the tree phases' dependent chains).
__device__ inline int pick_lane01(int v, int src) {
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 1);
    return src ? b : a;
}
__device__ inline float pick_lane01(float v, int src) { return __int_as_float(pick_lane01(__float_as_int(v), src)); }

// Random-word staging: for each of the wave's 64 trees, all 64 lanes cooperate on that tree's NEXT 64 words --
// lane j owns word (idx + j) mod 624; words not yet twisted (j >= ready) are twisted in place (every lane reads
// its three source words before any lane stores, which is exactly the sequential in-place order because the
// sources of word p are p, p+1 and p+397 and at most 64 consecutive words change) -- and the tempered words go to
// this wave's LDS tile [tree lane][word].  Global traffic is coalesced 256-byte segments.  Returns the lane's
// packed (ready << 16 | idx) after staging.
// Philox mode: the next words of the wave's trees are computed, not loaded -- lanes 0..15 of tree slot t evaluate the 16
// counter blocks that cover words [idx & ~3, (idx & ~3) + 64) of its stream and write those at or after idx to the tile
// row (Rng::load knows the row then holds 64 - (idx & 3) words).  `block`: each lane's own tree's block counter.
__device__ inline void philox_stage(const Params &P, int tree, bool valid, uint32_t *lds_tile, int packed, uint32_t block, int ntrees) {
    if (!lds_tile) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int tree0 = tree - lane;
    for (int t0 = 0; t0 < ntrees; t0 += 4) {                      // four trees per pass: 16 lanes each
        const int t = t0 + (lane >> 4), j = lane & 15;
        const int src = t < ntrees ? t : 0;
        const int pk = __shfl(packed, src);
        const uint32_t kb = (uint32_t)__shfl((int)block, src);
        const bool vt = t < ntrees && __shfl((int)valid, src) != 0;
        if (vt) {
            const int idx = pk & 0xffff;
            int q = (idx >> 2) + j;
            uint32_t b = kb;
            if (q >= kMtN / 4) { q -= kMtN / 4; ++b; }
            const uint64_t n = (uint64_t)b * (kMtN / 4) + (uint64_t)q;
            const PhiloxOut o = philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), P.rng_key[2 * (tree0 + t)], P.rng_key[2 * (tree0 + t) + 1]);
            const int pos = 4 * j - (idx & 3);
            uint32_t *row = lds_tile + t * kRngStride;
            if (pos >= 0) row[pos] = o.x;
            if (pos + 1 >= 0) row[pos + 1] = o.y;
            if (pos + 2 >= 0) row[pos + 2] = o.z;
            row[pos + 3] = o.w;
        }
    }
}

template <int U = 8, bool PHC = true>   // U trees in flight: their loads are all issued before the first dependent store
__device__ inline int wave_stage_rng_from(const Params &P, int tree, bool valid, uint32_t *lds_tile, int packed, uint32_t block = 0u) {
    const int lane = threadIdx.x & (kWave - 1);
    const int tree0 = tree - lane;
    if constexpr (PHC) {
        if (P.philox) {
            philox_stage(P, tree, valid, lds_tile, packed, block, P.tpw);
            return ((kRngStage) << 16) | (packed & 0xffff);
        }
    }
    for (int t0 = 0; t0 < P.tpw; t0 += U) {
        uint32_t w[U], b[U], c[U];
        int pos[U];
        bool tw[U], vt[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int pk = __shfl(packed, t0 + u);
            vt[u] = __shfl((int)valid, t0 + u) != 0;          // wave-uniform
            const int idx = pk & 0xffff, ready = pk >> 16;
            const uint32_t *mt = P.mt + (size_t)(tree0 + t0 + u) * kMtN;
            int p = idx + lane;
            if (p >= kMtN) p -= kMtN;
            pos[u] = p;
            tw[u] = vt[u] && lane >= ready;
            w[u] = b[u] = c[u] = 0u;
            if (vt[u]) w[u] = mt[p];
            if (tw[u]) {
                const int p1 = (p + 1 == kMtN) ? 0 : p + 1;
                int pm = p + kMtM;
                if (pm >= kMtN) pm -= kMtN;
                b[u] = mt[p1];
                c[u] = mt[pm];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (tw[u]) {
                w[u] = mt_twist(w[u], b[u], c[u]);
                (P.mt + (size_t)(tree0 + t0 + u) * kMtN)[pos[u]] = w[u];
            }
            if (vt[u] && lds_tile) lds_tile[(t0 + u) * kRngStride + lane] = mt_temper(w[u]);
        }
    }
    const int idx = packed & 0xffff, ready = packed >> 16;
    return ((ready > kRngStage ? ready : kRngStage) << 16) | idx;
}

// Half-width staging for wavefronts of 16+ trees (mid-size and large batches, MT19937): 32 words per tree and launch instead of 64 -- a
// simulation draws ~10 (the rare longer one falls back to the words in global memory, same values) -- two trees per pass, lanes
// 0..31 | 32..63.  At a million trees the 64-word window alone was 4-5 of the ~10 cache lines a tree reads per launch.
#ifndef SMZ_RNG_NARROW
#define SMZ_RNG_NARROW 32
#endif
constexpr int kRngStageNarrow = SMZ_RNG_NARROW;      // 16 or 32
__device__ inline int wave_stage_rng_narrow(const Params &P, int tree, bool valid, uint32_t *lds_tile) {
    constexpr int U = 8;
    constexpr int W = kRngStageNarrow, TP = kWave / W;      // trees per pass
    const int lane = threadIdx.x & (kWave - 1), half = lane / W, j = lane % W;
    const int tree0 = tree - lane;
    const int packed = valid ? P.rng_pos[tree] : 0;
    for (int t0 = 0; t0 < P.tpw; t0 += TP * U) {
        uint32_t w[U], b[U], c[U];
        int pos[U];
        bool tw[U], vt[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + TP * u + half;
            const int pk = __shfl(packed, t);
            vt[u] = __shfl((int)valid, t) != 0;
            const int idx = pk & 0xffff, ready = pk >> 16;
            const uint32_t *mt = P.mt + (size_t)(tree0 + t) * kMtN;
            int p = idx + j;
            if (p >= kMtN) p -= kMtN;
            pos[u] = p;
            tw[u] = vt[u] && j >= ready;
            w[u] = b[u] = c[u] = 0u;
            if (vt[u]) w[u] = mt[p];
            if (tw[u]) {
                const int p1 = (p + 1 == kMtN) ? 0 : p + 1;
                int pm = p + kMtM;
                if (pm >= kMtN) pm -= kMtN;
                b[u] = mt[p1];
                c[u] = mt[pm];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + TP * u + half;
            if (tw[u]) {
                w[u] = mt_twist(w[u], b[u], c[u]);
                (P.mt + (size_t)(tree0 + t) * kMtN)[pos[u]] = w[u];
            }
            if (vt[u]) lds_tile[t * kRngStride + j] = mt_temper(w[u]);
        }
    }
    const int idx = packed & 0xffff, ready = packed >> 16;
    return ((ready > kRngStageNarrow ? ready : kRngStageNarrow) << 16) | idx;
}

template <bool PHC = true>
__device__ inline int wave_stage_rng(const Params &P, int tree, bool valid, uint32_t *lds_tile) {
    return wave_stage_rng_from<8, PHC>(P, tree, valid, lds_tile, valid ? P.rng_pos[tree] : 0,
                                       (PHC && P.philox && valid) ? P.rng_block[tree] : 0u);
}

// SMZ_STAGE_STORES_LAST (round 5): stage_finish stores the twisted words after ALL trees' words went to LDS (=0: tree by tree).
#ifndef SMZ_STAGE_STORES_LAST
#define SMZ_STAGE_STORES_LAST 1
#endif
// The same staging split in two for waves that own at most U trees: stage_issue requests the source words (the loads
// stay in flight while the caller does unrelated work -- the network evaluation in k_search_mlp), stage_finish twists,
// stores and fills the LDS tile.  Nothing may draw random words of these trees in between.
template <int U>
struct StagePre {
    uint32_t w[U], b[U], c[U];
};
template <int U, bool PHC = true>
__device__ inline void stage_issue(const Params &P, int tree, bool valid, int packed, StagePre<U> &pre) {
    const int lane = threadIdx.x & (kWave - 1);
    const int tree0 = tree - lane;
    if (PHC && P.philox) return;                            // nothing to load: stage_finish computes the words
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int pk = __builtin_amdgcn_readlane(packed, u);              // (u is a constant after unrolling)
        const bool vt = u < P.tpw && __builtin_amdgcn_readlane((int)valid, u) != 0;          // wave-uniform
        const int idx = pk & 0xffff, ready = pk >> 16;
        const uint32_t *mt = P.mt + (size_t)(tree0 + u) * kMtN;
        int p = idx + lane;
        if (p >= kMtN) p -= kMtN;
        pre.w[u] = pre.b[u] = pre.c[u] = 0u;
        if (vt) pre.w[u] = mt[p];
        if (vt && lane >= ready) {
            const int p1 = (p + 1 == kMtN) ? 0 : p + 1;
            int pm = p + kMtM;
            if (pm >= kMtN) pm -= kMtN;
            pre.b[u] = mt[p1];
            pre.c[u] = mt[pm];
        }
    }
}
template <int U, bool PHC = true>
__device__ inline int stage_finish(const Params &P, int tree, bool valid, uint32_t *lds_tile, int packed, const StagePre<U> &pre,
                                   uint32_t block = 0u) {
    const int lane = threadIdx.x & (kWave - 1);
    const int tree0 = tree - lane;
    if constexpr (PHC) {
        if (P.philox) {
            philox_stage(P, tree, valid, lds_tile, packed, block, P.tpw < U ? P.tpw : U);
            return ((kRngStage) << 16) | (packed & 0xffff);
        }
    }
#if SMZ_STAGE_STORES_LAST
    // Every source word is consumed (twisted, tempered, handed to LDS) before the first twisted word is stored: a store between
    // one tree's words and the next's made the compiler wait for THAT STORE's completion (vmcnt counts stores on gfx9, and with
    // the twist under a branch its count is conservative: s_waitcnt vmcnt(0) after each global_store -- two serial store round
    // trips per round, profiles/r05_s_stage_waits.txt).
    uint32_t tw_w[U];
    int tw_p[U];
    bool tw_on[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int pk = __builtin_amdgcn_readlane(packed, u);
        const bool vt = u < P.tpw && __builtin_amdgcn_readlane((int)valid, u) != 0;
        const int idx = pk & 0xffff, ready = pk >> 16;
        int p = idx + lane;
        if (p >= kMtN) p -= kMtN;
        tw_on[u] = vt && lane >= ready;
        tw_p[u] = p;
        const uint32_t t = mt_twist(pre.w[u], pre.b[u], pre.c[u]);      // (lanes that do not twist hold zeros in b, c: value unused)
        tw_w[u] = tw_on[u] ? t : pre.w[u];
        if (vt) lds_tile[u * kRngStride + lane] = mt_temper(tw_w[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++)
        if (tw_on[u]) (P.mt + (size_t)(tree0 + u) * kMtN)[tw_p[u]] = tw_w[u];
#else
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int pk = __builtin_amdgcn_readlane(packed, u);
        const bool vt = u < P.tpw && __builtin_amdgcn_readlane((int)valid, u) != 0;
        const int idx = pk & 0xffff, ready = pk >> 16;
        int p = idx + lane;
        if (p >= kMtN) p -= kMtN;
        uint32_t w = pre.w[u];
        if (vt && lane >= ready) {
            w = mt_twist(w, pre.b[u], pre.c[u]);
            (P.mt + (size_t)(tree0 + u) * kMtN)[p] = w;
        }
        if (vt) lds_tile[u * kRngStride + lane] = mt_temper(w);
    }
#endif
    const int idx = packed & 0xffff, ready = packed >> 16;
    return ((ready > kRngStage ? ready : kRngStage) << 16) | idx;
}

// Row moves.  `lpr` consecutive lanes move one row of `width` floats; the rows of the wave's 64 trees are handed
// around with ds_bpermute.  Loads of kRowBatch rows are issued before the first store so that the (independent)
// row reads overlap instead of each waiting behind the previous row's may-alias store.
constexpr int kRowBatch = 8;

// Moves one row per tree of this wave.  src_row/dst_row are per-lane row pointers (of the lane's own tree).
// Every lane executes every shuffle (a lane whose column index falls outside the row still serves as a source).
__device__ inline void wave_copy_rows(const float *src_row, float *dst_row, bool valid, int width, int tpw) {
    const RowGeom g = row_geom(width);
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / g.lpr, li = lane % g.lpr;
    const unsigned long long s64 = (unsigned long long)src_row, d64 = (unsigned long long)dst_row;
    const int iters = (tpw + g.rows_per_iter - 1) / g.rows_per_iter;
    for (int it0 = 0; it0 < iters; it0 += kRowBatch) {
        unsigned long long sj[kRowBatch], dj[kRowBatch];
        bool ok[kRowBatch];
#pragma unroll
        for (int u = 0; u < kRowBatch; u++) {
            const int it = it0 + u;
            const int j = ((it < iters ? it : 0) * g.rows_per_iter + sub) & (kWave - 1);
            sj[u] = __shfl(s64, j);
            dj[u] = __shfl(d64, j);
            ok[u] = (it < iters) && (__shfl((int)valid, j) != 0);
        }
        for (int i = li; i < width; i += g.lpr) {                   // one pass per lpr-wide column slab
            float v[kRowBatch];
#pragma unroll
            for (int u = 0; u < kRowBatch; u++) v[u] = ok[u] ? ((const float *)sj[u])[i] : 0.f;
#pragma unroll
            for (int u = 0; u < kRowBatch; u++)
                if (ok[u]) ((float *)dj[u])[i] = v[u];
        }
    }
}

// Network-input gather of smz_select: parent hidden rows (+ one-hot of the last action) for the wave's 64 trees.
__device__ inline void wave_gather_inputs(const Params &P, int tree, bool valid, int parent, int act,
                                          float *parent_hidden, float *mlp_input) {
    const int S = P.S, W = S + P.A;
    const RowGeom g = row_geom(mlp_input ? W : S);
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / g.lpr, li = lane % g.lpr;
    const int tree0 = tree - lane;
    const int iters = (P.tpw + g.rows_per_iter - 1) / g.rows_per_iter;
    for (int it0 = 0; it0 < iters; it0 += kRowBatch) {
        int tj[kRowBatch], pj[kRowBatch], aj[kRowBatch];
        bool ok[kRowBatch];
#pragma unroll
        for (int u = 0; u < kRowBatch; u++) {
            const int it = it0 + u;
            const int j = ((it < iters ? it : 0) * g.rows_per_iter + sub) & (kWave - 1);
            pj[u] = __shfl(parent, j);
            aj[u] = __shfl(act, j);
            ok[u] = (it < iters) && (__shfl((int)valid, j) != 0);
            tj[u] = tree0 + j;
        }
        for (int i = li; i < g.width; i += g.lpr) {
            float v[kRowBatch];
#pragma unroll
            for (int u = 0; u < kRowBatch; u++) {
                v[u] = 0.f;
                if (ok[u]) v[u] = (i < S) ? P.hidden[((size_t)tj[u] * P.N + pj[u]) * P.hs + i] : ((i - S) == aj[u] ? 1.0f : 0.0f);
            }
#pragma unroll
            for (int u = 0; u < kRowBatch; u++) {
                if (ok[u]) {
                    if (mlp_input) mlp_input[(size_t)tj[u] * W + i] = v[u];
                    if (parent_hidden && i < S) parent_hidden[(size_t)tj[u] * S + i] = v[u];
                }
            }
        }
    }
}

__device__ inline void wave_add_stats(unsigned long long *stats, unsigned a, unsigned b, unsigned c, unsigned d) {
    if (!stats) return;
    for (int off = kWave / 2; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
        c += __shfl_down(c, off);
        d += __shfl_down(d, off);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicAdd(&stats[0], (unsigned long long)a);
        atomicAdd(&stats[1], (unsigned long long)b);
        atomicAdd(&stats[2], (unsigned long long)c);
        atomicAdd(&stats[3], (unsigned long long)d);
    }
}

// sqrt(n) * pb_c(n) table staged in LDS (dynamic shared memory) when it fits, so that the pUCT loop does not pay a
// dependent global load per level.
constexpr int kPbcLdsMax = 4096;
extern __shared__ double smz_dyn_lds[];
__device__ inline const double *stage_pbc(const Params &P) {
    const int n = P.sims + 2;
    if (n > kPbcLdsMax) return P.pbc_sqrt;
    for (int i = threadIdx.x; i < n; i += kWave) smz_dyn_lds[i] = P.pbc_sqrt[i];
    __syncthreads();
    return smz_dyn_lds;
}
// dynamic LDS of the step-wise tree kernels: [pbc table (when it fits)] [rng tile (when P.lds_stage)]
__device__ inline uint32_t *rng_tile_ptr(const Params &P) {
    if (!P.lds_stage) return nullptr;
    const int n = (P.sims + 2 <= kPbcLdsMax) ? P.sims + 2 : 0;
    return reinterpret_cast<uint32_t *>(smz_dyn_lds + n);
}

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
template <int MAXA>
__global__ void __launch_bounds__(kWave) k_root_init(Params P, const float *hidden, const float *policy,
                                                     const double *noise_override, int train) {
    uint32_t *rng_tile = rng_tile_ptr(P);
    const int n_staged = rng_tile ? kRngStage : 0;
    const int tree = blockIdx.x * P.tpw + threadIdx.x;
    const bool valid = (int)threadIdx.x < P.tpw && tree < P.B && tree_active(P, tree);
    const int packed = wave_stage_rng(P, tree, valid, rng_tile);
    if (valid) {
        Rng rng;
        rng.bind(P, tree, valid);
        rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + threadIdx.x * kRngStride, n_staged);
        root_init_tree<MAXA>(P, tree, rng, policy + (size_t)tree * P.A,
                             noise_override ? noise_override + (size_t)tree * P.A : nullptr, train != 0);
        P.rng_pos[tree] = rng.pack();
        rng.save(P, tree);
    }
    if (P.S > 0 && hidden) {
        const int t = valid ? tree : 0;
        wave_copy_rows(hidden + (size_t)t * P.S, P.hidden + (size_t)t * P.N * P.hs, valid, P.S, P.tpw);
    }
}
#endif

template <int MAXA, int KS, class RNG>
__device__ inline void select_phase(const Params &P, int tree, bool valid, RNG &rng, TreeHdr &h, const double *pbc_lds,
                                    float *parent_hidden, int32_t *last_action, uint8_t *branch, float *mlp_input) {
    Leaf L = {0, 0, 0, 0};
    unsigned n_dec = 0, n_chance = 0, n_children = 0;
    if (valid) {
        int len = 0;
        L = select_tree<MAXA, KS, true, false, false>(P, tree, rng, h, pbc_lds, len, n_dec, n_chance, n_children, path_col(P, tree));
        h.path_len = len;
        if (last_action) last_action[tree] = L.action;
        if (branch) branch[tree] = (uint8_t)L.branch;
    }
    if (P.ids_out && (int)threadIdx.x < P.tpw && tree < P.B) {
        P.ids_out[2 * (size_t)tree] = valid ? L.leaf_id : -1;
        P.ids_out[2 * (size_t)tree + 1] = valid ? L.parent_id : -1;
    }
    if (P.S > 0 && (parent_hidden || mlp_input))
        wave_gather_inputs(P, tree, valid, L.parent_id, L.action, parent_hidden, mlp_input);
    wave_add_stats(P.stats, n_dec, n_chance, valid ? 1u : 0u, n_children);
}

// Block geometry as a function of A and K (the formulas of smz_create): with compile-time A / K these fold to constants.
__device__ inline void fix_layout(Params &P, bool a_const, bool k_const) {
    if (a_const) { P.rp_off = (5 * P.A + 1) & ~1; P.rb_words = ((P.rp_off + 2 * P.A) + 15) & ~15; }
    if (k_const) P.eb_words = ((6 * P.K) + 15) & ~15;
}

// AEX (instantiated for the MAXA 2 and 4 buckets): the action count equals the bucket, so A (and K when KS > 0) are
// compile-time constants in everything inlined below (see k_search_mlp).
#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
template <int MAXA, int KS, bool AEX>
__global__ void __launch_bounds__(kWave, 4) k_select(Params Pin, float *parent_hidden, int32_t *last_action,
                                                  uint8_t *branch, float *mlp_input) {
    Params P = Pin;
    P.tree0 = 0;
    if (AEX) P.A = MAXA;
    if (KS > 0) P.K = KS;
    fix_layout(P, AEX, KS > 0);
    uint32_t *rng_tile = rng_tile_ptr(P);
    const bool narrow = rng_tile && P.tpw >= 16 && !P.philox;          // (wave-uniform) 16+ trees per wavefront: wave_stage_rng_narrow
    const int n_staged = rng_tile ? (narrow ? kRngStageNarrow : kRngStage) : 0;
    const int tree = blockIdx.x * P.tpw + threadIdx.x;
    const bool valid = (int)threadIdx.x < P.tpw && tree < P.B && tree_active(P, tree);
    const double *pbc_lds = stage_pbc(P);
    const int packed = narrow ? wave_stage_rng_narrow(P, tree, valid, rng_tile) : wave_stage_rng<!AEX>(P, tree, valid, rng_tile);
    RngT<!AEX> rng;          // (the specialised instantiations serve MT19937 handles only: see smz_select)
    rng.bind(P, tree, valid);
    TreeHdr h = {0, 0, 0.f, 0.f, 0, 0.f, 0, 0};
    if (valid) {
        rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + threadIdx.x * kRngStride, n_staged);
        h = P.hdr[tree];
    }
    select_phase<MAXA, KS>(P, tree, valid, rng, h, pbc_lds, parent_hidden, last_action, branch, mlp_input);
    if (valid) {
        P.rng_pos[tree] = rng.pack();
        rng.save(P, tree);
        P.hdr[tree] = h;
    }
}
#endif

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
// PHX (round 5, with AEX): the specialised kernel for Philox handles -- counter streams, no MT19937 state to load, twist or store
template <int MAXA, int KS, bool FUSE_SELECT, bool AEX, bool PHX = false>
__global__ void __launch_bounds__(kWave, MAXA > 16 ? 1 : SMZ_EB_WAVES) k_expand_backup(Params Pin, const float *hidden, const float *reward,
                                                         const float *policy, const float *value,
                                                         float *parent_hidden, int32_t *last_action, uint8_t *branch,
                                                         float *mlp_input) {
    Params P = Pin;
    P.tree0 = 0;
    if (AEX) P.A = MAXA;
    if (AEX) P.philox = PHX ? 1 : 0;          // (a constant in everything inlined below)
    if (KS > 0) P.K = KS;
    fix_layout(P, AEX, KS > 0);
    uint32_t *rng_tile = rng_tile_ptr(P);
    const bool narrow = rng_tile && P.tpw >= 16 && !P.philox;          // (wave-uniform) 16+ trees per wavefront: wave_stage_rng_narrow
    const int n_staged = rng_tile ? (narrow ? kRngStageNarrow : kRngStage) : 0;
    const int tree = blockIdx.x * P.tpw + threadIdx.x;
    const bool valid = (int)threadIdx.x < P.tpw && tree < P.B && tree_active(P, tree);
    const double *pbc_lds = stage_pbc(P);
    const int packed = narrow ? wave_stage_rng_narrow(P, tree, valid, rng_tile) : wave_stage_rng<!AEX || PHX>(P, tree, valid, rng_tile);
    RngT<!AEX || PHX> rng;
    rng.bind(P, tree, valid);
    TreeHdr h = {0, 0, 0.f, 0.f, 0, 0.f, 0, 0};
    int leaf = 0;
    if (valid) {
        rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + threadIdx.x * kRngStride, n_staged);
        h = P.hdr[tree];
        leaf = expand_backup_tree<MAXA, KS>(P, tree, rng, h, policy + (size_t)tree * P.A, reward ? reward[tree] : 0.0f,
                                            value[tree], path_col(P, tree));
    }
    if (P.S > 0 && hidden) {
        const int t = valid ? tree : 0;
        wave_copy_rows(hidden + (size_t)t * P.S, P.hidden + ((size_t)t * P.N + leaf) * P.hs, valid, P.S, P.tpw);
    }
    if (FUSE_SELECT) {
        // the leaf rows just written by other lanes of this wave may be the next parent rows
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        select_phase<MAXA, KS>(P, tree, valid, rng, h, pbc_lds, parent_hidden, last_action, branch, mlp_input);
    }
    if (valid) {
        P.rng_pos[tree] = rng.pack();
        rng.save(P, tree);
        P.hdr[tree] = h;
    }
}
#endif

// Counter-based generator of the built-in environments (reset states of later episodes, stand-in observations):
// splitmix64 of (seed, env, episode / step, component) -- the same value whatever the shard or launch geometry.
__host__ __device__ inline uint64_t smz_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline double smz_unit(uint64_t seed, uint64_t env, uint64_t epoch, uint64_t comp) {   // [0, 1)
    const uint64_t z = smz_mix64(smz_mix64(smz_mix64(seed ^ (env * 0xD1342543DE82EF95ull)) + epoch) + comp);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// One env step of the built-in CartPole-v1 shaped env + its trajectory record (k_traj_pack's layout, A = 2), with the game
// bookkeeping of self_play.py:79 / game.py:270-271 -- the ONE body behind smz_cartpole_step, _step_pack, _step_ctl and
// the tail of the single-launch search (smz_search_mlp_act_cartpole).  The flag written to flag_out / the record's
// `terminated` slot is 0 running, 1 terminated (Game.done True), 2 stopped by limit_of_game_play (the game is over but
// Game.done stays False, game.py:270-271), 3 no step taken (env switched off).  on_end: 1 = a finished env is switched off
// (active[e] = 0: the searches skip it from now on), 2 = it starts its next episode at once (state ~ U(-0.05, 0.05)^4 from
// the counter-based generator, episode[e] + 1).
__device__ inline void cartpole_env_step(const EnvStep &E, int e, int B, const int32_t *action, const double *policy,
                                         const double *child_visits, const float *root_value) {
    constexpr int A = 2, F = 4 + 3 * A + 3;
    double *r = E.traj ? E.traj + ((size_t)E.t * B + e) * F : nullptr;
    if (E.active && !E.active[e]) {
        if (E.flag_out) E.flag_out[e] = 3;
        if (E.reward_out) E.reward_out[e] = 0.f;
        if (r) { for (int k = 0; k < F; k++) r[k] = 0.0; r[5] = 3.0; }
        return;
    }
    const double g = 9.8, mc = 1.0, mp = 0.1, tm = mc + mp, len = 0.5, pml = mp * len, fm = 10.0, tau = 0.02;
    double *st = E.state + (size_t)e * 4;
    const double x = st[0], xd = st[1], th = st[2], thd = st[3];
    const int act = action[e];
    const double force = act == 1 ? fm : -fm;
    const double ct = cos(th), sn = sin(th);
    const double temp = (force + pml * thd * thd * sn) / tm;
    const double tha = (g * sn - ct * temp) / (len * (4.0 / 3.0 - mp * ct * ct / tm));
    const double xa = temp - pml * tha * ct / tm;
    double nx = x + tau * xd, nxd = xd + tau * xa, nth = th + tau * thd, nthd = thd + tau * tha;
    const bool term = fabs(nx) > 2.4 || fabs(nth) > 12.0 * 2.0 * 3.14159265358979323846 / 360.0;
    const int count = E.step_count ? E.step_count[e] + 1 : 0;
    const int flag = (E.limit > 0 && count == E.limit) ? 2 : (term ? 1 : 0);
    if (r) {
        r[0] = (double)(float)nx; r[1] = (double)(float)nxd; r[2] = (double)(float)nth; r[3] = (double)(float)nthd;
        r[4] = 1.0;
        r[5] = (double)flag;
        r[6] = policy[(size_t)e * A]; r[7] = policy[(size_t)e * A + 1];
        r[8] = act == 0 ? 1.0 : 0.0; r[9] = act == 1 ? 1.0 : 0.0;
        r[10] = (double)root_value[e];
        r[11] = child_visits[(size_t)e * A]; r[12] = child_visits[(size_t)e * A + 1];
    }
    if (E.reward_out) E.reward_out[e] = 1.0f;
    if (E.flag_out) E.flag_out[e] = (uint8_t)flag;
    int next_count = count;
    if (flag != 0 && E.on_end == 2) {
        const int ep = E.episode[e] + 1;
        E.episode[e] = ep;
        nx = -0.05 + 0.1 * smz_unit(E.reset_seed, (uint64_t)(E.first_env + e), (uint64_t)ep, 0);
        nxd = -0.05 + 0.1 * smz_unit(E.reset_seed, (uint64_t)(E.first_env + e), (uint64_t)ep, 1);
        nth = -0.05 + 0.1 * smz_unit(E.reset_seed, (uint64_t)(E.first_env + e), (uint64_t)ep, 2);
        nthd = -0.05 + 0.1 * smz_unit(E.reset_seed, (uint64_t)(E.first_env + e), (uint64_t)ep, 3);
        next_count = 0;
    } else if (flag != 0 && E.on_end == 1 && E.active) {
        E.active[e] = 0;
    }
    if (E.step_count) E.step_count[e] = next_count;
    st[0] = nx; st[1] = nxd; st[2] = nth; st[3] = nthd;
    if (E.obs_out) {
        float *o = E.obs_out + (size_t)e * 4;
        o[0] = (float)nx; o[1] = (float)nxd; o[2] = (float)nth; o[3] = (float)nthd;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Whole search in ONE launch (mlp_model heads): Monte_carlo_tree_search.run (mcts:311-349) for every tree.
// A workgroup = 8 wavefronts sharing one LDS copy of the network weights; a wavefront owns `tpw` trees: the first
// tpw lanes run the per-tree search code (root, select, expand, backup), then all 64 lanes evaluate the networks for
// the wave's leaves one row at a time (smz_mlp_device.hpp).  Leaf/parent hand-off, policies, values and rewards stay
// in LDS; only the child blocks, hidden rows and MT words touch global memory.  No inter-wave communication at all.
// ---------------------------------------------------------------------------------------------------------------
extern __shared__ float4 smz_search_lds4[];

// SMZ_BPS_HBM (round 5): the block-parallel selection also in the specialised kernels whose trees stay in global memory (more
// simulations than LDS holds: BASELINE configs[4]'s per-rank shape) -- every block of a tree is then ONE load from L2 issued by
// its own lane, instead of one dependent L2 round trip per level of the descent
#ifndef SMZ_BPS_HBM
#define SMZ_BPS_HBM 1
#endif
#ifndef SMZ_SELECT_BLOCKS
#define SMZ_SELECT_BLOCKS 1       // block-parallel selection in the kernels that keep their trees in LDS (A/B builds: 0)
#endif
// -DSMZ_BPS_PROBE (variant builds, tools/bps_probe.sh): s_memtime stamps inside the production LDS-resident kernel, summed
// over waves into smz_read_stats' slots 8.. (expand + backup | select: prepare, evaluate, chase, records | networks | staging)
// SMZ_EARLY_ROWS (round 5): in the block-parallel selection the wave's two parent rows are requested from global memory as soon
// as the pointer chase has named them (-DSMZ_EARLY_ROWS=0 builds keep the loads where the network inputs are assembled)
#ifndef SMZ_EARLY_ROWS
#define SMZ_EARLY_ROWS 1
#endif
// SMZ_PAIR_A4 (round 5): the paired descent (two lanes per tree, one child each) for FOUR actions too -- only the root has four
// children, which the two lanes split two and two (pick_decision_pair<Kids<4>>); every level below is the two-action code.
#ifndef SMZ_PAIR_A4
#define SMZ_PAIR_A4 1
#endif
// SMZ_SELECT_TWO_PASSES (round 5): the block-parallel selection on trees in global memory requests the blocks of two passes
// (64 per tree) before it decides the first pass' picks (-DSMZ_SELECT_TWO_PASSES=0: pass by pass; =3: three passes in flight).
#ifndef SMZ_SELECT_TWO_PASSES
#define SMZ_SELECT_TWO_PASSES 1
#endif
#ifndef SMZ_LEAF_FIRST
#define SMZ_LEAF_FIRST 1
#endif
// SMZ_EARLY_STAGE (round 5): the next round's MT19937 source words are requested together with the parent rows -- the stream
// position after the descent follows from the path length alone -- instead of after the selection's last phase.  Before, the
// wait for the (long landed) parent rows at the network inputs was a vmcnt(0) that also waited for the source words requested
// a few instructions earlier: one exposed L2 round trip per round (profiles/r05_s_stage_waits.txt).
#ifndef SMZ_EARLY_STAGE
#define SMZ_EARLY_STAGE 1
#endif
#ifdef SMZ_BPS_PROBE
#define SMZ_PROBE_DECL unsigned long long pb_t0 = 0, pb_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#define SMZ_PROBE_START pb_t0 = __builtin_amdgcn_s_memtime();
#define SMZ_PROBE_AT(i) { const unsigned long long pb_t1 = __builtin_amdgcn_s_memtime(); pb_acc[i] += pb_t1 - pb_t0; pb_t0 = pb_t1; }
// -DSMZ_BPS_PROBE (=1): stamp set A (expand + backup | select: prepare, evaluate, chase, records + leaf | networks | staging);
// -DSMZ_BPS_PROBE=2: set B (expansion | lane-parallel backup | whole select | network inputs (fence, word requests, x rows) |
//                          the two-row pass | head outputs + hand-off | staging) -- same seven slots
#if SMZ_BPS_PROBE + 0 == 2
#define SMZ_PROBE(i)
#define SMZ_PROBE_B(i) SMZ_PROBE_AT(i)
#else
#define SMZ_PROBE(i) SMZ_PROBE_AT(i)
#define SMZ_PROBE_B(i)
#endif
#define SMZ_PROBE_FLUSH(stats) if ((stats) && lane == 0) { for (int pb_i = 0; pb_i < 7; pb_i++) atomicAdd(&(stats)[8 + pb_i], pb_acc[pb_i]); }
#else
#define SMZ_PROBE_DECL
#define SMZ_PROBE_START
#define SMZ_PROBE(i)
#define SMZ_PROBE_B(i)
#define SMZ_PROBE_FLUSH(stats)
#endif

// LDS map of k_search_mlp (floats): weights | pbc table (doubles) | per wave, every part padded to 16 bytes:
//   mlp scratch | network inputs [tpw][K4in] | path records [tpw][P] uint4 | rng tile | head outputs
struct MegaLds {
    int pbc_off, wave_off, per_wave;                       // float offsets from the LDS base
    int x_off, pv_off, rng_off, out_off, sel_off, sel_n;   // float offsets inside a wave's region (sel: block-parallel select)
    int sel_on;                                            // the selection words are there
    int trees_off, tree_words;                             // TLDS: the workgroup's trees (words per tree, blocks packed at 6 K words)
};
__host__ __device__ inline int r4(int x) { return (x + 3) & ~3; }
// tlds: the instantiation that keeps the workgroup's trees in LDS for the search (and the weights in the compact image)
__host__ __device__ inline MegaLds mega_lds(const smz_mlp_desc &d, const Params &P, int tpw, bool tlds = false, int waves = 8) {
    MegaLds m;
    m.pbc_off = r4(tlds ? smz_mlp::compact_total_floats(d) : d.total_floats - smz_mlp::rep_floats(d));
    m.wave_off = m.pbc_off + r4(2 * 2 * (P.sims + 2));     // pb_c table + reciprocal table (div_by_count)
    m.x_off = r4(2 * smz_mlp::row_scratch_floats(d));     // two rows' scratch: a same-branch pair is evaluated together
    m.pv_off = m.x_off + tpw * smz_mlp::up4(P.S + P.A);
    m.rng_off = m.pv_off + tpw * P.P * 4;
    m.out_off = m.rng_off + r4(tpw * kRngStride);
    m.sel_off = m.out_off + r4(tpw * (P.A + 2));
    m.sel_n = (P.sims + 2 + 1) & ~1;                       // 16-bit words per tree: one per block (select_block / select_chase)
    // (picks per block + the path of the descent: 2 x tpw x sel_n 16-bit words.)  Round 5: the kernels whose trees stay in
    // global memory run the block-parallel selection too (SMZ_BPS_HBM) -- two actions, two children, up to 126 simulations,
    // and only when the words fit beside everything else (host and device evaluate this same function)
    bool sel = tlds;
    if (!tlds && SMZ_BPS_HBM && P.A == 2 && P.K == 2 && tpw == 2 && P.sims <= 126) {
        const int per = m.sel_off + r4(tpw * m.sel_n);
        sel = ((size_t)m.wave_off + (size_t)waves * per) * sizeof(float) <= (size_t)160 * 1024;
    }
    m.sel_on = sel ? 1 : 0;
    m.per_wave = m.sel_off + (sel ? r4(tpw * m.sel_n) : 0);
    m.tree_words = r4(P.rb_words + P.sims * 6 * P.K + (P.K == 2 ? 2 * P.sims : 0));   // (+ the chance thresholds, one double per block;
                                                                                      //  trees stay 16-byte aligned)
    m.trees_off = m.wave_off + waves * m.per_wave;
    return m;
}

// INSTR: instrumented build (level statistics, s_memtime phase stamps, SMZ_DEBUG_SKIP ablations); the production
// instantiation carries none of it -- the accumulators alone cost a dozen scalar registers in a kernel that spills them.
// AEX ("exact"): the specialised instantiation for the common case -- the action count equals the MAXA bucket
// (2, 4, 8, ...), a wavefront owns exactly two trees (the geometry of 4096 trees on 256 CUs) and the networks have the
// shape the reference ships (state_space_dimensions 31, hidden_layer_dimensions 64, number_of_hidden_layer 0: 11 of
// the 16 experiment configs under config/, and checkpoint 421).  A, tpw, K, S, H, L then are compile-time constants for
// everything inlined below: layer loops unroll, and the run-time `j < A` / `t < tpw` / `k < K4` predicates no longer
// live in hoisted 64-bit scalar masks (the generic instantiation spills ~90 SGPRs into VGPR lanes).  Any other
// geometry or shape takes the generic instantiation; both produce identical results (tests/test_gpu_end_to_end.py).
struct ActOut {              // smz_search_mlp_act: Game.policy_step folded into the tail of the launch (action == nullptr: off)
    double temperature;
    int32_t *action;
    double *policy, *child_visits;
    float *root_value;
};
constexpr int kFastTpw = 2, kFastS = 31, kFastH = 64, kFastL = 0;
#if SMZ_PART == 0 || SMZ_PART == 2 || SMZ_PART == 4 || SMZ_PART == 6
// MSK: the handle has a per-tree on / off array (smz_set_active).  Without one the validity of a tree slot is a comparison
// that is recomputed where needed; with one it is state that stays live through the search loop -- in the specialised
// instantiation that costs scalar registers it does not have (35 -> 45 spilled, -3 % measured), so it exists both ways.
// PHX: the specialised instantiation for SMZ_RNG_PHILOX handles (counter streams: no state words to load or store).
// TLDS (specialised instantiation, when it fits: ~53 simulations at 2 actions): the workgroup's 16 trees live in LDS for the
// search -- a descent level is an LDS round trip instead of an L2 one, the backup's stores stay on the CU -- and go to their
// place in global memory once, at the end.  LDS room comes from the compact weight image (smz_mlp::mat_op) and from packing
// expansion blocks at 6 K words instead of 64-byte granules.
// (round 6: num_simulations as a ninth, compile-time parameter of the headline instantiation -- LDS map, strides and table offsets
//  as immediates -- measured +0.2 %, inside the spread: profiles/r06_h_sims50_ab.txt; not kept)
template <int MAXA, int KS, int U, bool INSTR, bool AEX, bool MSK = true, bool PHX = false, bool TLDS = false>
__global__ void __launch_bounds__(SMZ_SEARCH_THREADS) k_search_mlp(Params Pin, smz_mlp_desc d, const float *weights, const float *obs,
                                                    int train, ActOut act, EnvStep env) {
    Params P = Pin;
    P.tree0 = 0;
    if (AEX) { P.A = MAXA; P.tpw = kFastTpw; d.A = MAXA; d.S = kFastS; d.H = kFastH; d.L = kFastL; d.OP = smz_mlp::kWave; P.S = kFastS; }
    if (AEX) P.philox = PHX ? 1 : 0;         // (a constant in everything inlined below)
    if (KS > 0) P.K = KS;
    fix_layout(P, AEX, KS > 0);
    if (AEX) P.hs = (kFastS + 15) & ~15;
    float *lds = reinterpret_cast<float *>(smz_search_lds4);
    static_assert(!TLDS || (AEX && KS > 0), "LDS-resident trees: the specialised instantiations only");
    // LDS copy: everything but the representation matrices (TLDS: in the compact image)
    const smz_mlp_desc dl = TLDS ? smz_mlp::lds_desc_compact(d) : smz_mlp::lds_desc_without_rep(d);
    if (TLDS) smz_mlp::stage_weights_compact(lds, weights, d);
    else smz_mlp::stage_weights_without_rep(lds, weights, d);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, waves = blockDim.x / kWave;
    const int A = P.A, S = P.S, tpw = P.tpw;
    const MegaLds ml = mega_lds(d, P, tpw, TLDS, waves);
    uint32_t *const nodes_global = P.nodes;
    const int gl_eb_words = P.eb_words;
    if (TLDS) {       // (compile-time block geometry: every tree access below becomes a ds_* instruction)
        uint32_t *nodes_lds = reinterpret_cast<uint32_t *>(lds + ml.trees_off);
        for (int i = threadIdx.x; i < waves * tpw * ml.tree_words; i += blockDim.x) nodes_lds[i] = 0u;
        P.nodes = nodes_lds;
        P.tree0 = blockIdx.x * waves * tpw;
        P.eb_words = 6 * KS;
        P.tree_words = ml.tree_words;
    }
    // Chance thresholds (select_tree / expand_backup_tree <THR>): in the padding of the 64-byte block when the trees are in
    // global memory, behind the tree's packed blocks when they are in LDS
    // ... and, with the trees in LDS, the children's value terms of the decision-flagged blocks (select_tree<YV>: one division
    // less per decision level; with the trees in global memory the extra store per backup level costs more than it saves:
    // 423 -> 415 M at 4096 x 100)
    constexpr bool THR = AEX && KS == 2, YV = THR && TLDS;
    if (THR) {
        P.thr_off = TLDS ? P.rb_words + P.sims * 6 * KS : P.rb_words + 12;
        P.thr_stride = TLDS ? 2 : P.eb_words;
        P.ry_off = (YV && MAXA <= 8 && P.rp_off + 3 * A <= P.rb_words) ? P.rp_off + 2 * A : -1;      // (select_tree<YV> reads it for MAXA <= 8)
    }
    double *pbc_lds = reinterpret_cast<double *>(lds + ml.pbc_off);
    const int n_pbc = P.sims + 2;
    for (int i = threadIdx.x; i < n_pbc; i += blockDim.x) {
        pbc_lds[i] = P.pbc_sqrt[i];
        pbc_lds[n_pbc + i] = i > 0 ? 1.0 / (double)i : 0.0;      // IEEE division: correctly rounded reciprocals
    }
    const int slot = A + 2;
    const int K4in = smz_mlp::up4(S + A);
    float *scratch = lds + ml.wave_off + wave * ml.per_wave;
    float *xall = scratch + ml.x_off;                                           // [tpw][K4in] network inputs of the round
    uint4 *pvals = reinterpret_cast<uint4 *>(scratch + ml.pv_off);              // [tpw][P] path records
    uint32_t *rng_tile = reinterpret_cast<uint32_t *>(scratch + ml.rng_off);
    float *outs = scratch + ml.out_off;                                         // [tpw][A + 2]: policy | value | reward
    __syncthreads();

    const int tree0 = (blockIdx.x * waves + wave) * tpw;
    const int tree = tree0 + lane;
    const bool valid = lane < tpw && tree < P.B && (!MSK || tree_active(P, tree));
    // a wave none of whose trees is searched (beyond B, or switched off with smz_set_active) is done: there is no
    // workgroup barrier after the weight staging above
    if (MSK && __ballot(valid) == 0ull) {
        // (smz_search_mlp_act_cartpole: the switched-off envs of this wave still get their "no step" record)
        if (env.state && lane < tpw && tree < P.B) cartpole_env_step(env, tree, P.B, act.action, act.policy, act.child_visits, act.root_value);
        return;
    }

    // ---- root: representation + prediction per row, then root expansion per lane ---------------------------------
    for (int t = 0; t < tpw; t++) {
        const int row = tree0 + t;
        if (row >= P.B) break;                                   // wave-uniform
        if (MSK && !__shfl((int)valid, t)) continue;             // wave-uniform: the tree is switched off
        smz_mlp::initial_row<U, TLDS>(lds, dl, weights, d, scratch, obs + (size_t)row * d.obs, P.hidden + (size_t)row * P.N * P.hs,
                                      nullptr, outs + t * slot);
    }
    constexpr bool PHC = !AEX || PHX;   // the specialised instantiations are compiled for one word source each
    int packed = wave_stage_rng<PHC>(P, tree, valid, rng_tile);
    RngT<PHC> rng;
    rng.bind(P, tree, valid);
    TreeHdr h = {0, 0, 0.f, 0.f, 0, 0.f, 0, 0};
    if (valid) {
        rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
        root_init_tree<MAXA>(P, tree, rng, outs + lane * slot, nullptr, train != 0);
        h = P.hdr[tree];
        packed = rng.pack();
    }
    unsigned n_dec = 0, n_chance = 0, n_children = 0, n_desc = 0;
    // (wave-uniform, read once: inside the rounds a lane picks its tree slot's flag with a select, not a cross-lane read)
    // (one scalar register: the kernel is short of them, and a wave-uniform bool costs a 64-bit lane mask)
    const int vmask = MSK ? __builtin_amdgcn_readlane((int)valid, 0) | (__builtin_amdgcn_readlane((int)valid, 1) << 1) : 3;
#define SMZ_SLOT_VALID(src) (MSK ? ((vmask >> (src)) & 1) != 0 : tree0 + (src) < P.B)
    const bool prof = INSTR && (P.dbg & 16) && P.stats;
    const int dbg = INSTR ? P.dbg : 0;
    unsigned long long t_stage = 0, t_expand = 0, t_select = 0, t_mlp = 0, t0 = 0, t1 = 0;
#define SMZ_STAMP(acc) if (INSTR && prof) { t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
    // ---- simulations -------------------------------------------------------------------------------------------------
    // Random words are staged once per round, for the NEXT round: the source words are requested right after the
    // selection (the stream position is final then) and the loads fly during the network evaluation.
    constexpr int SU = 2;
    const bool split = P.tpw <= SU;
    if (P.sims > 0 && !(dbg & 8)) packed = wave_stage_rng_from<4, PHC>(P, tree, valid, rng_tile, packed, rng.block());
    SMZ_PROBE_DECL
    for (int s = 0; s < P.sims; s++) {
        if (INSTR && prof) t0 = __builtin_amdgcn_s_memtime();
        SMZ_PROBE_START
        Leaf L = {0, 0, 0, 0};
        __builtin_amdgcn_s_setprio(SMZ_PRIO_TREE);
        // Specialised kernel: the expansion runs in the tree's lane, the backup with one lane per path level
        // (backup_levels_lanes: lanes t, t + 2, ..., t + 14 take eight levels of tree slot t per pass).
        constexpr bool LBKP = AEX && !INSTR;
        float leaf_rw = 0.f;
        if (valid) {
            rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
            if (s > 0 && !(dbg & 4)) expand_backup_tree<MAXA, KS, LBKP, THR, YV>(P, tree, rng, h, outs + lane * slot, outs[lane * slot + A + 1],
                                                      outs[lane * slot + A], pvals + lane * P.P, &leaf_rw);
        }
        SMZ_PROBE_B(0)
        if constexpr (LBKP) {
            if (s > 0) {
                const int src = lane & (kFastTpw - 1);
                const int len = pick_lane01(h.path_len, src);
                const float lrw = pick_lane01(leaf_rw, src);
                if (lane < 8 * kFastTpw && SMZ_SLOT_VALID(src)) {
                    const bool own = lane < kFastTpw;
                    float mn = own ? h.mn : __builtin_inff(), mx = own ? h.mx : -__builtin_inff(), v_root = 0.f;
                    backup_levels_lanes<kFastTpw, YV>(P, tree0 + src, lane / kFastTpw, len, outs[src * slot + A], lrw,
                                                  pvals + src * P.P, mn, mx, v_root);
                    if (own) {   // the root itself (reward 0)
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
        }
        SMZ_STAMP(t_expand)
        SMZ_PROBE(0)
        SMZ_PROBE_B(1)
        // Specialised two-action kernel: the descent runs on lanes 0..3 -- lane t and its helper t + 2 score one child
        // each (pick_decision_pair).  The helper works on a copy of the tree lane's stream position and MinMax bounds.
        // (round 6: also four children per block on a four-action tree -- KS = 4: the pair splits EVERY decision level two and
        //  two, pick_decision_pair<Kids<4>>, as it does the root)
        constexpr bool PAIR = AEX && (MAXA == 2 || (SMZ_PAIR_A4 && MAXA == 4)) && (KS == 2 || (SMZ_PAIR_K4 && KS == 4 && MAXA == 4)) && !INSTR;
        // Trees in LDS: block-parallel selection (smz_device.hpp, select_block / select_chase) -- every block of the wave's two
        // trees gets a lane that computes the block's pick from the words its level will read, then the tree's lane follows the
        // picks.  SMZ_SELECT_BLOCKS=0 (-DSMZ_SELECT_BLOCKS=0 builds) keeps the level-by-level descent.
        constexpr bool BPS = SMZ_SELECT_BLOCKS && AEX && KS == 2 && (TLDS || SMZ_BPS_HBM) && !INSTR && MAXA == 2;   // (four actions: the root is a
        // code path of its own beside the blocks' -- measured 409 against 458 M, profiles/r04_bps_ab.txt)
        bool bps_done = false;
        bool bps_all = false;                 // every tree of the wave went through the block-parallel selection this round
        float early_row[kFastTpw] = {0.f, 0.f};
        StagePre<SU> pre;
        bool staged_early = false;            // (wave-uniform) the next round's source words were requested with the parent rows
        if constexpr (BPS) if (TLDS || ml.sel_on) {
            uint16_t *selw = reinterpret_cast<uint16_t *>(scratch + ml.sel_off);          // [tpw][sel_n]
            const int SELN = ml.sel_n;
            if (valid && s > 0) selw[lane * SELN + h.n_exp] = (uint16_t)(h.path_len << 9);   // depth of the node the expansion created
            smz_mlp::lds_sync();
            const int src = lane & 1;
            const int nexp = pick_lane01(valid ? h.n_exp : -1, src), rvis = pick_lane01(h.root_visit, src);
            const float bmn = pick_lane01(h.mn, src), bmx = pick_lane01(h.mx, src);
            const int bused = pick_lane01(valid ? rng.used : 0, src), bstaged = pick_lane01(valid ? rng.staged : 0, src);
            const int bstage = pick_lane01(valid ? (int)(rng.stage - rng_tile) : 0, src);   // the tree's staged words, as its lane reads them
            const int nmax = max(__builtin_amdgcn_readlane(valid ? h.n_exp : -1, 0), __builtin_amdgcn_readlane(valid ? h.n_exp : -1, 1));
            const uint32_t *stb = tree_base(P, tree0 + src);
            SMZ_PROBE(1)
            if constexpr (!TLDS && SMZ_SELECT_LOADS_FIRST && SMZ_SELECT_TWO_PASSES) {
                // Trees in global memory: a pass of 32 blocks per tree is one L2 round trip, and a search of 100 simulations takes
                // up to four of them one after the other.  The blocks of TWO passes are requested before the first is decided.
                constexpr int NP = SMZ_SELECT_TWO_PASSES > 1 ? SMZ_SELECT_TWO_PASSES : 2;          // passes in flight
                for (int base = 0; base <= nmax; base += NP * (kWave / 2)) {
                    BlockRaw raw[NP];
#pragma unroll
                    for (int p = 0; p < NP; p++) {
                        const int b = base + p * (kWave / 2) + (lane >> 1);
                        if (b <= nexp) select_block_request<MAXA, YV>(P, stb, b, raw[p]);
                    }
#pragma unroll
                    for (int p = 0; p < NP; p++) {
                        const int b = base + p * (kWave / 2) + (lane >> 1);
                        if (b <= nexp) {
                            const int depth = b == 0 ? 0 : (int)(selw[src * SELN + b] >> 9);
                            const uint32_t r = select_block_decide<MAXA, YV, RngT<PHC>>(P, b, depth, rvis, bmn, bmx, rng_tile + bstage, bused, bstaged,
                                                                                         pbc_lds, raw[p]);
                            selw[src * SELN + b] = (uint16_t)((depth << 9) | r);
                        }
                    }
                }
            } else
            for (int base = 0; base <= nmax; base += kWave / 2) {
                const int b = base + (lane >> 1);
                if (b <= nexp) {
                    const int depth = b == 0 ? 0 : (int)(selw[src * SELN + b] >> 9);
                    const uint32_t r = select_block<MAXA, YV, RngT<PHC>, !TLDS>(P, stb, b, depth, rvis, bmn, bmx, rng_tile + bstage, bused, bstaged,
                                                                        pbc_lds);
                    selw[src * SELN + b] = (uint16_t)((depth << 9) | r);
                }
            }
            smz_mlp::lds_sync();
            SMZ_PROBE(2)
            // the descent: the tree's lane follows the picks (one dependent LDS read per level) and leaves the path in the upper half
            // of the tree's sel words (sel_n covers both); then one lane per level writes that level's path record.
            // (The same chase on the scalar unit -- picks kept in the lanes that computed them, a v_readlane per level, path entries
            // dropped into lanes by compare + select, no LDS traffic -- was built twice: round 4 at the 256-register limit (-8 %,
            // scratch) and round 5 with 58 registers to spare (-4 %: 106 scalar registers are the limit too, and a taken scalar
            // branch per level costs what the LDS round trip does); profiles/r05_f_pair_chase_ab.txt, commit "experiment: scalar
            // chase".)
            uint16_t *pathw = selw + tpw * SELN;                                        // [tpw][sel_n]
            int len = 0;
            if (valid) len = select_chase(selw + lane * SELN, pathw + lane * SELN);
            smz_mlp::lds_sync();
            SMZ_PROBE(3)
            const int blen = pick_lane01(len, src);
#if SMZ_EARLY_ROWS
            // Round 5: the leaf's PARENT is the path's last-but-one entry -- known here, before the path records, the leaf's
            // action and the stream position are worked out.  The loads of the wave's two parent rows (global memory: an L2 round
            // trip of ~1.5 k cycles that used to start only after all of that) are issued now and land in registers while the
            // LDS-only rest of the selection runs; the network inputs are written from the registers.  The tree phases of this
            // instantiation touch no global memory, so no later wait of the selection sits behind these loads (gfx950 returns
            // vector-memory loads in order: profiles/r03_ceiling.md 6b).  A tree that falls back to the sequential descent
            // (bps_all false) takes the old path.
            bps_all = __ballot(valid && len == 0) == 0ull && __ballot(valid) != 0ull;
#if SMZ_EARLY_STAGE
            if (valid && len > 0) {                                     // the words the descent's levels drew (all inside the staged window)
                const int nw = select_words(len, A);
                rng.used += nw; rng.ready -= nw; rng.idx += nw;
                if (rng.idx >= kMtN) { rng.idx -= kMtN; rng.wrapped(); }
                packed = rng.pack();
            }
#endif
            if (bps_all) {
                int par = 0;
                if (valid && len > 1) { const int loc = pathw[lane * SELN + len - 2], pb = loc >> 8; par = pb == 0 ? 1 + (loc & 3) : 1 + A + (pb - 1) * 2 + (loc & 3); }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // (rows stored in earlier rounds may be this round's parents)
#pragma unroll
                for (int t = 0; t < kFastTpw; t++) {
                    const int parent = __builtin_amdgcn_readlane(par, t);
                    const float *srow = P.hidden + ((size_t)(tree0 + t) * P.N + parent) * P.hs;
                    early_row[t] = (lane < S && tree0 + t < P.B) ? srow[lane] : 0.f;
                }
#if SMZ_EARLY_STAGE
                if (split && !(dbg & 8)) { stage_issue<SU, PHC>(P, tree, valid, packed, pre); staged_early = true; }
#endif
            }
#endif
            // (trees in global memory: the leaf's action word is requested BEFORE the path records' words, so the two L2 round trips
            //  overlap -- the records' loop waits for its own loads, and the leaf's used to start only behind that wait)
            if constexpr (!TLDS && SMZ_LEAF_FIRST) { if (valid && len > 0) L = select_leaf(P, stb, pathw + lane * SELN, len); }
            for (int d = lane >> 1; d < blen; d += kWave / 2) select_record(P, stb, pathw + src * SELN, d, pvals + src * P.P);
            if (valid && len > 0) {
                if constexpr (TLDS || !SMZ_LEAF_FIRST) L = select_leaf(P, stb, pathw + lane * SELN, len);
#if !(SMZ_EARLY_ROWS && SMZ_EARLY_STAGE)
                const int nw = select_words(len, A);                    // the words the descent's levels drew (all inside the staged window)
                rng.used += nw; rng.ready -= nw; rng.idx += nw;
                if (rng.idx >= kMtN) { rng.idx -= kMtN; rng.wrapped(); }
                packed = rng.pack();
#endif
                h.path_len = len;
                bps_done = true;
            }
        }
        if constexpr (BPS) {
            // (a level's words beyond the staged window -- a very deep path, or a window nearly used up: the sequential descent)
            if (valid && !bps_done) {
                int len = 0;
                L = select_tree<MAXA, KS, false, true, false, THR, YV>(P, tree, rng, h, pbc_lds, len, n_dec, n_chance, n_children, pvals + lane * P.P);
                h.path_len = len;
                packed = rng.pack();
            }
        } else if constexpr (PAIR) {
            const int src = lane & 1;
            const int pk = pick_lane01(valid ? rng.pack() : 0, src), us = pick_lane01(valid ? rng.used : 0, src);

            const float hmn = pick_lane01(h.mn, src), hmx = pick_lane01(h.mx, src);
            const int hrv = pick_lane01(h.root_visit, src);
            const uint32_t pb = (uint32_t)pick_lane01((int)rng.block(), src), pk0 = (uint32_t)pick_lane01((int)rng.key0(), src),
                           pk1 = (uint32_t)pick_lane01((int)rng.key1(), src);
            if (lane < 4 && SMZ_SLOT_VALID(src)) {
                TreeHdr hs = h;
                if (lane >= 2) {
                    rng.load(P.mt + (size_t)(tree0 + src) * kMtN, pk, rng_tile + src * kRngStride, kRngStage);
                    rng.used = us;
                    rng.follow(pb, pk0, pk1);      // (Philox: the helper draws beyond the staged window from the TREE's stream)
                    hs.mn = hmn; hs.mx = hmx; hs.root_visit = hrv;
                }
                int len = 0;
                const Leaf Lp = select_tree<MAXA, KS, false, true, true, THR, YV>(P, tree0 + src, rng, hs, pbc_lds, len, n_dec, n_chance,
                                                                             n_children, pvals + src * P.P, lane >> 1);
                if (lane < 2) {
                    L = Lp;
                    h.path_len = len;
                    packed = rng.pack();
                }
            }
        } else if (valid) {
            int len = 0;
            if (dbg & 2) { L.leaf_id = 1; L.parent_id = 0; L.action = 0; L.branch = 0; len = 1; }
            else L = select_tree<MAXA, KS, INSTR, true, false, THR, YV>(P, tree, rng, h, pbc_lds, len, n_dec, n_chance, n_children, pvals + lane * P.P);
            h.path_len = len;
            if (INSTR) n_desc++;
            packed = rng.pack();
        }
        SMZ_STAMP(t_select)
        SMZ_PROBE(4)
        SMZ_PROBE_B(2)
        __builtin_amdgcn_s_setprio(SMZ_PRIO_HEADS);
        // hidden rows written in earlier rounds (by any lane of this wave) may be this round's parent rows
        if (!(BPS && SMZ_EARLY_ROWS && bps_all)) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // (after the fence: it drains the vector-memory counter)
        if (split && !(dbg & 8) && !staged_early) stage_issue<SU, PHC>(P, tree, valid, packed, pre);
        // all rows' network inputs first (independent global loads, one latency), then the rows one after another
        if (BPS && SMZ_EARLY_ROWS && bps_all) {                 // (the parent rows are in registers already: see the selection)
#pragma unroll
            for (int t = 0; t < kFastTpw; t++) {
                if (tree0 + t >= P.B) break;
                const int act = __builtin_amdgcn_readlane(L.action, t);
                if (lane < K4in) xall[t * K4in + lane] = (lane < S) ? early_row[t] : ((lane < S + A && (lane - S) == act) ? 1.f : 0.f);
            }
        } else
        for (int t = 0; t < tpw; t++) {
            const int row = tree0 + t;
            if (row >= P.B) break;                               // wave-uniform
            const int parent = __builtin_amdgcn_readlane(L.parent_id, t), act = __builtin_amdgcn_readlane(L.action, t);
            const float *src = P.hidden + ((size_t)row * P.N + parent) * P.hs;
            for (int k = lane; k < K4in; k += kWave)
                xall[t * K4in + k] = (k < S) ? src[k] : ((k < S + A && (k - S) == act) ? 1.f : 0.f);
        }
        smz_mlp::lds_sync();
        SMZ_PROBE_B(3)
        bool paired = false;
        if (tpw == 2 && !(dbg & 1)) {
            // the wave's two leaves need the same pair of networks: one pass, weights read from LDS once for both rows
            const int b0 = __builtin_amdgcn_readlane(L.branch, 0), b1 = __builtin_amdgcn_readlane(L.branch, 1);
            if (tree0 + 1 < P.B) {
                const float *xin[2] = {xall, xall + K4in};
                const bool dyn[2] = {b0 != 0, b1 != 0};
                float *dh[2], *dp[2] = {outs, outs + slot};
                float reward[2], value[2];
#pragma unroll
                for (int r = 0; r < 2; r++)
                    dh[r] = P.hidden + ((size_t)(tree0 + r) * P.N + __builtin_amdgcn_readlane(L.leaf_id, r)) * P.hs;
                const bool live[2] = {MSK ? (vmask & 1) != 0 : true, MSK ? (vmask & 2) != 0 : true};
                if (b0 == b1) smz_mlp::recurrent_rows<U, 2, true, TLDS>(lds, dl, scratch, xin, dyn, live, dh, dp, reward, value);
                else smz_mlp::recurrent_rows<U, 2, false, TLDS>(lds, dl, scratch, xin, dyn, live, dh, dp, reward, value);
                SMZ_PROBE_B(4)
                if (lane == 0) {
                    outs[A] = value[0]; outs[A + 1] = reward[0];
                    outs[slot + A] = value[1]; outs[slot + A + 1] = reward[1];
                }
                paired = true;
            }
        }
        for (int t = 0; t < tpw && !paired; t += smz_mlp::kRows) {
            if (tree0 + t >= P.B) break;                         // wave-uniform
            constexpr int R = smz_mlp::kRows;
            const float *xin[R];
            bool dyn[R], live[R];
            float *dh[R], *dp[R];
            float reward[R], value[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                live[r] = (t + r < tpw) && (tree0 + t + r < P.B);
                const int tt = live[r] ? t + r : t;
                live[r] = live[r] && (!MSK || __shfl((int)valid, tt) != 0);
                const int row = tree0 + tt;
                const int leaf = __builtin_amdgcn_readlane(L.leaf_id, tt);
                dyn[r] = __builtin_amdgcn_readlane(L.branch, tt) != 0;
                xin[r] = xall + tt * K4in;
                dh[r] = P.hidden + ((size_t)row * P.N + leaf) * P.hs;
                dp[r] = outs + tt * slot;
            }
            if (!(dbg & 1)) smz_mlp::recurrent_rows<U, R, false, TLDS>(lds, dl, scratch, xin, dyn, live, dh, dp, reward, value);
#pragma unroll
            for (int r = 0; r < R; r++)
                if (live[r] && lane == 0) { outs[(t + r) * slot + A] = value[r]; outs[(t + r) * slot + A + 1] = reward[r]; }
        }
        smz_mlp::lds_sync();
        SMZ_STAMP(t_mlp)
        SMZ_PROBE(5)
        SMZ_PROBE_B(5)
        if (!(dbg & 8)) packed = split ? stage_finish<SU, PHC>(P, tree, valid, rng_tile, packed, pre, rng.block())
                                       : wave_stage_rng_from<4, PHC>(P, tree, valid, rng_tile, packed, rng.block());
        SMZ_STAMP(t_stage)
        SMZ_PROBE(6)
        SMZ_PROBE_B(6)
    }
    SMZ_PROBE_FLUSH(Pin.stats)
#undef SMZ_STAMP
#undef SMZ_SLOT_VALID
    if (INSTR && prof && lane == 0) {
        atomicAdd(&P.stats[4], t_stage); atomicAdd(&P.stats[5], t_expand);
        atomicAdd(&P.stats[6], t_select); atomicAdd(&P.stats[7], t_mlp);
    }
    if (valid) {
        if (P.sims > 0) {
            rng.load(P.mt + (size_t)tree * kMtN, packed, rng_tile + lane * kRngStride, kRngStage);
            expand_backup_tree<MAXA, KS, false, THR, YV>(P, tree, rng, h, outs + lane * slot, outs[lane * slot + A + 1], outs[lane * slot + A],
                                                     pvals + lane * P.P);
            // leave the last path where the step-wise entry points and the debug dump expect it
            for (int i = 0; i < h.path_len; i++) P.path[(size_t)i * P.B + tree] = pvals[lane * P.P + i];
            packed = rng.pack();
        }
        P.hdr[tree] = h;
        if (act.action) {
            // the post-search policy / action of game.py:179-232 on the finished tree: the same draws from the same
            // stream position as a separate smz_act launch (rng still holds this tree's position)
            act_tree<MAXA>(P, tree, rng, act.temperature, act.action, act.policy, act.child_visits, act.root_value);
            packed = rng.pack();
        }
        P.rng_pos[tree] = packed;
        rng.save(P, tree);
    }
    if constexpr (TLDS) {
        // the wave's finished trees back to their place in global memory (64-byte granules there: words 12..15 of a block are
        // padding).  Each wave moves its own trees: no other wave has touched them.
        smz_mlp::lds_sync();
        const int n_blk = 1 + P.sims;                                    // root block + one expansion block per simulation
        for (int t = 0; t < tpw; t++) {
            if (tree0 + t >= P.B) break;
            if (MSK && !__shfl((int)valid, t)) continue;             // a switched-off tree keeps what its last search left in HBM
            const uint32_t *src = P.nodes + (size_t)(wave * tpw + t) * P.tree_words;
            uint32_t *dst = nodes_global + (size_t)(tree0 + t) * (P.rb_words + (size_t)P.sims * gl_eb_words);
            for (int i = lane; i < P.rb_words; i += kWave) dst[i] = src[i];
            for (int i = lane; i < (n_blk - 1) * gl_eb_words; i += kWave) {
                const int b = i / gl_eb_words, w = i - b * gl_eb_words;
                dst[P.rb_words + i] = w < P.eb_words ? src[P.rb_words + b * P.eb_words + w] : 0u;
            }
        }
    }
    // smz_search_mlp_act_cartpole: the env step + trajectory record of this tree's env in the same lane (it reads back
    // the action / policy / child_visits / root value it has just written); a switched-off tree of a live wave gets its
    // "no step" record.  The next observation goes where this launch read the current one: only this wave reads that row.
    if (env.state && lane < tpw && tree < P.B)
        cartpole_env_step(env, tree, P.B, act.action, act.policy, act.child_visits, act.root_value);
    if (INSTR) wave_add_stats(P.stats, n_dec, n_chance, n_desc, n_children);
}
#endif

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(kWave) k_root_stats(Params P, int32_t *visits, double *priors, float *root_value,
                                                      float *child_reward) {
    const int tree = blockIdx.x * kWave + threadIdx.x;
    if (tree >= P.B) return;
    const uint32_t *rb = tree_base(P, tree);
    const double *rp = (const double *)(rb + P.rp_off);
    for (int a = 0; a < P.A; a++) {
        if (visits) visits[(size_t)tree * P.A + a] = (int32_t)rb[2 * a];
        if (priors) priors[(size_t)tree * P.A + a] = rp[a];
        if (child_reward) child_reward[(size_t)tree * P.A + a] = __uint_as_float(rb[2 * P.A + a]);
    }
    if (root_value) {
        const TreeHdr h = P.hdr[tree];
        root_value[tree] = h.root_visit ? h.root_value_sum / (float)h.root_visit : 0.0f;
    }
}
#endif

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
template <int MAXA>
__global__ void __launch_bounds__(kWave) k_act(Params P, double temperature, int32_t *action, double *policy,
                                               double *child_visits, float *root_value) {
    const int tree = blockIdx.x * kWave + threadIdx.x;
    if (tree >= P.B || !tree_active(P, tree)) return;
    Rng rng;
    rng.bind(P, tree, true);
    rng.load(P.mt + (size_t)tree * kMtN, P.rng_pos[tree], nullptr, 0);
    act_tree<MAXA>(P, tree, rng, temperature, action, policy, child_visits, root_value);
    P.rng_pos[tree] = rng.pack();
    rng.save(P, tree);
}
#endif

// ---- head epilogues: `lpr` lanes cooperate on one row (coalesced loads, shuffle reductions) ------------------------
__device__ inline float grp_max(float v, int lpr) {
    for (int off = lpr >> 1; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ inline float grp_min(float v, int lpr) {
    for (int off = lpr >> 1; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ inline float grp_sum(float v, int lpr) {
    for (int off = lpr >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// inverse_transform_with_support of one row by a lane group (muzero_model.py:575-591); every lane gets the result
__device__ inline float support_decode_group(const float *row, int S, int li, int lpr) {
    float m = -__builtin_inff();
    for (int i = li; i < S; i += lpr) m = fmaxf(m, row[i]);
    m = grp_max(m, lpr);
    float den = 0.f, num = 0.f;
    const int half = S / 2;
    for (int i = li; i < S; i += lpr) {
        const float e = expf(row[i] - m);
        den += e;
        num += (float)(i - half) * e;
    }
    den = grp_sum(den, lpr);
    num = grp_sum(num, lpr);
    const float y = num / den;
    const float sg = (y > 0.f) ? 1.f : ((y < 0.f) ? -1.f : 0.f);
    const float r = (sqrtf(1.f + 4.f * 0.001f * (fabsf(y) + 1.f + 0.001f)) - 1.f) / (2.f * 0.001f);
    return sg * (r * r - 1.f);
}

__device__ inline void softmax_group(const float *row, int A, float *out, int li, int lpr) {
    float m = -__builtin_inff();
    for (int i = li; i < A; i += lpr) m = fmaxf(m, row[i]);
    m = grp_max(m, lpr);
    float den = 0.f;
    for (int i = li; i < A; i += lpr) den += expf(row[i] - m);
    den = grp_sum(den, lpr);
    for (int i = li; i < A; i += lpr) out[i] = expf(row[i] - m) / den;
}

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(256) k_support_decode(const float *logits, int S, float *out, int B, int lpr) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / lpr, li = threadIdx.x % lpr;
    const int row = gid < B ? gid : B - 1;     // surplus groups recompute the last row (keeps shuffles convergent)
    const float v = support_decode_group(logits + (size_t)row * S, S, li, lpr);
    if (gid < B && li == 0) out[row] = v;
}
#endif

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(256) k_policy_softmax(const float *logits, int A, float *out, int B, int lpr) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / lpr, li = threadIdx.x % lpr;
    if (gid < B) softmax_group(logits + (size_t)gid * A, A, out + (size_t)gid * A, li, lpr);
    else { float dummy[1]; softmax_group(logits + (size_t)(B - 1) * A, 0, dummy, li, lpr); }
}
#endif

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(256) k_dynamics_epilogue(const float *state_dyn, const float *state_after,
                                                           const float *reward_logits, int ld, const uint8_t *branch,
                                                           int S, float *hidden_out, float *reward_out, int B, int lpr) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / lpr, li = threadIdx.x % lpr;
    const int row = gid < B ? gid : B - 1;
    const bool live = gid < B;
    const bool dyn = branch[row] != 0;
    const float *x = (dyn ? state_dyn : state_after) + (size_t)row * ld;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int i = li; i < S; i += lpr) { const float v = x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    mn = grp_min(mn, lpr);
    mx = grp_max(mx, lpr);
    float sc = mx - mn;
    if (sc < 1e-5f) sc += 1e-5f;  // neural_network_mlp_model.py:353
    if (live) for (int i = li; i < S; i += lpr) hidden_out[(size_t)row * S + i] = (x[i] - mn) / sc;
    if (reward_out) {
        float r = 0.f;
        if (reward_logits) r = support_decode_group(reward_logits + (size_t)row * ld, S, li, lpr);
        if (live && li == 0) reward_out[row] = dyn ? r : 0.f;
    }
}
#endif

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(256) k_prediction_epilogue(const float *pol_pred, const float *val_pred,
                                                             const float *pol_after, const float *val_after,
                                                             int ld, const uint8_t *branch, int A, int S,
                                                             float *policy_out, float *value_out, int B, int lpr) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / lpr, li = threadIdx.x % lpr;
    const int row = gid < B ? gid : B - 1;
    const bool live = gid < B;
    const bool dyn = branch[row] != 0;
    const float *pl = (dyn ? pol_pred : pol_after) + (size_t)row * ld;
    float m = -__builtin_inff();
    for (int i = li; i < A; i += lpr) m = fmaxf(m, pl[i]);
    m = grp_max(m, lpr);
    float den = 0.f;
    for (int i = li; i < A; i += lpr) den += expf(pl[i] - m);
    den = grp_sum(den, lpr);
    if (live) for (int i = li; i < A; i += lpr) policy_out[(size_t)row * A + i] = expf(pl[i] - m) / den;
    const float v = support_decode_group((dyn ? val_pred : val_after) + (size_t)row * ld, S, li, lpr);
    if (live && li == 0) value_out[row] = v;
}
#endif

// ---- synthetic env + trajectory record ---------------------------------------------------------------------------
#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
// one thread per env; traj != nullptr: also appends the step's record -- one launch less per env step
__global__ void __launch_bounds__(256) k_cartpole_step_env(EnvStep E, const int32_t *action, int B, const double *policy,
                                                           const double *child_visits, const float *root_value) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    cartpole_env_step(E, e, B, action, policy, child_visits, root_value);
}
#endif

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
// N(0,1) float32 observations of a stand-in env (Box-Muller on two counter-based uniforms), env-major [B][obs_dim]
__global__ void __launch_bounds__(256) k_synthetic_obs(float *obs, int B, int obs_dim, uint64_t seed, long long first_env,
                                                       long long t) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * obs_dim) return;
    const uint64_t e = (uint64_t)(first_env + (long long)(i / obs_dim)), k = (uint64_t)(i % obs_dim);
    const double u1 = 1.0 - smz_unit(seed, e, (uint64_t)t, 2 * k), u2 = smz_unit(seed, e, (uint64_t)t, 2 * k + 1);
    obs[i] = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2));
}
#endif

// record layout per (step, env):
//   [obs(obs_dim) | reward | terminated | policy(A) | action one-hot(A) | root_value | child_visits(A)]
// One thread per float64 of the step's [B][F] slab: writes are contiguous across the whole slab and the observation
// reads are contiguous per row, whatever obs_dim is (4 for CartPole, 28812 for a 98x98x3 frame).
#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(256) k_traj_pack(double *traj, int T, int t, int obs_dim, int A, const float *obs,
                                                   const float *reward, const uint8_t *terminated, const int32_t *action,
                                                   const double *policy, const double *child_visits,
                                                   const float *root_value, int B) {
    const int F = obs_dim + 3 * A + 3;
    const size_t total = (size_t)B * F;
    double *slab = traj + (size_t)t * total;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i / F), k = (int)(i % F);
        double v;
        if (k < obs_dim) v = (double)obs[(size_t)e * obs_dim + k];
        else if (k == obs_dim) v = reward ? (double)reward[e] : 0.0;
        else if (k == obs_dim + 1) v = terminated ? (double)terminated[e] : 0.0;     // the flag itself: 0 / 1 / 2 / 3
        else {
            const int j = k - obs_dim - 2;
            if (j < A) v = policy[(size_t)e * A + j];
            else if (j < 2 * A) v = (j - A == action[e]) ? 1.0 : 0.0;
            else if (j == 2 * A) v = (double)root_value[e];
            else v = child_visits[(size_t)e * A + (j - 2 * A - 1)];
        }
        slab[i] = v;
    }
}
#endif

// Game length of every env of a chunk: steps up to and including the first terminated one (chunk_to_games'/play_game's
// cut, self_play.py:79-94), or T.
#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(256) k_traj_lengths(const double *traj, int T, int obs_dim, int A, int B, int ignore_term,
                                                      int32_t *length) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    const int F = obs_dim + 3 * A + 3;
    int n = T;
    if (!ignore_term)
        for (int t = 0; t < T; t++)
            if (traj[((size_t)t * B + e) * F + obs_dim + 1] != 0.0) { n = t + 1; break; }
    length[e] = n;
}
#endif

// Several games per env and chunk (on_end = "reset"; chunk_to_games(after_end = "new_game")): game_end[t][e] = one past the
// last row of the game that row t of env e belongs to -- the row of its end flag (1 terminated / 2 step limit) + 1, or T
// for the unfinished game at the end of the chunk; -1 for a row without a step (flag 3).  new_game == 0: rows behind the
// first finished game belong to no game (the cut of k_traj_lengths).  length[e] = end of the env's first game.
#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(256) k_traj_game_ends(const double *traj, int T, int obs_dim, int A, int B, int ignore_term,
                                                        int new_game, int32_t *length, int32_t *game_end) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    const int F = obs_dim + 3 * A + 3;
    int end = T, first = T;
    for (int t = T - 1; t >= 0; t--) {
        const int flag = ignore_term ? 0 : (int)traj[((size_t)t * B + e) * F + obs_dim + 1];
        if (flag == 3) { end = t; game_end[(size_t)t * B + e] = -1; continue; }
        if (flag != 0) end = t + 1;
        game_end[(size_t)t * B + e] = end;
        first = end;
    }
    if (!new_game)
        for (int t = first; t < T; t++) game_end[(size_t)t * B + e] = -1;
    if (length) length[e] = first;
}
#endif

// n-step value target of every stored position (the value entry of Game.make_target and the target inside
// Game.make_priority, game.py:291-337), with the reference's scalar types: root values are numpy float32, rewards and
// discount powers Python floats, so under NEP 50 a bootstrapped chain (position + td_steps inside the game) runs in
// float32 -- f32(root_value) * f32(discount^td), then one f32 add per reward of the f64 product reward * discount^i
// rounded to f32 -- and a chain past the end of the game starts from a Python 0 and stays float64.
// abs_td = |float64(root_value[t]) - target| (make_priority before ** priority_scale).  Positions t >= length are 0.
#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
__global__ void __launch_bounds__(256) k_traj_targets(const double *traj, int T, int obs_dim, int A, int B, int td,
                                                      const double *disc_pow, const int32_t *length, double *target,
                                                      double *abs_td, const int32_t *game_end = nullptr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * B) return;
    const int t = (int)(i / B), e = (int)(i % B);
    // (a game that starts inside the chunk: positions, bootstrap index and end are all chunk rows, so the arithmetic of a
    // game-relative index is unchanged)
    const int F = obs_dim + 3 * A + 3, n = game_end ? game_end[i] : length[e];
    const size_t rv_off = obs_dim + 2 + 2 * A;
    double out = 0.0, err = 0.0;
    if (t < n) {
        const int b = t + td;
        if (b < n) {
            float v = (float)traj[((size_t)b * B + e) * F + rv_off] * (float)disc_pow[td];
            for (int k = 0; k < td; k++) v = v + (float)(traj[((size_t)(t + k) * B + e) * F + obs_dim] * disc_pow[k]);
            out = (double)v;
        } else {
            double v = 0.0;
            for (int k = 0; t + k < n; k++) v += traj[((size_t)(t + k) * B + e) * F + obs_dim] * disc_pow[k];
            out = v;
        }
        err = fabs(traj[i * F + rv_off] - out);
    }
    target[i] = out;
    if (abs_td) abs_td[i] = err;
}
#endif

#if SMZ_PART != 2 && SMZ_PART != 4 && SMZ_PART != 5 && SMZ_PART != 6
// div_by_count against the IEEE division, element-wise (inspection: tests pin the table-based quotients)
__global__ void __launch_bounds__(256) k_debug_div_by_count(const double *x, const int32_t *n, int count, int N, double *out) {
    for (int i = threadIdx.x; i < N; i += blockDim.x) smz_dyn_lds[i] = i > 0 ? 1.0 / (double)i : 0.0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
        out[i] = div_by_count(x[i], n[i], smz_dyn_lds);
}

// glibc's log / pow as the device restates them (smz_glibc_math.hpp), element-wise (inspection: tests compare with the host's libm)
__global__ void __launch_bounds__(256) k_debug_glibc_log_pow(const double *x, const double *y, int count, double *out_log,
                                                             double *out_pow) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        if (out_log) out_log[i] = smz_glibc_log(x[i]);
        if (out_pow) out_pow[i] = smz_glibc_pow(x[i], y[i]);
    }
}
#endif

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side: handle + C ABI
// ---------------------------------------------------------------------------------------------------------------
struct smz_handle {
    smz_config cfg;
    Params P;
    int K, N, Ppath;
    int maxa;  // template bucket
    bool root_ready, selected;
    uint64_t *d_seeds;
    double *d_pbc;
    double *d_pow;
    double pow_T;
    bool pow_valid;
    unsigned long long *d_stats;
    bool stats_on;
    uint32_t *d_mt_backup;
    int32_t *d_pos_backup;
    uint32_t *d_block_backup;
    bool has_backup;
    std::vector<void *> allocs;
    char last_kernel[96];      // the single-launch search instantiation launched last, as rocprofv3 prints it (smz_last_kernel)
};

// the last-error text is shared by the translation units this file is compiled into (SMZ_PART)
#if SMZ_PART == 0 || SMZ_PART == 1
thread_local char smz_g_err[512] = "";
#else
extern thread_local char smz_g_err[512];
#endif
#define g_err smz_g_err

namespace {

int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            snprintf(g_err, sizeof(g_err), "%s failed: %s", #expr, hipGetErrorString(e_)); \
            return SMZ_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

template <typename T>
int dev_alloc(smz_handle *h, T **out, size_t count) {
    void *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        return SMZ_ERR_NOMEM;
    }
    h->allocs.push_back(p);
    *out = (T *)p;
    return SMZ_OK;
}

inline dim3 tree_grid(int B) { return dim3((unsigned)((B + kWave - 1) / kWave)); }
inline dim3 wave_grid(const Params &P) { return dim3((unsigned)((P.B + P.tpw - 1) / P.tpw)); }
inline size_t tree_lds_bytes(const Params &P) {
    static const size_t pad = getenv("SMZ_DEBUG_LDS_PAD") ? (size_t)atoi(getenv("SMZ_DEBUG_LDS_PAD")) : 0;   // occupancy experiments
    return ((P.sims + 2 <= kPbcLdsMax) ? (size_t)(P.sims + 2) * sizeof(double) : 0) +
           (P.lds_stage ? (size_t)kWave * kRngStride * sizeof(uint32_t) : 0) + pad;
}
inline dim3 row_grid(int B) { return dim3((unsigned)((B + 255) / 256)); }
inline int group_lanes(int width) { int l = 1; while (l < width && l < kWave) l <<= 1; return l; }
inline dim3 group_grid(int B, int lpr) { const int per = 256 / lpr; return dim3((unsigned)((B + per - 1) / per)); }

int launch_check() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "kernel launch failed: %s", hipGetErrorString(e));
        return SMZ_ERR_HIP;
    }
    return SMZ_OK;
}

// dispatch on the per-lane scratch bucket (smallest MAXA >= A)
#define SMZ_DISPATCH(maxa, ...)               \
    switch (maxa) {                           \
        case 2: { constexpr int MA = 2; __VA_ARGS__; } break;   \
        case 4: { constexpr int MA = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int MA = 8; __VA_ARGS__; } break;   \
        case 16: { constexpr int MA = 16; __VA_ARGS__; } break; \
        default: { constexpr int MA = 32; __VA_ARGS__; } break; \
    }
// same, plus the exact-action-count specialisation for the two small buckets (aex: A == bucket)
#define SMZ_DISPATCH_AEX(maxa, aex, ...)                                                             \
    switch (maxa) {                                                                                  \
        case 2: if (aex) { constexpr int MA = 2; constexpr bool AEX = true; __VA_ARGS__; }            \
                else { constexpr int MA = 2; constexpr bool AEX = false; __VA_ARGS__; } break;        \
        case 4: if (aex) { constexpr int MA = 4; constexpr bool AEX = true; __VA_ARGS__; }            \
                else { constexpr int MA = 4; constexpr bool AEX = false; __VA_ARGS__; } break;        \
        case 8: { constexpr int MA = 8; constexpr bool AEX = false; __VA_ARGS__; } break;             \
        case 16: { constexpr int MA = 16; constexpr bool AEX = false; __VA_ARGS__; } break;           \
        default: { constexpr int MA = 32; constexpr bool AEX = false; __VA_ARGS__; } break;           \
    }
#define SMZ_DISPATCH2_AEX(maxa, k, aex, ...)                                          \
    if ((k) == 2) { constexpr int KS = 2; SMZ_DISPATCH_AEX(maxa, aex, __VA_ARGS__); }  \
    else { constexpr int KS = 0; SMZ_DISPATCH_AEX(maxa, aex, __VA_ARGS__); }
// ... and on the children-per-expansion specialisation (KS = 2: the static two-child code, 0: run-time count)
#define SMZ_DISPATCH2(maxa, k, ...)                                          \
    if ((k) == 2) { constexpr int KS = 2; SMZ_DISPATCH(maxa, __VA_ARGS__); }  \
    else { constexpr int KS = 0; SMZ_DISPATCH(maxa, __VA_ARGS__); }

}  // namespace

extern "C" {

#if SMZ_PART == 0 || SMZ_PART == 1
const char *smz_last_error(void) { return g_err; }
int smz_abi_version(void) { return SMZ_ABI_VERSION; }
int smz_build_features(void) { return 0; }
int smz_node_capacity(const smz_handle *h) { return h ? h->N : SMZ_ERR_INVALID; }
int smz_last_kernel(const smz_handle *h, char *buf, int cap) {
    if (!h || !buf || cap < 1) return fail(SMZ_ERR_INVALID, "smz_last_kernel: bad argument%s");
    snprintf(buf, (size_t)cap, "%s", h->last_kernel);
    return (int)strlen(h->last_kernel);
}

int smz_create(const smz_config *cfg, smz_handle **out) {
    if (!cfg || !out) return fail(SMZ_ERR_INVALID, "smz_create: null argument%s");
    *out = nullptr;
    // hyper-parameter checks of monte_carlo_tree_search.py:148-173
    if (cfg->pb_c_base < 1) return fail(SMZ_ERR_INVALID, "pb_c_base must be an int >= 1%s");
    if (!(cfg->pb_c_init >= 0)) return fail(SMZ_ERR_INVALID, "pb_c_init must be a float >= 0%s");
    if (!(cfg->discount >= 0)) return fail(SMZ_ERR_INVALID, "discount must be >= 0%s");
    if (!(cfg->root_dirichlet_alpha >= 0 && cfg->root_dirichlet_alpha <= 1))
        return fail(SMZ_ERR_INVALID, "root_dirichlet_alpha must be in [0, 1]%s");
    if (!(cfg->root_exploration_fraction >= 0 && cfg->root_exploration_fraction <= 1))
        return fail(SMZ_ERR_INVALID, "root_exploration_fraction must be in [0, 1]%s");
    if (cfg->max_action_sample < 1) return fail(SMZ_ERR_INVALID, "maxium_action_sample must be an int >= 1%s");
    if (cfg->num_simulations < 0) return fail(SMZ_ERR_INVALID, "num_simulations must be an int >= 0%s");
    if (cfg->num_trees < 1) return fail(SMZ_ERR_INVALID, "num_trees must be >= 1%s");
    if (cfg->num_actions < 1 || cfg->num_actions > SMZ_MAX_ACTIONS)
        return fail(SMZ_ERR_INVALID, "num_actions must be in [1, SMZ_MAX_ACTIONS]%s");
    if (cfg->hidden_size < 0) return fail(SMZ_ERR_INVALID, "hidden_size must be >= 0%s");
    if (cfg->num_simulations > 32000) return fail(SMZ_ERR_INVALID, "num_simulations above 32000 is not supported%s");
    if (cfg->rng_mode != SMZ_RNG_MT19937_NUMPY && cfg->rng_mode != SMZ_RNG_PHILOX)
        return fail(SMZ_ERR_INVALID, "rng_mode must be SMZ_RNG_MT19937_NUMPY or SMZ_RNG_PHILOX%s");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(SMZ_ERR_INVALID, "device ordinal out of range%s");
    DeviceGuard guard(cfg->device);

    smz_handle *h = new (std::nothrow) smz_handle();
    if (!h) return fail(SMZ_ERR_NOMEM, "host allocation failed%s");
    h->cfg = *cfg;
    const int B = cfg->num_trees, A = cfg->num_actions, S = cfg->hidden_size, sims = cfg->num_simulations;
    h->K = cfg->max_action_sample < A ? cfg->max_action_sample : A;
    h->N = 1 + A + sims * h->K;
    h->Ppath = sims + 2;
    h->maxa = A <= 2 ? 2 : A <= 4 ? 4 : A <= 8 ? 8 : A <= 16 ? 16 : 32;
    h->root_ready = h->selected = false;
    h->pow_valid = false;
    h->pow_T = 0.0;
    h->stats_on = false;

    Params &P = h->P;
    memset(&P, 0, sizeof(P));
    P.B = B; P.A = A; P.K = h->K; P.S = S; P.N = h->N; P.P = h->Ppath; P.sims = sims;
    P.dbg = getenv("SMZ_DEBUG_SKIP") ? atoi(getenv("SMZ_DEBUG_SKIP")) : 0;
    {   // trees per wavefront: spread small batches over the whole chip (>= 1 wave per SIMD before packing lanes)
        int tpw = kWave;
        while (tpw > 1 && (B + tpw - 1) / tpw < 1024) tpw >>= 1;
        if (const char *e = getenv("SMZ_TREES_PER_WAVE")) {
            const int v = atoi(e);
            if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) tpw = v;
        }
        P.tpw = tpw;
        // the staged words live in a 16.6 KB LDS tile; SMZ_LDS_STAGE=0 twists ahead only and draws from L1 (measured
        // 7 % slower at 1 M trees: the large-batch regime is bound by cache-line transactions, not by occupancy)
        P.lds_stage = 1;
        if (const char *e = getenv("SMZ_LDS_STAGE")) P.lds_stage = atoi(e) ? 1 : 0;
    }
    P.disc32 = (float)cfg->discount;
    P.keep32 = (float)(1.0 - cfg->root_exploration_fraction);
    P.frac = cfg->root_exploration_fraction;
    P.alpha = cfg->root_dirichlet_alpha;
    const size_t BN = (size_t)B * h->N;
    // child-block geometry (16-word = 64-byte granules so that a block never straddles more lines than it must)
    P.rp_off = (5 * A + 1) & ~1;
    P.rb_words = ((P.rp_off + 2 * A) + 15) & ~15;
    P.eb_words = ((6 * h->K) + 15) & ~15;
    P.tree_words = (int64_t)P.rb_words + (int64_t)sims * P.eb_words;
    int rc = SMZ_OK;
    auto A_ = [&](int r) { if (rc == SMZ_OK) rc = r; };
    A_(dev_alloc(h, &P.nodes, (size_t)B * (size_t)P.tree_words));
    P.hs = (S + 15) & ~15;        // hidden rows start on 64-byte lines (S = 31 -> 128-byte rows: two whole lines)
    A_(dev_alloc(h, &P.hidden, BN * (size_t)P.hs));
    A_(dev_alloc(h, &P.hdr, (size_t)B));
    A_(dev_alloc(h, &P.path, (size_t)B * h->Ppath));
    A_(dev_alloc(h, &P.mt, (size_t)B * kMtN));
    A_(dev_alloc(h, &P.rng_pos, (size_t)B));
    A_(dev_alloc(h, &h->d_seeds, (size_t)B));
    A_(dev_alloc(h, &h->d_pbc, (size_t)sims + 2));
    A_(dev_alloc(h, &h->d_pow, (size_t)sims + 1));
    A_(dev_alloc(h, &h->d_stats, (size_t)16));
    A_(dev_alloc(h, &h->d_mt_backup, (size_t)B * kMtN));
    A_(dev_alloc(h, &h->d_pos_backup, (size_t)B));
    A_(dev_alloc(h, &h->d_block_backup, (size_t)B));
    {
        uint32_t *key = nullptr;
        A_(dev_alloc(h, &P.rng_block, (size_t)B));
        A_(dev_alloc(h, &key, (size_t)2 * B));
        P.rng_key = key;
    }
    P.philox = cfg->rng_mode == SMZ_RNG_PHILOX ? 1 : 0;
    h->has_backup = false;
    h->last_kernel[0] = 0;
    if (rc != SMZ_OK) { smz_destroy(h); return rc; }
    P.pbc_sqrt = h->d_pbc;
    P.pow_table = nullptr;
    P.stats = nullptr;
    // defined contents before first use
    hipError_t e = hipMemset(P.nodes, 0, (size_t)B * (size_t)P.tree_words * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(P.hdr, 0, (size_t)B * sizeof(TreeHdr));
    if (e == hipSuccess) e = hipMemset(h->d_stats, 0, 16 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(P.rng_pos, 0, (size_t)B * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(P.mt, 0, (size_t)B * kMtN * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(P.rng_block, 0, (size_t)B * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(const_cast<uint32_t *>(P.rng_key), 0, (size_t)2 * B * sizeof(uint32_t));
    if (e != hipSuccess) { smz_destroy(h); return fail(SMZ_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e)); }
    std::vector<double> tab((size_t)sims + 2);
    for (int n = 0; n < sims + 2; n++)
        tab[n] = log(((double)n + (double)cfg->pb_c_base + 1.0) / (double)cfg->pb_c_base) + cfg->pb_c_init;
    rc = smz_set_pb_c_table(h, tab.data(), sims + 2);
    if (rc != SMZ_OK) { smz_destroy(h); return rc; }
    // default streams: numpy seed(i) for tree i
    std::vector<uint64_t> seeds((size_t)B);
    for (int i = 0; i < B; i++) seeds[i] = (uint64_t)i;
    rc = smz_seed(h, seeds.data(), nullptr);
    if (rc == SMZ_OK) { e = hipDeviceSynchronize(); if (e != hipSuccess) rc = fail(SMZ_ERR_HIP, "sync failed: %s", hipGetErrorString(e)); }
    if (rc != SMZ_OK) { smz_destroy(h); return rc; }
    *out = h;
    return SMZ_OK;
}

int smz_destroy(smz_handle *h) {
    if (!h) return SMZ_OK;
    DeviceGuard guard(h->cfg.device);
    (void)hipDeviceSynchronize();
    for (void *p : h->allocs) (void)hipFree(p);
    delete h;
    return SMZ_OK;
}

int smz_set_pb_c_table(smz_handle *h, const double *host_table, int n) {
    if (!h || !host_table) return fail(SMZ_ERR_INVALID, "smz_set_pb_c_table: null argument%s");
    const int need = h->cfg.num_simulations + 2;
    if (n < need) return fail(SMZ_ERR_INVALID, "smz_set_pb_c_table: table shorter than num_simulations + 2%s");
    DeviceGuard guard(h->cfg.device);
    std::vector<double> t((size_t)need);
    for (int i = 0; i < need; i++) t[i] = sqrt((double)i) * host_table[i];  // np.sqrt(Np) * pb_c (mcts:237)
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->d_pbc, t.data(), (size_t)need * sizeof(double), hipMemcpyHostToDevice));
    return SMZ_OK;
}

int smz_seed(smz_handle *h, const uint64_t *host_seeds, smz_stream stream) {
    if (!h || !host_seeds) return fail(SMZ_ERR_INVALID, "smz_seed: null argument%s");
    DeviceGuard guard(h->cfg.device);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipStreamSynchronize(s));  // d_seeds may still be read by an earlier smz_seed
    HIP_TRY(hipMemcpy(h->d_seeds, host_seeds, (size_t)h->cfg.num_trees * sizeof(uint64_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_seed, tree_grid(h->P.B), dim3(kWave), 0, s, h->P, h->d_seeds);
    return launch_check();
}

int smz_set_rng_state(smz_handle *h, int tree, const uint32_t *host_key, int pos) {
    if (!h || !host_key || tree < 0 || tree >= h->cfg.num_trees || pos < 0 || pos > kMtN)
        return fail(SMZ_ERR_INVALID, "smz_set_rng_state: bad argument%s");
    if (h->P.philox) return fail(SMZ_ERR_STATE, "smz_set_rng_state: numpy (MT19937) states belong to SMZ_RNG_MT19937_NUMPY handles%s");
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->P.mt + (size_t)tree * kMtN, host_key, kMtN * sizeof(uint32_t), hipMemcpyHostToDevice));
    // numpy block form (all 624 words of the current block, pos consumed) -> incremental form:
    // words pos..623 are already twisted and are handed out as they are.
    const int32_t packed = (pos == kMtN) ? 0 : (((kMtN - pos) << 16) | pos);
    HIP_TRY(hipMemcpy(h->P.rng_pos + tree, &packed, sizeof(int32_t), hipMemcpyHostToDevice));
    return SMZ_OK;
}

int smz_get_rng_state(smz_handle *h, int tree, uint32_t *host_key, int *pos) {
    if (!h || !host_key || !pos || tree < 0 || tree >= h->cfg.num_trees)
        return fail(SMZ_ERR_INVALID, "smz_get_rng_state: bad argument%s");
    if (h->P.philox) return fail(SMZ_ERR_STATE, "smz_get_rng_state: numpy (MT19937) states belong to SMZ_RNG_MT19937_NUMPY handles%s");
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    int32_t packed = 0;
    HIP_TRY(hipMemcpy(host_key, h->P.mt + (size_t)tree * kMtN, kMtN * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&packed, h->P.rng_pos + tree, sizeof(int32_t), hipMemcpyDeviceToHost));
    const int idx = packed & 0xffff, ready = packed >> 16;
    // Device form: words [idx, idx+ready) (cyclic) are twisted ahead of consumption, everything already consumed in
    // this pass is twisted, the rest still holds the previous block.  numpy's form wants one complete block + pos.
    auto twist_at = [&](int i) {
        const int i1 = (i + 1 == kMtN) ? 0 : i + 1;
        int im = i + kMtM;
        if (im >= kMtN) im -= kMtN;
        const uint32_t t = (host_key[i] & 0x80000000u) | (host_key[i1] & 0x7fffffffu);
        host_key[i] = host_key[im] ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
    };
    if (idx + ready <= kMtN) {
        if (idx == 0 && ready == 0) { *pos = kMtN; return SMZ_OK; }   // block boundary: numpy regenerates next
        for (int i = idx + ready; i < kMtN; i++) twist_at(i);          // finish the in-place pass
        *pos = idx;
        return SMZ_OK;
    }
    // The twist-ahead window wrapped: words [0, w) already hold NEXT-block values.  Recover the current-block words
    // they replaced by inverting the twist (new[k] ^ cur[k+397] gives the msb of cur[k] and the low 31 bits of
    // cur[k+1]; the low bits of cur[0] are never read again by the generator and are left zero).
    const int w = idx + ready - kMtN;
    if (w + kMtM > kMtN) return fail(SMZ_ERR_STATE, "smz_get_rng_state: twist-ahead window too long%s");
    std::vector<uint32_t> t((size_t)w);
    for (int k = 0; k < w; k++) {
        uint32_t y = host_key[k] ^ host_key[k + kMtM];
        uint32_t low = 0;
        if (y & 0x80000000u) { y ^= 0x9908b0dfu; low = 1u; }
        t[k] = (y << 1) | low;                                       // (cur[k] & UPPER) | (cur[k+1] & LOWER)
    }
    for (int k = 0; k < w; k++)
        host_key[k] = (t[k] & 0x80000000u) | (k > 0 ? (t[k - 1] & 0x7fffffffu) : 0u);
    *pos = idx;
    return SMZ_OK;
}

int smz_rng_snapshot(smz_handle *h, smz_stream stream) {
    if (!h) return fail(SMZ_ERR_INVALID, "smz_rng_snapshot: null handle%s");
    DeviceGuard guard(h->cfg.device);
    const size_t B = (size_t)h->cfg.num_trees;
    HIP_TRY(hipMemcpyAsync(h->d_mt_backup, h->P.mt, B * kMtN * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(h->d_pos_backup, h->P.rng_pos, B * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(h->d_block_backup, h->P.rng_block, B * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    h->has_backup = true;
    return SMZ_OK;
}

int smz_rng_restore(smz_handle *h, smz_stream stream) {
    if (!h) return fail(SMZ_ERR_INVALID, "smz_rng_restore: null handle%s");
    if (!h->has_backup) return fail(SMZ_ERR_STATE, "smz_rng_restore without smz_rng_snapshot%s");
    DeviceGuard guard(h->cfg.device);
    const size_t B = (size_t)h->cfg.num_trees;
    HIP_TRY(hipMemcpyAsync(h->P.mt, h->d_mt_backup, B * kMtN * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(h->P.rng_pos, h->d_pos_backup, B * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(h->P.rng_block, h->d_block_backup, B * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SMZ_OK;
}

int smz_root_init(smz_handle *h, const float *hidden_dev, const float *policy_dev, const double *noise_override_dev,
                  int train, smz_stream stream) {
    if (!h || !policy_dev) return fail(SMZ_ERR_INVALID, "smz_root_init: null argument%s");
    if (h->P.S > 0 && !hidden_dev) return fail(SMZ_ERR_INVALID, "smz_root_init: hidden_dev is required when hidden_size > 0%s");
    if (train && h->cfg.num_simulations > 0 && !(h->cfg.root_dirichlet_alpha > 0))
        return fail(SMZ_ERR_INVALID, "root_dirichlet_alpha must be > 0 to draw noise (numpy raises ValueError)%s");
    DeviceGuard guard(h->cfg.device);
    SMZ_DISPATCH(h->maxa, hipLaunchKernelGGL((k_root_init<MA>), wave_grid(h->P), dim3(kWave), tree_lds_bytes(h->P), (hipStream_t)stream,
                                             h->P, hidden_dev, policy_dev, noise_override_dev, train));
    h->root_ready = true;
    h->selected = false;
    return launch_check();
}

int smz_select(smz_handle *h, float *parent_hidden_dev, int32_t *last_action_dev, uint8_t *branch_dev,
               float *mlp_input_dev, smz_stream stream) {
    if (!h) return fail(SMZ_ERR_INVALID, "smz_select: null handle%s");
    if (!h->root_ready) return fail(SMZ_ERR_STATE, "smz_select before smz_root_init%s");
    DeviceGuard guard(h->cfg.device);
    SMZ_DISPATCH2_AEX(h->maxa, h->K, h->P.A == h->maxa && !h->P.philox, hipLaunchKernelGGL((k_select<MA, KS, AEX>), wave_grid(h->P), dim3(kWave), tree_lds_bytes(h->P), (hipStream_t)stream, h->P,
                                             parent_hidden_dev, last_action_dev, branch_dev, mlp_input_dev));
    h->selected = true;
    return launch_check();
}

int smz_expand_backup(smz_handle *h, const float *hidden_dev, const float *reward_dev, const float *policy_dev,
                      const float *value_dev, smz_stream stream) {
    if (!h || !policy_dev || !value_dev) return fail(SMZ_ERR_INVALID, "smz_expand_backup: null argument%s");
    if (!h->selected) return fail(SMZ_ERR_STATE, "smz_expand_backup without a preceding smz_select%s");
    DeviceGuard guard(h->cfg.device);
    SMZ_DISPATCH2_AEX(h->maxa, h->K, h->P.A == h->maxa && !h->P.philox, hipLaunchKernelGGL((k_expand_backup<MA, KS, false, AEX>), wave_grid(h->P), dim3(kWave), tree_lds_bytes(h->P),
                                             (hipStream_t)stream, h->P, hidden_dev, reward_dev, policy_dev, value_dev,
                                             (float *)nullptr, (int32_t *)nullptr, (uint8_t *)nullptr, (float *)nullptr));
    h->selected = false;
    return launch_check();
}

#endif

#if SMZ_PART == 0 || SMZ_PART == 3
int smz_expand_backup_select(smz_handle *h, const float *hidden_dev, const float *reward_dev, const float *policy_dev,
                             const float *value_dev, float *parent_hidden_dev, int32_t *last_action_dev,
                             uint8_t *branch_dev, float *mlp_input_dev, smz_stream stream) {
    if (!h || !policy_dev || !value_dev) return fail(SMZ_ERR_INVALID, "smz_expand_backup_select: null argument%s");
    if (!h->selected) return fail(SMZ_ERR_STATE, "smz_expand_backup_select without a preceding smz_select%s");
    DeviceGuard guard(h->cfg.device);
    if (h->P.philox && h->P.A == h->maxa && h->maxa <= 4) {       // Philox handles, exact action count: the specialised kernel too
#define SMZ_EBS_PHX(MA, KS)                                                                                                  \
        hipLaunchKernelGGL((k_expand_backup<MA, KS, true, true, true>), wave_grid(h->P), dim3(kWave), tree_lds_bytes(h->P),   \
                           (hipStream_t)stream, h->P, hidden_dev, reward_dev, policy_dev, value_dev, parent_hidden_dev,     \
                           last_action_dev, branch_dev, mlp_input_dev)
        if (h->maxa == 2) { if (h->K == 2) SMZ_EBS_PHX(2, 2); else SMZ_EBS_PHX(2, 0); }
        else { if (h->K == 2) SMZ_EBS_PHX(4, 2); else SMZ_EBS_PHX(4, 0); }
#undef SMZ_EBS_PHX
        return launch_check();
    }
    SMZ_DISPATCH2_AEX(h->maxa, h->K, h->P.A == h->maxa && !h->P.philox, hipLaunchKernelGGL((k_expand_backup<MA, KS, true, AEX>), wave_grid(h->P), dim3(kWave), tree_lds_bytes(h->P),
                                             (hipStream_t)stream, h->P, hidden_dev, reward_dev, policy_dev, value_dev,
                                             parent_hidden_dev, last_action_dev, branch_dev, mlp_input_dev));
    return launch_check();
}
#endif

#if SMZ_PART == 0 || SMZ_PART == 1

#endif  // SMZ_PART != 2

#if SMZ_PART == 0 || SMZ_PART == 2 || SMZ_PART == 4 || SMZ_PART == 6
// The instantiations are split over three translation units (compile time): SMZ_PART 2 holds the action buckets 2 and 4 and
// the ABI entry points, SMZ_PART 4 the buckets 8, 16 and 32, SMZ_PART 6 the instantiations with the trees in LDS.
#if SMZ_PART == 4
#define SMZ_SEARCH_LAUNCH smz_internal_search_launch_wide
#define SMZ_SEARCH_DISPATCH(maxa, ...)                                        \
    switch (maxa) {                                                           \
        case 8: { constexpr int MA = 8; __VA_ARGS__; } break;                 \
        case 16: { constexpr int MA = 16; __VA_ARGS__; } break;               \
        default: { constexpr int MA = 32; __VA_ARGS__; } break;               \
    }
#elif SMZ_PART == 2
#define SMZ_SEARCH_LAUNCH smz_internal_search_launch_narrow
#define SMZ_SEARCH_DISPATCH(maxa, ...)                                        \
    switch (maxa) {                                                           \
        case 2: { constexpr int MA = 2; __VA_ARGS__; } break;                 \
        default: { constexpr int MA = 4; __VA_ARGS__; } break;                \
    }
#else
#define SMZ_SEARCH_LAUNCH smz_internal_search_launch_narrow
#define SMZ_SEARCH_DISPATCH(maxa, ...) SMZ_DISPATCH(maxa, __VA_ARGS__)
#endif
#define SMZ_SEARCH_DISPATCH2(maxa, k, ...)                                          \
    if ((k) == 2) { constexpr int KS = 2; SMZ_SEARCH_DISPATCH(maxa, __VA_ARGS__); }  \
    else { constexpr int KS = 0; SMZ_SEARCH_DISPATCH(maxa, __VA_ARGS__); }
struct SearchActArgs {       // ActOut (+ the fused env step) across the translation-unit boundary (plain data)
    double temperature;
    int32_t *action;
    double *policy, *child_visits;
    float *root_value;
    EnvStep env;             // env.state == nullptr: no env step in the launch
};
}  // extern "C" (internal C++ linkage for the two launchers)
int smz_internal_search_launch_narrow(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev,
                                      int train, SearchActArgs a, const double *pow_table_host, smz_stream stream);
int smz_internal_search_launch_wide(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev,
                                    int train, SearchActArgs a, const double *pow_table_host, smz_stream stream);
// SMZ_PART 6: the instantiations that keep the workgroup's trees in LDS (P, geometry and LDS size come from the caller)
int smz_internal_search_launch_tlds(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev,
                                    int train, SearchActArgs a, Params P, int kWaves, int blocks, size_t lds_t, smz_stream stream);
#if SMZ_PART == 0 || SMZ_PART == 6
// k_search_mlp<MA, 2, 1, false, true, true, PHX, TLDS = true>: masked (smz_set_active) | Philox handles.  The caller
// (SMZ_SEARCH_LAUNCH of SMZ_PART 2) has validated everything and computed the geometry.
int smz_internal_search_launch_tlds(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev,
                                    int train, SearchActArgs a, Params P, int kWaves, int blocks, size_t lds_t, smz_stream stream) {
    const ActOut act = {a.temperature, a.action, a.policy, a.child_visits, a.root_value};
#define SMZ_LAUNCH_TLDS(MA, MSK, PHX)                                                                                  \
    {                                                                                                                  \
        static size_t granted_dev[64] = {};                                                                            \
        size_t &granted = granted_dev[h->cfg.device & 63];                                                             \
        if (lds_t > granted) {                                                                                         \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_mlp<MA, 2, 1, false, true, MSK, PHX, true>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t) != hipSuccess)             \
                return fail(SMZ_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed%s");                            \
            granted = lds_t;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((k_search_mlp<MA, 2, 1, false, true, MSK, PHX, true>), dim3(blocks), dim3(kWaves * kWave),  \
                           lds_t, (hipStream_t)stream, P, *desc, weights_dev, obs_dev, train, act, a.env);             \
        snprintf(h->last_kernel, sizeof(h->last_kernel), "k_search_mlp<%d, 2, 1, false, true, %s, %s, true>", MA,      \
                 MSK ? "true" : "false", PHX ? "true" : "false");                                                      \
    }
    if (h->maxa == 2) {
        if (P.philox) SMZ_LAUNCH_TLDS(2, true, true) else SMZ_LAUNCH_TLDS(2, true, false)
    } else {
        if (P.philox) SMZ_LAUNCH_TLDS(4, true, true) else SMZ_LAUNCH_TLDS(4, true, false)
    }
#undef SMZ_LAUNCH_TLDS
    return SMZ_OK;
}
#endif
#if SMZ_PART != 6
int SMZ_SEARCH_LAUNCH(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev, int train,
                      SearchActArgs a, const double *pow_table_host, smz_stream stream) {
    const ActOut act = {a.temperature, a.action, a.policy, a.child_visits, a.root_value};
#if SMZ_PART == 2
    if (h && h->maxa > 4) return smz_internal_search_launch_wide(h, desc, weights_dev, obs_dev, train, a, pow_table_host, stream);
#endif
    if (!h || !desc || !weights_dev || !obs_dev) return fail(SMZ_ERR_INVALID, "smz_search_mlp: null argument%s");
    smz_mlp_desc t = *desc;
    if (smz_mlp_layout(&t) != SMZ_OK || t.total_floats != desc->total_floats)
        return fail(SMZ_ERR_INVALID, "smz_search_mlp: descriptor does not describe an LDS-resident network%s");
    if (desc->A != h->P.A || desc->S != h->P.S)
        return fail(SMZ_ERR_INVALID, "smz_search_mlp: network dimensions differ from the handle's%s");
    if (train && h->cfg.num_simulations > 0 && !(h->cfg.root_dirichlet_alpha > 0))
        return fail(SMZ_ERR_INVALID, "root_dirichlet_alpha must be > 0 to draw noise (numpy raises ValueError)%s");
    DeviceGuard guard(h->cfg.device);
    int kWaves = 8;
    if (const char *e = getenv("SMZ_SEARCH_WAVES")) { const int v = atoi(e); if ((v == 1 || v == 2 || v == 4 || v == 8 || v == 12 || v == 16) && v * kWave <= SMZ_SEARCH_THREADS) kWaves = v; }
    Params P = h->P;
    if (act.action && pow_table_host && act.temperature >= 0.3) {       // as smz_act: the power table of this temperature
        if (!h->pow_valid || h->pow_T != act.temperature) {
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            HIP_TRY(hipMemcpy(h->d_pow, pow_table_host, ((size_t)h->cfg.num_simulations + 1) * sizeof(double),
                              hipMemcpyHostToDevice));
            h->pow_T = act.temperature;
            h->pow_valid = true;
        }
        P.pow_table = h->d_pow;
    }
    // trees per wave: the smallest power of two that covers B with 256 workgroups of 8 waves, capped at 2 -- beyond
    // 4096 trees the grid simply has more workgroups than CUs and they run one after another (each re-stages the
    // weights, ~1 % of its run time): per-wave LDS buffers stay small and the specialised instantiation applies.
    int tpw = 1;
    while (tpw < kFastTpw && (size_t)256 * kWaves * tpw < (size_t)P.B) tpw <<= 1;
    if (const char *e = getenv("SMZ_SEARCH_TPW")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) tpw = v; }
    P.tpw = tpw;
    const MegaLds ml = mega_lds(*desc, P, tpw, false, kWaves);
    const size_t lds = ((size_t)ml.wave_off + (size_t)kWaves * ml.per_wave) * sizeof(float);
    if (lds > 160 * 1024) return fail(SMZ_ERR_TOO_LARGE, "smz_search_mlp: working set exceeds the 160 KB LDS of a CU%s");
    const int blocks = (P.B + kWaves * tpw - 1) / (kWaves * tpw);
#define SMZ_LAUNCH_SEARCH(UU, INSTR, AEX, MSK, PHX)                                                                    \
    SMZ_SEARCH_DISPATCH2(h->maxa, h->K, {                                                                              \
        static size_t granted_dev[64] = {}; /* per instantiation and device: the opt-in is a host-side call */           \
        size_t &granted = granted_dev[h->cfg.device & 63];                                                             \
        if (lds > granted) {                                                                                           \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_mlp<MA, KS, UU, INSTR, AEX, MSK, PHX>),    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)               \
                return fail(SMZ_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed%s");                            \
            granted = lds;                                                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL((k_search_mlp<MA, KS, UU, INSTR, AEX, MSK, PHX>), dim3(blocks), dim3(kWaves * kWave), lds,  \
                           (hipStream_t)stream, P, *desc, weights_dev, obs_dev, train, act, a.env);                    \
        snprintf(h->last_kernel, sizeof(h->last_kernel), "k_search_mlp<%d, %d, %d, %s, %s, %s, %s, false>", MA, KS, UU, \
                 INSTR ? "true" : "false", AEX ? "true" : "false", MSK ? "true" : "false", PHX ? "true" : "false");     \
    })
    // smz_mlp_layout only accepts OP == 64 (one output neuron per lane): U = 1.  The instrumented instantiation runs
    // when level statistics are enabled (smz_enable_stats) or a SMZ_DEBUG_SKIP switch is set.
    // (the specialised instantiation is the parity-mode path: a Philox handle runs the generic one)
    const bool fast = P.A == h->maxa && tpw == kFastTpw && desc->S == kFastS && desc->H == kFastH && desc->L == kFastL;
#if SMZ_PART != 4
    {   // LDS-resident trees (k_search_mlp<..., TLDS>, compiled in their own translation unit, SMZ_PART 6): the specialised
        // instantiations -- plain, masked (smz_set_active) and Philox -- when the workgroup's trees fit next to the compact
        // weight image (checkpoint-421 shape, 2 actions: up to ~53 simulations); SMZ_SEARCH_TLDS=0 keeps the trees in global
        // memory (A/B runs)
        const MegaLds mt = mega_lds(*desc, P, tpw, true, kWaves);
        const size_t lds_t = ((size_t)mt.trees_off + (size_t)kWaves * tpw * mt.tree_words) * sizeof(float);
        const char *te = getenv("SMZ_SEARCH_TLDS");
        const bool tlds = fast && !(P.stats || P.dbg) && h->K == 2 && h->maxa <= 4 && lds_t <= 160 * 1024 && !(te && atoi(te) == 0) &&
                          P.sims <= 126;      // (the block-parallel selection's 7-bit block indices; reachable with 4-wave workgroups only)
#ifdef SMZ_BPS_PROBE
        if (tlds) P.stats = h->d_stats;              // (the production kernel carries no level statistics: only the probe's stamps land there)
#endif
        if (tlds && (P.active || P.philox)) {        // masked / Philox handles: SMZ_PART 6
            const int rc = smz_internal_search_launch_tlds(h, desc, weights_dev, obs_dev, train, a, P, kWaves, blocks, lds_t, stream);
            if (rc != SMZ_OK) return rc;
            h->root_ready = true;
            h->selected = false;
            return launch_check();
        }
#if SMZ_PART == 0 || SMZ_PART == 2
        if (tlds) {                                  // the plain instantiation (the headline's) stays in this translation unit
#define SMZ_LAUNCH_TLDS(MA)                                                                                            \
            {                                                                                                          \
                static size_t granted_dev[64] = {};                                                                    \
                size_t &granted = granted_dev[h->cfg.device & 63];                                                     \
                if (lds_t > granted) {                                                                                 \
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_mlp<MA, 2, 1, false, true, false, false, true>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t) != hipSuccess)     \
                        return fail(SMZ_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed%s");                    \
                    granted = lds_t;                                                                                   \
                }                                                                                                      \
                hipLaunchKernelGGL((k_search_mlp<MA, 2, 1, false, true, false, false, true>), dim3(blocks),            \
                                   dim3(kWaves * kWave), lds_t, (hipStream_t)stream, P, *desc, weights_dev, obs_dev, train, \
                                   act, a.env);                                                                        \
                snprintf(h->last_kernel, sizeof(h->last_kernel), "k_search_mlp<%d, 2, 1, false, true, false, false, true>", MA); \
            }
            if (h->maxa == 2) SMZ_LAUNCH_TLDS(2) else SMZ_LAUNCH_TLDS(4)
#undef SMZ_LAUNCH_TLDS
            h->root_ready = true;
            h->selected = false;
            return launch_check();
        }
#endif
    }
    if ((P.stats || P.dbg) && fast && !P.philox && (P.dbg & 32) && h->maxa == 2 && h->K == 2) {
        // phase stamps of the specialised instantiation itself (SMZ_DEBUG_SKIP=48), for the headline geometry only
        constexpr int MA = 2, KS = 2;
        hipLaunchKernelGGL((k_search_mlp<MA, KS, 1, true, true>), dim3(blocks), dim3(kWaves * kWave), lds, (hipStream_t)stream,
                           P, *desc, weights_dev, obs_dev, train, act, a.env);
        snprintf(h->last_kernel, sizeof(h->last_kernel), "k_search_mlp<2, 2, 1, true, true, true, false, false>");
    } else
#endif
#if (SMZ_PART == 0 || SMZ_PART == 2) && SMZ_KS4
    // Four children per expansion on a four-action tree (SURVEY 8d(3)'s stress shape; round 6): the child count is a compile-time
    // constant (KS = 4: load_kids_static<4>, unrolled scoring, the paired descent on every level) instead of the run-time-K
    // instantiation (KS = 0) -- 347 -> 409 M simulations/s on the K = 4 stress workload, profiles/r06_c_ab_noks4.txt.
    // Plain, masked (smz_set_active) and Philox handles.
    if (fast && !(P.stats || P.dbg) && h->maxa == 4 && h->K == 4) {
#define SMZ_LAUNCH_KS4(MSK, PHX)                                                                                       \
        {                                                                                                              \
            static size_t granted_dev[64] = {};                                                                        \
            size_t &granted = granted_dev[h->cfg.device & 63];                                                         \
            if (lds > granted) {                                                                                       \
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_mlp<4, 4, 1, false, true, MSK, PHX>),  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)           \
                    return fail(SMZ_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed%s");                        \
                granted = lds;                                                                                         \
            }                                                                                                          \
            hipLaunchKernelGGL((k_search_mlp<4, 4, 1, false, true, MSK, PHX>), dim3(blocks), dim3(kWaves * kWave), lds, \
                               (hipStream_t)stream, P, *desc, weights_dev, obs_dev, train, act, a.env);                \
            snprintf(h->last_kernel, sizeof(h->last_kernel), "k_search_mlp<4, 4, 1, false, true, %s, %s, false>",      \
                     MSK ? "true" : "false", PHX ? "true" : "false");                                                  \
        }
        if (P.philox) SMZ_LAUNCH_KS4(true, true)
        else if (P.active) SMZ_LAUNCH_KS4(true, false)
        else SMZ_LAUNCH_KS4(false, false)
#undef SMZ_LAUNCH_KS4
    } else
#endif
    if (P.stats || P.dbg) { SMZ_LAUNCH_SEARCH(1, true, false, true, false); }
    else if (fast && P.philox) { SMZ_LAUNCH_SEARCH(1, false, true, true, true); }
    else if (fast && !P.active) { SMZ_LAUNCH_SEARCH(1, false, true, false, false); }
    else if (fast) { SMZ_LAUNCH_SEARCH(1, false, true, true, false); }
    else { SMZ_LAUNCH_SEARCH(1, false, false, true, false); }
#undef SMZ_LAUNCH_SEARCH
    h->root_ready = true;
    h->selected = false;
    return launch_check();
}
#endif  // SMZ_PART != 6

extern "C" {
#if SMZ_PART != 4 && SMZ_PART != 6
int smz_search_mlp(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev, int train,
                   smz_stream stream) {
    return smz_internal_search_launch_narrow(h, desc, weights_dev, obs_dev, train,
                                             SearchActArgs{0.0, nullptr, nullptr, nullptr, nullptr, EnvStep{}}, nullptr, stream);
}

int smz_search_mlp_act(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev, int train,
                       double temperature, const double *pow_table_host, int32_t *action_dev, double *policy_dev,
                       double *child_visits_dev, float *root_value_dev, smz_stream stream) {
    if (!action_dev || !policy_dev || !child_visits_dev) return fail(SMZ_ERR_INVALID, "smz_search_mlp_act: null output%s");
    return smz_internal_search_launch_narrow(h, desc, weights_dev, obs_dev, train,
                                             SearchActArgs{temperature, action_dev, policy_dev, child_visits_dev, root_value_dev,
                                                           EnvStep{}},
                                             pow_table_host, stream);
}

int smz_search_mlp_act_cartpole(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, int train,
                                double temperature, const double *pow_table_host, int32_t *action_dev, double *policy_dev,
                                double *child_visits_dev, float *root_value_dev, const smz_cartpole_env *env,
                                smz_stream stream) {
    if (!action_dev || !policy_dev || !child_visits_dev || !root_value_dev || !env || !env->state_dev || !env->obs_dev)
        return fail(SMZ_ERR_INVALID, "smz_search_mlp_act_cartpole: null argument%s");
    if (!h || !desc || h->P.A != 2 || desc->obs != 4)
        return fail(SMZ_ERR_INVALID, "smz_search_mlp_act_cartpole: the built-in env has 4 observations and 2 actions%s");
    const smz_episode_ctl *c = env->ctl;
    if (c && (!c->step_count_dev || c->on_end < 0 || c->on_end > 2 || (c->on_end == 2 && !c->episode_dev) ||
              (c->on_end == 1 && !c->active_dev)))
        return fail(SMZ_ERR_INVALID, "smz_search_mlp_act_cartpole: bad smz_episode_ctl (as smz_cartpole_step_ctl)%s");
    if ((c ? c->active_dev : nullptr) != h->P.active)
        return fail(SMZ_ERR_INVALID, "smz_search_mlp_act_cartpole: ctl->active_dev must be the array given to smz_set_active%s");
    if (env->traj_dev && (env->t < 0 || env->t >= env->T))
        return fail(SMZ_ERR_INVALID, "smz_search_mlp_act_cartpole: bad trajectory argument%s");
    if (h->maxa > 4) return fail(SMZ_ERR_INVALID, "smz_search_mlp_act_cartpole: 2 actions%s");
    const EnvStep E = {env->state_dev, env->obs_dev, env->reward_dev, env->flag_dev, c ? c->step_count_dev : nullptr,
                       c ? c->episode_dev : nullptr, c ? c->active_dev : nullptr, c ? c->limit : 0, c ? c->on_end : 0,
                       c ? (uint64_t)c->reset_seed : 0, c ? (long long)c->first_env : 0, env->traj_dev, env->t};
    return smz_internal_search_launch_narrow(h, desc, weights_dev, env->obs_dev, train,
                                             SearchActArgs{temperature, action_dev, policy_dev, child_visits_dev, root_value_dev, E},
                                             pow_table_host, stream);
}
#endif  // SMZ_PART != 4 && SMZ_PART != 6

#endif  // SMZ_PART != 1

#if SMZ_PART == 0 || SMZ_PART == 1
int smz_root_stats(smz_handle *h, int32_t *visits_dev, double *priors_dev, float *root_value_dev,
                   float *child_reward_dev, smz_stream stream) {
    if (!h) return fail(SMZ_ERR_INVALID, "smz_root_stats: null handle%s");
    if (!h->root_ready) return fail(SMZ_ERR_STATE, "smz_root_stats before smz_root_init%s");
    DeviceGuard guard(h->cfg.device);
    hipLaunchKernelGGL(k_root_stats, tree_grid(h->P.B), dim3(kWave), 0, (hipStream_t)stream, h->P, visits_dev, priors_dev,
                       root_value_dev, child_reward_dev);
    return launch_check();
}

int smz_act(smz_handle *h, double temperature, const double *pow_table_host, int32_t *action_dev, double *policy_dev,
            double *child_visits_dev, float *root_value_dev, smz_stream stream) {
    if (!h) return fail(SMZ_ERR_INVALID, "smz_act: null handle%s");
    if (!h->root_ready) return fail(SMZ_ERR_STATE, "smz_act before smz_root_init%s");
    DeviceGuard guard(h->cfg.device);
    Params P = h->P;
    if (pow_table_host && temperature >= 0.3) {
        if (!h->pow_valid || h->pow_T != temperature) {
            // a new temperature: synchronous upload (not capturable); the table is reused while T is unchanged
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            HIP_TRY(hipMemcpy(h->d_pow, pow_table_host, ((size_t)h->cfg.num_simulations + 1) * sizeof(double),
                              hipMemcpyHostToDevice));
            h->pow_T = temperature;
            h->pow_valid = true;
        }
        P.pow_table = h->d_pow;
    }
    SMZ_DISPATCH(h->maxa, hipLaunchKernelGGL((k_act<MA>), tree_grid(P.B), dim3(kWave), 0, (hipStream_t)stream, P, temperature,
                                             action_dev, policy_dev, child_visits_dev, root_value_dev));
    return launch_check();
}

int smz_support_decode(const float *logits_dev, int S, float *out_dev, int B, smz_stream stream) {
    if (!logits_dev || !out_dev || S < 1 || B < 1) return fail(SMZ_ERR_INVALID, "smz_support_decode: bad argument%s");
    const int lpr = group_lanes(S);
    hipLaunchKernelGGL(k_support_decode, group_grid(B, lpr), dim3(256), 0, (hipStream_t)stream, logits_dev, S, out_dev, B, lpr);
    return launch_check();
}

int smz_policy_softmax(const float *logits_dev, int A, float *out_dev, int B, smz_stream stream) {
    if (!logits_dev || !out_dev || A < 1 || B < 1) return fail(SMZ_ERR_INVALID, "smz_policy_softmax: bad argument%s");
    const int lpr = group_lanes(A);
    hipLaunchKernelGGL(k_policy_softmax, group_grid(B, lpr), dim3(256), 0, (hipStream_t)stream, logits_dev, A, out_dev, B, lpr);
    return launch_check();
}

int smz_dynamics_epilogue(const float *state_dyn_dev, const float *state_after_dev, const float *reward_logits_dev,
                          int ld, const uint8_t *branch_dev, int S, float *hidden_out_dev, float *reward_out_dev, int B,
                          smz_stream stream) {
    if (!state_dyn_dev || !state_after_dev || !branch_dev || !hidden_out_dev || S < 1 || B < 1 || ld < S)
        return fail(SMZ_ERR_INVALID, "smz_dynamics_epilogue: bad argument%s");
    const int lpr = group_lanes(S);
    hipLaunchKernelGGL(k_dynamics_epilogue, group_grid(B, lpr), dim3(256), 0, (hipStream_t)stream, state_dyn_dev,
                       state_after_dev, reward_logits_dev, ld, branch_dev, S, hidden_out_dev, reward_out_dev, B, lpr);
    return launch_check();
}

int smz_prediction_epilogue(const float *policy_logits_pred_dev, const float *value_logits_pred_dev,
                            const float *policy_logits_after_dev, const float *value_logits_after_dev,
                            int ld, const uint8_t *branch_dev, int A, int S, float *policy_out_dev,
                            float *value_out_dev, int B, smz_stream stream) {
    if (!policy_logits_pred_dev || !value_logits_pred_dev || !policy_logits_after_dev || !value_logits_after_dev ||
        !branch_dev || !policy_out_dev || !value_out_dev || A < 1 || S < 1 || B < 1 || ld < 1)
        return fail(SMZ_ERR_INVALID, "smz_prediction_epilogue: bad argument%s");
    const int lpr = group_lanes(S > A ? S : A);
    hipLaunchKernelGGL(k_prediction_epilogue, group_grid(B, lpr), dim3(256), 0, (hipStream_t)stream, policy_logits_pred_dev,
                       value_logits_pred_dev, policy_logits_after_dev, value_logits_after_dev, ld, branch_dev, A, S,
                       policy_out_dev, value_out_dev, B, lpr);
    return launch_check();
}

int smz_cartpole_step(double *state_dev, const int32_t *action_dev, float *obs_out_dev, float *reward_out_dev,
                      uint8_t *terminated_out_dev, int B, smz_stream stream) {
    if (!state_dev || !action_dev || B < 1) return fail(SMZ_ERR_INVALID, "smz_cartpole_step: bad argument%s");
    const EnvStep E = {state_dev, obs_out_dev, reward_out_dev, terminated_out_dev, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, 0};
    hipLaunchKernelGGL(k_cartpole_step_env, row_grid(B), dim3(256), 0, (hipStream_t)stream, E, action_dev, B,
                       (const double *)nullptr, (const double *)nullptr, (const float *)nullptr);
    return launch_check();
}

int smz_cartpole_step_pack(double *state_dev, const int32_t *action_dev, float *obs_out_dev, float *reward_out_dev,
                           uint8_t *terminated_out_dev, double *traj_dev, int T, int t, const double *policy_dev,
                           const double *child_visits_dev, const float *root_value_dev, int B, smz_stream stream) {
    if (!state_dev || !action_dev || !traj_dev || !policy_dev || !child_visits_dev || !root_value_dev || t < 0 || t >= T || B < 1)
        return fail(SMZ_ERR_INVALID, "smz_cartpole_step_pack: bad argument%s");
    const EnvStep E = {state_dev, obs_out_dev, reward_out_dev, terminated_out_dev, nullptr, nullptr, nullptr, 0, 0, 0, 0, traj_dev, t};
    hipLaunchKernelGGL(k_cartpole_step_env, row_grid(B), dim3(256), 0, (hipStream_t)stream, E, action_dev, B, policy_dev,
                       child_visits_dev, root_value_dev);
    return launch_check();
}

int smz_cartpole_step_ctl(double *state_dev, const int32_t *action_dev, float *obs_out_dev, float *reward_out_dev,
                          uint8_t *flag_out_dev, const smz_episode_ctl *ctl, double *traj_dev, int T, int t,
                          const double *policy_dev, const double *child_visits_dev, const float *root_value_dev, int B,
                          smz_stream stream) {
    if (!state_dev || !action_dev || !ctl || !ctl->step_count_dev || B < 1)
        return fail(SMZ_ERR_INVALID, "smz_cartpole_step_ctl: bad argument%s");
    if (ctl->on_end < 0 || ctl->on_end > 2 || (ctl->on_end == 2 && !ctl->episode_dev) || (ctl->on_end == 1 && !ctl->active_dev))
        return fail(SMZ_ERR_INVALID, "smz_cartpole_step_ctl: on_end 1 needs active_dev, on_end 2 needs episode_dev%s");
    if (traj_dev && (!policy_dev || !child_visits_dev || !root_value_dev || t < 0 || t >= T))
        return fail(SMZ_ERR_INVALID, "smz_cartpole_step_ctl: bad trajectory argument%s");
    const EnvStep E = {state_dev, obs_out_dev, reward_out_dev, flag_out_dev, ctl->step_count_dev, ctl->episode_dev,
                       ctl->active_dev, ctl->limit, ctl->on_end, (uint64_t)ctl->reset_seed, (long long)ctl->first_env, traj_dev, t};
    hipLaunchKernelGGL(k_cartpole_step_env, row_grid(B), dim3(256), 0, (hipStream_t)stream, E, action_dev, B, policy_dev,
                       child_visits_dev, root_value_dev);
    return launch_check();
}

int smz_cartpole_reset_state(uint64_t reset_seed, int64_t env, int64_t episode, double state_out[4]) {
    if (!state_out || episode < 1) return fail(SMZ_ERR_INVALID, "smz_cartpole_reset_state: bad argument%s");
    for (int c = 0; c < 4; c++) state_out[c] = -0.05 + 0.1 * smz_unit(reset_seed, (uint64_t)env, (uint64_t)episode, (uint64_t)c);
    return SMZ_OK;
}

int smz_host_cartpole_step(double *state_host, const int32_t *action_host, float *obs_out_host, float *reward_out_host,
                           uint8_t *flag_out_host, int32_t *step_count_host, int32_t limit, int B) {
    if (!state_host || !action_host || B < 1) return fail(SMZ_ERR_INVALID, "smz_host_cartpole_step: bad argument%s");
    const double g = 9.8, mc = 1.0, mp = 0.1, tm = mc + mp, len = 0.5, pml = mp * len, fm = 10.0, tau = 0.02;
    for (int e = 0; e < B; e++) {
        double *st = state_host + (size_t)e * 4;
        const double x = st[0], xd = st[1], th = st[2], thd = st[3];
        const double force = action_host[e] == 1 ? fm : -fm;
        const double ct = cos(th), sn = sin(th);
        const double temp = (force + pml * thd * thd * sn) / tm;
        const double tha = (g * sn - ct * temp) / (len * (4.0 / 3.0 - mp * ct * ct / tm));
        const double xa = temp - pml * tha * ct / tm;
        const double nx = x + tau * xd, nxd = xd + tau * xa, nth = th + tau * thd, nthd = thd + tau * tha;
        st[0] = nx; st[1] = nxd; st[2] = nth; st[3] = nthd;
        if (obs_out_host) {
            float *o = obs_out_host + (size_t)e * 4;
            o[0] = (float)nx; o[1] = (float)nxd; o[2] = (float)nth; o[3] = (float)nthd;
        }
        if (reward_out_host) reward_out_host[e] = 1.0f;
        const bool term = fabs(nx) > 2.4 || fabs(nth) > 12.0 * 2.0 * 3.14159265358979323846 / 360.0;
        int flag = term ? 1 : 0;
        if (step_count_host) {
            const int count = ++step_count_host[e];
            if (limit > 0 && count == limit) flag = 2;
        }
        if (flag_out_host) flag_out_host[e] = (uint8_t)flag;
    }
    return SMZ_OK;
}

int smz_synthetic_obs(float *obs_dev, int B, int obs_dim, uint64_t seed, int64_t first_env, int64_t t, smz_stream stream) {
    if (!obs_dev || B < 1 || obs_dim < 1) return fail(SMZ_ERR_INVALID, "smz_synthetic_obs: bad argument%s");
    const size_t n = (size_t)B * obs_dim;
    hipLaunchKernelGGL(k_synthetic_obs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs_dev, B,
                       obs_dim, seed, (long long)first_env, (long long)t);
    return launch_check();
}

int smz_set_active(smz_handle *h, const uint8_t *active_dev) {
    if (!h) return fail(SMZ_ERR_INVALID, "smz_set_active: null handle%s");
    h->P.active = active_dev;
    return SMZ_OK;
}

int smz_set_leaf_ids_out(smz_handle *h, int32_t *ids_dev) {
    if (!h) return fail(SMZ_ERR_INVALID, "smz_set_leaf_ids_out: null handle%s");
    h->P.ids_out = ids_dev;
    return SMZ_OK;
}

int smz_get_hidden_layout(smz_handle *h, float **hidden_dev_out, int *nodes_per_tree_out, int *row_stride_out) {
    if (!h || !hidden_dev_out || !nodes_per_tree_out || !row_stride_out) return fail(SMZ_ERR_INVALID, "smz_get_hidden_layout: null argument%s");
    *hidden_dev_out = h->P.hidden;
    *nodes_per_tree_out = h->P.N;
    *row_stride_out = h->P.hs;
    return SMZ_OK;
}

int smz_traj_floats(int obs_dim, int A) { return obs_dim + 3 * A + 3; }

int smz_traj_pack(double *traj_dev, int T, int t, int obs_dim, int A, const float *obs_dev, const float *reward_dev,
                  const uint8_t *terminated_dev, const int32_t *action_dev, const double *policy_dev, const double *child_visits_dev,
                  const float *root_value_dev, int B, smz_stream stream) {
    if (!traj_dev || (!obs_dev && obs_dim > 0) || !action_dev || !policy_dev || !child_visits_dev || !root_value_dev || t < 0 ||
        t >= T || B < 1 || A < 1 || obs_dim < 0)
        return fail(SMZ_ERR_INVALID, "smz_traj_pack: bad argument%s");
    const size_t slab = (size_t)B * (obs_dim + 3 * A + 3);
    const unsigned blocks = (unsigned)std::min<size_t>((slab + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_traj_pack, dim3(blocks), dim3(256), 0, (hipStream_t)stream, traj_dev, T, t, obs_dim, A, obs_dev,
                       reward_dev, terminated_dev, action_dev, policy_dev, child_visits_dev, root_value_dev, B);
    return launch_check();
}

int smz_traj_targets(const double *traj_dev, int T, int obs_dim, int A, int B, int td_steps, const double *discount_pow_dev,
                     int ignore_termination, int32_t *length_dev, double *value_target_dev, double *abs_td_error_dev,
                     smz_stream stream) {
    if (!traj_dev || !discount_pow_dev || !length_dev || !value_target_dev || T < 1 || B < 1 || A < 1 || obs_dim < 0 ||
        td_steps < 0)
        return fail(SMZ_ERR_INVALID, "smz_traj_targets: bad argument%s");
    hipLaunchKernelGGL(k_traj_lengths, row_grid(B), dim3(256), 0, (hipStream_t)stream, traj_dev, T, obs_dim, A, B,
                       ignore_termination, length_dev);
    const size_t cells = (size_t)T * B;
    hipLaunchKernelGGL(k_traj_targets, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, (hipStream_t)stream, traj_dev, T,
                       obs_dim, A, B, td_steps, discount_pow_dev, (const int32_t *)length_dev, value_target_dev,
                       abs_td_error_dev, (const int32_t *)nullptr);
    return launch_check();
}

int smz_traj_targets_games(const double *traj_dev, int T, int obs_dim, int A, int B, int td_steps,
                           const double *discount_pow_dev, int ignore_termination, int new_game, int32_t *length_dev,
                           int32_t *game_end_dev, double *value_target_dev, double *abs_td_error_dev, smz_stream stream) {
    if (!traj_dev || !discount_pow_dev || !game_end_dev || !value_target_dev || T < 1 || B < 1 || A < 1 || obs_dim < 0 ||
        td_steps < 0)
        return fail(SMZ_ERR_INVALID, "smz_traj_targets_games: bad argument%s");
    hipLaunchKernelGGL(k_traj_game_ends, row_grid(B), dim3(256), 0, (hipStream_t)stream, traj_dev, T, obs_dim, A, B,
                       ignore_termination, new_game, length_dev, game_end_dev);
    const size_t cells = (size_t)T * B;
    hipLaunchKernelGGL(k_traj_targets, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, (hipStream_t)stream, traj_dev, T,
                       obs_dim, A, B, td_steps, discount_pow_dev, (const int32_t *)nullptr, value_target_dev, abs_td_error_dev,
                       (const int32_t *)game_end_dev);
    return launch_check();
}

int smz_debug_div_by_count(const double *x_dev, const int32_t *n_dev, int count, int table_size, double *out_dev,
                           smz_stream stream) {
    if (!x_dev || !n_dev || !out_dev || count < 1 || table_size < 2 || table_size > kPbcLdsMax)
        return fail(SMZ_ERR_INVALID, "smz_debug_div_by_count: bad argument%s");
    hipLaunchKernelGGL(k_debug_div_by_count, dim3(256), dim3(256), (size_t)table_size * sizeof(double), (hipStream_t)stream,
                       x_dev, n_dev, count, table_size, out_dev);
    return launch_check();
}

int smz_debug_glibc_log_pow(const double *x_dev, const double *y_dev, int count, double *out_log_dev, double *out_pow_dev,
                            smz_stream stream) {
    if (!x_dev || count < 1 || (!out_log_dev && !out_pow_dev) || (out_pow_dev && !y_dev))
        return fail(SMZ_ERR_INVALID, "smz_debug_glibc_log_pow: bad argument%s");
    hipLaunchKernelGGL(k_debug_glibc_log_pow, dim3(1024), dim3(256), 0, (hipStream_t)stream, x_dev, y_dev, count, out_log_dev,
                       out_pow_dev);
    return launch_check();
}

int smz_debug_dump_tree(smz_handle *h, int tree, smz_node_view *nodes, int cap, float *minmax_out, int32_t *path_out,
                        int cap_path, int32_t *path_len_out, double *root_priors_out) {
    if (!h || tree < 0 || tree >= h->cfg.num_trees) return fail(SMZ_ERR_INVALID, "smz_debug_dump_tree: bad argument%s");
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const Params &P = h->P;
    const int A = P.A, K = P.K;
    TreeHdr hdr;
    HIP_TRY(hipMemcpy(&hdr, P.hdr + tree, sizeof(hdr), hipMemcpyDeviceToHost));
    std::vector<uint32_t> blob((size_t)P.tree_words);
    HIP_TRY(hipMemcpy(blob.data(), P.nodes + (size_t)tree * P.tree_words, (size_t)P.tree_words * 4, hipMemcpyDeviceToHost));
    const int n = 1 + A + hdr.n_exp * K;
    auto f32 = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    if (nodes && cap > 0) {
        // flatten the child blocks back to creation-order node ids (root 0, its children 1..A, expansion e at 1+A+e*K)
        if (cap > 0) nodes[0] = smz_node_view{hdr.root_visit, hdr.root_value_sum, 0.f, 0.f, 1, 0};
        for (int a = 0; a < A && 1 + a < cap; a++) {
            const uint32_t *rb = blob.data();
            const int c = (int)rb[4 * A + a];
            nodes[1 + a] = smz_node_view{(int32_t)rb[2 * a], f32(rb[2 * a + 1]), f32(rb[2 * A + a]), f32(rb[3 * A + a]),
                                         c ? 1 + A + (c - 1) * K : 0, a};
        }
        for (int e = 0; e < hdr.n_exp; e++) {
            const uint32_t *eb = blob.data() + P.rb_words + (size_t)e * P.eb_words;
            for (int j = 0; j < K; j++) {
                const int id = 1 + A + e * K + j;
                if (id >= cap) break;
                const int c = (int)eb[4 * K + j];
                nodes[id] = smz_node_view{(int32_t)eb[2 * j], f32(eb[2 * j + 1]), f32(eb[2 * K + j]), f32(eb[3 * K + j]),
                                          c ? 1 + A + (c - 1) * K : 0, (int32_t)eb[5 * K + j]};
            }
        }
    }
    if (minmax_out) { minmax_out[0] = hdr.mn; minmax_out[1] = hdr.mx; }
    // the recorded path as node ids, root first (the device stores (block << 8 | slot) without the root)
    const int plen = hdr.path_len > 0 ? hdr.path_len + 1 : 0;
    if (path_len_out) *path_len_out = plen;
    if (path_out && cap_path > 0 && plen > 0) {
        std::vector<uint4> recs((size_t)hdr.path_len);
        HIP_TRY(hipMemcpy2D(recs.data(), sizeof(uint4), P.path + tree, (size_t)P.B * sizeof(uint4), sizeof(uint4), (size_t)hdr.path_len, hipMemcpyDeviceToHost));
        path_out[0] = 0;
        for (int i = 0; i < hdr.path_len && i + 1 < cap_path; i++) {
            const int blk = (int)recs[i].x >> 8, slot = (int)recs[i].x & 0xff;
            path_out[i + 1] = blk == 0 ? 1 + slot : 1 + A + (blk - 1) * K + slot;
        }
    }
    if (root_priors_out) memcpy(root_priors_out, blob.data() + P.rp_off, (size_t)A * 8);
    return n;
}

int smz_philox_words(uint64_t seed, uint32_t block, int idx, int n, uint32_t *host_out) {
    if (!host_out || n < 0 || idx < 0 || idx >= kMtN) return fail(SMZ_ERR_INVALID, "smz_philox_words: bad argument%s");
    for (int i = 0; i < n; i++) {
        host_out[i] = philox_word(block, idx, (uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32));
        if (++idx == kMtN) { idx = 0; ++block; }
    }
    return SMZ_OK;
}

int smz_get_philox_position(smz_handle *h, int tree, uint32_t *block_out, int *idx_out) {
    if (!h || tree < 0 || tree >= h->cfg.num_trees || !block_out || !idx_out)
        return fail(SMZ_ERR_INVALID, "smz_get_philox_position: bad argument%s");
    if (!h->P.philox) return fail(SMZ_ERR_STATE, "smz_get_philox_position: the handle draws from MT19937%s");
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    int32_t packed = 0;
    HIP_TRY(hipMemcpy(&packed, h->P.rng_pos + tree, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(block_out, h->P.rng_block + tree, sizeof(uint32_t), hipMemcpyDeviceToHost));
    *idx_out = packed & 0xffff;
    return SMZ_OK;
}

int smz_enable_stats(smz_handle *h, int on) {
    if (!h) return fail(SMZ_ERR_INVALID, "smz_enable_stats: null handle%s");
    h->stats_on = on != 0;
    h->P.stats = h->stats_on ? h->d_stats : nullptr;
    return SMZ_OK;
}

int smz_read_stats(smz_handle *h, uint64_t levels_out[4], int reset) {
    if (!h || !levels_out) return fail(SMZ_ERR_INVALID, "smz_read_stats: null argument%s");
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long v[16];
    HIP_TRY(hipMemcpy(v, h->d_stats, sizeof(v), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) levels_out[i] = (uint64_t)v[i];
    if (getenv("SMZ_DEBUG_SKIP") && (atoi(getenv("SMZ_DEBUG_SKIP")) & 16))
        fprintf(stderr, "[smz phase cycles, summed over waves] stage %llu expand %llu select %llu mlp %llu  (k_search_vision: tree, conv, wait, towers+tails)\n", v[4], v[5], v[6], v[7]);
#ifdef SMZ_BPS_PROBE
    if (v[8])
        fprintf(stderr, "[smz k_search_mlp probe, cycles summed over waves] expand+backup %llu | select: prepare %llu evaluate %llu chase %llu records+leaf %llu "
                        "| networks %llu | staging %llu\n", v[8], v[9], v[10], v[11], v[12], v[13], v[14]);
#endif
    if (getenv("SMZ_DEBUG_SKIP") && (atoi(getenv("SMZ_DEBUG_SKIP")) & 16) && v[8])
        fprintf(stderr, "[smz k_search_vision phases] tree %llu load %llu conv_transition %llu conv_prediction %llu wait %llu layer1 %llu hidden_out %llu tails_stage %llu\n",
                v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);
    if (reset) HIP_TRY(hipMemset(h->d_stats, 0, sizeof(v)));
    return SMZ_OK;
}

#endif  // SMZ_PART != 2

}  // extern "C"
