// smz_device.hpp -- device-side building blocks of the gfx950 search kernels.
//
// Execution shape: one search tree per lane, 64 trees per wavefront.  The per-tree logic below is scalar code run
// by one lane; the wave-cooperative parts (random-word staging, hidden-row moves) live in smz_kernels.hip.
// The arithmetic follows, operation by operation and with the same float32/float64 roundings, the reference's
// monte_carlo_tree_search.py and the numpy legacy RandomState it draws from.  This translation unit MUST be
// compiled with -ffp-contract=off: a fused multiply-add would change roundings the reference performs separately.
//
// Data layout (HBM).  A tree is a header plus "child blocks"; a block holds the children of ONE node as a small
// structure-of-arrays, so that one descent level touches one contiguous 64/128-byte block:
//     root block   (A children): (visit, value_sum)[A] | reward[A] | prior32[A] | child[A] | (pad) | prior64[A]
//     expansion e  (K children): (visit, value_sum)[K] | reward[K] | prior32[K] | child[K] | action[K]
// (the visit count and value sum of a child are adjacent: the backup updates both with ONE 8-byte store per level)
// `child` = 1 + index of the expansion block holding that node's own children (0 = not expanded), so the block
// just loaded already names the next block to load: one dependent memory round trip per level.  Node ids (used for
// the hidden-state rows and by the debug dump) stay the creation-order ids of the reference-side goldens:
// root 0, root children 1..A, children of expansion e at 1 + A + e*K + j.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smz_glibc_math.hpp"

namespace smz {

constexpr int kMtN = 624;
constexpr int kMtM = 397;
constexpr int kWave = 64;
constexpr int kRngStage = 64;   // MT words staged per tree per launch
constexpr int kRngStride = kRngStage + 1;

struct TreeHdr {          // one 32-byte record per tree
    int32_t n_exp;        // expansion blocks allocated so far
    int32_t path_len;     // non-root nodes on the recorded search path
    float mn, mx;         // MinMaxStats (monte_carlo_tree_search.py:24-36)
    int32_t root_visit;   // root.visit_count
    float root_value_sum; // root.value_sum
    int32_t pad0, pad1;
};

struct Params {
    int32_t B, A, K, S, N, P, sims;
    int32_t hs;         // floats between consecutive hidden rows: S rounded up to a 64-byte line
    int32_t lds_stage;  // 1: stage the twisted words in LDS (small batches); 0: twist ahead only, draw from L1 (occupancy)
    int32_t dbg;        // timing-ablation switches (SMZ_DEBUG_SKIP, diagnostics only; results are then meaningless)
    int32_t tpw;        // trees per wavefront (power of two <= 64): lanes >= tpw only help in the cooperative phases
    int32_t rb_words;   // words in a root block (multiple of 16)
    int32_t eb_words;   // words in an expansion block (multiple of 16)
    int32_t rp_off;     // word offset of prior64[] inside the root block (even)
    int64_t tree_words; // words per tree = rb_words + sims * eb_words
    float disc32;       // float32(discount): python float * np.float32 -> float32 under NEP 50 (mcts:239, :308)
    float keep32;       // float32(1 - root_exploration_fraction) (mcts:224-225)
    double frac, alpha;
    uint32_t *nodes;        // [B][tree_words]
    float *hidden;          // [B][N][hs]
    TreeHdr *hdr;           // [B]
    uint4 *path;            // [P][B]   records (block << 8 | slot, visit, value_sum, reward); block 0 = root block.  Level-major:
                            //          the 64 trees of a wavefront write / read a level as one contiguous 1 KB piece (PathCol)
    uint32_t *mt;           // [B][624]
    int32_t *rng_pos;       // [B]  (ready << 16) | idx
    const double *pbc_sqrt; // [sims+2]  sqrt(n) * pb_c(n)
    const double *pow_table;  // [sims+1] or nullptr
    unsigned long long *stats;  // [4] or nullptr
    const uint8_t *active;      // [B] or nullptr: trees with active[i] == 0 are skipped by every entry point (smz_set_active)
    int32_t philox;             // SMZ_RNG_PHILOX: words come from Philox4x32-10 (key = the tree's seed) instead of MT19937
    uint32_t *rng_block;        // [B]  (philox) 624-word blocks consumed so far: word (block, idx) of a tree is component
                                //      idx & 3 of philox(counter = block * 156 + idx / 4, key); rng_pos keeps idx
    const uint32_t *rng_key;    // [B][2] (philox)
    int32_t *ids_out;           // [B][2] or nullptr: every selection also writes (leaf node id, parent node id) per tree
                                // ({-1, -1} for a switched-off tree): smz_set_leaf_ids_out
    int32_t tree0;              // index of the tree whose blocks `nodes` points at: 0, except in a kernel that keeps its workgroup's
                                // trees in LDS for the search (k_search_vision) and points `nodes` there
    int32_t thr_off, thr_stride;  // THR kernels (two children per block): two spare words per expansion block b at word
                                  // thr_off + (b - 1) * thr_stride of its tree.  Block of a chance-flagged node: its float64 sampling
                                  // threshold (chance_threshold2).  Block of a decision-flagged node: float32 y[2], the children's
                                  // reward + discount * value_sum / visits as of their last backup (value_term)
    int32_t ry_off;               // ... and y[A] of the root's children at this word of the root block (-1: no room, not kept)
};
#ifdef SMZ_NO_MASK
__device__ inline bool tree_active(const Params &, int) { return true; }
#else
__device__ inline bool tree_active(const Params &P, int tree) { return !P.active || P.active[tree] != 0; }
#endif

__device__ inline uint32_t *tree_base(const Params &P, int tree) { return P.nodes + (size_t)(tree - P.tree0) * P.tree_words; }
__device__ inline uint32_t *block_ptr(const Params &P, uint32_t *tb, int blk) {
    return blk == 0 ? tb : tb + P.rb_words + (size_t)(blk - 1) * P.eb_words;
}
__device__ inline int loc_node_id(const Params &P, int loc) {
    const int blk = loc >> 8, slot = loc & 0xff;
    return blk == 0 ? 1 + slot : 1 + P.A + (blk - 1) * P.K + slot;
}

// ---------------------------------------------------------------------------------------------------------------
// numpy legacy RandomState: MT19937 advanced one word at a time.
// The block regeneration numpy performs every 624 draws is equivalent to twisting word i in place when draw i is
// requested (word i needs old[i], old[i+1] and old-or-new[i+397 mod 624] exactly as the in-place block loop has
// them).  `ready` counts words at idx.. that are ALREADY twisted: the kernels twist ahead cooperatively (one
// coalesced pass per tree) and stage the next kRngStage tempered words of every tree in LDS, so that drawing is
// an LDS read; a lane that needs more than the staged words falls back to twisting in global memory.
// ---------------------------------------------------------------------------------------------------------------
__device__ inline uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ inline uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t t = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter (c0, c1, 0, 0), key (k0, k1).
// (Scalars in, four scalars out: small local arrays indexed at run time end up in scratch memory on this compiler.)
struct PhiloxOut { uint32_t x, y, z, w; };
__device__ __host__ inline PhiloxOut philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1) {
    uint32_t c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return PhiloxOut{c0, c1, c2, c3};
}
// word `idx` of 624-word block `block` of the stream with key (k0, k1)
__device__ __host__ inline uint32_t philox_word(uint32_t block, int idx, uint32_t k0, uint32_t k1) {
    const uint64_t n = (uint64_t)block * (kMtN / 4) + (uint64_t)(idx >> 2);
    const PhiloxOut o = philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), k0, k1);
    const int c = idx & 3;
    return c == 0 ? o.x : (c == 1 ? o.y : (c == 2 ? o.z : o.w));
}

// the same word for the rare draw beyond the staged words: rounds in a rolled loop (this body is inlined at every draw
// site of the philox-capable kernels)
__device__ inline uint32_t philox_word_compact(uint32_t block, int idx, uint32_t k0, uint32_t k1) {
    const uint64_t n = (uint64_t)block * (kMtN / 4) + (uint64_t)(idx >> 2);
    uint32_t c0 = (uint32_t)n, c1 = (uint32_t)(n >> 32), c2 = 0u, c3 = 0u;
#pragma unroll 1
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const int c = idx & 3;
    return c == 0 ? c0 : (c == 1 ? c1 : (c == 2 ? c2 : c3));
}

// PHC ("philox capable"): RngT<true> can draw from either generator (a run-time flag per handle); RngT<false> is the
// MT19937-only form with no extra state -- what the specialised (AEX) kernels of the parity-mode headline path
// instantiate, so that the second generator costs them nothing (a Philox handle runs the generic instantiations).
template <bool PHC> struct PhiloxState { uint32_t block, k0, k1; bool on; };
template <> struct PhiloxState<false> {};

template <bool PHC>
struct RngT {
    uint32_t *mt;           // this tree's 624 words in global memory
    const uint32_t *stage;  // this lane's staged (tempered) words in LDS, or nullptr
    int idx, ready, used, staged;
    PhiloxState<PHC> ph;    // (PHC) blocks consumed, key, generator switch
    __device__ __forceinline__ bool philox() const { if constexpr (PHC) return ph.on; else return false; }
    __device__ __forceinline__ uint32_t block() const { if constexpr (PHC) return ph.block; else return 0u; }
    __device__ __forceinline__ uint32_t key0() const { if constexpr (PHC) return ph.k0; else return 0u; }
    __device__ __forceinline__ uint32_t key1() const { if constexpr (PHC) return ph.k1; else return 0u; }
    // a helper lane (paired descent) takes over the tree lane's counter stream: block and key.  bind() gave it none -- it is not a
    // tree lane -- so a draw of the helper beyond the 64 staged words (a path of 20+ levels) would have come from key 0.  Round 6,
    // found by reading; no search that hits it was found (tests/test_gpu_end_to_end.py test_philox_paired_descent_on_deep_paths
    // passes with and without: 2 % of its four-action trees end 20-22 levels deep), so this is a precaution, not a measured fix.
    __device__ __forceinline__ void follow(uint32_t block, uint32_t k0, uint32_t k1) { if constexpr (PHC) { ph.block = block; ph.k0 = k0; ph.k1 = k1; } }
    __device__ __forceinline__ void wrapped() { if constexpr (PHC) ++ph.block; }
    __device__ __forceinline__ void load(uint32_t *state, int packed, const uint32_t *lds_row, int n_staged) {
        mt = state; idx = packed & 0xffff; ready = packed >> 16; stage = lds_row; staged = n_staged; used = 0;
        if (philox()) { staged = n_staged - (idx & 3); if (staged < 0) staged = 0; }   // philox tiles start on a 4-word boundary
    }
    // per-kernel set-up, before the first load(): which generator, and (philox) this tree's key and block counter
    __device__ __forceinline__ void bind(const Params &P, int tree, bool valid) {
        if constexpr (PHC) {
            ph.on = P.philox != 0; ph.block = 0u; ph.k0 = ph.k1 = 0u;
            if (ph.on && valid) { ph.block = P.rng_block[tree]; ph.k0 = P.rng_key[2 * tree]; ph.k1 = P.rng_key[2 * tree + 1]; }
        }
    }
    __device__ __forceinline__ void save(const Params &P, int tree) const { if constexpr (PHC) { if (ph.on) P.rng_block[tree] = ph.block; } }
    __device__ __forceinline__ int pack() const { return (ready << 16) | idx; }
    // (forced inline: an out-of-line call takes `this` by address and the whole generator state moves to scratch memory)
    __device__ __forceinline__ uint32_t next32() {
        if (__builtin_expect(used < staged, 1)) {   // fast path: word was twisted and tempered by the staging pass
            const uint32_t y = stage[used++];
            --ready;
            if (idx + 1 == kMtN) { idx = 0; wrapped(); } else ++idx;
            return y;
        }
        const int i = idx;
        if constexpr (PHC) {
            if (ph.on) {
                const uint32_t y = philox_word_compact(ph.block, i, ph.k0, ph.k1);
                if (i + 1 == kMtN) { idx = 0; ++ph.block; } else idx = i + 1;
                return y;
            }
        }
        uint32_t y;
        if (ready > 0) {
            y = mt[i];
            --ready;
        } else {
            const int i1 = (i + 1 == kMtN) ? 0 : i + 1;
            int im = i + kMtM;
            if (im >= kMtN) im -= kMtN;
            y = mt_twist(mt[i], mt[i1], mt[im]);
            mt[i] = y;
        }
        idx = (i + 1 == kMtN) ? 0 : i + 1;
        return mt_temper(y);
    }
    // N consecutive words with ONE staged-words check and one bookkeeping update (a pUCT level needs 2 per child)
    template <int N>
    __device__ __forceinline__ void take(uint32_t (&out)[N]) {
        if (__builtin_expect(used + N <= staged, 1)) {
#pragma unroll
            for (int i = 0; i < N; i++) out[i] = stage[used + i];
            used += N;
            ready -= N;
            idx += N;
            if (idx >= kMtN) { idx -= kMtN; wrapped(); }
        } else {
#pragma unroll
            for (int i = 0; i < N; i++) out[i] = next32();
        }
    }
    // RandomState.random_sample(): (a * 2^26 + b) / 2^53 with a = 27 bits, b = 26 bits
    __device__ static double to_double(uint32_t w0, uint32_t w1) {
        const int32_t a = (int32_t)(w0 >> 5);
        const int32_t b = (int32_t)(w1 >> 6);
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ __forceinline__ double random_sample() {
        uint32_t w[2];
        take<2>(w);
        return to_double(w[0], w[1]);
    }
};
using Rng = RngT<true>;      // the general form (step-wise root / act kernels, generic instantiations)
using RngMt = RngT<false>;   // MT19937 only (specialised instantiations)

// ndarray.sum() of n <= 128 contiguous elements: sequential below 8, else numpy's 8-lane unrolled pairwise block.
template <typename T, int MAXA>
__device__ inline T np_sum(const T *a, int n) {
    if (MAXA < 8 || n < 8) {
        T r = (T)0;
#pragma unroll
        for (int i = 0; i < MAXA; i++) if (i < n) r += a[i];
        return r;
    }
    T r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

// cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, side='right')  == number of entries <= u
template <int MAXA>
__device__ inline int sample_cdf(const double *p, int n, double u) {
    double cdf[MAXA];
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < MAXA; i++) if (i < n) { acc += p[i]; cdf[i] = acc; }
    const double last = acc;
    // the last entry is last / last: 1.0 (or NaN when the sum is 0 / inf), never <= u for u in [0, 1): neither the
    // division nor the comparison can change the count
    int k = 0;
#pragma unroll
    for (int i = 0; i < MAXA; i++) if (i < n - 1) k += ((cdf[i] / last) <= u) ? 1 : 0;
    return k;
}

// register-array helpers: element access by a run-time index as a compare/select chain, so that small per-lane arrays
// stay in registers (a dynamically indexed array is placed in scratch memory, i.e. behind a global round trip)
template <typename T, int N>
__device__ inline T reg_get(const T (&a)[N], int i) {
    T v = a[0];
#pragma unroll
    for (int j = 1; j < N; j++) v = (j == i) ? a[j] : v;
    return v;
}
template <typename T, int N>
__device__ inline void reg_set(T (&a)[N], int i, T v) {
#pragma unroll
    for (int j = 0; j < N; j++) a[j] = (j == i) ? v : a[j];
}

// RandomState.choice(n, size, p=p, replace=False) -- picks in draw order.  `p` is clobbered.
template <int MAXA, class RNG>
__device__ inline void choice_noreplace(RNG &rng, double (&p)[MAXA], int n, int size, int32_t (&out)[MAXA]) {
    constexpr bool REG = MAXA <= 8;     // small arrays: select chains; larger ones: plain indexing
    double cdf[MAXA], x[MAXA];
    int32_t cand[MAXA];
    int n_uniq = 0;
    if constexpr (MAXA == 2) {
        // Both of two entries (the reference's default maxium_action_sample on a two-action env, and every two-action root):
        // the result is {0, 1} in some order and only the number of draws is open.  Round one draws twice against
        // t = cdf[0] / cdf[1]; if both draws fall on the same side, round two zeroes the entry found and draws once more --
        // its cdf is then {0, p1} / p1 or {p0, p0} / p0 and the other entry comes out whatever that draw is.
        const double last = (0.0 + p[0]) + p[1];
        if (n == 2 && size == 2 && last > 0.0 && last < __builtin_inf()) {
            uint32_t w[4];
            rng.template take<4>(w);
            const double t = (0.0 + p[0]) / last;
            const bool c0 = t <= RNG::to_double(w[0], w[1]), c1 = t <= RNG::to_double(w[2], w[3]);
            if (c0 == c1) (void)rng.random_sample();
            out[0] = c0 ? 1 : 0;
            out[1] = c0 ? 0 : 1;
            return;
        }
    }
    while (n_uniq < size) {
        const int m = size - n_uniq;
        if (MAXA <= 4 && m == MAXA) {            // a full round of draws: one staged-words check for all of them
            uint32_t w[MAXA <= 4 ? 2 * MAXA : 2];
            rng.template take<MAXA <= 4 ? 2 * MAXA : 2>(w);
#pragma unroll
            for (int i = 0; i < MAXA; i++) x[i] = RNG::to_double(w[MAXA <= 4 ? 2 * i : 0], w[MAXA <= 4 ? 2 * i + 1 : 1]);
        } else {
#pragma unroll
            for (int i = 0; i < MAXA; i++) if (i < m) x[i] = rng.random_sample();
        }
        if (REG) {
#pragma unroll
            for (int j = 0; j < MAXA; j++) {          // p[out[i]] = 0 for every pick so far
                bool taken = false;
#pragma unroll
                for (int i = 0; i < MAXA; i++) taken = taken || (i < n_uniq && out[i] == j);
                if (taken) p[j] = 0.0;
            }
        } else {
            for (int i = 0; i < n_uniq; i++) p[out[i]] = 0.0;
        }
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < MAXA; i++) if (i < n) { acc += p[i]; cdf[i] = acc; }
        const double last = acc;
#pragma unroll
        for (int i = 0; i < MAXA; i++) if (i < n - 1) cdf[i] = cdf[i] / last;     // entry n-1 would be 1.0 / NaN: see sample_cdf
#pragma unroll
        for (int i = 0; i < MAXA; i++) {
            if (i < m) {
                int k = 0;
#pragma unroll
                for (int j = 0; j < MAXA; j++) if (j < n - 1) k += (cdf[j] <= x[i]) ? 1 : 0;
                cand[i] = k;
            }
        }
#pragma unroll
        for (int i = 0; i < MAXA; i++) {
            if (i < m) {
                bool dup = false;
#pragma unroll
                for (int j = 0; j < MAXA; j++) dup = dup || (j < i && cand[j] == cand[i]);
                if (!dup) {
                    if (REG) reg_set<int32_t, MAXA>(out, n_uniq, cand[i]); else out[n_uniq] = cand[i];
                    n_uniq++;
                }
            }
        }
    }
}

// np.sort of the first `size` entries (insertion sort; select chains for small arrays)
template <int MAXA>
__device__ inline void sort_picks(int32_t (&v)[MAXA], int size) {
    if (MAXA <= 8) {
#pragma unroll
        for (int pass = 0; pass < MAXA - 1; pass++) {
#pragma unroll
            for (int j = 0; j + 1 < MAXA; j++) {
                if (j + 1 < size && v[j] > v[j + 1]) { const int32_t t = v[j]; v[j] = v[j + 1]; v[j + 1] = t; }
            }
        }
    } else {
        for (int i = 1; i < size; i++) {
            const int32_t x = v[i];
            int j = i - 1;
            while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; j--; }
            v[j + 1] = x;
        }
    }
}

// legacy_standard_gamma for shape <= 1 (numpy/random/src/legacy/legacy-distributions.c).  log / pow are glibc's own routines,
// restated operation by operation (smz_glibc_math.hpp): the sample -- and with it the float64 root priors -- is numpy's bit for
// bit (rounds 1-5 used the device library's, correct to < 1 ulp but not glibc's roundings: priors within 1e-13 only).
#ifndef SMZ_DEVICE_LIBM
#define SMZ_DEVICE_LIBM 0        // 1: the device library's log / pow as in rounds 1-5 (A/B builds only: priors to 1e-13)
#endif
#if SMZ_DEVICE_LIBM
#define smz_glibc_log log
#define smz_glibc_pow pow
#endif
template <class RNG>
__device__ inline double legacy_gamma(RNG &rng, double shape) {
    if (shape == 1.0) return -smz_glibc_log(1.0 - rng.random_sample());
    if (shape == 0.0) return 0.0;
    for (;;) {
        const double U = rng.random_sample();
        const double V = -smz_glibc_log(1.0 - rng.random_sample());
        if (U <= 1.0 - shape) {
            const double X = smz_glibc_pow(U, 1.0 / shape);
            if (X <= V) return X;
        } else {
            const double Y = -smz_glibc_log((1.0 - U) / shape);
            const double X = smz_glibc_pow(1.0 - shape + shape * Y, 1.0 / shape);
            if (X <= (V + Y)) return X;
        }
    }
}

// p = (policy + 1e-12) / sum, float32 (monte_carlo_tree_search.py:205-206, 291-292)
template <int MAXA>
__device__ inline void normalise_policy(const float *policy, int A, float *p) {
#pragma unroll
    for (int a = 0; a < MAXA; a++) p[a] = (a < A) ? policy[a] + 1e-12f : 0.f;
    const float s = np_sum<float, MAXA>(p, A);
#pragma unroll
    for (int a = 0; a < MAXA; a++) if (a < A) p[a] = p[a] / s;
}

__device__ inline int depth_flag(int depth) { return (depth >> 1) & 1; }  // F F T T ...

// ---------------------------------------------------------------------------------------------------------------
// root (monte_carlo_tree_search.py:179-225)
// ---------------------------------------------------------------------------------------------------------------
template <int MAXA, class RNG>
__device__ inline void root_init_tree(const Params &P, int tree, RNG &rng, const float *policy_row,
                                      const double *noise_override_row, bool train) {
    const int A = P.A;
    uint32_t *rb = tree_base(P, tree);
    float p[MAXA];
    double p64[MAXA];
    int32_t picks[MAXA];
    float pol[MAXA];
    for (int a = 0; a < A; a++) pol[a] = policy_row[a];
    normalise_policy<MAXA>(pol, A, p);
    for (int a = 0; a < A; a++) p64[a] = (double)p[a];
    choice_noreplace<MAXA>(rng, p64, A, A, picks);  // sorted result is 0..A-1; only the draws matter (mcts:208)
    int32_t *vi = (int32_t *)rb;
    float *fs = (float *)rb;
    for (int a = 0; a < A; a++) {
        vi[2 * a] = 0;             // visit
        fs[2 * a + 1] = 0.f;       // value_sum (interleaved with the visit count: one 8-byte store per backup level)
        fs[2 * A + a] = 0.f;       // reward
        fs[3 * A + a] = p[a];      // float32 prior
        vi[4 * A + a] = 0;         // child
    }
    double *rp = (double *)(rb + P.rp_off);
    if (train && P.sims > 0) {
        double noise[MAXA];
        double acc = 0.0;
        for (int a = 0; a < A; a++) { noise[a] = legacy_gamma(rng, P.alpha); acc = acc + noise[a]; }
        const double inv = 1.0 / acc;
        for (int a = 0; a < A; a++) {
            const double n = noise_override_row ? noise_override_row[a] : noise[a] * inv;
            const float scaled = p[a] * P.keep32;
            rp[a] = (double)scaled + n * P.frac;
        }
    } else {
        for (int a = 0; a < A; a++) rp[a] = (double)p[a];
    }
    TreeHdr h;
    h.n_exp = 0;
    h.path_len = 0;
    h.mn = __builtin_inff();
    h.mx = -__builtin_inff();
    h.root_visit = 0;
    h.root_value_sum = 0.f;
    h.pad0 = h.pad1 = 0;
    P.hdr[tree] = h;
}

// ---------------------------------------------------------------------------------------------------------------
// selection (monte_carlo_tree_search.py:228-267): one block load per level, everything else in registers / LDS.
// KS = 2 compiles the expansion-block code for exactly two children (maxium_action_sample's default, every reference
// config): the block is three 16-byte loads at fixed offsets and no lane carries per-child predicates; KS = 0 is the
// general run-time child count (<= MAXA).  The root level (A children, float64 priors) is peeled off the loop.
// Every level appends one record (block << 8 | slot, visit, value_sum, reward of the chosen child) to `rec` (LDS in
// the single-launch kernel, global otherwise), so the backup that follows needs no loads from the tree.
// ---------------------------------------------------------------------------------------------------------------
struct Leaf {
    int32_t leaf_id, parent_id, action, branch;
};

template <int N>
struct Kids {
    int32_t vis[N], chd[N], act[N];
    float vsum[N], rew[N], pri[N];
    double pri64[N];
    float yv[N];     // YV kernels: reward + discount * value_sum / visits as stored by the last backup (meaningful when vis > 0)
};

// run-time child count: predicated dword loads
template <int N>
__device__ inline void load_kids_dyn(const uint32_t *bp, int cnt, bool root, int rp_off, Kids<N> &k) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (j < cnt) {
            k.vis[j] = (int32_t)bp[2 * j];
            k.vsum[j] = __uint_as_float(bp[2 * j + 1]);
            k.rew[j] = __uint_as_float(bp[2 * cnt + j]);
            k.pri[j] = __uint_as_float(bp[3 * cnt + j]);
            k.chd[j] = (int32_t)bp[4 * cnt + j];
            k.act[j] = root ? j : (int32_t)bp[5 * cnt + j];
            k.pri64[j] = root ? ((const double *)(bp + rp_off))[j] : (double)k.pri[j];
        } else {
            k.vis[j] = 0; k.chd[j] = 0; k.act[j] = 0; k.vsum[j] = 0.f; k.rew[j] = 0.f; k.pri[j] = 0.f; k.pri64[j] = 0.0;
        }
    }
}

// exactly N children in an expansion block: 6N words as 16-byte loads (blocks are 64-byte aligned and padded)
template <int N>
__device__ inline void load_kids_static(const uint32_t *bp, Kids<N> &k) {
    constexpr int NV = (6 * N + 3) / 4;
    uint32_t w[NV * 4];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const uint4 q = reinterpret_cast<const uint4 *>(bp)[v];
        w[4 * v] = q.x; w[4 * v + 1] = q.y; w[4 * v + 2] = q.z; w[4 * v + 3] = q.w;
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        k.vis[j] = (int32_t)w[2 * j];
        k.vsum[j] = __uint_as_float(w[2 * j + 1]);
        k.rew[j] = __uint_as_float(w[2 * N + j]);
        k.pri[j] = __uint_as_float(w[3 * N + j]);
        k.chd[j] = (int32_t)w[4 * N + j];
        k.act[j] = (int32_t)w[5 * N + j];
        k.pri64[j] = (double)k.pri[j];
    }
}

// chance-flagged node: sample an outcome from the smoothed priors (mcts:247-255)
template <int N, class RNG>
__device__ inline int pick_chance(const Kids<N> &k, int cnt, RNG &rng) {
    float tmp[N];
    double q64[N];
#pragma unroll
    for (int j = 0; j < N; j++) { const float om = 1.0f - k.pri[j]; tmp[j] = om + 1e-12f; }
    const float s = np_sum<float, N>(tmp, cnt);
    const float r = fabsf((float)((double)s / (double)cnt));
#pragma unroll
    for (int j = 0; j < N; j++) tmp[j] = k.pri[j] + r;
    const float qs = np_sum<float, N>(tmp, cnt);
#pragma unroll
    for (int j = 0; j < N; j++) q64[j] = (double)(tmp[j] / qs);
    return sample_cdf<N>(q64, cnt, rng.random_sample());
}

// A chance-flagged node with TWO children picks child 1 iff t <= u, where u is the draw and t = cdf[0] / cdf[1] of
// pick_chance / sample_cdf: a function of the two priors alone.  The priors of a block never change after its expansion,
// so the specialised search kernel computes t ONCE, when it writes the block, instead of on every visit of the node
// (two float32 quotients and a float64 one leave the descent's dependent chain; a chance node is visited ~4 times in a
// 50-simulation search).  Same operations in the same order as pick_chance<2> + sample_cdf<2>: the pick is the same bit for bit.
__device__ inline double chance_threshold2(float p0, float p1) {
    float tmp[2] = {(1.0f - p0) + 1e-12f, (1.0f - p1) + 1e-12f};
    const float s = np_sum<float, 2>(tmp, 2);
    const float r = fabsf((float)((double)s / 2.0));
    tmp[0] = p0 + r; tmp[1] = p1 + r;
    const float qs = np_sum<float, 2>(tmp, 2);
    const double q0 = (double)(tmp[0] / qs), q1 = (double)(tmp[1] / qs);
    double acc = 0.0;
    acc += q0;
    const double c0 = acc;
    acc += q1;
    return c0 / acc;
}

// Correctly rounded x / n for a visit count n in [1, sims + 1] through a table of correctly rounded reciprocals:
// q = RN(x r) is a faithful quotient, e = x - q n is exact in an FMA, RN(q + e r) = RN(x / n) (Markstein's theorem; its
// one exception, a divisor whose significand is all ones, cannot be a small integer).  x = sqrt(N) pb_c prior is a
// normal number of moderate magnitude or exactly 0, so the correction term cannot underflow.  Three dependent
// operations instead of the ~14 of the IEEE sequence, on the longest dependent chain of a tree level.
// (tests/test_gpu_tree_parity.py::test_small_integer_division_is_correctly_rounded pins it against the IEEE quotient.)
__device__ inline double div_by_count(double x, int n, const double *r64) {
    const double r = r64[n], dn = (double)n;
    const double q = x * r;
    const double e = fma(-q, dn, x);
    return fma(e, r, q);
}

// The value term of a child before the MinMax normalisation: reward + discount * value() (mcts:239-241).  The backup that
// last touched the child has computed value() = value_sum / visits for the MinMax update; YV kernels keep the term it leads
// to beside the block (Params::thr_off) and the next descent reads it instead of dividing again -- the same two operands, the
// same three operations, one IEEE division less on the dependent chain of every decision level.
__device__ inline float value_term(float reward, float disc32, float qv) {
    const float dv = disc32 * qv;
    return reward + dv;
}
// decision-flagged node: pUCT argmax (mcts:235-243, 257-259)
// r64: reciprocal table (1/n at r64[n], n <= sims + 1) or nullptr (compile-time at every call site) for the IEEE division
template <int N, bool YV = false>
__device__ inline double puct_score(const Kids<N> &k, int j, double sp, bool norm, float mn, float span, float disc32,
                                    double u, const double *r64) {
    const int Nc = k.vis[j];
    const double prior_score = r64 ? div_by_count(sp * k.pri64[j], Nc + 1, r64) : (sp * k.pri64[j]) / (double)(Nc + 1);
    double value_score = 0.0;
    if (Nc > 0) {
        float x = YV ? k.yv[j] : value_term(k.rew[j], disc32, k.vsum[j] / (float)Nc);
        if (norm) { const float num = x - mn; x = num / span; }
        value_score = (double)x;
    }
    const double jitter = 1e-7 + (2e-7 - 1e-7) * u;
    return (prior_score + value_score) + jitter;
}
template <int N, bool YV = false, class RNG>
__device__ inline int pick_decision(const Kids<N> &k, int cnt, double sp, bool norm, float mn, float span, float disc32,
                                    RNG &rng, const double *r64 = nullptr) {
    double best = 0.0;
    int pick = 0;
    // the jitter words of all children at once when the child count is the (small) compile-time N: one staged-words
    // check per level instead of one per draw; draw order is unchanged (child 0's pair first)
    constexpr bool kBatch = N <= 4;
    uint32_t jw[kBatch ? 2 * N : 2];
    const bool batched = kBatch && cnt == N;
    if (batched) rng.template take<kBatch ? 2 * N : 2>(jw);
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (j < cnt) {
            const double u = batched ? RNG::to_double(jw[kBatch ? 2 * j : 0], jw[kBatch ? 2 * j + 1 : 1]) : rng.random_sample();
            const double score = puct_score<N, YV>(k, j, sp, norm, mn, span, disc32, u, r64);
            if (j == 0 || score >= best) { best = score; pick = j; }  // exact tie -> larger action
        }
    }
    return pick;
}

// The same decision for a two-child level when TWO lanes of a quad work on the tree: lane `me` (0 / 1 = the tree's
// own lane / its helper, two lanes further in the quad) scores child `me`, the scores are exchanged with a quad-perm
// DPP and both lanes take the same pick.  Every lane consumes the level's four random words (its copy of the
// tree's stream stays in step) and uses the pair belonging to its child.
__device__ inline double quad_partner(double v) {          // value held by lane ^ 2
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <bool YV = false, class RNG>
__device__ inline int pick_decision_pair(const Kids<2> &k, int me, double sp, bool norm, float mn, float span,
                                         float disc32, RNG &rng, const double *r64) {
    uint32_t jw[4];
    rng.template take<4>(jw);
    Kids<2> mine;                 // slot 0 = this lane's child
    mine.vis[0] = me ? k.vis[1] : k.vis[0];
    mine.vsum[0] = me ? k.vsum[1] : k.vsum[0];
    mine.rew[0] = me ? k.rew[1] : k.rew[0];
    mine.pri64[0] = me ? k.pri64[1] : k.pri64[0];
    if (YV) mine.yv[0] = me ? k.yv[1] : k.yv[0];
    const double u = RNG::to_double(me ? jw[2] : jw[0], me ? jw[3] : jw[1]);
    const double s_me = puct_score<2, YV>(mine, 0, sp, norm, mn, span, disc32, u, r64);
    const double s_other = quad_partner(s_me);
    const double s0 = me ? s_other : s_me, s1 = me ? s_me : s_other;
    return s1 >= s0 ? 1 : 0;      // exact tie -> larger action
}

// The FOUR-child root level on the same two lanes (round 5; below the root every block has two children and
// pick_decision_pair above applies): lane `me` scores children 2 me and 2 me + 1, keeps the better (ties to the larger
// action, as the sequential loop does), the two lanes exchange their best score and pick, and the pair {2, 3} wins ties
// against {0, 1}: the last maximum in action order, which is what `score >= best` in pick_decision leaves.  Both lanes
// consume the level's eight words.
__device__ inline int quad_partner(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false); }
template <bool YV = false, class RNG>
__device__ inline int pick_decision_pair(const Kids<4> &k, int me, double sp, bool norm, float mn, float span,
                                         float disc32, RNG &rng, const double *r64) {
    uint32_t jw[8];
    rng.template take<8>(jw);
    Kids<4> mine;                 // slots 0, 1 = this lane's children
#pragma unroll
    for (int j = 0; j < 2; j++) {
        mine.vis[j] = me ? k.vis[2 + j] : k.vis[j];
        mine.vsum[j] = me ? k.vsum[2 + j] : k.vsum[j];
        mine.rew[j] = me ? k.rew[2 + j] : k.rew[j];
        mine.pri64[j] = me ? k.pri64[2 + j] : k.pri64[j];
        if (YV) mine.yv[j] = me ? k.yv[2 + j] : k.yv[j];
    }
    const double u0 = RNG::to_double(me ? jw[4] : jw[0], me ? jw[5] : jw[1]);
    const double u1 = RNG::to_double(me ? jw[6] : jw[2], me ? jw[7] : jw[3]);
    const double sa = puct_score<4, YV>(mine, 0, sp, norm, mn, span, disc32, u0, r64);
    const double sb = puct_score<4, YV>(mine, 1, sp, norm, mn, span, disc32, u1, r64);
    const bool second = sb >= sa;
    const double best = second ? sb : sa;
    const int lp = second ? 1 : 0;
    const double obest = quad_partner(best);
    const int olp = quad_partner(lp);
    const double b01 = me ? obest : best, b23 = me ? best : obest;
    const int p01 = me ? olp : lp, p23 = me ? lp : olp;
    return b23 >= b01 ? 2 + p23 : p01;
}

// The path records of ONE tree inside the level-major global array: record i of tree t = path[i * B + t].  (The single-launch
// kernels keep a tree's records in LDS as a plain array and pass a pointer.)
struct PathCol {
    uint4 *p;
    size_t stride;
    __device__ __forceinline__ uint4 &operator[](int i) const { return p[(size_t)i * stride]; }
};
__device__ __forceinline__ PathCol path_col(const Params &P, int tree) { return PathCol{P.path + tree, (size_t)P.B}; }

// LUT: the reciprocal table of div_by_count follows the pb_c table (pbc_sqrt[sims + 2 + n] = 1 / n)
// PAIR (MAXA == 2, KS == 2, A == 2): two lanes per tree, see pick_decision_pair; `me` = 0 for the tree's lane (which
// alone writes the path records), 1 for its helper.  Chance levels are evaluated redundantly by both lanes.
// THR (KS == 2): chance levels compare the draw with the block's stored threshold (chance_threshold2, written by
// expand_backup_tree<..., THR>) instead of recomputing it.  YV (needs THR's two words per block): decision levels read the
// children's value terms as the last backup left them (value_term) instead of dividing value_sum by visits again.
template <int MAXA, int KS, bool STATS = true, bool LUT = false, bool PAIR = false, bool THR = false, bool YV = false, class RNG = Rng, class REC = uint4 *>
__device__ inline Leaf select_tree(const Params &P, int tree, RNG &rng, const TreeHdr &h, const double *pbc_sqrt,
                                   int &path_len_out, unsigned &n_dec, unsigned &n_chance, unsigned &n_children,
                                   REC rec, int me = 0) {
    static_assert(!THR || KS == 2, "stored chance thresholds: two children per block");
    static_assert(!YV || THR, "stored value terms live in the threshold words of decision-flagged blocks");
    constexpr bool RY = YV && MAXA <= 8;       // the root block has room for its children's value terms (Params::ry_off)
    constexpr int NK = KS > 0 ? KS : MAXA;     // register arrays of the expansion levels
    const int A = P.A, K = P.K;
    uint32_t *tb = tree_base(P, tree);
    const float mn = h.mn, mx = h.mx;
    const bool norm = mx > mn;
    const float span = mx - mn;
    int depth = 0, cur_visit = h.root_visit, action = 0, leaf_id = 0, parent_id = 0, c = 0;
    {   // ---- root level: decision-flagged, A children, float64 priors ----------------------------------------------
        Kids<MAXA> k;
        load_kids_dyn<MAXA>(tb, A, true, P.rp_off, k);
        if constexpr (RY) {
#pragma unroll
            for (int j = 0; j < MAXA; j++) k.yv[j] = j < A ? __uint_as_float(tb[P.ry_off + j]) : 0.f;
        }
        int pick;
        if constexpr (PAIR) pick = pick_decision_pair<RY>(k, me, pbc_sqrt[cur_visit], norm, mn, span, P.disc32, rng, LUT ? pbc_sqrt + P.sims + 2 : nullptr);
        else pick = pick_decision<MAXA, RY>(k, A, pbc_sqrt[cur_visit], norm, mn, span, P.disc32, rng, LUT ? pbc_sqrt + P.sims + 2 : nullptr);
        float pv = 0.f, pr = 0.f;
#pragma unroll
        for (int j = 0; j < MAXA; j++) if (j == pick) { c = k.chd[j]; cur_visit = k.vis[j]; action = j; pv = k.vsum[j]; pr = k.rew[j]; }
        if (!PAIR || me == 0) rec[0] = make_uint4((uint32_t)pick, (uint32_t)cur_visit, __float_as_uint(pv), __float_as_uint(pr));
        leaf_id = 1 + pick;
        depth = 1;
    }
    while (c != 0) {
        const int blk = c;
        const uint32_t *bp = tb + P.rb_words + (size_t)(blk - 1) * P.eb_words;
        Kids<NK> k;
        if (KS > 0) load_kids_static<NK>(bp, k);
        else load_kids_dyn<NK>(bp, K, false, 0, k);
        const int cnt = KS > 0 ? KS : K;
        int pick;
        if (depth_flag(depth)) {
            if constexpr (THR) {
                const double t = *reinterpret_cast<const double *>(tb + P.thr_off + (size_t)(blk - 1) * P.thr_stride);
                pick = (t <= rng.random_sample()) ? 1 : 0;
            } else {
                pick = pick_chance<NK>(k, cnt, rng);
            }
        } else {
            if constexpr (YV) {
                const uint2 y2 = *reinterpret_cast<const uint2 *>(tb + P.thr_off + (size_t)(blk - 1) * P.thr_stride);
                k.yv[0] = __uint_as_float(y2.x); k.yv[1] = __uint_as_float(y2.y);
            }
            if constexpr (PAIR) pick = pick_decision_pair<YV>(k, me, pbc_sqrt[cur_visit], norm, mn, span, P.disc32, rng, LUT ? pbc_sqrt + P.sims + 2 : nullptr);
            else pick = pick_decision<NK, YV>(k, cnt, pbc_sqrt[cur_visit], norm, mn, span, P.disc32, rng, LUT ? pbc_sqrt + P.sims + 2 : nullptr);
        }
        float pv = 0.f, pr = 0.f;
#pragma unroll
        for (int j = 0; j < NK; j++) if (j == pick) { c = k.chd[j]; cur_visit = k.vis[j]; action = k.act[j]; pv = k.vsum[j]; pr = k.rew[j]; }
        if (!PAIR || me == 0) rec[depth] = make_uint4((uint32_t)((blk << 8) | pick), (uint32_t)cur_visit, __float_as_uint(pv), __float_as_uint(pr));
        parent_id = leaf_id;
        leaf_id = 1 + A + (blk - 1) * K + pick;
        depth++;
    }
    path_len_out = depth;
    if (STATS) {   // level statistics in closed form (levels 0..depth-1 carry the flags F F T T F F T T ...): counters updated
        // inside the descent loop were placed in scratch memory by the compiler -- a memory round trip per level
        const unsigned d = (unsigned)depth, ch = 2u * (d >> 2) + ((d & 3u) > 2u ? (d & 3u) - 2u : 0u), dec = d - ch;
        n_chance += ch;
        n_dec += dec;
        n_children += (unsigned)A + (dec - 1u) * (unsigned)(KS > 0 ? KS : K);
    }
    Leaf L;
    L.leaf_id = leaf_id;
    L.parent_id = parent_id;
    L.action = action;
    L.branch = depth_flag(depth - 1);
    return L;
}

// ---------------------------------------------------------------------------------------------------------------
// Block-parallel selection (round 4; two children per expansion block).  What a level of the descent draws does not depend
// on the path: a decision-flagged node draws one uniform per child (A at the root, K below), a chance-flagged one draws one,
// and the flag is a function of the depth alone (depth_flag).  So the words level d reads sit at a FIXED offset behind the
// stream position the descent starts from (select_words), and the pick of every node -- given its block, its depth, the
// tree's MinMax bounds and those words -- can be computed without knowing whether the descent will come by: one lane per
// block evaluates ALL blocks of the tree at once (select_block: the arithmetic of select_tree's levels, term by term), and
// the descent itself shrinks to a pointer chase over one 16-bit word per block.  A wavefront instruction costs the same for
// 2 active lanes as for 64, so the ~5.5 dependent level evaluations of a descent become one.
// Word per block: depth << 9 (kept from the expansion) | ok << 8 | pick << 7 | next block (0: the picked child is a leaf):
// two actions, at most 126 simulations (7-bit depths and block indices; the launcher keeps longer searches off this path).
// ---------------------------------------------------------------------------------------------------------------
__device__ inline int select_words(int d, int A) {          // random words consumed by levels 0 .. d-1 (K = 2 below the root)
    if (d <= 0) return 0;
    const int ch = 2 * (d >> 2) + ((d & 3) > 2 ? (d & 3) - 2 : 0);     // chance-flagged levels among 0 .. d-1
    return 2 * A + (d - ch - 1) * 4 + ch * 2;
}
template <int N, bool YV, class RNG>
__device__ inline int pick_decision_words(const Kids<N> &k, int cnt, double sp, bool norm, float mn, float span, float disc32,
                                          const uint32_t *jw, const double *r64) {
    double best = 0.0;
    int pick = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (j < cnt) {
            const double u = RNG::to_double(jw[2 * j], jw[2 * j + 1]);
            const double score = puct_score<N, YV>(k, j, sp, norm, mn, span, disc32, u, r64);
            if (j == 0 || score >= best) { best = score; pick = j; }  // exact tie -> larger action
        }
    }
    return pick;
}
// the pick of block `b` (0 = root) whose node sits at `depth`; stage[used ..] = the words the descent starts from, `staged` of
// them valid.  Returns ok << 8 | pick << 7 | next, or 0 when the level's words lie beyond the staged window.
// LF ("loads first", round 5; the kernels whose trees are in GLOBAL memory): every word the block's branches read is requested
// with the children's fields -- see select_block_request.  Trees in LDS: measured slower (-0.3 %: four more registers, and reads
// every lane issues whether its branch wants them), so those instantiations keep the reads inside the branches.
#ifndef SMZ_SELECT_LOADS_FIRST
#define SMZ_SELECT_LOADS_FIRST 1
#endif
// What select_block reads of a block, as loaded (nothing is computed on it here: a conversion behind the loads would be a wait
// between this block's requests and the next one's -- the kernel requests the blocks of TWO passes before it decides the first).
struct BlockRaw {
    uint4 q[3];          // the block's twelve words (load_kids_static<2>'s)
    uint2 aux2;          // a chance-flagged block's threshold / a decision-flagged one's value terms -- the same eight bytes
    double rp0, rp1;     // the root's float64 priors (every lane reads them: one broadcast address)
};
// Inside select_block's branches each of these was a dependent round trip of its own behind the children's fields -- and the
// branches of a wavefront run one after the other: up to three serial L2 round trips per pass instead of one (4096 x 100:
// +0.8 %, profiles/r05_aa_select_loads_ab.txt).
template <int MAXA, bool YV>
__device__ inline void select_block_request(const Params &P, const uint32_t *tb, int b, BlockRaw &in) {
    constexpr bool RY = YV && MAXA <= 8;
    const bool root = b == 0;
    const uint32_t *bp = root ? tb : tb + P.rb_words + (size_t)(b - 1) * P.eb_words;
    const uint32_t *aux = root ? tb + (RY ? P.ry_off : 0) : tb + P.thr_off + (size_t)(b - 1) * P.thr_stride;
#pragma unroll
    for (int v = 0; v < 3; v++) in.q[v] = reinterpret_cast<const uint4 *>(bp)[v];
    in.aux2 = *reinterpret_cast<const uint2 *>(aux);
    const double *rpp = reinterpret_cast<const double *>(tb + P.rp_off);        // (8-byte aligned: rp_off is even)
    in.rp0 = rpp[0]; in.rp1 = rpp[1];
}
template <int MAXA, bool YV, class RNG>
__device__ inline uint32_t select_block_decide(const Params &P, int b, int depth, int root_visit, float mn, float mx,
                                               const uint32_t *stage, int used, int staged, const double *pbc_sqrt, const BlockRaw &in) {
    const int A = P.A;
    const bool chance = depth_flag(depth) != 0;
    const int w0 = used + select_words(depth, A), need = chance ? 2 : (b == 0 ? 2 * A : 4);
    if (w0 + need > staged) return 0u;
    const uint32_t *w = stage + w0;
    const bool norm = mx > mn;
    const float span = mx - mn;
    const double *r64 = pbc_sqrt + P.sims + 2;
    const bool root = b == 0;
    const uint32_t wd[12] = {in.q[0].x, in.q[0].y, in.q[0].z, in.q[0].w, in.q[1].x, in.q[1].y, in.q[1].z, in.q[1].w,
                             in.q[2].x, in.q[2].y, in.q[2].z, in.q[2].w};
    Kids<2> k;
#pragma unroll
    for (int j = 0; j < 2; j++) {                       // (load_kids_static<2>'s unpacking)
        k.vis[j] = (int32_t)wd[2 * j];
        k.vsum[j] = __uint_as_float(wd[2 * j + 1]);
        k.rew[j] = __uint_as_float(wd[4 + j]);
        k.pri[j] = __uint_as_float(wd[6 + j]);
        k.chd[j] = (int32_t)wd[8 + j];
        k.act[j] = (int32_t)wd[10 + j];
        k.pri64[j] = (double)k.pri[j];
    }
    int pick;
    if (chance) {
        pick = (__hiloint2double((int)in.aux2.y, (int)in.aux2.x) <= RNG::to_double(w[0], w[1])) ? 1 : 0;
    } else {
        if constexpr (YV) { k.yv[0] = __uint_as_float(in.aux2.x); k.yv[1] = __uint_as_float(in.aux2.y); }
        if (root) { k.pri64[0] = in.rp0; k.pri64[1] = in.rp1; }
        const int np = root ? root_visit : 1 + k.vis[0] + k.vis[1];
        pick = pick_decision_words<2, YV, RNG>(k, 2, pbc_sqrt[np], norm, mn, span, P.disc32, w, r64);
    }
    const int c = pick ? k.chd[1] : k.chd[0];
    return 0x100u | ((uint32_t)pick << 7) | (uint32_t)c;
}
template <int MAXA, bool YV, class RNG, bool LF = false>
__device__ inline uint32_t select_block(const Params &P, const uint32_t *tb, int b, int depth, int root_visit, float mn, float mx,
                                        const uint32_t *stage, int used, int staged, const double *pbc_sqrt) {
    static_assert(MAXA == 2, "one bit for the pick: the block-parallel selection is built for two actions");
    if constexpr (LF && SMZ_SELECT_LOADS_FIRST) {
        BlockRaw in;
        select_block_request<MAXA, YV>(P, tb, b, in);
        return select_block_decide<MAXA, YV, RNG>(P, b, depth, root_visit, mn, mx, stage, used, staged, pbc_sqrt, in);
    }
    constexpr bool RY = YV && MAXA <= 8;
    const int A = P.A;
    const bool chance = depth_flag(depth) != 0;
    const int w0 = used + select_words(depth, A), need = chance ? 2 : (b == 0 ? 2 * A : 4);
    if (w0 + need > staged) return 0u;
    const uint32_t *w = stage + w0;
    const bool norm = mx > mn;
    const float span = mx - mn;
    const double *r64 = pbc_sqrt + P.sims + 2;
    int pick = 0, c = 0;
    {
        // two actions: the root block has the expansion blocks' field offsets (A == K == 2), so the root is ONE code path with
        // the decision-flagged blocks -- its float64 priors, its value terms and its visit count come from their own places
        const bool root = b == 0;
        const uint32_t *bp = root ? tb : tb + P.rb_words + (size_t)(b - 1) * P.eb_words;
        const uint32_t *aux = root ? tb + (RY ? P.ry_off : 0) : tb + P.thr_off + (size_t)(b - 1) * P.thr_stride;
        Kids<2> k;
        load_kids_static<2>(bp, k);
        if (chance) {
            pick = (*reinterpret_cast<const double *>(aux) <= RNG::to_double(w[0], w[1])) ? 1 : 0;
        } else {
            if constexpr (YV) {
                const uint2 y2 = *reinterpret_cast<const uint2 *>(aux);
                k.yv[0] = __uint_as_float(y2.x); k.yv[1] = __uint_as_float(y2.y);
            }
            if (root) {
                const double *rp = reinterpret_cast<const double *>(tb + P.rp_off);
                k.pri64[0] = rp[0]; k.pri64[1] = rp[1];
            }
            // visits of the node that owns the block: its expansion + one per later descent through it, each of which went on to
            // one of its children -- what the sequential descent carries along as the picked child's count
            const int np = root ? root_visit : 1 + k.vis[0] + k.vis[1];
            pick = pick_decision_words<2, YV, RNG>(k, 2, pbc_sqrt[np], norm, mn, span, P.disc32, w, r64);
        }
        c = pick ? k.chd[1] : k.chd[0];
    }
    return 0x100u | ((uint32_t)pick << 7) | (uint32_t)c;
}
// The descent over the evaluated blocks, in two steps.  (1) select_chase, by the tree's lane: follow sel[] from the root --
// ONE dependent LDS read per level -- and leave (block << 8 | pick) of every level in `path`; returns the depth (0: a block on
// the way was not evaluated).  (2) select_record, one lane per level: the path record of level d (the picked child's visit
// count, value sum and reward, for the backup) from the block itself -- these reads are off the chain; select_leaf names the
// leaf from the path's last two entries.
// the leaf the path ends in, read back from the path (tracking it inside the chase loop instead costs the kernel its gain: the
// register allocation of the whole kernel changes -- 453 against 462 M on one box, profiles/r04_bps_ab.txt)
__device__ inline Leaf select_leaf(const Params &P, const uint32_t *tb, const uint16_t *path, int depth) {
    const int A = P.A;
    auto node = [&](int loc) { const int b = loc >> 8, pk = loc & 3; return b == 0 ? 1 + pk : 1 + A + (b - 1) * 2 + pk; };
    const int last = path[depth - 1], lb = last >> 8, lp = last & 3;
    Leaf L;
    L.leaf_id = node(last);
    L.parent_id = depth > 1 ? node(path[depth - 2]) : 0;
    L.action = lb == 0 ? lp : (int)tb[P.rb_words + (size_t)(lb - 1) * P.eb_words + 5 * 2 + lp];
    L.branch = depth_flag(depth - 1);
    return L;
}
__device__ inline int select_chase(const uint16_t *sel, uint16_t *path) {
    int b = 0, depth = 0;
    for (;;) {
        const uint32_t s = sel[b];
        if (!(s & 0x100u)) return 0;
        path[depth++] = (uint16_t)((b << 8) | ((s >> 7) & 1u));
        b = (int)(s & 127u);
        if (b == 0) return depth;
    }
}
template <class REC>
__device__ inline void select_record(const Params &P, const uint32_t *tb, const uint16_t *path, int d, REC rec) {
    const int loc = path[d], b = loc >> 8, pick = loc & 3;
    const uint32_t *bp = b == 0 ? tb : tb + P.rb_words + (size_t)(b - 1) * P.eb_words;
    const uint2 vv = *reinterpret_cast<const uint2 *>(bp + 2 * pick);                     // (visit, value_sum) of the picked child
    rec[d] = make_uint4((uint32_t)loc, vv.x, vv.y, bp[2 * (b == 0 ? P.A : 2) + pick]);
}
// YV kernels: the value term of the child in slot `sl` of block `b`, chosen at path level `level` (= the depth of the block's
// node), goes beside the block -- unless that node samples its children (the two words hold its threshold then, and a value
// term there would never be read).
__device__ inline void store_value_term(const Params &P, uint32_t *tb, int b, int sl, int level, float y) {
    if (b == 0) {
        if (P.ry_off >= 0) tb[P.ry_off + sl] = __float_as_uint(y);
    } else if (!depth_flag(level)) {
        tb[P.thr_off + (size_t)(b - 1) * P.thr_stride + sl] = __float_as_uint(y);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// expansion + backup (monte_carlo_tree_search.py:289-308); the leaf's hidden row is stored by the caller.
// The path records of the preceding select carry every visited node's (visit, value_sum, reward): the backup only
// STORES into the tree.  Returns the leaf's node id.
// ---------------------------------------------------------------------------------------------------------------
// EXPAND_ONLY: stop after the expansion and return the leaf reward through *leaf_reward_out -- the caller runs the
// backup with backup_levels_lanes (several lanes per tree).
template <int MAXA, int KS, bool EXPAND_ONLY = false, bool THR = false, bool YV = false, class RNG = Rng, class REC = const uint4 *>
__device__ inline int expand_backup_tree(const Params &P, int tree, RNG &rng, TreeHdr &h, const float *policy_row,
                                         float reward, float value, REC rec, float *leaf_reward_out = nullptr) {
    static_assert(!THR || KS == 2, "stored chance thresholds: two children per block");
    constexpr int CH = 8;   // path records fetched per round trip
    const int A = P.A, K = P.K;
    uint32_t *tb = tree_base(P, tree);
    const int len = h.path_len;
    const uint4 leaf_rec = rec[len - 1];
    const int leaf_loc = (int)leaf_rec.x;
    const int pflag = depth_flag(len - 1);       // flag of the leaf's parent (depth len-1; root is depth 0)
    // ---- expansion ---------------------------------------------------------------------------------------------
    float p[MAXA], pol[MAXA];
    double p64[MAXA];
    int32_t picks[MAXA];
    for (int a = 0; a < A; a++) pol[a] = policy_row[a];
    normalise_policy<MAXA>(pol, A, p);
    for (int a = 0; a < A; a++) p64[a] = (double)p[a];
    choice_noreplace<MAXA>(rng, p64, A, K, picks);
    sort_picks<MAXA>(picks, K);   // np.sort of the K picks
    const int e = h.n_exp;
    h.n_exp = e + 1;
    {
        uint32_t *nb = tb + P.rb_words + (size_t)e * P.eb_words;
        if (KS == 2) {   // 12 words at fixed offsets: three 16-byte stores
            float pj[2] = {0.f, 0.f};
            for (int a = 0; a < A; a++) { if (a == picks[0]) pj[0] = p[a]; if (a == picks[1]) pj[1] = p[a]; }
            uint4 *nb4 = reinterpret_cast<uint4 *>(nb);
            nb4[0] = make_uint4(0u, 0u, 0u, 0u);                                              // visit, value_sum
            nb4[1] = make_uint4(0u, 0u, __float_as_uint(pj[0]), __float_as_uint(pj[1]));      // reward, prior
            nb4[2] = make_uint4(0u, 0u, (uint32_t)picks[0], (uint32_t)picks[1]);              // child, action
            if (P.eb_words >= 16) nb4[3] = make_uint4(0u, 0u, 0u, 0u);                        // (padding: the block leaves as one whole 64-byte line;
                                                                                              //  LDS-resident trees pack blocks at 12 words)
            if constexpr (THR) {      // the new node sits at depth `len`: its children are sampled iff that depth is chance-flagged
                if (depth_flag(len))
                    *reinterpret_cast<double *>(tb + P.thr_off + (size_t)e * P.thr_stride) = chance_threshold2(pj[0], pj[1]);
            }
        } else if (KS == 4) {   // 24 words at fixed offsets: six 16-byte stores (stored chance thresholds as for K = 2 were measured:
                                // 413 against 415 M, profiles/r06_f_thr4_ab.txt -- three more stores per block cost what the visits save)
            float pj[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++) for (int a = 0; a < A; a++) if (a == picks[j]) pj[j] = p[a];
            uint4 *nb4 = reinterpret_cast<uint4 *>(nb);
            nb4[0] = make_uint4(0u, 0u, 0u, 0u);                                              // (visit, value_sum) x 4
            nb4[1] = make_uint4(0u, 0u, 0u, 0u);
            nb4[2] = make_uint4(0u, 0u, 0u, 0u);                                              // reward
            nb4[3] = make_uint4(__float_as_uint(pj[0]), __float_as_uint(pj[1]), __float_as_uint(pj[2]), __float_as_uint(pj[3]));   // prior
            nb4[4] = make_uint4(0u, 0u, 0u, 0u);                                              // child
            nb4[5] = make_uint4((uint32_t)picks[0], (uint32_t)picks[1], (uint32_t)picks[2], (uint32_t)picks[3]);                   // action
        } else {
            for (int j = 0; j < K; j++) {
                nb[2 * j] = 0u;                              // visit
                nb[2 * j + 1] = __float_as_uint(0.f);        // value_sum
                nb[2 * K + j] = __float_as_uint(0.f);        // reward
                float pj = 0.f;
                for (int a = 0; a < A; a++) if (a == picks[j]) pj = p[a];
                nb[3 * K + j] = __float_as_uint(pj);         // prior = un-renormalised p[a]
                nb[4 * K + j] = 0u;                          // child
                nb[5 * K + j] = (uint32_t)picks[j];          // action
            }
        }
    }
    const float leaf_reward = pflag ? reward : 0.0f;     // the afterstate branch never assigns one (mcts:338-342)
    {
        const int lb = leaf_loc >> 8, ls = leaf_loc & 0xff;
        const int lc = (lb == 0) ? A : K;
        uint32_t *lp = block_ptr(P, tb, lb);
        lp[4 * lc + ls] = (uint32_t)(e + 1);             // leaf.children now live in expansion e
        lp[2 * lc + ls] = __float_as_uint(leaf_reward);
    }
    if constexpr (EXPAND_ONLY) {
        *leaf_reward_out = leaf_reward;
        return loc_node_id(P, leaf_loc);
    }
    // ---- backup, leaf -> root: stores only -----------------------------------------------------------------------
    float v = value;
    float mn = h.mn, mx = h.mx;
    for (int i0 = len - 1; i0 >= 0; i0 -= CH) {
        uint4 r4[CH];
#pragma unroll
        for (int q = 0; q < CH; q++) r4[q] = (i0 - q >= 0) ? rec[i0 - q] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int q = 0; q < CH; q++) {
            if (i0 - q >= 0) {
                const uint4 e4 = r4[q];
                const int b = (int)e4.x >> 8, sl = (int)e4.x & 0xff;
                const int cnt = (b == 0) ? A : K;
                uint32_t *np = block_ptr(P, tb, b) + 2 * sl;
                const float r = (i0 - q == len - 1) ? leaf_reward : __uint_as_float(e4.w);
                const float nvs = __uint_as_float(e4.z) + v;
                const int nvc = (int)e4.y + 1;
                *reinterpret_cast<uint2 *>(np) = make_uint2((uint32_t)nvc, __float_as_uint(nvs));     // (visit, value_sum)
                const float qv = nvs / (float)nvc;
                if (qv > mx) mx = qv;
                if (qv < mn) mn = qv;
                if constexpr (YV) store_value_term(P, tb, b, sl, i0 - q, value_term(r, P.disc32, qv));
                const float dv = P.disc32 * v;
                v = r + dv;
            }
        }
    }
    {   // the root itself (reward 0)
        const float nvs = h.root_value_sum + v;
        const int nvc = h.root_visit + 1;
        h.root_value_sum = nvs;
        h.root_visit = nvc;
        const float qv = nvs / (float)nvc;
        if (qv > mx) mx = qv;
        if (qv < mn) mn = qv;
    }
    h.mn = mn;
    h.mx = mx;
    return loc_node_id(P, leaf_loc);
}

// The backup of expand_backup_tree with one lane per path level: lane (t + TPW * j) of the wave handles level
// (base - j) of tree slot t, eight levels per pass.  The value recursion v <- r + disc v is a two-operation chain that
// every lane repeats; the per-level work (visit / value_sum stores, the MinMax quotient) is the same arithmetic as in
// the one-lane loop, so the tree and the bounds come out bit-identical; min / max are order-free.  Called by lanes
// [0, 8 TPW); returns the tree's new MinMax bounds and the value arriving at the root in the lanes j == 0.
template <int SH>
__device__ inline void minmax_shl(float &mn, float &mx) {      // mn = min(mn, mn of lane + SH), mx likewise (same 16-lane row)
    const float smn = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(mn), __float_as_int(mn), 0x100 + SH, 0xf, 0xf, false));
    const float smx = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(mx), __float_as_int(mx), 0x100 + SH, 0xf, 0xf, false));
    mn = fminf(mn, smn);
    mx = fmaxf(mx, smx);
}
template <int TPW, bool YV = false>
__device__ inline void backup_levels_lanes(const Params &P, int tree, int j, int len, float value, float leaf_reward,
                                           const uint4 *rec, float &mn, float &mx, float &v_root) {
    const int A = P.A, K = P.K;
    uint32_t *tb = tree_base(P, tree);
    float v = value;
    for (int base = len - 1; base >= 0; base -= 8) {
        float vin = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = base - q;
            if (i >= 0) {                                   // uniform over the lanes of a tree
                if (q == j) vin = v;
                const float r = (i == len - 1) ? leaf_reward : __uint_as_float(rec[i].w);
                const float dv = P.disc32 * v;
                v = r + dv;
            }
        }
        const int i = base - j;
        if (i >= 0) {
            const uint4 e4 = rec[i];
            const int b = (int)e4.x >> 8, sl = (int)e4.x & 0xff;
            const int cnt = (b == 0) ? A : K;
            uint32_t *np = block_ptr(P, tb, b) + 2 * sl;
            const float nvs = __uint_as_float(e4.z) + vin;
            const int nvc = (int)e4.y + 1;
            *reinterpret_cast<uint2 *>(np) = make_uint2((uint32_t)nvc, __float_as_uint(nvs));         // (visit, value_sum)
            const float qv = nvs / (float)nvc;
            if (qv > mx) mx = qv;
            if (qv < mn) mn = qv;
            if constexpr (YV)
                store_value_term(P, tb, b, sl, i, value_term((i == len - 1) ? leaf_reward : __uint_as_float(e4.w), P.disc32, qv));
        }
    }
    v_root = v;
    // min / max over the tree's eight lanes (stride TPW inside one 16-lane row): shift-left reductions bring them to j == 0
    // (on the DPP crossbar: row_shl by a constant inside the 16-lane row, a lane without a source keeps its own value)
    static_assert(8 * TPW <= 16, "the tree's eight lanes must sit in one DPP row");
    minmax_shl<TPW>(mn, mx);
    minmax_shl<2 * TPW>(mn, mx);
    minmax_shl<4 * TPW>(mn, mx);
}

// ---------------------------------------------------------------------------------------------------------------
// post-search policy / action (game.py:179-232)
// ---------------------------------------------------------------------------------------------------------------
template <int MAXA, class RNG>
__device__ inline void act_tree(const Params &P, int tree, RNG &rng, double temperature, int32_t *action_out,
                                double *policy_out, double *child_visits_out, float *root_value_out) {
    const int A = P.A;
    const uint32_t *rb = tree_base(P, tree);
    const double *rp = (const double *)(rb + P.rp_off);
    double pol[MAXA], vis[MAXA], pri[MAXA];
    int32_t vc[MAXA];
    for (int a = 0; a < A; a++) { vc[a] = (int32_t)rb[2 * a]; vis[a] = (double)vc[a]; pri[a] = rp[a]; }
    const double vsum = np_sum<double, MAXA>(vis, A);
    const bool from_visits = !(vsum <= 1.0);
    for (int a = 0; a < A; a++) pol[a] = from_visits ? vis[a] : pri[a];
    if (temperature >= 0.3) {
        for (int a = 0; a < A; a++)
            pol[a] = (from_visits && P.pow_table) ? P.pow_table[vc[a]] : smz_glibc_pow(pol[a], 1.0 / temperature);   // (numpy's ** = libm's pow)
    }
    const double ps = np_sum<double, MAXA>(pol, A);
    for (int a = 0; a < A; a++) pol[a] = pol[a] / ps;
    bool all_equal = true;
    for (int a = 1; a < A; a++) all_equal = all_equal && (pol[a] == pol[0]);
    int pick = 0;
    if (temperature > 0.1 || all_equal) {
        pick = sample_cdf<MAXA>(pol, A, rng.random_sample());
    } else {
        for (int a = 1; a < A; a++) if (pol[a] > pol[pick]) pick = a;  // np.argmax: first maximum
    }
    if (action_out) action_out[tree] = pick;
    if (policy_out) for (int a = 0; a < A; a++) policy_out[(size_t)tree * A + a] = pol[a];
    if (child_visits_out) {
        if (vsum >= 3.0) {
            for (int a = 0; a < A; a++) child_visits_out[(size_t)tree * A + a] = vis[a] / vsum;
        } else {
            const double s = np_sum<double, MAXA>(pri, A);
            for (int a = 0; a < A; a++) child_visits_out[(size_t)tree * A + a] = pri[a] / s;
        }
    }
    if (root_value_out) {
        const TreeHdr h = P.hdr[tree];
        root_value_out[tree] = h.root_visit ? h.root_value_sum / (float)h.root_visit : 0.0f;
    }
}

}  // namespace smz
