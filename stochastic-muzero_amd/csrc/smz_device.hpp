// smz_device.hpp -- device-side building blocks of the gfx950 search kernels.
//
// Everything here is per-tree scalar logic executed by ONE lane per tree (64 trees per wavefront); the wave-wide
// parts (row gathers/scatters of hidden states) live in smz_kernels.hip.  The arithmetic follows, operation by
// operation and with the same float32/float64 roundings, the reference's monte_carlo_tree_search.py and the numpy
// legacy RandomState it draws from.  This translation unit MUST be compiled with -ffp-contract=off: a fused
// multiply-add would change the roundings the reference performs separately.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smz {

constexpr int kMtN = 624;
constexpr int kMtM = 397;
constexpr int kWave = 64;

struct TreeHdr {       // one 16-byte record per tree
    int32_t alloc;     // next free node index
    int32_t path_len;  // nodes on the recorded search path
    float mn, mx;      // MinMaxStats (monte_carlo_tree_search.py:24-36)
};

// Kernel parameter block (by value).  Node fields are structure-of-arrays, [B][N] each; children of a node are
// contiguous (child_base .. child_base + count), count = A for the root and K for every other node.
struct Params {
    int32_t B, A, K, S, N, P, sims;
    float disc32;   // float32(discount): python float * np.float32 -> float32 under NEP 50 (mcts:239, :308)
    float keep32;   // float32(1 - root_exploration_fraction) (mcts:224-225)
    double frac, alpha;
    int32_t *visit;
    float *value_sum, *reward, *prior;
    int32_t *child_base, *action;
    float *hidden;          // [B][N][S]
    double *root_prior;     // [B][A]  float64 priors of the root children (after noise)
    TreeHdr *hdr;           // [B]
    int32_t *path;          // [B][P]
    uint32_t *mt;           // [B][624]
    int32_t *rng_pos;       // [B]  (ready << 16) | idx
    const double *pbc_sqrt; // [sims+2]  sqrt(n) * pb_c(n)
    const double *pow_table;  // [sims+1] or nullptr
    unsigned long long *stats;  // [4] or nullptr
};

// ---------------------------------------------------------------------------------------------------------------
// numpy legacy RandomState: MT19937 advanced one word at a time.
// The block regeneration numpy performs every 624 draws is equivalent to twisting word i in place when draw i is
// requested (word i needs old[i], old[i+1] and old-or-new[i+397 mod 624] exactly as the in-place block loop has
// them), which removes the 624-iteration divergent refill.  `ready` counts words at idx.. that are ALREADY
// twisted (non-zero only right after importing a numpy state whose pos < 624).
// ---------------------------------------------------------------------------------------------------------------
struct Rng {
    uint32_t *mt;
    int idx, ready;
    __device__ void load(uint32_t *state, int packed) { mt = state; idx = packed & 0xffff; ready = packed >> 16; }
    __device__ int pack() const { return (ready << 16) | idx; }
    __device__ uint32_t next32() {
        const int i = idx;
        uint32_t y;
        if (ready > 0) {
            y = mt[i];
            --ready;
        } else {
            const int i1 = (i + 1 == kMtN) ? 0 : i + 1;
            int im = i + kMtM;
            if (im >= kMtN) im -= kMtN;
            const uint32_t a = mt[i], b = mt[i1], c = mt[im];
            const uint32_t t = (a & 0x80000000u) | (b & 0x7fffffffu);
            y = c ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
            mt[i] = y;
        }
        idx = (i + 1 == kMtN) ? 0 : i + 1;
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    // RandomState.random_sample(): (a * 2^26 + b) / 2^53 with a = 27 bits, b = 26 bits
    __device__ double random_sample() {
        const int32_t a = (int32_t)(next32() >> 5);
        const int32_t b = (int32_t)(next32() >> 6);
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
};

// ndarray.sum() of n <= 128 contiguous elements: sequential below 8, else numpy's 8-lane unrolled pairwise block.
template <typename T, int MAXA>
__device__ inline T np_sum(const T *a, int n) {
    if (MAXA < 8 || n < 8) {
        T r = (T)0;
        for (int i = 0; i < n; i++) r += a[i];
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
    for (int i = 0; i < n; i++) { acc += p[i]; cdf[i] = acc; }
    const double last = cdf[n - 1];
    int k = 0;
    for (int i = 0; i < n; i++) k += ((cdf[i] / last) <= u) ? 1 : 0;
    return k;
}

// RandomState.choice(n, size, p=p, replace=False) -- picks in draw order.  `p` is clobbered.
template <int MAXA>
__device__ inline void choice_noreplace(Rng &rng, double *p, int n, int size, int32_t *out) {
    double cdf[MAXA], x[MAXA];
    int32_t cand[MAXA];
    int n_uniq = 0;
    while (n_uniq < size) {
        const int m = size - n_uniq;
        for (int i = 0; i < m; i++) x[i] = rng.random_sample();
        for (int i = 0; i < n_uniq; i++) p[out[i]] = 0.0;
        double acc = 0.0;
        for (int i = 0; i < n; i++) { acc += p[i]; cdf[i] = acc; }
        const double last = cdf[n - 1];
        for (int i = 0; i < n; i++) cdf[i] = cdf[i] / last;
        for (int i = 0; i < m; i++) {
            int k = 0;
            for (int j = 0; j < n; j++) k += (cdf[j] <= x[i]) ? 1 : 0;
            cand[i] = k;
        }
        for (int i = 0; i < m; i++) {
            bool dup = false;
            for (int j = 0; j < i; j++) dup = dup || (cand[j] == cand[i]);
            if (!dup) out[n_uniq++] = cand[i];
        }
    }
}

// legacy_standard_gamma for shape <= 1 (numpy/random/src/legacy/legacy-distributions.c); log/pow are the device
// library's (correct to < 1 ulp, not guaranteed bit-identical to glibc's -- see DESIGN.md "Dirichlet noise").
__device__ inline double legacy_gamma(Rng &rng, double shape) {
    if (shape == 1.0) return -log(1.0 - rng.random_sample());
    if (shape == 0.0) return 0.0;
    for (;;) {
        const double U = rng.random_sample();
        const double V = -log(1.0 - rng.random_sample());
        if (U <= 1.0 - shape) {
            const double X = pow(U, 1.0 / shape);
            if (X <= V) return X;
        } else {
            const double Y = -log((1.0 - U) / shape);
            const double X = pow(1.0 - shape + shape * Y, 1.0 / shape);
            if (X <= (V + Y)) return X;
        }
    }
}

// p = (policy + 1e-12) / sum, float32 (monte_carlo_tree_search.py:205-206, 291-292)
template <int MAXA>
__device__ inline void normalise_policy(const float *policy, int A, float *p) {
    for (int a = 0; a < A; a++) p[a] = policy[a] + 1e-12f;
    const float s = np_sum<float, MAXA>(p, A);
    for (int a = 0; a < A; a++) p[a] = p[a] / s;
}

__device__ inline int depth_flag(int depth) { return (depth >> 1) & 1; }  // F F T T ...

// ---------------------------------------------------------------------------------------------------------------
// root (monte_carlo_tree_search.py:179-225)
// ---------------------------------------------------------------------------------------------------------------
template <int MAXA>
__device__ inline void root_init_tree(const Params &P, int tree, Rng &rng, const float *policy_row,
                                      const double *noise_override_row, bool train) {
    const int A = P.A;
    const size_t nb = (size_t)tree * P.N;
    float p[MAXA];
    double p64[MAXA];
    int32_t picks[MAXA];
    float pol[MAXA];
    for (int a = 0; a < A; a++) pol[a] = policy_row[a];
    normalise_policy<MAXA>(pol, A, p);
    for (int a = 0; a < A; a++) p64[a] = (double)p[a];
    choice_noreplace<MAXA>(rng, p64, A, A, picks);  // sorted result is 0..A-1; only the draws matter (mcts:208)
    P.visit[nb] = 0;
    P.value_sum[nb] = 0.f;
    P.reward[nb] = 0.f;
    P.prior[nb] = 0.f;
    P.child_base[nb] = 1;
    P.action[nb] = 0;
    for (int a = 0; a < A; a++) {
        const size_t c = nb + 1 + a;
        P.visit[c] = 0;
        P.value_sum[c] = 0.f;
        P.reward[c] = 0.f;
        P.prior[c] = p[a];
        P.child_base[c] = 0;
        P.action[c] = a;
    }
    double *rp = P.root_prior + (size_t)tree * A;
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
    h.alloc = 1 + A;
    h.path_len = 0;
    h.mn = __builtin_inff();
    h.mx = -__builtin_inff();
    P.hdr[tree] = h;
}

// ---------------------------------------------------------------------------------------------------------------
// selection (monte_carlo_tree_search.py:228-267)
// ---------------------------------------------------------------------------------------------------------------
struct Leaf {
    int32_t leaf, parent, action, branch;
};

template <int MAXA>
__device__ inline Leaf select_tree(const Params &P, int tree, Rng &rng, float mn, float mx, int &path_len_out,
                                   unsigned &n_dec, unsigned &n_chance, unsigned &n_children) {
    const int A = P.A, K = P.K;
    const size_t nb = (size_t)tree * P.N;
    int32_t *path = P.path + (size_t)tree * P.P;
    int node = 0, depth = 0, len = 1, parent = 0;
    path[0] = 0;
    int cb = P.child_base[nb];
    const bool norm = mx > mn;
    const float span = mx - mn;
    while (cb != 0) {
        const int cnt = (node == 0) ? A : K;
        int pick = 0;
        if (depth_flag(depth)) {
            // chance-flagged: sample an outcome from the smoothed priors (mcts:247-255)
            float pr[MAXA], tmp[MAXA];
            double q64[MAXA];
            for (int j = 0; j < cnt; j++) pr[j] = P.prior[nb + cb + j];
            for (int j = 0; j < cnt; j++) { const float om = 1.0f - pr[j]; tmp[j] = om + 1e-12f; }
            const float s = np_sum<float, MAXA>(tmp, cnt);
            const float r = fabsf((float)((double)s / (double)cnt));
            for (int j = 0; j < cnt; j++) tmp[j] = pr[j] + r;
            const float qs = np_sum<float, MAXA>(tmp, cnt);
            for (int j = 0; j < cnt; j++) q64[j] = (double)(tmp[j] / qs);
            pick = sample_cdf<MAXA>(q64, cnt, rng.random_sample());
            n_chance++;
        } else {
            // decision-flagged: pUCT argmax (mcts:235-243, 257-259)
            const int Np = P.visit[nb + node];
            const double sp = P.pbc_sqrt[Np];
            double best = 0.0;
            for (int j = 0; j < cnt; j++) {
                const size_t c = nb + cb + j;
                const int Nc = P.visit[c];
                const double prior = (node == 0) ? P.root_prior[(size_t)tree * A + j] : (double)P.prior[c];
                const double prior_score = (sp * prior) / (double)(Nc + 1);
                double value_score = 0.0;
                if (Nc > 0) {
                    const float qv = P.value_sum[c] / (float)Nc;
                    const float dv = P.disc32 * qv;
                    float x = P.reward[c] + dv;
                    if (norm) { const float num = x - mn; x = num / span; }
                    value_score = (double)x;
                }
                const double jitter = 1e-7 + (2e-7 - 1e-7) * rng.random_sample();
                const double score = (prior_score + value_score) + jitter;
                if (j == 0 || score >= best) { best = score; pick = j; }  // exact tie -> larger action
            }
            n_dec++;
            n_children += (unsigned)cnt;
        }
        parent = node;
        node = cb + pick;
        depth++;
        path[len++] = node;
        cb = P.child_base[nb + node];
    }
    path_len_out = len;
    Leaf L;
    L.leaf = node;
    L.parent = parent;
    L.action = P.action[nb + node];
    L.branch = depth_flag(depth - 1);
    return L;
}

// ---------------------------------------------------------------------------------------------------------------
// expansion + backup (monte_carlo_tree_search.py:289-308); the leaf's hidden row is stored by the caller.
// ---------------------------------------------------------------------------------------------------------------
template <int MAXA>
__device__ inline void expand_backup_tree(const Params &P, int tree, Rng &rng, TreeHdr &h, const float *policy_row,
                                          float reward, float value) {
    const int A = P.A, K = P.K;
    const size_t nb = (size_t)tree * P.N;
    const int32_t *path = P.path + (size_t)tree * P.P;
    const int len = h.path_len;
    const int leaf = path[len - 1];
    const int pflag = depth_flag(len - 2);
    float p[MAXA], pol[MAXA];
    double p64[MAXA];
    int32_t picks[MAXA];
    for (int a = 0; a < A; a++) pol[a] = policy_row[a];
    normalise_policy<MAXA>(pol, A, p);
    for (int a = 0; a < A; a++) p64[a] = (double)p[a];
    choice_noreplace<MAXA>(rng, p64, A, K, picks);
    for (int i = 1; i < K; i++) {  // np.sort of the K picks
        const int32_t x = picks[i];
        int j = i - 1;
        while (j >= 0 && picks[j] > x) { picks[j + 1] = picks[j]; j--; }
        picks[j + 1] = x;
    }
    const int cb = h.alloc;
    h.alloc = cb + K;
    P.child_base[nb + leaf] = cb;
    P.reward[nb + leaf] = pflag ? reward : 0.0f;  // the afterstate branch never assigns a reward (mcts:338-342)
    for (int j = 0; j < K; j++) {
        const size_t c = nb + cb + j;
        P.visit[c] = 0;
        P.value_sum[c] = 0.f;
        P.reward[c] = 0.f;
        P.prior[c] = p[picks[j]];
        P.child_base[c] = 0;
        P.action[c] = picks[j];
    }
    float v = value;
    float mn = h.mn, mx = h.mx;
    for (int i = len - 1; i >= 0; i--) {
        const size_t n = nb + path[i];
        const float vs = P.value_sum[n] + v;
        const int vc = P.visit[n] + 1;
        P.value_sum[n] = vs;
        P.visit[n] = vc;
        const float qv = vs / (float)vc;
        if (qv > mx) mx = qv;
        if (qv < mn) mn = qv;
        const float dv = P.disc32 * v;
        v = P.reward[n] + dv;
    }
    h.mn = mn;
    h.mx = mx;
}

// ---------------------------------------------------------------------------------------------------------------
// post-search policy / action (game.py:179-232)
// ---------------------------------------------------------------------------------------------------------------
template <int MAXA>
__device__ inline void act_tree(const Params &P, int tree, Rng &rng, double temperature, int32_t *action_out,
                                double *policy_out, double *child_visits_out, float *root_value_out) {
    const int A = P.A;
    const size_t nb = (size_t)tree * P.N;
    double pol[MAXA], vis[MAXA], pri[MAXA];
    int32_t vc[MAXA];
    for (int a = 0; a < A; a++) { vc[a] = P.visit[nb + 1 + a]; vis[a] = (double)vc[a]; pri[a] = P.root_prior[(size_t)tree * A + a]; }
    const double vsum = np_sum<double, MAXA>(vis, A);
    const bool from_visits = !(vsum <= 1.0);
    for (int a = 0; a < A; a++) pol[a] = from_visits ? vis[a] : pri[a];
    if (temperature >= 0.3) {
        for (int a = 0; a < A; a++)
            pol[a] = (from_visits && P.pow_table) ? P.pow_table[vc[a]] : pow(pol[a], 1.0 / temperature);
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
        const int rv = P.visit[nb];
        root_value_out[tree] = rv ? P.value_sum[nb] / (float)rv : 0.0f;
    }
}

}  // namespace smz
