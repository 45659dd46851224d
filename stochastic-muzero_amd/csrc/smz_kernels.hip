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
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/smz.h"
#include "smz_device.hpp"

using namespace smz;

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
__global__ void __launch_bounds__(kWave) k_seed(Params P, const uint64_t *seeds) {
    const int tree = blockIdx.x * kWave + threadIdx.x;
    if (tree >= P.B) return;
    uint32_t s = (uint32_t)(seeds[tree] & 0xffffffffull);
    uint32_t *mt = P.mt + (size_t)tree * kMtN;
    for (int i = 0; i < kMtN; i++) {
        mt[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    P.rng_pos[tree] = 0;  // idx 0, nothing pre-twisted: the first draw twists word 0 (== numpy pos 624)
}

// Moves one row per tree of this wave.  src_row/dst_row are per-lane row pointers (of the lane's own tree);
// rows are handed around with ds_bpermute so that `lpr` consecutive lanes move one row.
__device__ inline void wave_copy_rows(const float *src_row, float *dst_row, bool valid, int width) {
    const RowGeom g = row_geom(width);
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane / g.lpr, li = lane % g.lpr;
    const unsigned long long s64 = (unsigned long long)src_row, d64 = (unsigned long long)dst_row;
    for (int j0 = 0; j0 < kWave; j0 += g.rows_per_iter) {
        const int j = j0 + sub;
        const unsigned long long sj = __shfl(s64, j), dj = __shfl(d64, j);
        const int vj = __shfl((int)valid, j);
        if (vj) {
            const float *s = (const float *)sj;
            float *d = (float *)dj;
            for (int i = li; i < width; i += g.lpr) d[i] = s[i];
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
    for (int j0 = 0; j0 < kWave; j0 += g.rows_per_iter) {
        const int j = j0 + sub;
        const int pj = __shfl(parent, j), aj = __shfl(act, j), vj = __shfl((int)valid, j);
        if (vj) {
            const int t = tree0 + j;
            const float *src = P.hidden + ((size_t)t * P.N + pj) * S;
            for (int i = li; i < g.width; i += g.lpr) {
                const float v = (i < S) ? src[i] : ((i - S) == aj ? 1.0f : 0.0f);
                if (mlp_input) mlp_input[(size_t)t * W + i] = v;
                if (parent_hidden && i < S) parent_hidden[(size_t)t * S + i] = v;
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

template <int MAXA>
__global__ void __launch_bounds__(kWave) k_root_init(Params P, const float *hidden, const float *policy,
                                                     const double *noise_override, int train) {
    const int tree = blockIdx.x * kWave + threadIdx.x;
    const bool valid = tree < P.B;
    if (valid) {
        Rng rng;
        rng.load(P.mt + (size_t)tree * kMtN, P.rng_pos[tree]);
        root_init_tree<MAXA>(P, tree, rng, policy + (size_t)tree * P.A,
                             noise_override ? noise_override + (size_t)tree * P.A : nullptr, train != 0);
        P.rng_pos[tree] = rng.pack();
    }
    if (P.S > 0 && hidden) {
        const int t = valid ? tree : 0;
        wave_copy_rows(hidden + (size_t)t * P.S, P.hidden + (size_t)t * P.N * P.S, valid, P.S);
    }
}

template <int MAXA>
__device__ inline void select_phase(const Params &P, int tree, bool valid, Rng &rng, TreeHdr &h, float *parent_hidden,
                                    int32_t *last_action, uint8_t *branch, float *mlp_input) {
    Leaf L = {0, 0, 0, 0};
    unsigned n_dec = 0, n_chance = 0, n_children = 0;
    if (valid) {
        int len = 0;
        L = select_tree<MAXA>(P, tree, rng, h.mn, h.mx, len, n_dec, n_chance, n_children);
        h.path_len = len;
        if (last_action) last_action[tree] = L.action;
        if (branch) branch[tree] = (uint8_t)L.branch;
    }
    if (P.S > 0 && (parent_hidden || mlp_input))
        wave_gather_inputs(P, tree, valid, L.parent, L.action, parent_hidden, mlp_input);
    wave_add_stats(P.stats, n_dec, n_chance, valid ? 1u : 0u, n_children);
}

template <int MAXA>
__global__ void __launch_bounds__(kWave) k_select(Params P, float *parent_hidden, int32_t *last_action,
                                                  uint8_t *branch, float *mlp_input) {
    const int tree = blockIdx.x * kWave + threadIdx.x;
    const bool valid = tree < P.B;
    Rng rng;
    TreeHdr h = {0, 0, 0.f, 0.f};
    if (valid) {
        rng.load(P.mt + (size_t)tree * kMtN, P.rng_pos[tree]);
        h = P.hdr[tree];
    }
    select_phase<MAXA>(P, tree, valid, rng, h, parent_hidden, last_action, branch, mlp_input);
    if (valid) {
        P.rng_pos[tree] = rng.pack();
        P.hdr[tree] = h;
    }
}

template <int MAXA, bool FUSE_SELECT>
__global__ void __launch_bounds__(kWave) k_expand_backup(Params P, const float *hidden, const float *reward,
                                                         const float *policy, const float *value,
                                                         float *parent_hidden, int32_t *last_action, uint8_t *branch,
                                                         float *mlp_input) {
    const int tree = blockIdx.x * kWave + threadIdx.x;
    const bool valid = tree < P.B;
    Rng rng;
    TreeHdr h = {0, 0, 0.f, 0.f};
    int leaf = 0;
    if (valid) {
        rng.load(P.mt + (size_t)tree * kMtN, P.rng_pos[tree]);
        h = P.hdr[tree];
        leaf = P.path[(size_t)tree * P.P + h.path_len - 1];
        expand_backup_tree<MAXA>(P, tree, rng, h, policy + (size_t)tree * P.A, reward ? reward[tree] : 0.0f,
                                 value[tree]);
    }
    if (P.S > 0 && hidden) {
        const int t = valid ? tree : 0;
        wave_copy_rows(hidden + (size_t)t * P.S, P.hidden + ((size_t)t * P.N + leaf) * P.S, valid, P.S);
    }
    if (FUSE_SELECT) {
        // the leaf rows just written by other lanes of this wave may be the next parent rows
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        select_phase<MAXA>(P, tree, valid, rng, h, parent_hidden, last_action, branch, mlp_input);
    }
    if (valid) {
        P.rng_pos[tree] = rng.pack();
        P.hdr[tree] = h;
    }
}

__global__ void __launch_bounds__(kWave) k_root_stats(Params P, int32_t *visits, double *priors, float *root_value,
                                                      float *child_reward) {
    const int tree = blockIdx.x * kWave + threadIdx.x;
    if (tree >= P.B) return;
    const size_t nb = (size_t)tree * P.N;
    for (int a = 0; a < P.A; a++) {
        if (visits) visits[(size_t)tree * P.A + a] = P.visit[nb + 1 + a];
        if (priors) priors[(size_t)tree * P.A + a] = P.root_prior[(size_t)tree * P.A + a];
        if (child_reward) child_reward[(size_t)tree * P.A + a] = P.reward[nb + 1 + a];
    }
    if (root_value) {
        const int rv = P.visit[nb];
        root_value[tree] = rv ? P.value_sum[nb] / (float)rv : 0.0f;
    }
}

template <int MAXA>
__global__ void __launch_bounds__(kWave) k_act(Params P, double temperature, int32_t *action, double *policy,
                                               double *child_visits, float *root_value) {
    const int tree = blockIdx.x * kWave + threadIdx.x;
    if (tree >= P.B) return;
    Rng rng;
    rng.load(P.mt + (size_t)tree * kMtN, P.rng_pos[tree]);
    act_tree<MAXA>(P, tree, rng, temperature, action, policy, child_visits, root_value);
    P.rng_pos[tree] = rng.pack();
}

// ---- head epilogues ------------------------------------------------------------------------------------------------
// inverse_transform_with_support on one row held by one lane (muzero_model.py:575-591)
__device__ inline float support_decode_row(const float *row, int S) {
    float m = row[0];
    for (int i = 1; i < S; i++) m = fmaxf(m, row[i]);
    float den = 0.f, num = 0.f;
    const int half = S / 2;
    for (int i = 0; i < S; i++) {
        const float e = expf(row[i] - m);
        den += e;
        num += (float)(i - half) * e;
    }
    const float y = num / den;
    const float sg = (y > 0.f) ? 1.f : ((y < 0.f) ? -1.f : 0.f);
    const float r = (sqrtf(1.f + 4.f * 0.001f * (fabsf(y) + 1.f + 0.001f)) - 1.f) / (2.f * 0.001f);
    return sg * (r * r - 1.f);
}

__global__ void __launch_bounds__(256) k_support_decode(const float *logits, int S, float *out, int B) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < B) out[row] = support_decode_row(logits + (size_t)row * S, S);
}

__device__ inline void softmax_row(const float *row, int A, float *out) {
    float m = row[0];
    for (int i = 1; i < A; i++) m = fmaxf(m, row[i]);
    float den = 0.f;
    for (int i = 0; i < A; i++) den += expf(row[i] - m);
    for (int i = 0; i < A; i++) out[i] = expf(row[i] - m) / den;
}

__global__ void __launch_bounds__(256) k_policy_softmax(const float *logits, int A, float *out, int B) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < B) softmax_row(logits + (size_t)row * A, A, out + (size_t)row * A);
}

__global__ void __launch_bounds__(256) k_dynamics_epilogue(const float *state_dyn, const float *state_after,
                                                           const float *reward_logits, int ld, const uint8_t *branch,
                                                           int S, float *hidden_out, float *reward_out, int B) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    const bool dyn = branch[row] != 0;
    const float *x = (dyn ? state_dyn : state_after) + (size_t)row * ld;
    float mn = x[0], mx = x[0];
    for (int i = 1; i < S; i++) { mn = fminf(mn, x[i]); mx = fmaxf(mx, x[i]); }
    float sc = mx - mn;
    if (sc < 1e-5f) sc += 1e-5f;  // neural_network_mlp_model.py:353
    for (int i = 0; i < S; i++) hidden_out[(size_t)row * S + i] = (x[i] - mn) / sc;
    if (reward_out) reward_out[row] = (dyn && reward_logits) ? support_decode_row(reward_logits + (size_t)row * ld, S) : 0.f;
}

__global__ void __launch_bounds__(256) k_prediction_epilogue(const float *pol_pred, const float *val_pred,
                                                             const float *pol_after, const float *val_after,
                                                             int ld, const uint8_t *branch, int A, int S,
                                                             float *policy_out, float *value_out, int B) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    const bool dyn = branch[row] != 0;
    softmax_row((dyn ? pol_pred : pol_after) + (size_t)row * ld, A, policy_out + (size_t)row * A);
    value_out[row] = support_decode_row((dyn ? val_pred : val_after) + (size_t)row * ld, S);
}

// ---- synthetic env + trajectory record ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cartpole_step(double *state, const int32_t *action, float *obs_out,
                                                       float *reward_out, uint8_t *term_out, int B) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    const double g = 9.8, mc = 1.0, mp = 0.1, tm = mc + mp, len = 0.5, pml = mp * len, fm = 10.0, tau = 0.02;
    double *st = state + (size_t)e * 4;
    const double x = st[0], xd = st[1], th = st[2], thd = st[3];
    const double force = action[e] == 1 ? fm : -fm;
    const double ct = cos(th), sn = sin(th);
    const double temp = (force + pml * thd * thd * sn) / tm;
    const double tha = (g * sn - ct * temp) / (len * (4.0 / 3.0 - mp * ct * ct / tm));
    const double xa = temp - pml * tha * ct / tm;
    const double nx = x + tau * xd, nxd = xd + tau * xa, nth = th + tau * thd, nthd = thd + tau * tha;
    st[0] = nx; st[1] = nxd; st[2] = nth; st[3] = nthd;
    if (obs_out) {
        float *o = obs_out + (size_t)e * 4;
        o[0] = (float)nx; o[1] = (float)nxd; o[2] = (float)nth; o[3] = (float)nthd;
    }
    if (reward_out) reward_out[e] = 1.0f;
    if (term_out) term_out[e] = (fabs(nx) > 2.4 || fabs(nth) > 12.0 * 2.0 * 3.14159265358979323846 / 360.0) ? 1 : 0;
}

// record layout per (step, env):
//   [obs(obs_dim) | reward | terminated | policy(A) | action one-hot(A) | root_value | child_visits(A)]
__global__ void __launch_bounds__(256) k_traj_pack(double *traj, int T, int t, int obs_dim, int A, const float *obs,
                                                   const float *reward, const uint8_t *terminated, const int32_t *action,
                                                   const double *policy, const double *child_visits,
                                                   const float *root_value, int B) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    const int F = obs_dim + 3 * A + 3;
    double *r = traj + ((size_t)t * B + e) * F;
    for (int i = 0; i < obs_dim; i++) r[i] = (double)obs[(size_t)e * obs_dim + i];
    r[obs_dim] = reward ? (double)reward[e] : 0.0;
    r[obs_dim + 1] = (terminated && terminated[e]) ? 1.0 : 0.0;
    double *p = r + obs_dim + 2;
    for (int a = 0; a < A; a++) p[a] = policy[(size_t)e * A + a];
    for (int a = 0; a < A; a++) p[A + a] = (a == action[e]) ? 1.0 : 0.0;
    p[2 * A] = (double)root_value[e];
    for (int a = 0; a < A; a++) p[2 * A + 1 + a] = child_visits[(size_t)e * A + a];
}

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
    bool has_backup;
    std::vector<void *> allocs;
};

namespace {

thread_local char g_err[512] = "";

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
inline dim3 row_grid(int B) { return dim3((unsigned)((B + 255) / 256)); }

int launch_check() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "kernel launch failed: %s", hipGetErrorString(e));
        return SMZ_ERR_HIP;
    }
    return SMZ_OK;
}

// dispatch on the per-lane scratch bucket (smallest MAXA >= A)
#define SMZ_DISPATCH(maxa, CALL)              \
    switch (maxa) {                           \
        case 2: { constexpr int MA = 2; CALL; } break;   \
        case 4: { constexpr int MA = 4; CALL; } break;   \
        case 8: { constexpr int MA = 8; CALL; } break;   \
        case 16: { constexpr int MA = 16; CALL; } break; \
        default: { constexpr int MA = 32; CALL; } break; \
    }

}  // namespace

extern "C" {

const char *smz_last_error(void) { return g_err; }
int smz_abi_version(void) { return SMZ_ABI_VERSION; }
int smz_node_capacity(const smz_handle *h) { return h ? h->N : SMZ_ERR_INVALID; }

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
    if (cfg->rng_mode != SMZ_RNG_MT19937_NUMPY)
        return fail(SMZ_ERR_INVALID, "rng_mode: only SMZ_RNG_MT19937_NUMPY is built%s");
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
    P.disc32 = (float)cfg->discount;
    P.keep32 = (float)(1.0 - cfg->root_exploration_fraction);
    P.frac = cfg->root_exploration_fraction;
    P.alpha = cfg->root_dirichlet_alpha;
    const size_t BN = (size_t)B * h->N;
    int rc = SMZ_OK;
    auto A_ = [&](int r) { if (rc == SMZ_OK) rc = r; };
    A_(dev_alloc(h, &P.visit, BN));
    A_(dev_alloc(h, &P.value_sum, BN));
    A_(dev_alloc(h, &P.reward, BN));
    A_(dev_alloc(h, &P.prior, BN));
    A_(dev_alloc(h, &P.child_base, BN));
    A_(dev_alloc(h, &P.action, BN));
    A_(dev_alloc(h, &P.hidden, BN * (size_t)S));
    A_(dev_alloc(h, &P.root_prior, (size_t)B * A));
    A_(dev_alloc(h, &P.hdr, (size_t)B));
    A_(dev_alloc(h, &P.path, (size_t)B * h->Ppath));
    A_(dev_alloc(h, &P.mt, (size_t)B * kMtN));
    A_(dev_alloc(h, &P.rng_pos, (size_t)B));
    A_(dev_alloc(h, &h->d_seeds, (size_t)B));
    A_(dev_alloc(h, &h->d_pbc, (size_t)sims + 2));
    A_(dev_alloc(h, &h->d_pow, (size_t)sims + 1));
    A_(dev_alloc(h, &h->d_stats, (size_t)4));
    A_(dev_alloc(h, &h->d_mt_backup, (size_t)B * kMtN));
    A_(dev_alloc(h, &h->d_pos_backup, (size_t)B));
    h->has_backup = false;
    if (rc != SMZ_OK) { smz_destroy(h); return rc; }
    P.pbc_sqrt = h->d_pbc;
    P.pow_table = nullptr;
    P.stats = nullptr;
    // defined contents before first use (child_base == 0 <=> not expanded)
    hipError_t e = hipMemset(P.child_base, 0, BN * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(P.hdr, 0, (size_t)B * sizeof(TreeHdr));
    if (e == hipSuccess) e = hipMemset(h->d_stats, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(P.rng_pos, 0, (size_t)B * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(P.mt, 0, (size_t)B * kMtN * sizeof(uint32_t));
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
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    int32_t packed = 0;
    HIP_TRY(hipMemcpy(host_key, h->P.mt + (size_t)tree * kMtN, kMtN * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&packed, h->P.rng_pos + tree, sizeof(int32_t), hipMemcpyDeviceToHost));
    const int idx = packed & 0xffff, ready = packed >> 16;
    if (ready > 0) { *pos = idx; return SMZ_OK; }       // still inside an imported block
    if (idx == 0) { *pos = kMtN; return SMZ_OK; }        // block boundary: numpy regenerates on the next draw
    // words [0, idx) belong to the new block, [idx, 624) to the previous one: finish the in-place twist
    for (int i = idx; i < kMtN; i++) {
        const int i1 = (i + 1 == kMtN) ? 0 : i + 1;
        int im = i + kMtM;
        if (im >= kMtN) im -= kMtN;
        const uint32_t t = (host_key[i] & 0x80000000u) | (host_key[i1] & 0x7fffffffu);
        host_key[i] = host_key[im] ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
    }
    *pos = idx;
    return SMZ_OK;
}

int smz_rng_snapshot(smz_handle *h, smz_stream stream) {
    if (!h) return fail(SMZ_ERR_INVALID, "smz_rng_snapshot: null handle%s");
    DeviceGuard guard(h->cfg.device);
    const size_t B = (size_t)h->cfg.num_trees;
    HIP_TRY(hipMemcpyAsync(h->d_mt_backup, h->P.mt, B * kMtN * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(h->d_pos_backup, h->P.rng_pos, B * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
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
    return SMZ_OK;
}

int smz_root_init(smz_handle *h, const float *hidden_dev, const float *policy_dev, const double *noise_override_dev,
                  int train, smz_stream stream) {
    if (!h || !policy_dev) return fail(SMZ_ERR_INVALID, "smz_root_init: null argument%s");
    if (h->P.S > 0 && !hidden_dev) return fail(SMZ_ERR_INVALID, "smz_root_init: hidden_dev is required when hidden_size > 0%s");
    if (train && h->cfg.num_simulations > 0 && !(h->cfg.root_dirichlet_alpha > 0))
        return fail(SMZ_ERR_INVALID, "root_dirichlet_alpha must be > 0 to draw noise (numpy raises ValueError)%s");
    DeviceGuard guard(h->cfg.device);
    SMZ_DISPATCH(h->maxa, hipLaunchKernelGGL((k_root_init<MA>), tree_grid(h->P.B), dim3(kWave), 0, (hipStream_t)stream,
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
    SMZ_DISPATCH(h->maxa, hipLaunchKernelGGL((k_select<MA>), tree_grid(h->P.B), dim3(kWave), 0, (hipStream_t)stream, h->P,
                                             parent_hidden_dev, last_action_dev, branch_dev, mlp_input_dev));
    h->selected = true;
    return launch_check();
}

int smz_expand_backup(smz_handle *h, const float *hidden_dev, const float *reward_dev, const float *policy_dev,
                      const float *value_dev, smz_stream stream) {
    if (!h || !policy_dev || !value_dev) return fail(SMZ_ERR_INVALID, "smz_expand_backup: null argument%s");
    if (!h->selected) return fail(SMZ_ERR_STATE, "smz_expand_backup without a preceding smz_select%s");
    DeviceGuard guard(h->cfg.device);
    SMZ_DISPATCH(h->maxa, hipLaunchKernelGGL((k_expand_backup<MA, false>), tree_grid(h->P.B), dim3(kWave), 0,
                                             (hipStream_t)stream, h->P, hidden_dev, reward_dev, policy_dev, value_dev,
                                             (float *)nullptr, (int32_t *)nullptr, (uint8_t *)nullptr, (float *)nullptr));
    h->selected = false;
    return launch_check();
}

int smz_expand_backup_select(smz_handle *h, const float *hidden_dev, const float *reward_dev, const float *policy_dev,
                             const float *value_dev, float *parent_hidden_dev, int32_t *last_action_dev,
                             uint8_t *branch_dev, float *mlp_input_dev, smz_stream stream) {
    if (!h || !policy_dev || !value_dev) return fail(SMZ_ERR_INVALID, "smz_expand_backup_select: null argument%s");
    if (!h->selected) return fail(SMZ_ERR_STATE, "smz_expand_backup_select without a preceding smz_select%s");
    DeviceGuard guard(h->cfg.device);
    SMZ_DISPATCH(h->maxa, hipLaunchKernelGGL((k_expand_backup<MA, true>), tree_grid(h->P.B), dim3(kWave), 0,
                                             (hipStream_t)stream, h->P, hidden_dev, reward_dev, policy_dev, value_dev,
                                             parent_hidden_dev, last_action_dev, branch_dev, mlp_input_dev));
    return launch_check();
}

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
    hipLaunchKernelGGL(k_support_decode, row_grid(B), dim3(256), 0, (hipStream_t)stream, logits_dev, S, out_dev, B);
    return launch_check();
}

int smz_policy_softmax(const float *logits_dev, int A, float *out_dev, int B, smz_stream stream) {
    if (!logits_dev || !out_dev || A < 1 || B < 1) return fail(SMZ_ERR_INVALID, "smz_policy_softmax: bad argument%s");
    hipLaunchKernelGGL(k_policy_softmax, row_grid(B), dim3(256), 0, (hipStream_t)stream, logits_dev, A, out_dev, B);
    return launch_check();
}

int smz_dynamics_epilogue(const float *state_dyn_dev, const float *state_after_dev, const float *reward_logits_dev,
                          int ld, const uint8_t *branch_dev, int S, float *hidden_out_dev, float *reward_out_dev, int B,
                          smz_stream stream) {
    if (!state_dyn_dev || !state_after_dev || !branch_dev || !hidden_out_dev || S < 1 || B < 1 || ld < S)
        return fail(SMZ_ERR_INVALID, "smz_dynamics_epilogue: bad argument%s");
    hipLaunchKernelGGL(k_dynamics_epilogue, row_grid(B), dim3(256), 0, (hipStream_t)stream, state_dyn_dev, state_after_dev,
                       reward_logits_dev, ld, branch_dev, S, hidden_out_dev, reward_out_dev, B);
    return launch_check();
}

int smz_prediction_epilogue(const float *policy_logits_pred_dev, const float *value_logits_pred_dev,
                            const float *policy_logits_after_dev, const float *value_logits_after_dev,
                            int ld, const uint8_t *branch_dev, int A, int S, float *policy_out_dev,
                            float *value_out_dev, int B, smz_stream stream) {
    if (!policy_logits_pred_dev || !value_logits_pred_dev || !policy_logits_after_dev || !value_logits_after_dev ||
        !branch_dev || !policy_out_dev || !value_out_dev || A < 1 || S < 1 || B < 1 || ld < 1)
        return fail(SMZ_ERR_INVALID, "smz_prediction_epilogue: bad argument%s");
    hipLaunchKernelGGL(k_prediction_epilogue, row_grid(B), dim3(256), 0, (hipStream_t)stream, policy_logits_pred_dev,
                       value_logits_pred_dev, policy_logits_after_dev, value_logits_after_dev, ld, branch_dev, A, S,
                       policy_out_dev, value_out_dev, B);
    return launch_check();
}

int smz_cartpole_step(double *state_dev, const int32_t *action_dev, float *obs_out_dev, float *reward_out_dev,
                      uint8_t *terminated_out_dev, int B, smz_stream stream) {
    if (!state_dev || !action_dev || B < 1) return fail(SMZ_ERR_INVALID, "smz_cartpole_step: bad argument%s");
    hipLaunchKernelGGL(k_cartpole_step, row_grid(B), dim3(256), 0, (hipStream_t)stream, state_dev, action_dev, obs_out_dev,
                       reward_out_dev, terminated_out_dev, B);
    return launch_check();
}

int smz_traj_floats(int obs_dim, int A) { return obs_dim + 3 * A + 3; }

int smz_traj_pack(double *traj_dev, int T, int t, int obs_dim, int A, const float *obs_dev, const float *reward_dev,
                  const uint8_t *terminated_dev, const int32_t *action_dev, const double *policy_dev, const double *child_visits_dev,
                  const float *root_value_dev, int B, smz_stream stream) {
    if (!traj_dev || !obs_dev || !action_dev || !policy_dev || !child_visits_dev || !root_value_dev || t < 0 || t >= T ||
        B < 1 || A < 1 || obs_dim < 1)
        return fail(SMZ_ERR_INVALID, "smz_traj_pack: bad argument%s");
    hipLaunchKernelGGL(k_traj_pack, row_grid(B), dim3(256), 0, (hipStream_t)stream, traj_dev, T, t, obs_dim, A, obs_dev,
                       reward_dev, terminated_dev, action_dev, policy_dev, child_visits_dev, root_value_dev, B);
    return launch_check();
}

int smz_debug_dump_tree(smz_handle *h, int tree, smz_node_view *nodes, int cap, float *minmax_out, int32_t *path_out,
                        int cap_path, int32_t *path_len_out, double *root_priors_out) {
    if (!h || tree < 0 || tree >= h->cfg.num_trees) return fail(SMZ_ERR_INVALID, "smz_debug_dump_tree: bad argument%s");
    DeviceGuard guard(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const Params &P = h->P;
    TreeHdr hdr;
    HIP_TRY(hipMemcpy(&hdr, P.hdr + tree, sizeof(hdr), hipMemcpyDeviceToHost));
    const int n = hdr.alloc;
    if (nodes && cap > 0) {
        const int m = n < cap ? n : cap;
        std::vector<int32_t> vi((size_t)m), cb((size_t)m), ac((size_t)m);
        std::vector<float> vs((size_t)m), rw((size_t)m), pr((size_t)m);
        const size_t off = (size_t)tree * P.N;
        if (m > 0) {
            HIP_TRY(hipMemcpy(vi.data(), P.visit + off, (size_t)m * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(vs.data(), P.value_sum + off, (size_t)m * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(rw.data(), P.reward + off, (size_t)m * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(pr.data(), P.prior + off, (size_t)m * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(cb.data(), P.child_base + off, (size_t)m * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(ac.data(), P.action + off, (size_t)m * 4, hipMemcpyDeviceToHost));
        }
        for (int i = 0; i < m; i++) nodes[i] = smz_node_view{vi[i], vs[i], rw[i], pr[i], cb[i], ac[i]};
    }
    if (minmax_out) { minmax_out[0] = hdr.mn; minmax_out[1] = hdr.mx; }
    if (path_len_out) *path_len_out = hdr.path_len;
    if (path_out && cap_path > 0 && hdr.path_len > 0) {
        const int m = hdr.path_len < cap_path ? hdr.path_len : cap_path;
        HIP_TRY(hipMemcpy(path_out, P.path + (size_t)tree * P.P, (size_t)m * 4, hipMemcpyDeviceToHost));
    }
    if (root_priors_out)
        HIP_TRY(hipMemcpy(root_priors_out, P.root_prior + (size_t)tree * P.A, (size_t)P.A * 8, hipMemcpyDeviceToHost));
    return n;
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
    unsigned long long v[4];
    HIP_TRY(hipMemcpy(v, h->d_stats, sizeof(v), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) levels_out[i] = (uint64_t)v[i];
    if (reset) HIP_TRY(hipMemset(h->d_stats, 0, sizeof(v)));
    return SMZ_OK;
}

}  // extern "C"
