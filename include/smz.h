/*
 * smz.h -- C ABI of libsmz.so, the MI355X (gfx950) batched Stochastic-MuZero search engine.
 *
 * This is the drop-in boundary for the reference's search hot path.  The reference (DHDev0/Stochastic-muzero) is
 * pure Python with no FFI layer of its own; the seam it offers is the duck-typed object pair
 *     Monte_carlo_tree_search(**cfg).run(observation, model, train) -> root      monte_carlo_tree_search.py:76-85, 311-349
 *     Game.policy_step(root, temperature, ...) / Game.store_search_statistics(root)  game.py:179-273
 * and each entry point below cites the reference lines it replaces.  The Python binding a maintainer would add on
 * the reference side (ctypes) is shown in INTEGRATION.md; the binding this repo ships is stochastic-muzero_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success or a negative smz_status; smz_last_error() gives the text (thread-local);
 *   - nothing throws across the boundary;
 *   - every `dev` pointer is caller-owned DEVICE memory (e.g. a torch tensor's data_ptr()) that stays valid until
 *     the work enqueued on `stream` has run; `host` pointers are ordinary host memory, read before the call returns;
 *   - all launches are asynchronous on the caller's HIP stream (a hipStream_t passed as void*; NULL = the default
 *     stream) and are legal inside a stream capture (no allocation, no synchronisation), EXCEPT the functions
 *     marked [sync], which synchronise the stream / device themselves and must not be captured;
 *   - one handle owns the structure-of-arrays node buffers, search-path buffers, MinMax statistics and the
 *     per-tree random streams of `num_trees` independent search trees on ONE device.  A handle is not thread-safe;
 *     distinct handles are independent.
 *   - tree i of a handle is, by contract, the reference's single tree run under `np.random.seed(seed_i)`.
 */
#ifndef SMZ_H
#define SMZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMZ_ABI_VERSION 1
#define SMZ_MAX_ACTIONS 32 /* action_dimension limit of this build (per-lane scratch is sized by it) */
#define SMZ_MT_WORDS 624   /* MT19937 state words per tree */

typedef enum {
    SMZ_OK = 0,
    SMZ_ERR_INVALID = -1,   /* bad argument / hyper-parameter (the reference raises AssertionError, mcts:148-173) */
    SMZ_ERR_HIP = -2,       /* a HIP runtime call failed */
    SMZ_ERR_NOMEM = -3,     /* device allocation failed */
    SMZ_ERR_STATE = -4,     /* call order violated (e.g. select before root_init) */
    SMZ_ERR_TOO_LARGE = -5  /* the working set of a single-launch kernel exceeds a CU's LDS for this batch geometry:
                               nothing was launched, use the step-wise entry points */
} smz_status;

typedef enum {
    SMZ_RNG_MT19937_NUMPY = 0, /* per-tree numpy-legacy RandomState stream: bit-parity with the reference */
    SMZ_RNG_PHILOX = 1         /* counter-based stream (throughput mode): the SAME numpy-legacy algorithms (random_sample,
                                  choice, dirichlet ...) drawing their 32-bit words from Philox4x32-10 keyed by the tree's
                                  64-bit seed instead of from MT19937 -- same distributions, different numbers, no
                                  generator state in memory.  Word i of the stream = component i & 3 of
                                  philox(counter = i / 4, key = seed); smz_philox_words evaluates it on the host. */
} smz_rng_mode;

typedef void *smz_stream; /* hipStream_t */
typedef struct smz_handle smz_handle;

/* Hyper-parameters: the kwargs of Monte_carlo_tree_search.__init__ (monte_carlo_tree_search.py:76-85), plus the
 * batch geometry.  Validation mirrors monte_carlo_tree_search.py:148-173. */
typedef struct {
    int32_t num_trees;                 /* B: independent trees (envs) on this device */
    int32_t num_actions;               /* A: action_dimension (policy width, muzero_model.py:260) */
    int32_t max_action_sample;         /* maxium_action_sample; children per expansion K = min(this, A) (mcts:293) */
    int32_t hidden_size;               /* S: floats per hidden state (31 for ckpt 421, 147 for the vision nets) */
    int32_t num_simulations;           /* mcts:82 */
    int32_t pb_c_base;                 /* mcts:77 */
    double pb_c_init;                  /* mcts:78 */
    double discount;                   /* mcts:79 */
    double root_dirichlet_alpha;       /* mcts:80, 0 < alpha <= 1 */
    double root_exploration_fraction;  /* mcts:81 */
    int32_t rng_mode;                  /* smz_rng_mode */
    int32_t device;                    /* HIP device ordinal */
} smz_config;

/* One node as seen by smz_debug_dump_tree (host side, parity tests). */
typedef struct {
    int32_t visit_count;  /* Node.visit_count  mcts:8  */
    float value_sum;      /* Node.value_sum    mcts:10 */
    float reward;         /* Node.reward       mcts:13 */
    float prior;          /* Node.prior        mcts:9 (float32 policy entry; root children: see root priors) */
    int32_t child_base;   /* index of the first child, 0 = not expanded (children are contiguous) */
    int32_t action;       /* key of this node in its parent's `children` dict */
} smz_node_view;

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
/* [sync] Allocates all device state for cfg->num_trees trees.  Replaces Monte_carlo_tree_search.__init__/reset
 * (mcts:76-177).  SMZ_ERR_INVALID where the reference asserts. */
int smz_create(const smz_config *cfg, smz_handle **out);
/* [sync] */
int smz_destroy(smz_handle *h);
int smz_abi_version(void);
/* Optional pieces compiled into this library (bit mask; none at present: the register-resident experimental search kernel
 * of rounds 2-3, a measured dead end, left the tree in round 4 -- profiles/r02_reg_kernel_ab.txt, git history). */
int smz_build_features(void);
const char *smz_last_error(void);
/* Number of nodes each tree can hold: 1 + A + num_simulations * K. */
int smz_node_capacity(const smz_handle *h);

/* [sync] pb_c[n] = log((n + pb_c_base + 1)/pb_c_base) + pb_c_init for n = 0..num_simulations+1 (mcts:236).
 * smz_create fills it with libm's log; a Python host passes the table as numpy evaluates it so that the scores
 * are the reference's to the last bit on that machine.  `n` must be >= num_simulations + 2. */
int smz_set_pb_c_table(smz_handle *h, const double *host_table, int n);

/* ---- random streams (numpy.random.seed / get_state / set_state; used at mcts:208,220,243,254,294, game.py:213) - */
/* Seeds tree i with host_seeds[i] (numpy `seed(int)`: init_genrand on the low 32 bits).  Asynchronous on `stream`
 * after an internal host->device copy of the seeds that completes before return. */
int smz_seed(smz_handle *h, const uint64_t *host_seeds, smz_stream stream);
/* [sync] numpy get_state()/set_state() of one tree: key[624] and pos in numpy's convention (pos == 624 means
 * "regenerate before the next draw").  Lets a single-tree caller continue the process-global numpy stream. */
int smz_set_rng_state(smz_handle *h, int tree, const uint32_t *host_key, int pos);
int smz_get_rng_state(smz_handle *h, int tree, uint32_t *host_key, int *pos);

/* SMZ_RNG_PHILOX: host evaluation of n consecutive words of the stream keyed by `seed`, starting at word `idx` (< 624) of
 * 624-word block `block`, and the position (block, idx) a tree of a Philox handle has reached [sync]. */
int smz_philox_words(uint64_t seed, uint32_t block, int idx, int n, uint32_t *host_out);
int smz_get_philox_position(smz_handle *h, int tree, uint32_t *block_out, int *idx_out);

/* Asynchronous device-side snapshot / restore of ALL trees' streams (one backup slot per handle).  Lets a caller
 * run throw-away warm-up searches (e.g. before capturing a HIP graph) without disturbing the per-tree streams. */
int smz_rng_snapshot(smz_handle *h, smz_stream stream);
int smz_rng_restore(smz_handle *h, smz_stream stream);

/* Per-tree on/off switch: active_dev [B] u8 (caller-owned device memory that must outlive its use; NULL = all on).
 * Every search phase, smz_search_mlp(_act) and smz_act skip the trees whose byte is 0: their nodes, outputs and random
 * streams stay untouched.  This is the batched form of the reference loop's `while not environment.terminal`
 * (self_play.py:79): a finished game stops consuming simulations.  The array is read when the kernels run, so the
 * env-step kernel may clear entries on the same stream (smz_cartpole_step_ctl). */
int smz_set_active(smz_handle *h, const uint8_t *active_dev);
/* Large batches: the row moves of a simulation round can be left to the network kernel.  When ids_dev is set, every
 * selection (smz_select, smz_expand_backup_select) also writes, per tree, the node ids of the selected leaf and of its
 * parent to ids_dev [B][2] i32 ({-1, -1} for a tree switched off with smz_set_active); a network kernel then reads the
 * parent's hidden row from the handle's storage and writes the leaf's row into it (smz_mlp_recurrent_rows), and the tree
 * entry points are called with NULL for their row arguments (parent_hidden / mlp_input outputs, hidden input).  NULL
 * switches it off.  ids_dev is read by later launches: it must stay alive.  mcts:270-286 (leaf I/O of the nets). */
int smz_set_leaf_ids_out(smz_handle *h, int32_t *ids_dev);
/* Hidden-state storage of the handle: the row of node n of tree t starts at hidden + ((size_t)t * nodes_per_tree + n) *
 * row_stride floats (hidden_size floats used). */
int smz_get_hidden_layout(smz_handle *h, float **hidden_dev_out, int *nodes_per_tree_out, int *row_stride_out);

/* ---- the search (one call per phase of Monte_carlo_tree_search.run, mcts:311-349) ----------------------------- */
/* Root: resets the trees and MinMaxStats, stores the root hidden state, normalises the root policy, creates all A
 * children (consuming the draws of np.random.choice, mcts:203-211) and, when `train` and num_simulations > 0,
 * mixes Dirichlet noise into the priors (mcts:214-225).
 *   hidden_dev  [B,S] f32  representation_function_inference output (mcts:179-183)
 *   policy_dev  [B,A] f32  softmaxed prediction policy (mcts:197-200; the value head's output is discarded, :319)
 *   noise_override_dev [B,A] f64 or NULL: when given, these values replace the device-drawn Dirichlet sample
 *     (the stream still advances exactly as if it had been drawn) -- a parity-test affordance. */
int smz_root_init(smz_handle *h, const float *hidden_dev, const float *policy_dev, const double *noise_override_dev,
                  int train, smz_stream stream);

/* Selection: full root->leaf descent for every tree (mcts:228-267: pUCT argmax at decision-flagged nodes,
 * prior sampling at chance-flagged nodes), recording the search path, then gathers what the networks need
 * (mcts:270-286): any output pointer may be NULL.
 *   parent_hidden_dev [B,S]  f32  hidden state of the leaf's parent
 *   last_action_dev   [B]    i32  history[-1]
 *   branch_dev        [B]    u8   parent.is_chance: 1 -> dynamics + prediction, 0 -> afterstate_dynamics +
 *                                 afterstate_prediction (mcts:333-342)
 *   mlp_input_dev     [B,S+A] f32 [parent hidden | one_hot(last action)] exactly as neural_network_mlp_model.py:123,205
 *                                 concatenates it (muzero_model.py:496-509) */
int smz_select(smz_handle *h, float *parent_hidden_dev, int32_t *last_action_dev, uint8_t *branch_dev,
               float *mlp_input_dev, smz_stream stream);

/* Expansion + backup for the leaves chosen by the last smz_select (mcts:289-308): stores hidden state and reward
 * on the leaf (reward only on the dynamics branch), samples K children without replacement from the normalised
 * policy, then backs the value up the recorded path updating visit counts, value sums and MinMaxStats.
 *   hidden_dev [B,S] f32, reward_dev [B] f32 (may be NULL = 0), policy_dev [B,A] f32 (softmaxed), value_dev [B] f32 */
int smz_expand_backup(smz_handle *h, const float *hidden_dev, const float *reward_dev, const float *policy_dev,
                      const float *value_dev, smz_stream stream);

/* Fused smz_expand_backup + the next smz_select in one launch (the two halves touch the same tree from the same
 * lane, so no grid-wide ordering is needed).  Outputs as smz_select. */
int smz_expand_backup_select(smz_handle *h, const float *hidden_dev, const float *reward_dev, const float *policy_dev,
                             const float *value_dev, float *parent_hidden_dev, int32_t *last_action_dev,
                             uint8_t *branch_dev, float *mlp_input_dev, smz_stream stream);

/* Root statistics: what game.py reads off the returned root (game.py:181-204).  Any pointer may be NULL.
 *   visits_dev [B,A] i32, priors_dev [B,A] f64, root_value_dev [B] f32 (Node.value(), mcts:20-21),
 *   child_reward_dev [B,A] f32 */
int smz_root_stats(smz_handle *h, int32_t *visits_dev, double *priors_dev, float *root_value_dev,
                   float *child_reward_dev, smz_stream stream);

/* Post-search policy and action (Game.policy_step up to the env step, game.py:197-235, and
 * Game.store_search_statistics, game.py:179-195); draws from each tree's stream where the reference would
 * (game.py:212-213).  pow_table_host[v] = float64(v) ** (1/temperature), v = 0..num_simulations, as numpy
 * evaluates game.py:208 (NULL: device pow); read before return.
 *   action_dev [B] i32, policy_dev [B,A] f64 (Game.policies entry), child_visits_dev [B,A] f64, root_value_dev [B] f32 */
int smz_act(smz_handle *h, double temperature, const double *pow_table_host, int32_t *action_dev, double *policy_dev,
            double *child_visits_dev, float *root_value_dev, smz_stream stream);

/* ---- head epilogues (what muzero_model.py does around the torch modules) --------------------------------------- */
/* inverse_transform_with_support (muzero_model.py:575-591): softmax over S bins, expectation over the integer
 * support, inverse h-transform.  logits_dev [B,S] f32 -> out_dev [B] f32. */
int smz_support_decode(const float *logits_dev, int S, float *out_dev, int B, smz_stream stream);
/* Softmax over the last dimension (muzero_model.py:837,855): logits_dev [B,A] f32 -> out_dev [B,A] f32. */
int smz_policy_softmax(const float *logits_dev, int A, float *out_dev, int B, smz_stream stream);
/* scale_to_bound_action (neural_network_mlp_model.py:349-357) on the candidate next state of the branch each tree
 * took, plus the reward decode of the dynamics branch.  The three inputs are row-major with a common row stride
 * `ld` (in floats, >= S) so that they may be column slices of one fused GEMM output:
 *   state_dyn_dev, state_after_dev, reward_logits_dev: [B] rows of S pre-activation head outputs; branch_dev [B] u8
 *   -> hidden_out_dev [B,S] contiguous (scaled), reward_out_dev [B] (0 on the afterstate branch, mcts:338-342) */
int smz_dynamics_epilogue(const float *state_dyn_dev, const float *state_after_dev, const float *reward_logits_dev,
                          int ld, const uint8_t *branch_dev, int S, float *hidden_out_dev, float *reward_out_dev, int B,
                          smz_stream stream);
/* policy softmax + value decode of the branch each tree took (muzero_model.py:837-839, 855-856); inputs row-major
 * with common row stride `ld` floats:
 *   policy_logits_{pred,after}_dev rows of A, value_logits_{pred,after}_dev rows of S, branch_dev [B] u8
 *   -> policy_out_dev [B,A] f32, value_out_dev [B] f32 (contiguous) */
int smz_prediction_epilogue(const float *policy_logits_pred_dev, const float *value_logits_pred_dev,
                            const float *policy_logits_after_dev, const float *value_logits_after_dev, int ld,
                            const uint8_t *branch_dev, int A, int S, float *policy_out_dev, float *value_out_dev,
                            int B, smz_stream stream);

/* ---- fused `mlp_model` heads (neural_network_mlp_model.py:5-250 + muzero_model.py:802-909) ---------------------- */
/* One launch evaluates, for every tree, the pair of networks its leaf needs (mcts:333-342) -- dynamics + prediction
 * or afterstate_dynamics + afterstate_prediction -- including scale_to_bound_action, the policy softmax and the
 * support decodes; weights are staged once per workgroup in LDS.  Available when the packed weights fit in LDS
 * (smz_mlp_layout reports the size); otherwise the caller evaluates the heads with its own GEMMs and the epilogue
 * kernels above.
 *
 * Packed weight buffer: float32, every matrix stored input-major and 4-way interleaved along the input index so
 * that lane o reads four consecutive input weights with one 16-byte LDS read:
 *     element (k, o) of a K x O matrix  ->  base + ((k / 4) * OP + o) * 4 + (k % 4),   OP = 64 (max(H, 2S, A+S) <= 64),
 * K zero-padded to a multiple of 4, O zero-padded to OP; a bias vector is OP floats.  Matrices (off[] index):
 *   0 dyn_in  (S+A x H)   1 ady_in (S+A x H)   2 dyn_mid (H x H)  3 ady_mid (H x H)      [mid only used when L > 0]
 *   4 dyn_out (H x 2S: reward logits | next state)     5 ady_out (H x S: next state)
 *   6 pre_in  (S x H)     7 apr_in (S x H)     8 pre_mid (H x H)  9 apr_mid (H x H)
 *  10 pre_out (H x A+S: policy logits | value logits)  11 apr_out (H x A+S)
 *  12 rep_in  (obs x H)  13 rep_mid (H x H)   14 rep_out (H x S)
 * followed by the 15 bias vectors in the same order (off[15 + i]). */
typedef struct {
    int32_t obs, A, S, H, L;
    int32_t OP;            /* padded output width */
    int32_t total_floats;  /* size of the packed buffer */
    int32_t off[30];       /* float offsets: 15 matrices then 15 biases */
} smz_mlp_desc;
/* Fills OP, total_floats and off[] from obs/A/S/H/L.  Returns SMZ_ERR_INVALID when the recurrent working set does not
 * fit the 160 KB LDS of a CU (use the GEMM path then). */
int smz_mlp_layout(smz_mlp_desc *desc);
/* representation + root prediction: obs_dev [B,obs] -> hidden_out_dev [B,S] (scaled), policy_out_dev [B,A] (softmax) */
int smz_mlp_initial(const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev, float *hidden_out_dev,
                    float *policy_out_dev, int B, smz_stream stream);
/* recurrent step for all trees: mlp_input_dev [B,S+A] and branch_dev [B] as written by smz_select ->
 * hidden_out_dev [B,S], reward_out_dev [B] (0 on the afterstate branch), policy_out_dev [B,A], value_out_dev [B] */
int smz_mlp_recurrent(const smz_mlp_desc *desc, const float *weights_dev, const float *mlp_input_dev,
                      const uint8_t *branch_dev, float *hidden_out_dev, float *reward_out_dev, float *policy_out_dev,
                      float *value_out_dev, int B, smz_stream stream);

/* ---- fused heads of the `vision_model` family (neural_network_vision_model.py:41-515) ---------------------------- */
/* Replaces, for the reference's ResNet-v2 family on 98x98x3 frames (hidden state 3x7x7 = 147 floats, channel major),
 * the same five *_inference calls (muzero_model.py:802-909) as the smz_mlp_* entry points do for `mlp_model`.
 * Packed weight buffer (float32, every piece starts at a multiple of 4 floats; off[] holds the float offsets):
 *   convolutions  : torch layout [out][in][ky][kx] flattened;
 *   batch-norms   : eval-mode affine form, [scale(C) | shift(C)] with scale = weight / sqrt(running_var + eps),
 *                   shift = bias - running_mean * scale (float32 arithmetic, as ATen's CPU kernel folds them);
 *   1x1 "mix" convolutions: weight [out][in] then bias [out] as separate pieces;
 *   towers        : three Linear layers (147 -> H, H -> H (shared, applied L times), H -> n_out) each as
 *                   W in the 4-way interleaved input-major layout of smz_mlp_desc (Wp[k/4][o][k%4], o padded to OP)
 *                   followed by its bias piece (OP floats).
 * Indices into off[]: transition nets (dynamics, afterstate dynamics) at SMZ_V_TRANS_BASE + net * SMZ_V_TRANS_STRIDE +
 * SMZ_VT_*; prediction nets (prediction, afterstate prediction) at SMZ_V_PRED_BASE + net * SMZ_V_PRED_STRIDE + SMZ_VP_*;
 * representation at SMZ_V_REP_BASE + SMZ_VR_*.  The afterstate dynamics net has no reward branch: its SMZ_VT_MIX_* and
 * tower pieces stay zero. */
enum { SMZ_V_DYN = 0, SMZ_V_ADY = 1, SMZ_V_PRE = 2, SMZ_V_APR = 3 };
enum { SMZ_VT_CONV_IN = 0, SMZ_VT_BN_IN, SMZ_VT_RES_A, SMZ_VT_RES_B, SMZ_VT_RES_BN, SMZ_VT_MIX_W, SMZ_VT_MIX_B,
       SMZ_VT_TOWER /* 6 entries: W1,b1,Wm,bm,Wo,bo */, SMZ_V_TRANS_STRIDE = 13 };
enum { SMZ_VP_RES_A = 0, SMZ_VP_RES_B, SMZ_VP_RES_BN, SMZ_VP_VMIX_W, SMZ_VP_VMIX_B, SMZ_VP_VTOWER /* 6 */,
       SMZ_VP_PMIX_W = 11, SMZ_VP_PMIX_B, SMZ_VP_PTOWER /* 6 */, SMZ_V_PRED_STRIDE = 19 };
enum { SMZ_VR_STEM = 0, SMZ_VR_NARROW_A, SMZ_VR_NARROW_B, SMZ_VR_NARROW_BN, SMZ_VR_WIDEN, SMZ_VR_WIDE_A, SMZ_VR_WIDE_B,
       SMZ_VR_WIDE_BN, SMZ_VR_LAST_A, SMZ_VR_LAST_B, SMZ_VR_LAST_BN };
enum { SMZ_V_TRANS_BASE = 0, SMZ_V_PRED_BASE = 26, SMZ_V_REP_BASE = 64, SMZ_V_OFFSETS = 80 };
typedef struct smz_vision_desc {
    int32_t A, S, H, L;    /* actions, support size (state_space_dimensions), tower width, number_of_hidden_layer */
    int32_t OP;            /* padded tower output width (64) */
    int32_t total_floats;  /* size of the packed buffer */
    int32_t small_floats;  /* [0, small_floats): convolution / batch-norm / 1x1 pieces of the four recurrent nets,
                              contiguous (staged in LDS by smz_vision_recurrent); towers and representation follow */
    int32_t off[SMZ_V_OFFSETS];
} smz_vision_desc;
/* Fills OP, total_floats, small_floats and off[] from A/S/H/L.  SMZ_ERR_INVALID when A, S or H exceed 64 (one neuron
 * per lane). */
int smz_vision_layout(smz_vision_desc *desc);
/* representation + root prediction: frames_dev [B,3,98,98] f32 (8-byte aligned) -> hidden_out_dev [B,147] (scaled per
 * pixel across channels), policy_out_dev [B,A] (softmax).  One 256-thread workgroup per frame. */
int smz_vision_initial(const smz_vision_desc *desc, const float *weights_dev, const float *frames_dev,
                       float *hidden_out_dev, float *policy_out_dev, int B, smz_stream stream);
/* ... which also appends the frames it reads to the trajectory record: frames_copy_dev [B,3,98,98] f32 (NULL: no copy)
 * receives a bit-exact copy of frames_dev -- Game.observations.append of the frame the search is about to run on
 * (game.py:263-264: the observation after the previous action).  The copy moves in 16-byte pieces alongside the residual
 * blocks of the launch (the stem has just read the frame, the re-read comes from the caches), not as a second pass over HBM.
 * frames_dev 8-byte aligned; with a copy both frame pointers 16-byte aligned (SMZ_ERR_INVALID otherwise). */
int smz_vision_initial_record(const smz_vision_desc *desc, const float *weights_dev, const float *frames_dev,
                              float *frames_copy_dev, float *hidden_out_dev, float *policy_out_dev, int B, smz_stream stream);
/* Frame ingest: n_frames rendered frames [n][H][W][3] uint8 (as a host environment uploads them) -> the
 * [*,3,out_h,out_w] float32 tensor smz_vision_initial reads: ToTensor (CHW, / 255) + bilinear Resize without antialias,
 * align_corners = False -- Game.transform_rgb of the reference (game.py:82-89, 142-143), ATen's upsample_bilinear2d
 * arithmetic in float32.  Frame i goes to output row rows_dev[i] (rows_dev NULL: row i), so a few frames can be patched into
 * a full batch.  SMZ_ERR_TOO_LARGE for rows wider than ~10 000 pixels. */
int smz_frames_resize_u8(const uint8_t *frames_dev, int n_frames, int H, int W, int out_h, int out_w, const int32_t *rows_dev,
                         float *out_dev, smz_stream stream);
/* The same resize from TAP-COMPACTED frames (round 4): taps_dev [n][2 out_h][2 out_w][3] uint8 holds of each H x W frame only
 * the pixels the resize reads -- row 2 oy + r = source row y_r(oy), column 2 ox + q = source column x_q(ox), the (i0, i1) pairs
 * of ATen's align_corners = False index rule (host_envs.tap_index) -- 115 KB instead of 720 KB per 400 x 600 frame over PCIe.
 * Bit-identical to smz_frames_resize_u8 on the full frames. */
int smz_frames_resize_taps_u8(const uint8_t *taps_dev, int n_frames, int H, int W, int out_h, int out_w, const int32_t *rows_dev,
                              float *out_dev, smz_stream stream);
/* Host-buffer boundary (replaces the per-frame / per-observation torch.tensor(...) of game.py:145-167 and the Ray
 * object store of self_play.py:240-256 for envs that live on the host): [sync] page-lock / release a caller-owned host mapping
 * (e.g. the shared-memory block the env worker processes write their rows into), and copy between it and device memory
 * asynchronously on the caller's stream (to_device != 0: host -> device). */
int smz_host_register(void *host_ptr, size_t bytes);
int smz_host_unregister(void *host_ptr);
int smz_copy_async(void *dst, const void *src, size_t bytes, int to_device, smz_stream stream);
/* recurrent step for all trees, one wavefront per leaf: parent_hidden_dev [B,ld] (first 147 floats of each row),
 * last_action_dev [B], branch_dev [B] as written by smz_select -> hidden_out_dev [B,147], reward_out_dev [B] (0 on the
 * afterstate branch), policy_out_dev [B,A], value_out_dev [B].  The action enters as the constant plane (a+1)/A
 * (muzero_model.py:511-522). */
int smz_vision_recurrent(const smz_vision_desc *desc, const float *weights_dev, const float *parent_hidden_dev, int ld,
                         const int32_t *last_action_dev, const uint8_t *branch_dev, float *hidden_out_dev,
                         float *reward_out_dev, float *policy_out_dev, float *value_out_dev, int B, smz_stream stream);

/* The whole Monte_carlo_tree_search.run (mcts:311-349) of every tree in ONE launch, for `mlp_model` heads that fit
 * in LDS: representation + root prediction, root expansion and noise, then num_simulations x (select, the pair of
 * networks the leaf needs, expansion, backup).  Network weights are staged once per workgroup; leaf hand-off and
 * network outputs never leave LDS.  Results are read as after the step-wise calls (smz_root_stats / smz_act /
 * smz_debug_dump_tree).  obs_dev [B,obs] f32.  SMZ_ERR_TOO_LARGE when the working set exceeds a CU's LDS. */
int smz_search_mlp(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev, int train,
                   smz_stream stream);

/* smz_search_mlp followed by smz_act (same arguments, same results, same stream position afterwards) in ONE launch:
 * the action selection runs in the tail of the search kernel, on the lanes that own the trees. */
int smz_search_mlp_act(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, const float *obs_dev, int train,
                       double temperature, const double *pow_table_host, int32_t *action_dev, double *policy_dev,
                       double *child_visits_dev, float *root_value_dev, smz_stream stream);

/* The same for the `vision_model` family: root expansion and noise, then num_simulations x (select, the leaf's
 * (afterstate) dynamics + (afterstate) prediction networks, expansion, backup) in ONE launch.  The root's hidden state and
 * policy come from smz_vision_initial (hidden0_dev [B,147], policy0_dev [B,A]): the representation network works on whole
 * 98x98 frames, one workgroup per frame, and stays its own launch.  A workgroup of 4 wavefronts owns 4 trees (kept in LDS
 * for the whole search); the convolutional part of a leaf is evaluated by its tree's wavefront, the five 147 -> H -> S/A
 * towers for the workgroup's 4 leaves at once on the matrix cores (v_mfma_f32_4x4x1 chains, weights in registers; f32 in /
 * f32 accumulate: bit-identical to smz_vision_recurrent).  Results are read as after
 * the step-wise calls.  SMZ_ERR_TOO_LARGE outside the kernel's limits (maxium_action_sample == 2, A <= 4, S <= 32,
 * H <= 64, working set <= 160 KB of LDS; MT19937 and -- round 6 -- Philox handles alike): use the step-wise entry points then.
 * monte_carlo_tree_search.py:311-349, neural_network_vision_model.py:41-515, muzero_model.py:802-909. */
int smz_search_vision(smz_handle *h, const smz_vision_desc *desc, const float *weights_dev, const float *hidden0_dev,
                      const float *policy0_dev, int train, smz_stream stream);
/* ... followed by smz_act in the tail of the same launch (as smz_search_mlp_act). */
int smz_search_vision_act(smz_handle *h, const smz_vision_desc *desc, const float *weights_dev, const float *hidden0_dev,
                          const float *policy0_dev, int train, double temperature, const double *pow_table_host,
                          int32_t *action_dev, double *policy_dev, double *child_visits_dev, float *root_value_dev,
                          smz_stream stream);

/* smz_mlp_recurrent on rows that live in a handle's hidden-state storage (smz_get_hidden_layout): leaf i's network input
 * is the hidden row of node ids_dev[2i+1] of tree i plus the one-hot of last_action_dev[i], its new hidden row is written
 * to node ids_dev[2i] of tree i; rows with ids < 0 are skipped.  Matrix-core kernel of the shipped network shape
 * (state_space_dimensions 31, hidden_layer_dimensions 64, number_of_hidden_layer 0, 2 or 4 actions): SMZ_ERR_TOO_LARGE for
 * any other shape -- use smz_mlp_recurrent then.  Same outputs, bit for bit.  muzero_model.py:844-909. */
int smz_mlp_recurrent_rows(const smz_mlp_desc *desc, const float *weights_dev, float *hidden_dev, int nodes_per_tree,
                           int row_stride, const int32_t *ids_dev, const int32_t *last_action_dev, const uint8_t *branch_dev,
                           float *reward_out_dev, float *policy_out_dev, float *value_out_dev, int B, smz_stream stream);

/* mlp_model networks too wide for LDS residency (H <= 128, 2 S <= 128, A + S <= 128, any number_of_hidden_layer: the
 * reference's config/experiment_434_config.json, S 61 / H 126 / L 0, and its shipped checkpoint 450, S 61 / H 126 / L 4).
 * smz_mlp_layout_wide fills the same descriptor for a 128-outputs-wide packed image (matrix m: rows padded to a multiple of
 * 8 inputs, element (k, o) at off[m] + ((k / 4) * 128 + o) * 4 + k % 4; biases at off[15 + m]; with number_of_hidden_layer
 * > 0 each trunk's ONE shared Linear(H, H) is matrix *_MID, applied L times); SMZ_ERR_TOO_LARGE beyond those limits.
 * smz_mlp_recurrent_wide = smz_mlp_recurrent on such a buffer: 16-leaf tiles on the matrix cores, weights streamed from L2.
 * neural_network_mlp_model.py:5-250, muzero_model.py:844-909. */
int smz_mlp_layout_wide(smz_mlp_desc *desc);
int smz_mlp_recurrent_wide(const smz_mlp_desc *desc, const float *weights_dev, const float *mlp_input_dev,
                           const uint8_t *branch_dev, float *hidden_out_dev, float *reward_out_dev, float *policy_out_dev,
                           float *value_out_dev, int B, smz_stream stream);

/* ---- synthetic environment + trajectory record (self_play.py:63-98 loop body around the search) -------------- */
/* CartPole-v1 shaped Euler step on device (float64 state, float32 observation), used for the synthetic
 * fixed-length episodes of the benchmark: state_dev [B,4] f64 in/out, action_dev [B] i32,
 * obs_out_dev [B,4] f32, reward_out_dev [B] f32, terminated_out_dev [B] u8 (any output may be NULL). */
int smz_cartpole_step(double *state_dev, const int32_t *action_dev, float *obs_out_dev, float *reward_out_dev,
                      uint8_t *terminated_out_dev, int B, smz_stream stream);
/* smz_cartpole_step followed by smz_traj_pack (obs_dim 4, A 2) in one launch: same record, one launch less per env step. */
int smz_cartpole_step_pack(double *state_dev, const int32_t *action_dev, float *obs_out_dev, float *reward_out_dev,
                           uint8_t *terminated_out_dev, double *traj_dev, int T, int t, const double *policy_dev,
                           const double *child_visits_dev, const float *root_value_dev, int B, smz_stream stream);
/* The same step with the per-env game bookkeeping of the reference's loop (`while not environment.terminal and counter <
 * environment.limit_of_game_play`, self_play.py:79; Game.done, game.py:270-271) kept on the device, so that a batched
 * loop needs no host round trip to learn which games have ended.  The flag written to flag_out_dev [B] u8 and to the
 * record's `terminated` slot is
 *     0 running | 1 terminated (Game.done True) | 2 stopped by limit_of_game_play (game over, Game.done stays False,
 *     game.py:270-271) | 3 no step taken (the env is switched off: active_dev[e] == 0).
 * smz_episode_ctl (host struct, read before return; the arrays are caller-owned device memory):
 *   step_count_dev [B] i32 in/out  steps taken in the current game (the reference's `counter`)
 *   episode_dev    [B] i32 in/out  games finished by this env so far (needed for on_end 2)
 *   active_dev     [B] u8  in/out  envs still playing (needed for on_end 1; may be NULL otherwise).  Hand the same array
 *                                  to smz_set_active and finished envs stop consuming simulations.
 *   limit          limit_of_game_play (<= 0: none)
 *   on_end         0: keep stepping past the end (fixed-length synthetic episodes of the benchmark)
 *                  1: switch the env off (active_dev[e] = 0)
 *                  2: start the env's next game at once: state ~ U(-0.05, 0.05)^4 (CartPole-v1's reset distribution)
 *                     drawn from a counter-based generator keyed by (reset_seed, first_env + e, episode) -- independent
 *                     of the shard and of the launch geometry; smz_cartpole_reset_state gives the same values on the host.
 * traj_dev may be NULL (no record). */
typedef struct {
    int32_t *step_count_dev;
    int32_t *episode_dev;
    uint8_t *active_dev;
    int32_t limit;
    int32_t on_end;
    uint64_t reset_seed;
    int64_t first_env;
} smz_episode_ctl;
int smz_cartpole_step_ctl(double *state_dev, const int32_t *action_dev, float *obs_out_dev, float *reward_out_dev,
                          uint8_t *flag_out_dev, const smz_episode_ctl *ctl, double *traj_dev, int T, int t,
                          const double *policy_dev, const double *child_visits_dev, const float *root_value_dev, int B,
                          smz_stream stream);
/* ONE launch per env step of the built-in env (the loop body of self_play.py:79-94 for every env): smz_search_mlp_act on the
 * observations in env->obs_dev, then -- in the tail of the same kernel, in the lane that owns the tree -- the env step and
 * the trajectory record of smz_cartpole_step / _step_pack / _step_ctl with the action just chosen.  Same results, bit for
 * bit, as the two launches.  obs_dev [B,4] f32 is read at the start of the launch and receives the NEXT observation;
 * state_dev [B,4] f64 in/out; reward_dev [B] f32, flag_dev [B] u8 (may be NULL); ctl NULL = plain steps (flag =
 * terminated), else the game bookkeeping of smz_cartpole_step_ctl -- ctl->active_dev must then be the array the handle
 * got through smz_set_active (or both NULL); traj_dev [T][B][13] f64 or NULL, row t.  2 actions, 4 observations. */
typedef struct {
    double *state_dev;
    float *obs_dev;
    float *reward_dev;
    uint8_t *flag_dev;
    const smz_episode_ctl *ctl;
    double *traj_dev;
    int32_t T;
    int32_t t;
} smz_cartpole_env;
int smz_search_mlp_act_cartpole(smz_handle *h, const smz_mlp_desc *desc, const float *weights_dev, int train,
                                double temperature, const double *pow_table_host, int32_t *action_dev, double *policy_dev,
                                double *child_visits_dev, float *root_value_dev, const smz_cartpole_env *env,
                                smz_stream stream);
/* Host evaluation of the reset state smz_cartpole_step_ctl gives env `env` for its game number `episode` (>= 1). */
int smz_cartpole_reset_state(uint64_t reset_seed, int64_t env, int64_t episode, double state_out[4]);
/* The same CartPole-v1 Euler step for environments that live on the HOST (plain C loop, no GPU call): the compiled counterpart
 * of a gymnasium vector env behind the host-buffer variant of the boundary (envs.HostCartPoleVec: actions down and
 * observations up through pinned memory every env step).  state_host [B,4] f64 in/out; flag 0 running | 1 terminated | 2
 * stopped by `limit` (<= 0: none; needs step_count_host [B] i32 in/out); any output may be NULL.  gymnasium's
 * cartpole.py (dependency of game.py:123-131). */
int smz_host_cartpole_step(double *state_host, const int32_t *action_host, float *obs_out_host, float *reward_out_host,
                           uint8_t *flag_out_host, int32_t *step_count_host, int32_t limit, int B);
/* Observation-only stand-in env (LunarLander-shaped benchmark workload; Box2D / gymnasium are not part of this build):
 * obs_dev [B][obs_dim] f32 ~ N(0,1), element (env, k) of step t a pure function of (seed, first_env + env, t, k). */
int smz_synthetic_obs(float *obs_dev, int B, int obs_dim, uint64_t seed, int64_t first_env, int64_t t, smz_stream stream);

/* Appends one env step of every tree to a fixed-length trajectory buffer laid out [T][B][F] (step-major, so one
 * step is one contiguous, coalesced slab and a finished chunk is one message for the trajectory gather):
 * what Game.policy_step / store_search_statistics append to their lists (game.py:193-195, 263-267).  Record of F =
 * smz_traj_floats(obs_dim, A) float64 values:
 *   [ observation AFTER the step (obs_dim) | reward | terminated | policy (A) | action one-hot (A) | root value |
 *     child_visits (A) ]
 * float64 keeps Game.policies / Game.child_visits exact; float32 fields widen exactly.  step t in [0,T).
 * obs_dim 0 (obs_dev may be NULL): the record without the observation -- image observations (28 812 floats per frame) are
 * kept by the caller in a float32 buffer of their own (selfplay.TrajectoryChunk.obs) instead of being widened to float64. */
int smz_traj_floats(int obs_dim, int A);
int smz_traj_pack(double *traj_dev, int T, int t, int obs_dim, int A, const float *obs_dev, const float *reward_dev,
                  const uint8_t *terminated_dev, const int32_t *action_dev, const double *policy_dev, const double *child_visits_dev,
                  const float *root_value_dev, int B, smz_stream stream);

/* Replay ingest, vectorised: for every stored position of a [T][B][F] chunk, the game length of its env
 * (steps up to and including the first terminated one; T when ignore_termination), the n-step value target that
 * Game.make_target / Game.make_priority compute (game.py:291-337) with the reference's scalar types (float32 chain when
 * the bootstrap position lies inside the game, float64 past its end), and |root_value - target| (make_priority before
 * `** priority_scale`).  discount_pow_dev [td_steps+1] f64 = discount ** i as computed by the host language (Python's
 * float pow).  length_dev [B] i32, value_target_dev [T][B] f64, abs_td_error_dev [T][B] f64 (may be NULL); positions
 * at or beyond the game length are 0. */
int smz_traj_targets(const double *traj_dev, int T, int obs_dim, int A, int B, int td_steps, const double *discount_pow_dev,
                     int ignore_termination, int32_t *length_dev, double *value_target_dev, double *abs_td_error_dev,
                     smz_stream stream);

/* The same for chunks that hold SEVERAL games per env (envs that restart at once, smz_cartpole_step_ctl on_end = 2;
 * selfplay.chunk_to_games(after_end = "new_game")): game_end_dev [T][B] i32 receives, per row, one past the last row of the
 * game the row belongs to (the row carrying its end flag 1 / 2, or T for the unfinished game at the chunk's end; -1 for a
 * row without a step, flag 3), and every game gets its own targets (game.py:291-337 applied per game).  new_game == 0: rows
 * behind an env's first finished game belong to no game (smz_traj_targets' cut).  length_dev [B] (may be NULL) = end of the
 * env's first game. */
int smz_traj_targets_games(const double *traj_dev, int T, int obs_dim, int A, int B, int td_steps,
                           const double *discount_pow_dev, int ignore_termination, int new_game, int32_t *length_dev,
                           int32_t *game_end_dev, double *value_target_dev, double *abs_td_error_dev, smz_stream stream);

/* ---- inspection ------------------------------------------------------------------------------------------------ */
/* Name of the single-launch search kernel instantiation this handle launched last, spelled as rocprofv3 prints it (e.g.
 * "k_search_mlp<2, 2, 1, false, true, false, false>"; "" before the first launch): measurement code attaches profiler
 * evidence to the kernel that actually ran.  Returns the length. */
int smz_last_kernel(const smz_handle *h, char *buf, int cap);

/* [sync] Copies one tree to the host: up to `cap` nodes into `nodes`; minmax_out[2] = {min, max} (may be NULL);
 * path_out (cap_path entries) / path_len_out = the last recorded search path; root_priors_out [A] f64.
 * Returns the number of allocated nodes (>= 0) or a negative status. */
int smz_debug_dump_tree(smz_handle *h, int tree, smz_node_view *nodes, int cap, float *minmax_out, int32_t *path_out,
                        int cap_path, int32_t *path_len_out, double *root_priors_out);
/* Element-wise out = x / n computed the way the single-launch search divides the pUCT prior term by a visit count
 * (correctly rounded reciprocal table + one FMA correction: smz_device.hpp div_by_count), n in [1, table_size).
 * For tests: the results must equal the IEEE quotients bit for bit. */
int smz_debug_div_by_count(const double *x_dev, const int32_t *n_dev, int count, int table_size, double *out_dev,
                           smz_stream stream);
/* Element-wise out_log = log(x), out_pow = pow(x, y) as the device computes them inside the Dirichlet root noise: glibc's own
 * routines restated operation by operation (csrc/smz_glibc_math.hpp; numpy's legacy gamma sampler calls libm's, the reference
 * draws its noise with it: monte_carlo_tree_search.py:220).  Either output may be NULL (y_dev may then be NULL too for pow).
 * pow's domain: x >= 0, y > 0 finite; NaN outside.  For tests: the results must equal the host libm's bit for bit. */
int smz_debug_glibc_log_pow(const double *x_dev, const double *y_dev, int count, double *out_log_dev, double *out_pow_dev,
                            smz_stream stream);
/* Per-level counters accumulated by smz_select since the last reset (device-side atomics, off by default):
 * levels_out[0] = decision levels, [1] = chance levels, [2] = descents, [3] = children scored.  [sync] */
int smz_enable_stats(smz_handle *h, int on);
int smz_read_stats(smz_handle *h, uint64_t levels_out[4], int reset);

#ifdef __cplusplus
}
#endif
#endif /* SMZ_H */
