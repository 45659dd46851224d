"""Batched evaluation of the learned heads for all pending leaves of a SearchEngine.

The reference calls five tiny networks once per simulation with batch size 1 (muzero_model.py:802-909).  Here the
same five functions are evaluated once per simulation round for ALL B trees.  Which pair of networks a tree needs
depends on its leaf's parent flag (monte_carlo_tree_search.py:333-342); to keep every shape static (so a whole
search can be captured in one HIP graph) both pairs are evaluated for every tree and the HIP epilogue kernels pick
per tree (smz_dynamics_epilogue / smz_prediction_epilogue).  FLOPs are irrelevant here (~30 kFLOP per leaf).

PyTorch-ROCm is used for the dense layers only; softmax, support decode, min-max scaling and the branch select are
libsmz kernels.

Two implementations of the same interface:
  FusedMlpHeads  -- `mlp_model` family (neural_network_mlp_model.py:5-250): the two dynamics trunks share their
                    input and the two prediction trunks share theirs, so each pair is evaluated as ONE stacked
                    linear layer + ONE block-structured output layer (4 GEMMs per round instead of 10).
  ModuleHeads    -- any head family following the reference's module signatures (vision, lstm, ...): calls the
                    five torch modules batched and uses the same epilogues.
Interface:
  initial(obs)                      -> hidden [B,S] f32, policy [B,A] f32 (softmaxed; the root value is unused, mcts:319)
  recurrent(mlp_input/hidden, last_action, branch) -> hidden' [B,S], reward [B], policy [B,A], value [B]
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib

# weight names of the mlp_model family as plain arrays: <head>_<layer>_{w,b}
MLP_ARRAYS = ["rep_in", "rep_mid", "rep_out",
              "pre_in", "pre_mid", "pre_pol", "pre_val",
              "apr_in", "apr_mid", "apr_pol", "apr_val",
              "ady_in", "ady_mid", "ady_st",
              "dyn_in", "dyn_mid", "dyn_rw", "dyn_st"]


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class FusedMlpHeads:
    wants_mlp_input, wants_parent_hidden = True, False

    def __init__(self, weights, dims, device):
        """weights: dict name -> array-like ('rep_in_w', 'rep_in_b', ...); dims: obs, A, S, H, L."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.obs, self.A, self.S, self.H, self.L = (int(dims[k]) for k in ("obs", "A", "S", "H", "L"))
        w = {k: torch.as_tensor(v, dtype=torch.float32).to(self.device) for k, v in weights.items()}
        self.w = w
        A, S, H = self.A, self.S, self.H
        z = lambda r, c: torch.zeros(r, c, device=self.device)
        # representation (mlp:5-42): Linear -> ELU -> [Linear(H,H) -> ELU] x L (one shared module) -> Linear -> scale
        self.rep = [(w["rep_in_w"].t().contiguous(), w["rep_in_b"])] + \
                   [(w["rep_mid_w"].t().contiguous(), w["rep_mid_b"])] * self.L
        self.rep_out = (w["rep_out_w"].t().contiguous(), w["rep_out_b"])
        # root prediction (mlp:47-83): policy and value share the trunk weights
        self.pre = [(w["pre_in_w"].t().contiguous(), w["pre_in_b"])] + \
                   [(w["pre_mid_w"].t().contiguous(), w["pre_mid_b"])] * self.L
        self.pre_pol = (w["pre_pol_w"].t().contiguous(), w["pre_pol_b"])
        # dynamics || afterstate_dynamics, both fed [hidden | one-hot action] (mlp:122-124, 204-206)
        self.dyn_in = (torch.cat([w["dyn_in_w"], w["ady_in_w"]], 0).t().contiguous(),
                       torch.cat([w["dyn_in_b"], w["ady_in_b"]]))
        self.dyn_mid = (torch.block_diag(w["dyn_mid_w"], w["ady_mid_w"]).t().contiguous(),
                        torch.cat([w["dyn_mid_b"], w["ady_mid_b"]]))
        w2 = torch.cat([torch.cat([w["dyn_rw_w"], z(S, H)], 1),     # reward logits      <- dynamics trunk
                        torch.cat([w["dyn_st_w"], z(S, H)], 1),     # next state         <- dynamics trunk
                        torch.cat([z(S, H), w["ady_st_w"]], 1)], 0)  # afterstate         <- afterstate trunk
        self.dyn_out = (w2.t().contiguous(), torch.cat([w["dyn_rw_b"], w["dyn_st_b"], w["ady_st_b"]]))
        # prediction || afterstate_prediction on the chosen next state
        self.prd_in = (torch.cat([w["pre_in_w"], w["apr_in_w"]], 0).t().contiguous(),
                       torch.cat([w["pre_in_b"], w["apr_in_b"]]))
        self.prd_mid = (torch.block_diag(w["pre_mid_w"], w["apr_mid_w"]).t().contiguous(),
                        torch.cat([w["pre_mid_b"], w["apr_mid_b"]]))
        w3 = torch.cat([torch.cat([w["pre_pol_w"], z(A, H)], 1), torch.cat([w["pre_val_w"], z(S, H)], 1),
                        torch.cat([z(A, H), w["apr_pol_w"]], 1), torch.cat([z(S, H), w["apr_val_w"]], 1)], 0)
        self.prd_out = (w3.t().contiguous(),
                        torch.cat([w["pre_pol_b"], w["pre_val_b"], w["apr_pol_b"], w["apr_val_b"]]))
        self._buf = {}

    @classmethod
    def from_npz(cls, path, device):
        import numpy as np
        z = np.load(path)
        dims = {k: int(z["dim_" + k]) for k in ("obs", "A", "S", "H", "L")}
        weights = {n + s: z[n + s] for n in MLP_ARRAYS for s in ("_w", "_b")}
        return cls(weights, dims, device)

    def _out(self, name, shape, dtype=torch.float32):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._buf[name] = torch.empty(*shape, dtype=dtype, device=self.device)
        return t

    @staticmethod
    def _trunk(x, layers):
        for wt, b in layers:
            x = F.elu(torch.addmm(b, x, wt))
        return x

    def initial(self, obs):
        B = obs.shape[0]
        pre = torch.addmm(self.rep_out[1], self._trunk(obs, self.rep), self.rep_out[0])     # [B,S] pre-scale
        hidden = self._out("h0", (B, self.S))
        zero = self._out("zero_branch", (B,), torch.uint8)
        zero.zero_()
        # scale_to_bound_action through the same epilogue (afterstate slot, branch 0, no reward)
        _lib.check(self.lib.smz_dynamics_epilogue(_ptr(pre), _ptr(pre), None, self.S, _ptr(zero), self.S, _ptr(hidden),
                                                  None, B, _stream(self.device)))
        logits = torch.addmm(self.pre_pol[1], self._trunk(hidden, self.pre), self.pre_pol[0])
        policy = self._out("p0", (B, self.A))
        _lib.check(self.lib.smz_policy_softmax(_ptr(logits), self.A, _ptr(policy), B, _stream(self.device)))
        return hidden, policy

    def recurrent(self, engine):
        """Consumes engine.mlp_input / engine.branch (outputs of the last select); returns the four tensors that
        smz_expand_backup takes."""
        x, branch = engine.mlp_input, engine.branch
        B, A, S = x.shape[0], self.A, self.S
        t = F.elu(torch.addmm(self.dyn_in[1], x, self.dyn_in[0]))
        for _ in range(self.L):
            t = F.elu(torch.addmm(self.dyn_mid[1], t, self.dyn_mid[0]))
        o = torch.addmm(self.dyn_out[1], t, self.dyn_out[0])                                 # [B, 3S]
        hidden = self._out("h", (B, S))
        reward = self._out("r", (B,))
        fs = o.element_size()
        base = o.data_ptr()
        _lib.check(self.lib.smz_dynamics_epilogue(C.c_void_p(base + S * fs), C.c_void_p(base + 2 * S * fs),
                                                  C.c_void_p(base), 3 * S, _ptr(branch), S, _ptr(hidden), _ptr(reward),
                                                  B, _stream(self.device)))
        u = F.elu(torch.addmm(self.prd_in[1], hidden, self.prd_in[0]))
        for _ in range(self.L):
            u = F.elu(torch.addmm(self.prd_mid[1], u, self.prd_mid[0]))
        q = torch.addmm(self.prd_out[1], u, self.prd_out[0])                                 # [B, 2A+2S]
        policy = self._out("p", (B, A))
        value = self._out("v", (B,))
        qb = q.data_ptr()
        _lib.check(self.lib.smz_prediction_epilogue(C.c_void_p(qb), C.c_void_p(qb + A * fs), C.c_void_p(qb + (A + S) * fs),
                                                    C.c_void_p(qb + (2 * A + S) * fs), 2 * A + 2 * S, _ptr(branch), A, S,
                                                    _ptr(policy), _ptr(value), B, _stream(self.device)))
        return hidden, reward, policy, value


class HipMlpHeads:
    """`mlp_model` heads evaluated by ONE hand-written HIP kernel per phase (smz_mlp_initial / smz_mlp_recurrent):
    weights packed once into the LDS layout described in include/smz.h, no library GEMMs, no torch ops.
    Raises ValueError when the networks do not fit a CU's LDS (use FusedMlpHeads then)."""
    wants_mlp_input, wants_parent_hidden = True, False
    IN_PLACE_MIN = 8192       # from this many trees on (shipped network shape) the rows stay in the tree: see bind_engine
    # off[] order of smz_mlp_desc: (name of the input-major matrix, [names of the output heads concatenated])
    _MATS = [("dyn_in", ["dyn_in"]), ("ady_in", ["ady_in"]), ("dyn_mid", ["dyn_mid"]), ("ady_mid", ["ady_mid"]),
             ("dyn_out", ["dyn_rw", "dyn_st"]), ("ady_out", ["ady_st"]), ("pre_in", ["pre_in"]), ("apr_in", ["apr_in"]),
             ("pre_mid", ["pre_mid"]), ("apr_mid", ["apr_mid"]), ("pre_out", ["pre_pol", "pre_val"]),
             ("apr_out", ["apr_pol", "apr_val"]), ("rep_in", ["rep_in"]), ("rep_mid", ["rep_mid"]), ("rep_out", ["rep_out"])]

    def __init__(self, weights, dims, device):
        import numpy as np
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.obs, self.A, self.S, self.H, self.L = (int(dims[k]) for k in ("obs", "A", "S", "H", "L"))
        d = _lib.MlpDesc(self.obs, self.A, self.S, self.H, self.L)
        if self.lib.smz_mlp_layout(C.byref(d)) != 0:
            raise ValueError("mlp heads do not fit the LDS-resident kernel (use FusedMlpHeads)")
        self.desc = d
        buf = np.zeros(d.total_floats, np.float32)
        OP = d.OP
        for m, (_, parts) in enumerate(self._MATS):
            if "_mid" in parts[0] and self.L == 0:
                continue
            W = np.concatenate([np.asarray(weights[p + "_w"], np.float32) for p in parts], 0)      # [O, K] (torch layout)
            b = np.concatenate([np.asarray(weights[p + "_b"], np.float32) for p in parts], 0)
            O, K = W.shape
            K4 = (K + 3) & ~3
            blk = np.zeros((K4 // 4, OP, 4), np.float32)
            Wt = np.zeros((K4, OP), np.float32)
            Wt[:K, :O] = W.T
            blk[:] = Wt.reshape(K4 // 4, 4, OP).transpose(0, 2, 1)
            buf[d.off[m]:d.off[m] + K4 * OP] = blk.reshape(-1)
            buf[d.off[15 + m]:d.off[15 + m] + O] = b
        self.weights = torch.from_numpy(buf).to(self.device)
        self._buf = {}

    def _out(self, name, shape, dtype=torch.float32):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._buf[name] = torch.empty(*shape, dtype=dtype, device=self.device)
        return t

    def initial(self, obs):
        B = obs.shape[0]
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.obs
        hidden, policy = self._out("h0", (B, self.S)), self._out("p0", (B, self.A))
        _lib.check(self.lib.smz_mlp_initial(C.byref(self.desc), _ptr(self.weights), _ptr(obs), _ptr(hidden), _ptr(policy),
                                            B, _stream(self.device)))
        return hidden, policy

    def bind_engine(self, engine):
        """Called by the step-wise search before its first selection.  Large batches of the shipped network shape
        (S 31, H 64, L 0; 2 or 4 actions): the matrix-core kernel reads each leaf's parent row from the tree's
        hidden-state storage and writes the new row into it (smz_mlp_recurrent_rows), so the tree kernels gather and
        scatter no rows -- a quarter of their memory traffic at 10^6 trees.  Returns the row arguments of the tree calls."""
        import os
        lim = int(os.environ.get("SMZ_MLP_IN_PLACE_MIN", self.IN_PLACE_MIN))
        shape_ok = self.S == 31 and self.H == 64 and self.L == 0 and self.A in (2, 4)
        in_place = shape_ok and lim >= 0 and engine.B >= lim and hasattr(self.lib, "smz_mlp_recurrent_rows")
        engine.enable_leaf_ids(in_place)
        self._in_place = engine if in_place else None
        return dict(want_mlp_input=not in_place, want_parent_hidden=False)

    def recurrent(self, engine):
        in_place = getattr(self, "_in_place", None) is engine
        B = engine.B if in_place else engine.mlp_input.shape[0]
        reward, policy, value = self._out("r", (B,)), self._out("p", (B, self.A)), self._out("v", (B,))
        if in_place:
            base, n, hs = engine.hidden_layout()
            _lib.check(self.lib.smz_mlp_recurrent_rows(C.byref(self.desc), _ptr(self.weights), base, n, hs, _ptr(engine.leaf_ids),
                                                       _ptr(engine.last_action), _ptr(engine.branch), _ptr(reward), _ptr(policy),
                                                       _ptr(value), B, _stream(self.device)))
            return None, reward, policy, value
        x, branch = engine.mlp_input, engine.branch
        hidden = self._out("h", (B, self.S))
        _lib.check(self.lib.smz_mlp_recurrent(C.byref(self.desc), _ptr(self.weights), _ptr(x), _ptr(branch), _ptr(hidden),
                                              _ptr(reward), _ptr(policy), _ptr(value), B, _stream(self.device)))
        return hidden, reward, policy, value


class HipMlpTileHeads(FusedMlpHeads):
    """`mlp_model` heads too wide for the LDS-resident kernels (H <= 128, 2 S <= 128, any number_of_hidden_layer: the
    reference's config 434, S 61 / H 126 / L 0, and its checkpoint 450, L 4): the recurrent networks as ONE hand-written kernel per simulation round
    (smz_mlp_recurrent_wide: 16-leaf tiles on the matrix cores, weights streamed from L2 in the 128-wide packed layout of
    smz_mlp_layout_wide) instead of ten library launches; the root evaluation stays on the torch GEMMs of FusedMlpHeads
    (once per search).  Raises ValueError outside the kernel's limits."""
    wants_mlp_input, wants_parent_hidden = True, False

    def __init__(self, weights, dims, device):
        import numpy as np
        super().__init__(weights, dims, device)
        d = _lib.MlpDesc(int(dims["obs"]), self.A, self.S, self.H, self.L)
        if not hasattr(self.lib, "smz_mlp_layout_wide") or self.lib.smz_mlp_layout_wide(C.byref(d)) != 0:
            raise ValueError("mlp heads outside the limits of the wide tile kernel (use FusedMlpHeads)")
        self.wide_desc = d      # (not `desc`: that name marks heads the single-launch search kernel can take)
        buf = np.zeros(d.total_floats, np.float32)
        OP = d.OP
        for m, (_, parts) in enumerate(HipMlpHeads._MATS):
            if "_mid" in parts[0] and self.L == 0:
                continue
            W = np.concatenate([np.asarray(weights[p + "_w"], np.float32) for p in parts], 0)      # [O, K] (torch layout)
            b = np.concatenate([np.asarray(weights[p + "_b"], np.float32) for p in parts], 0)
            O, K = W.shape
            K8 = (K + 7) & ~7
            Wt = np.zeros((K8, OP), np.float32)
            Wt[:K, :O] = W.T
            buf[d.off[m]:d.off[m] + K8 * OP] = Wt.reshape(K8 // 4, 4, OP).transpose(0, 2, 1).reshape(-1)
            buf[d.off[15 + m]:d.off[15 + m] + O] = b
        self.packed = torch.from_numpy(buf).to(self.device)

    def recurrent(self, engine):
        x, branch = engine.mlp_input, engine.branch
        B = x.shape[0]
        hidden, reward = self._out("h", (B, self.S)), self._out("r", (B,))
        policy, value = self._out("p", (B, self.A)), self._out("v", (B,))
        _lib.check(self.lib.smz_mlp_recurrent_wide(C.byref(self.wide_desc), _ptr(self.packed), _ptr(x), _ptr(branch), _ptr(hidden),
                                                   _ptr(reward), _ptr(policy), _ptr(value), B, _stream(self.device)))
        return hidden, reward, policy, value


class HipVisionHeads:
    """`vision_model` heads (the reference's ResNet-v2 family, compat_vision.py) evaluated by hand-written HIP kernels:
    smz_vision_initial (one workgroup per frame) and smz_vision_recurrent (one wavefront per leaf).  Weights are
    packed once from the five modules into the layout documented at smz_vision_desc (include/smz.h)."""
    wants_mlp_input, wants_parent_hidden = False, True
    is_rgb = True
    records_frames = True       # initial(obs, record=...) appends the frames to the trajectory record in the same launch
    # index constants of include/smz.h
    _T = dict(CONV_IN=0, BN_IN=1, RES_A=2, RES_B=3, RES_BN=4, MIX_W=5, MIX_B=6, TOWER=7, STRIDE=13, BASE=0)
    _P = dict(RES_A=0, RES_B=1, RES_BN=2, VMIX_W=3, VMIX_B=4, VTOWER=5, PMIX_W=11, PMIX_B=12, PTOWER=13, STRIDE=19, BASE=26)
    _R = dict(STEM=0, NARROW_A=1, NARROW_B=2, NARROW_BN=3, WIDEN=4, WIDE_A=5, WIDE_B=6, WIDE_BN=7, LAST_A=8, LAST_B=9,
              LAST_BN=10, BASE=64)

    def __init__(self, representation, prediction, afterstate_prediction, afterstate_dynamics, dynamics, num_actions,
                 support_size, device):
        import numpy as np
        self.lib = _lib.load()
        self.device = torch.device(device)
        tower = dynamics.sequential_reward[2]
        linears = [m for m in tower if isinstance(m, torch.nn.Linear)]
        self.A, self.Ssup, self.H = int(num_actions), int(support_size), int(linears[0].out_features)
        self.L = (len(tower) - 3) // 2                  # [Linear, relu] + [Linear, relu] * L + [Linear]
        if linears[0].in_features != 147 or not getattr(representation, "down_sampling", True):
            raise ValueError("the HIP vision kernels cover the down-sampling family (98x98x3 frames, 3x7x7 hidden state)")
        d = _lib.VisionDesc(self.A, self.Ssup, self.H, self.L)
        if self.lib.smz_vision_layout(C.byref(d)) != 0:
            raise ValueError("vision heads outside the kernel's limits (A, S, H <= 64)")
        self.desc, self.S = d, 147
        buf = np.zeros(d.total_floats, np.float32)
        OP = d.OP

        def put(idx, arr):
            a = np.asarray(arr.detach().cpu().numpy() if torch.is_tensor(arr) else arr, np.float32).reshape(-1)
            buf[d.off[idx]:d.off[idx] + a.size] = a

        def put_bn(idx, bn):
            # ATen's CPU eval-mode batch-norm: alpha = weight * (1 / sqrt(var + eps)), beta = bias - mean * alpha (float32)
            var, mean = bn.running_var.detach().cpu().numpy().astype(np.float32), bn.running_mean.detach().cpu().numpy().astype(np.float32)
            invstd = (np.float32(1) / np.sqrt(var + np.float32(bn.eps))).astype(np.float32)
            w = bn.weight.detach().cpu().numpy().astype(np.float32) if bn.affine else np.ones_like(var)
            b = bn.bias.detach().cpu().numpy().astype(np.float32) if bn.affine else np.zeros_like(var)
            alpha = (w * invstd).astype(np.float32)
            put(idx, np.concatenate([alpha, (b - mean * alpha).astype(np.float32)]))

        def put_linear(idx, lin):
            W, b = lin.weight.detach().cpu().numpy().astype(np.float32), lin.bias.detach().cpu().numpy().astype(np.float32)
            O, K = W.shape
            K4 = (K + 3) & ~3
            Wt = np.zeros((K4, OP), np.float32)
            Wt[:K, :O] = W.T
            put(idx, Wt.reshape(K4 // 4, 4, OP).transpose(0, 2, 1))
            put(idx + 1, b)

        def put_tower(idx, seq):
            lins = [m for m in seq if isinstance(m, torch.nn.Linear)]
            put_linear(idx, lins[0])
            if self.L > 0:
                put_linear(idx + 2, lins[1])
            put_linear(idx + 4, lins[-1])

        def put_block(ia, ib, ibn, block):
            sc = block.sequential_container         # bn, relu, convA, bn, relu, convB, bn, relu, convA
            put(ia, sc[2].weight); put(ib, sc[5].weight); put_bn(ibn, sc[0])

        T, P, R = self._T, self._P, self._R
        for n, mod in enumerate((dynamics, afterstate_dynamics)):
            b = T["BASE"] + n * T["STRIDE"]
            sc = mod.sequential_container           # conv, bn, relu, [block] * L, relu
            put(b + T["CONV_IN"], sc[0].weight); put_bn(b + T["BN_IN"], sc[1])
            if self.L > 0:
                put_block(b + T["RES_A"], b + T["RES_B"], b + T["RES_BN"], sc[3])
            if n == 0:
                rw = mod.sequential_reward          # conv1x1, flatten, tower
                put(b + T["MIX_W"], rw[0].weight); put(b + T["MIX_B"], rw[0].bias); put_tower(b + T["TOWER"], rw[2])
        for n, mod in enumerate((prediction, afterstate_prediction)):
            b = P["BASE"] + n * P["STRIDE"]
            if self.L > 0:
                put_block(b + P["RES_A"], b + P["RES_B"], b + P["RES_BN"], mod.resnet[0])
            put(b + P["VMIX_W"], mod.nn_value[0].weight); put(b + P["VMIX_B"], mod.nn_value[0].bias)
            put_tower(b + P["VTOWER"], mod.nn_value[2])
            put(b + P["PMIX_W"], mod.nn_policy[0].weight); put(b + P["PMIX_B"], mod.nn_policy[0].bias)
            put_tower(b + P["PTOWER"], mod.nn_policy[2])
        down = representation.sequential_downsampler[0].sequential_container   # stem, R1, R1, widen, R2, R2, pool, R2 x3, pool
        b = R["BASE"]
        put(b + R["STEM"], down[0].weight); put_block(b + R["NARROW_A"], b + R["NARROW_B"], b + R["NARROW_BN"], down[1])
        put(b + R["WIDEN"], down[3].weight); put_block(b + R["WIDE_A"], b + R["WIDE_B"], b + R["WIDE_BN"], down[4])
        put_block(b + R["LAST_A"], b + R["LAST_B"], b + R["LAST_BN"], representation.sequential_downsampler[1])
        self.weights = torch.from_numpy(buf).to(self.device)
        self._buf = {}

    def _out(self, name, shape, dtype=torch.float32):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._buf[name] = torch.empty(*shape, dtype=dtype, device=self.device)
        return t

    def initial(self, obs, record=None):
        """`record` ([B, 3*98*98] or [B,3,98,98] float32, contiguous): the launch also copies the frames there -- the trajectory
        record of the observation, written by the threads that read it (smz_vision_initial_record)."""
        B = obs.shape[0]
        assert obs.dtype == torch.float32 and obs.is_contiguous() and tuple(obs.shape[1:]) == (3, 98, 98)
        assert record is None or (record.dtype == torch.float32 and record.is_contiguous() and record.numel() == obs.numel())
        hidden, policy = self._out("h0", (B, 147)), self._out("p0", (B, self.A))
        _lib.check(self.lib.smz_vision_initial_record(C.byref(self.desc), _ptr(self.weights), _ptr(obs),
                                                      None if record is None else _ptr(record), _ptr(hidden),
                                                      _ptr(policy), B, _stream(self.device)))
        return hidden, policy

    def recurrent(self, engine):
        ph = engine.parent_hidden
        B = ph.shape[0]
        hidden, reward = self._out("h", (B, 147)), self._out("r", (B,))
        policy, value = self._out("p", (B, self.A)), self._out("v", (B,))
        _lib.check(self.lib.smz_vision_recurrent(C.byref(self.desc), _ptr(self.weights), _ptr(ph), ph.stride(0),
                                                 _ptr(engine.last_action), _ptr(engine.branch), _ptr(hidden), _ptr(reward),
                                                 _ptr(policy), _ptr(value), B, _stream(self.device)))
        return hidden, reward, policy, value


class ModuleHeads:
    """Heads given as five torch modules with the reference's signatures:
         representation(obs) -> hidden                                   (already scaled)
         prediction(h), afterstate_prediction(h) -> (policy_logits, value_logits)
         afterstate_dynamics(h, action_encoding) -> hidden
         dynamics(h, action_encoding) -> (reward_logits, hidden)
       `encode_action(last_action[B] int64, hidden) -> tensor` builds the action input (one-hot for vector
       observations, constant plane (a+1)/A for RGB: muzero_model.py:496-523)."""

    wants_mlp_input, wants_parent_hidden = False, True

    def __init__(self, representation, prediction, afterstate_prediction, afterstate_dynamics, dynamics, num_actions,
                 support_size, device, is_rgb=False):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.rep, self.pre, self.apr, self.ady, self.dyn = (m.to(self.device).eval() for m in (
            representation, prediction, afterstate_prediction, afterstate_dynamics, dynamics))
        self.A, self.Ssup, self.is_rgb = int(num_actions), int(support_size), bool(is_rgb)
        self._buf = {}

    def _out(self, name, shape, dtype=torch.float32):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._buf[name] = torch.empty(*shape, dtype=dtype, device=self.device)
        return t

    def encode_action(self, action, hidden):
        if not self.is_rgb:
            return F.one_hot(action.long(), self.A).to(hidden.dtype)
        plane = (action.to(hidden.dtype) + 1) / self.A
        return plane.view(-1, 1, 1, 1).expand(-1, 1, hidden.shape[2], hidden.shape[3]).contiguous()

    @torch.no_grad()
    def initial(self, obs):
        hidden = self.rep(obs)
        self.hidden_shape = tuple(hidden.shape[1:])
        logits, _ = self.pre(hidden)
        B = obs.shape[0]
        policy = self._out("p0", (B, self.A))
        logits = logits.float().contiguous()
        _lib.check(self.lib.smz_policy_softmax(_ptr(logits), self.A, _ptr(policy), B, _stream(self.device)))
        return hidden.reshape(B, -1).float().contiguous(), policy

    @torch.no_grad()
    def recurrent(self, engine):
        B = engine.B
        h = engine.parent_hidden[:, :engine.S].reshape((B,) + self.hidden_shape)
        enc = self.encode_action(engine.last_action, h)
        reward_logits, s_dyn = self.dyn(h, enc)
        s_aft = self.ady(h, enc)
        m = engine.branch.bool()
        hidden = torch.where(m.view((B,) + (1,) * (s_dyn.dim() - 1)), s_dyn, s_aft).contiguous()
        rl = reward_logits.float().contiguous()
        rdec = self._out("rdec", (B,))
        _lib.check(self.lib.smz_support_decode(_ptr(rl), self.Ssup, _ptr(rdec), B, _stream(self.device)))
        reward = torch.where(m, rdec, torch.zeros_like(rdec))
        pp, vp = self.pre(hidden)
        pa, va = self.apr(hidden)
        pp, vp, pa, va = (t.float().contiguous() for t in (pp, vp, pa, va))
        policy = self._out("p", (B, self.A))
        value = self._out("v", (B,))
        # policy rows (A wide) and value rows (S wide) have different strides: one epilogue call per width
        _lib.check(self.lib.smz_policy_softmax(_ptr(torch.where(m[:, None], pp, pa).contiguous()), self.A, _ptr(policy),
                                               B, _stream(self.device)))
        _lib.check(self.lib.smz_support_decode(_ptr(torch.where(m[:, None], vp, va).contiguous()), self.Ssup,
                                               _ptr(value), B, _stream(self.device)))
        return hidden.reshape(B, -1).float().contiguous(), reward.contiguous(), policy, value
