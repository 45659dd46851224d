"""Inference half of the reference's `Muzero` wrapper and its checkpoint surface.

Keeps what the self-play path touches (SURVEY section 8b "Model object" and "Checkpoint / config"):
  * files  `{dir}/{tag}_muzero_{representation,prediction,afterstate_prediction,afterstate_dynamics,dynamics,
    encoder}_function.pt` (whole pickled modules) + `{dir}/{tag}_muzero_init_variables.json` with the reference's
    keys (muzero_model.py:911-996);
  * the five batch-1 `*_inference` methods with the reference's return types (muzero_model.py:802-909), so the
    object can be handed to any code written against the reference's model;
  * `.heads(device)`: the batched evaluator the GPU engine uses (heads.py).
Training (losses, optimiser, `train()`) is out of scope of this engine and is not provided.
"""
import json
import os
import sys

import numpy as np
import torch

from . import compat_mlp, compat_vision
from .heads import MLP_ARRAYS, FusedMlpHeads, HipMlpHeads, HipMlpTileHeads, HipVisionHeads, ModuleHeads

_FUNCS = ("representation", "prediction", "afterstate_prediction", "afterstate_dynamics", "dynamics", "encoder")
_FAMILY_MODULE = {"mlp_model": "neural_network_mlp_model", "lstm_model": "neural_network_lstm_model",
                  "vision_model": "neural_network_vision_model",
                  "vision_conv_lstm_model": "neural_network_vision_conv_lstm_model",
                  "transformer_model": "neural_network_transformer_decoder_model"}

for _n in ("Representation_function", "Prediction_function", "Afterstate_prediction_function",
           "Afterstate_dynamics_function", "Dynamics_function", "Encoder_function", "StraightThroughEstimator",
           "Onehot_argmax"):
    getattr(compat_mlp, _n).__module__ = "neural_network_mlp_model"   # the name the reference's pickles carry
for _n in ("Representation_function", "Prediction_function", "Afterstate_prediction_function",
           "Afterstate_dynamics_function", "Dynamics_function", "Encoder_function", "Residual_block", "Down_sample",
           "Onehot_argmax"):
    getattr(compat_vision, _n).__module__ = "neural_network_vision_model"


_COMPAT = {"neural_network_mlp_model": compat_mlp, "neural_network_vision_model": compat_vision}


def _is_compat_class(cls):
    """True for this package's re-declaration of a head class (its __module__ carries the reference's module name, so
    only identity tells it from the reference's own class)."""
    return cls.__module__ in _COMPAT and getattr(_COMPAT[cls.__module__], cls.__name__, None) is cls


class _compat_modules_bound:
    """While inside: `neural_network_{mlp,vision}_model` resolve to this package's re-declarations, so that torch.save
    writes -- and torch.load of a reference checkpoint finds -- the class paths the reference's pickles carry.  Whatever
    was registered under those names before (the reference's real modules, when a process has imported them) is put
    back afterwards; nothing stays registered globally."""

    def __init__(self, force=True):
        self.force = force

    def __enter__(self):
        self.saved = {n: sys.modules.get(n) for n in _COMPAT}
        for n, m in _COMPAT.items():
            if self.force or self.saved[n] is None:
                sys.modules[n] = m
        return self

    def __exit__(self, *exc):
        for n, old in self.saved.items():
            if old is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = old
        return False


def _linears(seq):
    return [m for m in seq if isinstance(m, torch.nn.Linear)]


def mlp_arrays_from_modules(rep, pre, apr, ady, dyn):
    """name -> float32 tensor, names as heads.MLP_ARRAYS (+ _w/_b), from modules with the reference's attributes."""
    out = {}

    def trunk(prefix, seq):
        lin = _linears(seq)
        H = lin[0].weight.shape[0]
        out[prefix + "_in_w"], out[prefix + "_in_b"] = lin[0].weight, lin[0].bias
        if len(lin) > 2:
            out[prefix + "_mid_w"], out[prefix + "_mid_b"] = lin[1].weight, lin[1].bias
        else:
            out[prefix + "_mid_w"], out[prefix + "_mid_b"] = torch.zeros(H, H), torch.zeros(H)
        return lin[-1], len(lin) - 2 if len(lin) > 2 else 0

    def unwrap(m):
        return m.module if isinstance(m, torch.nn.DataParallel) else m
    rep, pre, apr, ady, dyn = (unwrap(m) for m in (rep, pre, apr, ady, dyn))
    last, _ = trunk("rep", rep.state_norm); out["rep_out_w"], out["rep_out_b"] = last.weight, last.bias
    for tag, mod in (("pre", pre), ("apr", apr)):
        last, _ = trunk(tag, mod.policy); out[tag + "_pol_w"], out[tag + "_pol_b"] = last.weight, last.bias
        v = _linears(mod.value)[-1]; out[tag + "_val_w"], out[tag + "_val_b"] = v.weight, v.bias
    last, _ = trunk("ady", ady.next_state_normalized); out["ady_st_w"], out["ady_st_b"] = last.weight, last.bias
    last, _ = trunk("dyn", dyn.next_state_normalized); out["dyn_st_w"], out["dyn_st_b"] = last.weight, last.bias
    r = _linears(dyn.reward)[-1]; out["dyn_rw_w"], out["dyn_rw_b"] = r.weight, r.bias
    return {k: v.detach().float().cpu() for k, v in out.items()}


class Muzero:
    """Inference-side stand-in for the reference's Muzero object."""

    def __init__(self, model_structure="mlp_model", observation_space_dimensions=None, action_space_dimensions=None,
                 state_space_dimensions=9, hidden_layer_dimensions=16, number_of_hidden_layer=1, device="cpu",
                 load=False, action_map=None, random_tag=None, extra_init_variables=None):
        assert model_structure in _FAMILY_MODULE, "model_structure ∈ {mlp_model,lstm_model,vision_model,vision_conv_lstm_model,transformer_model}"
        self.model_structure = model_structure
        self.state_dimension = int(state_space_dimensions)
        self.hidden_layer_dimension = int(hidden_layer_dimensions)
        self.number_of_hidden_layer = int(number_of_hidden_layer)
        self.device = device
        self.is_RGB = "vision" in model_structure
        self.type_format = torch.float32
        self.use_amp = False
        self.random_tag = int(random_tag) if random_tag is not None else int(np.random.RandomState().randint(0, 100000000))
        self.extra = dict(extra_init_variables or {})
        self._heads = {}
        self._heads_version = {}
        if load:
            return
        if model_structure not in ("mlp_model", "vision_model"):
            raise NotImplementedError("fresh construction is provided for mlp_model and vision_model; other "
                                      "families load from checkpoints written by the reference")
        n_act = int(action_space_dimensions)
        self.action_dictionnary = list(action_map) if action_map is not None else list(range(n_act))
        self.action_dimension = len(self.action_dictionnary)
        if self.is_RGB:
            family = compat_vision
            self.observation_dimension = compat_vision.FRAME          # muzero_model.py:400-404: fixed for vision
        else:
            family = compat_mlp
            self.observation_dimension = int(observation_space_dimensions)
        kw = dict(state_dimension=self.state_dimension, action_dimension=self.action_dimension,
                  observation_space_dimensions=self.observation_dimension,
                  hidden_layer_dimensions=self.hidden_layer_dimension, number_of_hidden_layer=self.number_of_hidden_layer)
        # construction order and generator draws = muzero_model.py:300-335 (compat_mlp._unused_draws), so an equal torch
        # seed gives the reference's initial weights (pinned for both families by goldens the reference initialised)
        self.representation_function = family.Representation_function(**kw)
        self.prediction_function = family.Prediction_function(**kw)
        self.afterstate_prediction_function = family.Afterstate_prediction_function(**kw)
        self.afterstate_dynamics_function = family.Afterstate_dynamics_function(**kw)
        self.dynamics_function = family.Dynamics_function(**kw)
        self.encoder_function = family.Encoder_function(**kw)
        for f in _FUNCS:
            if not self.is_RGB:
                getattr(self, f + "_function").apply(compat_mlp.weights_init)     # muzero_model.py:350-358 (mlp only)
            getattr(self, f + "_function").eval()                                  # inference side: :803-804

    # ---- checkpoint surface (muzero_model.py:911-996) ---------------------------------------------------------
    def init_variables(self):
        base = {"model_structure": self.model_structure, "observation_space_dimensions": self.observation_dimension,
                "action_space_dimensions": self.action_dimension, "state_space_dimensions": self.state_dimension,
                "k_hypothetical_steps": 10, "learning_rate": 0.01, "optimizer": "adam", "loss_type": "general",
                "lr_scheduler": None, "num_of_epoch": 1, "device": "cpu" if self.device == "cpu" else "cuda",
                "hidden_layer_dimensions": self.hidden_layer_dimension,
                "number_of_hidden_layer": self.number_of_hidden_layer, "random_tag": self.random_tag,
                "action_map": self.action_dictionnary, "use_amp": False, "priority_scale": 1, "rescale_value_loss": 1}
        base.update({k: v for k, v in self.extra.items() if k in base and k not in (
            "model_structure", "observation_space_dimensions", "action_space_dimensions", "state_space_dimensions",
            "hidden_layer_dimensions", "number_of_hidden_layer", "random_tag", "action_map")})
        return base

    def save_model(self, directory="model_checkpoint", tag=None, model_update_or_backtrack=None):
        if model_update_or_backtrack is not None:
            return
        os.makedirs(directory, exist_ok=True)
        if tag != 0:                  # (the reference's rule, muzero_model.py:917-918)
            self.random_tag = tag
        ours = all(_is_compat_class(type(getattr(self, f + "_function"))) for f in _FUNCS)
        # modules of this package pickle under the reference's class paths; modules that ARE the reference's (a model
        # trained by the reference and handed over) pickle through whatever is registered
        with _compat_modules_bound(force=ours):
            for f in _FUNCS:
                torch.save(getattr(self, f + "_function"), f"{directory}/{self.random_tag}_muzero_{f}_function.pt")
        with open(f"{directory}/{self.random_tag}_muzero_init_variables.json", "w") as fh:
            json.dump(self.init_variables(), fh)

    def load_model(self, model_directory="model_checkpoint", tag=0, observation_space_dimensions=None,
                   type_format=torch.float32, device=None):
        with open(f"{model_directory}/{tag}_muzero_init_variables.json", "r") as fh:
            iv = json.load(fh)
        self.extra = dict(iv)
        self.model_structure = iv["model_structure"]
        self.state_dimension = iv["state_space_dimensions"]
        self.hidden_layer_dimension = iv["hidden_layer_dimensions"]
        self.number_of_hidden_layer = iv["number_of_hidden_layer"]
        self.observation_dimension = iv["observation_space_dimensions"]
        self.action_dictionnary = iv["action_map"]
        self.action_dimension = len(self.action_dictionnary)
        self.device = device if device is not None else iv["device"]
        self.is_RGB = self.model_structure == "vision_model"
        with _compat_modules_bound(force=False):      # the reference's own modules win when they are importable
            for f in _FUNCS:
                path = f'{model_directory}/{iv["random_tag"]}_muzero_{f}_function.pt'
                mod = torch.load(path, map_location="cpu", weights_only=False)   # whole-module pickle
                setattr(self, f + "_function", mod.to(torch.float32).eval())
        self.random_tag = tag if tag > 0 else iv["random_tag"]
        self._heads = {}
        return self

    @classmethod
    def from_checkpoint(cls, model_directory="model_checkpoint", tag=0, device="cpu"):
        m = cls(load=True, device=device)
        return m.load_model(model_directory=model_directory, tag=tag, device=device)

    @classmethod
    def from_state_dicts(cls, path_or_dict, device="cpu"):
        """Builds a model from the six modules' state_dicts stored flat as "<function>/<key>" arrays plus
        `meta_*` scalars (model_structure, A, S, H, L, obs) -- the layout of tests/golden/visionnet_*.npz."""
        z = np.load(path_or_dict) if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
        m = cls(model_structure=str(z["meta_model_structure"]), observation_space_dimensions=int(z["meta_obs"]),
                action_space_dimensions=int(z["meta_A"]), state_space_dimensions=int(z["meta_S"]),
                hidden_layer_dimensions=int(z["meta_H"]), number_of_hidden_layer=int(z["meta_L"]), device=device,
                random_tag=0)
        for f in _FUNCS:
            sd = {k[len(f) + 1:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(f + "/")}
            getattr(m, f + "_function").load_state_dict(sd, strict=True)
        return m

    @classmethod
    def from_arrays(cls, path_or_dict, device="cpu"):
        """Builds an mlp_model from plain weight arrays (the layout of tests/golden/weights_*.npz)."""
        z = np.load(path_or_dict) if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
        d = {k: int(z["dim_" + k]) for k in ("obs", "A", "S", "H", "L")}
        m = cls(model_structure="mlp_model", observation_space_dimensions=d["obs"], action_space_dimensions=d["A"],
                state_space_dimensions=d["S"], hidden_layer_dimensions=d["H"], number_of_hidden_layer=d["L"],
                device=device, random_tag=0)

        def put(seq, idx, name):
            lin = _linears(seq)[idx]
            with torch.no_grad():
                lin.weight.copy_(torch.as_tensor(z[name + "_w"])); lin.bias.copy_(torch.as_tensor(z[name + "_b"]))
        L = d["L"]
        for seq, pre, outs in ((m.representation_function.state_norm, "rep", [("rep_out", None)]),
                               (m.prediction_function.policy, "pre", []), (m.afterstate_prediction_function.policy, "apr", []),
                               (m.afterstate_dynamics_function.next_state_normalized, "ady", []),
                               (m.dynamics_function.next_state_normalized, "dyn", [])):
            put(seq, 0, pre + "_in")
            if L > 0:
                put(seq, 1, pre + "_mid")
        put(m.representation_function.state_norm, -1, "rep_out")
        put(m.prediction_function.policy, -1, "pre_pol"); put(m.prediction_function.value, -1, "pre_val")
        put(m.afterstate_prediction_function.policy, -1, "apr_pol"); put(m.afterstate_prediction_function.value, -1, "apr_val")
        put(m.afterstate_dynamics_function.next_state_normalized, -1, "ady_st")
        put(m.dynamics_function.next_state_normalized, -1, "dyn_st"); put(m.dynamics_function.reward, -1, "dyn_rw")
        return m

    # ---- batched heads for the GPU engine -----------------------------------------------------------------------
    def weights_version(self):
        """Changes whenever a parameter or buffer of the five search-side modules is replaced or written in place
        (torch bumps a tensor's _version on every in-place write: optimizer steps, copy_, load_state_dict)."""
        out = []
        for f in _FUNCS[:5]:
            mod = getattr(self, f + "_function", None)
            if mod is None:
                continue
            for t in list(mod.parameters()) + list(mod.buffers()):
                out.append((id(t), t._version))
        return hash(tuple(out))

    def refresh_heads(self):
        """Drops every packed / copied evaluator: the next heads() call packs the modules' current weights.  heads()
        does this by itself when weights_version() changed; call it after swapping whole modules by other means."""
        self._heads = {}
        self._heads_version = {}

    def heads(self, device, instance=0, backend="auto"):
        """Batched evaluator on `device`.  backend: "hip" = the fused LDS-resident HIP kernel (mlp_model only, when
        the networks fit a CU's LDS), "torch" = torch-ROCm GEMMs + HIP epilogues, "auto" = hip when possible, else (mlp_model) the wide tile
        kernel for the recurrent networks when the shape is within its limits, else torch.
        `instance` distinguishes evaluators that must not share output buffers (one per concurrent stream group)."""
        key = (str(device), instance, backend)
        version = self.weights_version()
        if key in self._heads and self._heads_version.get(key) != version:
            del self._heads[key]              # the modules were updated in place (optimizer step, load_state_dict, ...)
        self._heads_version[key] = version
        if key not in self._heads:
            if self.model_structure == "mlp_model" and backend in ("auto", "hip"):
                arrays = mlp_arrays_from_modules(self.representation_function, self.prediction_function,
                                                 self.afterstate_prediction_function, self.afterstate_dynamics_function,
                                                 self.dynamics_function)
                dims = dict(obs=self.observation_dimension, A=self.action_dimension, S=self.state_dimension,
                            H=self.hidden_layer_dimension, L=self.number_of_hidden_layer)
                try:
                    self._heads[key] = HipMlpHeads(arrays, dims, device)
                    return self._heads[key]
                except ValueError:
                    if backend == "hip":
                        raise
                try:        # too wide for LDS residency: tiles on the matrix cores with the weights streamed from L2
                    self._heads[key] = HipMlpTileHeads(arrays, dims, device)
                    return self._heads[key]
                except ValueError:
                    pass
            if self.model_structure == "mlp_model":
                arrays = mlp_arrays_from_modules(self.representation_function, self.prediction_function,
                                                 self.afterstate_prediction_function, self.afterstate_dynamics_function,
                                                 self.dynamics_function)
                dims = dict(obs=self.observation_dimension, A=self.action_dimension, S=self.state_dimension,
                            H=self.hidden_layer_dimension, L=self.number_of_hidden_layer)
                assert set(n + s for n in MLP_ARRAYS for s in ("_w", "_b")) <= set(arrays)
                self._heads[key] = FusedMlpHeads(arrays, dims, device)
            elif self.model_structure == "vision_model" and backend in ("auto", "hip"):
                mods = [getattr(self, f + "_function") for f in _FUNCS[:5]]
                self._heads[key] = HipVisionHeads(*mods, num_actions=self.action_dimension,
                                                  support_size=self.state_dimension, device=device)
            else:
                import copy
                mods = [copy.deepcopy(getattr(self, f + "_function")) for f in _FUNCS[:5]]
                self._heads[key] = ModuleHeads(*mods, num_actions=self.action_dimension,
                                               support_size=self.state_dimension, device=device, is_rgb=self.is_RGB)
        return self._heads[key]

    # ---- the reference's batch-1 inference API (muzero_model.py:802-909) ------------------------------------------
    def _t(self, x):
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.asarray(x, dtype=np.float32))
        return x.to(torch.float32)

    def inverse_transform_with_support(self, logits):
        S = self.state_dimension
        half = S // 2
        p = torch.softmax(logits, dim=1)
        sup = torch.arange(-half, -half + S, dtype=p.dtype)
        y = torch.sum(sup * p, dim=1, keepdim=True)
        return torch.sign(y) * (((torch.sqrt(1 + 4 * 0.001 * (torch.abs(y) + 1 + 0.001)) - 1) / (2 * 0.001)) ** 2 - 1)

    def one_hot_encode(self, action, counter_part):
        a = torch.as_tensor(action).to(torch.int64).reshape(-1)
        if not self.is_RGB:
            return torch.nn.functional.one_hot(a, num_classes=self.action_dimension).to(torch.float32)
        plane = torch.ones(1, 1, counter_part.shape[2], counter_part.shape[3])
        return torch.cat([((s + 1) / self.action_dimension) * plane for s in a], dim=0)

    @torch.no_grad()
    def representation_function_inference(self, state):
        return self.representation_function(self._t(state)).detach().cpu()

    def _pv(self, module, h):
        policy, value = module(self._t(h))
        policy = torch.softmax(policy, dim=-1).detach().cpu().numpy()
        value = self.inverse_transform_with_support(value).detach().flatten().float().cpu().numpy()[0]
        return policy, value

    @torch.no_grad()
    def prediction_function_inference(self, state_normalized):
        return self._pv(self.prediction_function, state_normalized)

    @torch.no_grad()
    def afterstate_prediction_function_inference(self, state_normalized):
        return self._pv(self.afterstate_prediction_function, state_normalized)

    @torch.no_grad()
    def afterstate_dynamics_function_inference(self, state_normalized, action):
        h = self._t(state_normalized)
        return self.afterstate_dynamics_function(h, self.one_hot_encode(action, h)).detach().cpu()

    @torch.no_grad()
    def dynamics_function_inference(self, state_normalized, action):
        h = self._t(state_normalized)
        reward, nxt = self.dynamics_function(h, self.one_hot_encode(action, h))
        reward = self.inverse_transform_with_support(reward.float()).detach().flatten().float().cpu().numpy()[0]
        return reward, nxt.detach().cpu()

    def train(self, *a, **k):
        raise NotImplementedError("training is outside this engine's scope; use the reference's Muzero.train with "
                                  "the games this engine produces")
