"""Head modules of the `mlp_model` family under the reference's class names.

The reference stores checkpoints as whole pickled nn.Modules (muzero_model.py:920-925), so a `.pt` file names the
classes `neural_network_mlp_model.{Representation,Prediction,Afterstate_prediction,Afterstate_dynamics,Dynamics,
Encoder}_function`.  To load those files where the reference's source is not installed, checkpoint.py registers
this module under that name; pickle then restores each module's attribute dictionary onto the classes below
(their constructors are not run on load), so only the attribute names used in forward() have to agree:
`state_norm`, `policy`, `value`, `reward`, `next_state_normalized`, `encoder`.

Architecture facts restated from neural_network_mlp_model.py:5-250: every trunk is Linear(in,H) -> ELU followed by
the SAME Linear(H,H) -> ELU pair repeated `number_of_hidden_layer` times; heads that share a trunk (policy/value,
reward/next state) share the trunk's module objects; hidden states leave through a per-row min-max scaling.
"""
import torch
import torch.nn as nn


def scale_to_bound_action(x):
    """Row-wise (x - min) / (max - min); ranges below 1e-5 get +1e-5 (neural_network_mlp_model.py:349-357)."""
    lo = x.amin(dim=1, keepdim=True)
    span = x.amax(dim=1, keepdim=True) - lo
    span = torch.where(span < 1e-5, span + 1e-5, span)
    return (x - lo) / span


def _trunk(n_in, width, depth):
    first, mid, act = nn.Linear(n_in, width), nn.Linear(width, width), nn.ELU()
    return [first, act] + [mid, act] * depth


def _unused_draws(width, n_out):
    """The reference's Representation_function and Encoder_function construct one Linear(H, S) that never enters a
    Sequential (neural_network_mlp_model.py:17, 222) before the output layer they do use; its default initialisation
    consumes torch's global generator, so an equal torch seed only reproduces the reference's fresh weights if the same
    draws are made here (tests/test_abi_and_host.py pins this against weights the reference itself initialised)."""
    nn.Linear(width, n_out)


class _TwoHeadTrunk(nn.Module):
    def _build(self, n_in, width, depth, names_and_outs):
        layers = _trunk(n_in, width, depth)
        for name, n_out in names_and_outs:
            setattr(self, name, nn.Sequential(*layers, nn.Linear(width, n_out)))


class Representation_function(nn.Module):
    def __init__(self, observation_space_dimensions, state_dimension, action_dimension, hidden_layer_dimensions,
                 number_of_hidden_layer):
        super().__init__()
        self.action_space = action_dimension
        self.scale = nn.Tanh()
        layers = _trunk(observation_space_dimensions, hidden_layer_dimensions, number_of_hidden_layer)
        _unused_draws(hidden_layer_dimensions, state_dimension)
        self.state_norm = nn.Sequential(*layers, nn.Linear(hidden_layer_dimensions, state_dimension))

    def forward(self, state):
        return scale_to_bound_action(self.state_norm(state))


class Prediction_function(_TwoHeadTrunk):
    def __init__(self, state_dimension, action_dimension, observation_space_dimensions, hidden_layer_dimensions,
                 number_of_hidden_layer):
        super().__init__()
        self._build(state_dimension, hidden_layer_dimensions, number_of_hidden_layer,
                    [("policy", action_dimension), ("value", state_dimension)])

    def forward(self, state_normalized):
        return self.policy(state_normalized), self.value(state_normalized)


class Afterstate_prediction_function(Prediction_function):
    pass


class Afterstate_dynamics_function(_TwoHeadTrunk):
    def __init__(self, state_dimension, action_dimension, observation_space_dimensions, hidden_layer_dimensions,
                 number_of_hidden_layer):
        super().__init__()
        self.action_space = action_dimension
        self._build(state_dimension + action_dimension, hidden_layer_dimensions, number_of_hidden_layer,
                    [("reward", state_dimension), ("next_state_normalized", state_dimension)])

    def forward(self, state_normalized, action):
        x = torch.cat([state_normalized, action], dim=1)
        return scale_to_bound_action(self.next_state_normalized(x))


class Dynamics_function(Afterstate_dynamics_function):
    def forward(self, state_normalized, action):
        x = torch.cat([state_normalized, action], dim=1)
        return self.reward(x), scale_to_bound_action(self.next_state_normalized(x))


class StraightThroughEstimator(nn.Module):
    def forward(self, x):
        return Onehot_argmax.apply(x)


class Onehot_argmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return torch.zeros_like(x).scatter_(-1, x.argmax(dim=-1, keepdim=True), 1.0)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class Encoder_function(nn.Module):
    """Chance-outcome encoder (training only; never called by the search)."""

    def __init__(self, observation_space_dimensions, state_dimension, action_dimension, hidden_layer_dimensions,
                 number_of_hidden_layer):
        super().__init__()
        self.action_space = action_dimension
        self.scale = nn.Tanh()
        layers = _trunk(observation_space_dimensions, hidden_layer_dimensions, number_of_hidden_layer)
        _unused_draws(hidden_layer_dimensions, state_dimension)
        self.encoder = nn.Sequential(*layers, nn.Linear(hidden_layer_dimensions, action_dimension))
        self.onehot_argmax = StraightThroughEstimator()

    def forward(self, o_i):
        c_e_t = torch.softmax(self.encoder(o_i), dim=-1)
        return torch.zeros_like(c_e_t).scatter_(-1, c_e_t.argmax(dim=-1, keepdim=True), 1.0), c_e_t


def weights_init(m):
    """N(0, 1/137.035999) for Linear/Conv2d weights and biases (neural_network_mlp_model.py:495-508)."""
    if isinstance(m, (nn.Linear, nn.Conv2d)):
        nn.init.normal_(m.weight, mean=0.0, std=1 / 137.035999)
        nn.init.normal_(m.bias, mean=0.0, std=1 / 137.035999)
