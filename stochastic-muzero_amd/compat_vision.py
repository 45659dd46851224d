"""Head modules of the `vision_model` family (ResNet-v2 towers over 98x98x3 frames) under the reference's class names.

Same role as compat_mlp.py: the reference pickles whole nn.Modules (muzero_model.py:920-925), so its vision
checkpoints name `neural_network_vision_model.<Class>`; model.py registers this module under that name, pickle
restores the attribute dictionaries, and the forward() methods below only rely on the attribute names the reference
uses (`sequential_container`, `sequential_downsampler`, `sequential_convolution_activation`, `sequential_reward`,
`resnet`, `nn_value`, `nn_policy`, `encoder`) and on the order of the entries inside each nn.Sequential.

Architecture facts restated from neural_network_vision_model.py:41-515 (module SHARING is part of the definition --
a layer object that appears several times in a Sequential is one set of weights applied several times):

* residual block (v2, :41-79): y = x + F(x), F = bn,relu,convA, bn,relu,convB, bn,relu,convA with ONE BatchNorm2d
  and convA used twice; all convolutions 3x3, padding 1, no bias, C -> C.
* down-sampler (:81-121): conv3x3/s2 (Cin -> C/2), R1, R1, conv3x3/s2 (C/2 -> C), R2, R2, avgpool3/s2/p1,
  R2, R2, R2, avgpool3/s2/p1 with R1 / R2 single shared blocks: 98 -> 49 -> 25 -> 13 -> 7.
* representation (:124-163): down-sampler + one more block, then the per-pixel channel min-max scaling.
* (afterstate) dynamics (:166-234, :398-452): x = cat(state, action plane) (C+1 channels); next state =
  conv3x3 (C+1 -> C, no bias), bn, relu, [R]*L, relu, scaled; reward = conv1x1 (C+1 -> C, bias), flatten,
  Linear(C*7*7 -> H) relu [Linear(H -> H) relu]*L Linear(H -> S) (dynamics only).
* (afterstate) prediction (:236-311, :455-515): t = [R]*L (state); value = conv1x1, flatten, MLP -> S;
  policy = conv1x1, flatten, MLP -> A.
* encoder (:313-395): down-sampler, R, [R]*L (the same R), conv1x1, flatten, MLP -> A; softmax and a
  straight-through one-hot of its argmax.

Inference runs these in eval mode (muzero_model.py:803-804 etc.): batch-norm uses its running statistics, so rows of a
batch are independent -- the property the batched engine needs.
"""
import torch
import torch.nn as nn

CHANNELS = 3          # num_channels default of every class in the reference family
FRAME = (98, 98, 3)   # muzero_model.py:400-404 fixes the frame size for every "vision" structure


def scale_to_bound_action(x):
    """(x - min) / (max - min) along dim 1, i.e. across CHANNELS for each pixel; ranges below 1e-5 get +1e-5
    (neural_network_vision_model.py:494-503)."""
    lo = x.amin(dim=1, keepdim=True)
    span = x.amax(dim=1, keepdim=True) - lo
    span = torch.where(span < 1e-5, span + 1e-5, span)
    return (x - lo) / span


def _conv3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def _tower(n_in, width, depth, n_out):
    """Linear(n_in,W) relu, the SAME Linear(W,W) relu pair `depth` times, Linear(W,n_out)."""
    relu = nn.ReLU()
    first, mid, last = nn.Linear(n_in, width), nn.Linear(width, width), nn.Linear(width, n_out)    # creation order =
    return nn.Sequential(first, relu, *([mid, relu] * depth), last)        # the reference's, so equal seeds give equal nets


def _flat_features(observation_space_dimensions, down_sampling=True, reduced=1):
    h, w, c = observation_space_dimensions[:3]
    return reduced * (int(h / 14) * int(w / 14) * c if down_sampling else h * w * c)


class Residual_block(nn.Module):
    def __init__(self, num_channels, stride=1):
        super().__init__()
        conv_b, conv_a = _conv3(num_channels, num_channels, stride), _conv3(num_channels, num_channels, stride)
        norm, relu = nn.BatchNorm2d(num_channels), nn.ReLU()
        self.sequential_container = nn.Sequential(norm, relu, conv_a, norm, relu, conv_b, norm, relu, conv_a)
        self.last_layer = relu

    def forward(self, state):
        return self.sequential_container(state) + state


class Down_sample(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        half = out_channels // 2
        stem, widen = _conv3(in_channels, half, 2), _conv3(half, out_channels, 2)
        narrow, wide = Residual_block(half), Residual_block(out_channels)
        pool = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.sequential_container = nn.Sequential(stem, narrow, narrow, widen, wide, wide, pool, wide, wide, wide, pool)

    def forward(self, x):
        return self.sequential_container(x)


class Representation_function(nn.Module):
    def __init__(self, observation_space_dimensions, state_dimension, action_dimension, hidden_layer_dimensions,
                 number_of_hidden_layer, num_channels=CHANNELS, stacked_observations=1, down_sampling=True):
        super().__init__()
        self.action_space = action_dimension
        self.down_sampling = down_sampling
        planes = observation_space_dimensions[-1]
        down, conv, norm = Down_sample(planes, num_channels), _conv3(planes, num_channels), nn.BatchNorm2d(num_channels)
        block = Residual_block(num_channels)
        self.sequential_downsampler = nn.Sequential(down, block)
        self.sequential_convolution_activation = nn.Sequential(conv, norm, nn.ReLU(), block)

    def forward(self, state):
        net = self.sequential_downsampler if self.down_sampling else self.sequential_convolution_activation
        return scale_to_bound_action(net(state))


class _Transition(nn.Module):
    """Shared constructor of the two dynamics networks."""

    def _build(self, state_dimension, action_dimension, observation_space_dimensions, hidden_layer_dimensions,
               number_of_hidden_layer, num_channels, down_sampling, with_reward):
        self.action_space = action_dimension
        relu = nn.ReLU()
        conv, norm, block = _conv3(num_channels + 1, num_channels), nn.BatchNorm2d(num_channels), Residual_block(num_channels)
        # the afterstate network builds (and drops) the reward branch too: same parameter draws as the reference
        reward = nn.Sequential(nn.Conv2d(num_channels + 1, num_channels, 1), nn.Flatten(1, -1),
                               _tower(_flat_features(observation_space_dimensions, down_sampling),
                                      hidden_layer_dimensions, number_of_hidden_layer, state_dimension))
        self.sequential_container = nn.Sequential(conv, norm, relu, *([block] * number_of_hidden_layer), relu)
        if with_reward:
            self.sequential_reward = reward


class Dynamics_function(_Transition):
    def __init__(self, state_dimension, action_dimension, observation_space_dimensions, hidden_layer_dimensions,
                 number_of_hidden_layer, num_channels=CHANNELS, reduced_channels_reward=1, down_sampling=True):
        super().__init__()
        self._build(state_dimension, action_dimension, observation_space_dimensions, hidden_layer_dimensions,
                    number_of_hidden_layer, num_channels, down_sampling, with_reward=True)

    def forward(self, state_normalized, action):
        x = torch.cat([state_normalized, action], dim=1)
        return self.sequential_reward(x), scale_to_bound_action(self.sequential_container(x))


class Afterstate_dynamics_function(_Transition):
    def __init__(self, state_dimension, action_dimension, observation_space_dimensions, hidden_layer_dimensions,
                 number_of_hidden_layer, num_channels=CHANNELS, reduced_channels_reward=1, down_sampling=True):
        super().__init__()
        self._build(state_dimension, action_dimension, observation_space_dimensions, hidden_layer_dimensions,
                    number_of_hidden_layer, num_channels, down_sampling, with_reward=False)

    def forward(self, state_normalized, action):
        return scale_to_bound_action(self.sequential_container(torch.cat([state_normalized, action], dim=1)))


class Prediction_function(nn.Module):
    def __init__(self, state_dimension, action_dimension, observation_space_dimensions, hidden_layer_dimensions,
                 number_of_hidden_layer, down_sampling=True, reduced_channels_value=1, reduced_channels_policy=1,
                 num_channels=CHANNELS):
        super().__init__()
        feats = _flat_features(observation_space_dimensions, down_sampling)
        block = Residual_block(num_channels)
        mix_v, mix_p = nn.Conv2d(num_channels, num_channels, 1), nn.Conv2d(num_channels, num_channels, 1)
        tower_v = _tower(feats, hidden_layer_dimensions, number_of_hidden_layer, state_dimension)
        tower_p = _tower(feats, hidden_layer_dimensions, number_of_hidden_layer, action_dimension)
        self.resnet = nn.Sequential(*([block] * number_of_hidden_layer))
        self.nn_value = nn.Sequential(mix_v, nn.Flatten(1, -1), tower_v)
        self.nn_policy = nn.Sequential(mix_p, nn.Flatten(1, -1), tower_p)

    def forward(self, state_normalize):
        t = self.resnet(state_normalize)
        value = self.nn_value(t)
        return self.nn_policy(t), value


class Afterstate_prediction_function(Prediction_function):
    pass


class Onehot_argmax(torch.autograd.Function):
    """One-hot of the arg-max with a straight-through gradient (neural_network_vision_model.py:506-514)."""

    @staticmethod
    def forward(ctx, probs):
        return torch.zeros_like(probs).scatter_(-1, probs.argmax(dim=-1, keepdim=True), 1.0)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class Encoder_function(nn.Module):
    def __init__(self, state_dimension, action_dimension, observation_space_dimensions, hidden_layer_dimensions,
                 number_of_hidden_layer, down_sampling=True, reduced_channels_value=1, reduced_channels_policy=1,
                 num_channels=CHANNELS):
        super().__init__()
        self.action_space = action_dimension
        self.down_sampling = down_sampling
        down = Down_sample(observation_space_dimensions[-1], num_channels)
        Residual_block(num_channels)          # the reference constructs one block it never uses (:334, :343)
        block = Residual_block(num_channels)
        self.encoder = nn.Sequential(
            down, block, *([block] * number_of_hidden_layer), nn.Conv2d(num_channels, num_channels, 1),
            nn.Flatten(1, -1), _tower(_flat_features(observation_space_dimensions, down_sampling),
                                      hidden_layer_dimensions, number_of_hidden_layer, action_dimension))

    def forward(self, o_i):
        c_e_t = torch.softmax(self.encoder(o_i), dim=-1)
        return Onehot_argmax.apply(c_e_t), c_e_t
