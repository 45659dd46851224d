"""stochastic-muzero_amd: MI355X-native batched Stochastic-MuZero self-play / MCTS engine.

The directory name carries a hyphen (it mirrors the reference's repository name), so import it through the
alias module at the repository root:  `import stochastic_muzero_amd as smz`.
"""
from . import _lib  # noqa: F401
from .engine import SearchEngine, pb_c_table, pow_table  # noqa: F401
