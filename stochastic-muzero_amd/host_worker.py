"""Entry of a host-env worker process (envs.HostVecEnv(workers=N)): python host_worker.py <spec file>.

The worker must stay light -- numpy and the env's own module, never torch or the GPU runtime -- so the package's __init__
(which loads libsmz.so) is kept out: a bare package object under the package's name makes `stochastic-muzero_amd.host_envs`
importable on its own, which is also the module path the pickled envs and adapters of the spec refer to."""
import importlib
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.getcwd()):
    if p not in sys.path:
        sys.path.insert(0, p)
pkg_name = os.path.basename(HERE)
if pkg_name not in sys.modules:
    pkg = types.ModuleType(pkg_name)
    pkg.__path__ = [HERE]
    sys.modules[pkg_name] = pkg
    sys.modules["stochastic_muzero_amd"] = pkg            # (the alias module of the repository root)

if __name__ == "__main__":
    sys.exit(importlib.import_module(pkg_name + ".host_envs").worker_main(sys.argv[1]))
