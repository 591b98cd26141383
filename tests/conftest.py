import os
import sys
import pytest

# torch bundles its own libamdhip64 with the same SONAME as /opt/rocm's: whichever is loaded first
# serves the whole process.  Tests that hand torch CUDA tensors to libsnkhip.so (the multi-GPU
# exchange buffers) need torch's runtime to come first, exactly as in bench.py.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver on the GPU box)')


def pytest_runtest_logstart(nodeid, location):
    """Name the test about to run on stderr (unbuffered) and in gpurun_out/current_test.txt: if the process
    dies inside the runtime, the last line before the runtime's message says where."""
    try:
        os.write(2, ('\n[snk-test] %s\n' % nodeid).encode())      # the descriptor itself: never captured, never buffered
    except OSError:
        pass
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(d):
            with open(os.path.join(d, 'current_test.txt'), 'w') as f:
                f.write(nodeid + '\n')
    except OSError:
        pass


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'reference_mini.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def mini_voice(golden):
    """Weighted DB of the golden mini voice, built by the oracle from the raw DB arrays."""
    import numpy as np
    import snk_oracle as o
    dims = {'mag': 60, 'real': 45, 'imag': 45, 'lf0': 1}
    tw, jw = o.apply_jcw(golden['target_stream_weights'], golden['join_stream_weights'],
                         float(golden['join_cost_weight']))
    wt = o.stream_weight_vector(list(tw), ['mag', 'lf0'], dims)
    wj = o.stream_weight_vector(list(jw), ['mag', 'real', 'imag', 'lf0'], dims)
    F, E, S = o.weighted_db(golden['F_unw'], golden['JC_unw'], wt, wj)
    return dict(F=F, E=E, S=S, wt=wt, wj=wj, F_unw=golden['F_unw'], JC_unw=golden['JC_unw'])
