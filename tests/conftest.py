import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """the CPU oracle (test infrastructure), built on demand with gcc"""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def summary_golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "summary_golden.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def summary_hp_golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "summary_hp_golden.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def hip_ctx():
    """one device context for the whole GPU session; fails loudly if the extension is missing"""
    from pepper_thesis_amd import runtime
    ctx = runtime.Context(0)
    yield ctx
    ctx.close()


OPTION_DEFAULTS = {"lstm_split": 1, "lstm_rows": 0, "tail_rows": 0, "head_splits": 0, "head_map": 1, "gru_rows": 0, "gru_split": 1,
                   "gru_usplit": 1, "shared_device": 0, "exchange_spin_log2": 18, "debug_drop_part": -1, "p1_bf16_min_batch": 513}


@pytest.fixture
def opts(hip_ctx):
    """set kernel-form options on the session context for one test (pv_set_option); defaults are restored afterwards"""
    def set_(**kw):
        for k, v in kw.items():
            hip_ctx.set_option(k, v)
    yield set_
    for k, v in OPTION_DEFAULTS.items():
        hip_ctx.set_option(k, v)
