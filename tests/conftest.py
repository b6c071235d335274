import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_losses():
    return load_golden("losses.npz")


def loss_case(z, name):
    """Inputs + expected outputs of one golden loss case as a dict of numpy arrays."""
    pre = name + "/"
    d = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    if "gen_args" in d:      # big case: inputs regenerated from the seed (see make_golden.case_inputs)
        sys.path.insert(0, GOLDEN)
        from make_golden import case_inputs
        nb, D, NC, ppc, seed, unk, rag = [int(v) for v in d["gen_args"]]
        ci = case_inputs(nb, D, NC, ppc, seed, unk, bool(rag))
        chk = np.array([ci["fv"].double().sum().item(), ci["centers"].double().abs().sum().item(),
                        ci["proxies"].double().abs().sum().item(), float(ci["labels"].sum()),
                        float(ci["distortion"].sum())])
        if not np.allclose(chk, d["input_checksums"], rtol=1e-9, atol=1e-9):
            pytest.skip("torch RNG stream differs from the one the golden was generated with")
        d.update(fv=ci["fv"].numpy(), labels=ci["labels"], distortion=ci["distortion"].numpy(),
                 centers=ci["centers"].numpy(), centers_labels=ci["centers_labels"],
                 proxies=ci["proxies"].numpy(), proxies_labels=ci["proxies_labels"])
    return d
