import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ---- parity ledger -------------------------------------------------------------------------------------------
# Some byte-level comparisons only apply when the product's integer side equals the reference run's on EVERY element
# (round() / the scale-table search are discontinuous; with 2.65 M latents a full-size frame practically always holds
# an element within an ulp of a boundary).  Such a comparison must never be a silent `if`: every conditional parity
# check reports here whether it RAN, and the session prints the ledger (also with -q) and writes
# gpurun_out/parity_ledger.json.
_LEDGER = []


class _Ledger:
    def ran(self, name, detail=""):
        _LEDGER.append(dict(check=name, ran=True, detail=detail))

    def not_applicable(self, name, detail):
        _LEDGER.append(dict(check=name, ran=False, detail=detail))


@pytest.fixture(scope="session")
def ledger():
    return _Ledger()


def pytest_terminal_summary(terminalreporter):
    if not _LEDGER:
        return
    tr = terminalreporter
    tr.section("parity ledger: conditional byte / integer comparisons")
    tr.write_line('("reference-python-written stream" = written by the reference\'s own compress() Python with the oracle coder '
                  'plugged in as `compressai.ans`: the reference\'s compiled coder cannot be built here - DESIGN.md section 2)')
    for e in _LEDGER:
        tr.write_line(("RAN            " if e["ran"] else "NOT APPLICABLE ") + e["check"] + (": " + e["detail"] if e["detail"] else ""))
    n_ran = sum(e["ran"] for e in _LEDGER)
    tr.write_line(f"{n_ran} of {len(_LEDGER)} conditional comparisons ran")
    try:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(_LEDGER, open(os.path.join(out, "parity_ledger.json"), "w"), indent=1)
    except OSError:
        pass
