# -*- coding: utf-8 -*-
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (gfx950) device; run with -m gpu on the GPU box")


def _device_count():
    from celerite_amd import batch  # ImportError here is a build problem: stay loud

    return batch.device_count()


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items:
        return
    if _device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no gfx950 device visible (GPU tests run with -m gpu on the GPU box)")
    for it in gpu_items:
        it.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    try:
        from _cases import MEASURED
    except Exception:
        return
    if not MEASURED:
        return
    terminalreporter.section("measured deviations (worst / asserted tolerance / checks)")
    for name in sorted(MEASURED):
        worst, tol, n = MEASURED[name]
        terminalreporter.write_line("%-72s %.2e / %.0e / %d" % (name, worst, tol, n))
