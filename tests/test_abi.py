"""CPU: the C-ABI shared library loads and exports every symbol include/sfmba.h declares;
without a GPU the product path fails loudly instead of falling back."""
import ctypes
import os
import re

import pytest

from structure_from_motion_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sfmba.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sfmba_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(engine.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "structure_from_motion_amd", "libsfmba.so"))
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_default_options_match_reference_call():
    o = engine.default_options()
    # bundle_adjustment.py:141: ftol=1e-4, xtol=1e-12; SciPy default gtol=1e-8
    assert (o.ftol, o.xtol, o.gtol) == (1e-4, 1e-12, 1e-8)
    assert o.nranks == 1 and o.rank == 0 and o.device == -1
    assert o.pcg_mode == 0 and o.pcg_blocks == 0 and o.linear_solver == 0 and o.schur_mode == 0 and o.camera_model == 0 and o.pcg_coarse == 0


def test_struct_layouts_match_header():
    text = open(os.path.join(ROOT, "include", "sfmba.h")).read()
    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [m for m in re.findall(r"\b(?:double|int32_t|int64_t)\s+([a-z_0-9]+)\s*;", body)]
    assert fields("sfmba_options") == [n for n, _ in engine.Options._fields_]
    assert fields("sfmba_stats") == [n for n, _ in engine.Stats._fields_]


def _gpu_present():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    with pytest.raises(engine.EngineError):
        engine.Engine()
    from structure_from_motion_amd import bundle_adjustment as ba
    import numpy as np
    with pytest.raises(engine.EngineError):
        ba.project(np.zeros((2, 3)), np.tile(np.r_[np.eye(3).ravel(), 0, 0, 1.0], (2, 1)), np.eye(3))
