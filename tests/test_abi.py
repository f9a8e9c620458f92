"""The C-ABI library loads and exports every symbol include/cfdh.h declares (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

from cfd_hemodynamic_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "cfdh.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cfdh_[a-z0-9_]+)\s*\(", src)) - {"cfdh_allreduce_fn", "cfdh_exchange_fn"})


def test_every_declared_symbol_is_exported_and_bound():
    L = _lib.lib()
    names = _header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libcfdh.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names, "python binding list out of sync with include/cfdh.h"
    assert L.cfdh_abi_version() == 1


def test_struct_layouts_match_header():
    o = _lib.Options()
    _lib.lib().cfdh_default_options(ctypes.byref(o))
    # PETSc defaults + caps of stabilized_schur.py:269-274
    assert (o.snes_rtol, o.snes_atol, o.snes_stol, o.snes_max_it) == (1e-8, 1e-50, 1e-8, 100)
    assert (o.ksp_rtol, o.ksp_max_it, o.ksp_restart) == (1e-5, 1000, 200)
    assert (o.cheb_degree, o.amg_smooth_degree, o.schur_full, o.remove_p_mean, o.verbose) == (3, 1, 2, 1, 0)
    assert (o.pc_type, o.cc_smooth_degree) == (1, 2)
    assert ctypes.sizeof(_lib.Stats) == 4 * 4 + 6 * 8


def test_bad_arguments_return_error_codes_not_crashes():
    L = _lib.lib()
    h = ctypes.c_void_p()
    x = np.zeros((3, 2))
    cells = np.array([[0, 1, 2]], dtype=np.int32)
    rc = L.cfdh_create(ctypes.byref(h), 0, 4, 3, 3, 1, _lib._ip(cells), _lib._dp(x), 0, None, None, None)
    assert rc == -1 and b"gdim" in L.cfdh_last_error(None)
    assert L.cfdh_solve_step(None, None) == -1
    assert L.cfdh_info(None, 0) == -1


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product path must fail loudly, never fall back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from util import dfg_case, make_ctx
    with pytest.raises(RuntimeError, match="no HIP device|HIP"):
        make_ctx(dfg_case(4))
