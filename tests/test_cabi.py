"""The C-ABI library loads without a GPU and exports every symbol
include/asr_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared(header='asr_amd.h'):
    text = open(os.path.join(ROOT, 'include', header)).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(asr_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from att_speech import _native
    names = _declared()
    assert len(names) >= 9
    handle = ctypes.CDLL(_native.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), n
    assert set(_native._SIGNATURES) == set(names)
    assert _native.lib().asr_abi_version() == _native.ABI_VERSION
    assert b'invalid' in _native.lib().asr_strerror(_native.ASR_EINVAL)


def test_experiments_stay_out_of_the_default_library():
    """include/asr_amd_experiments.h: variants that lost against the default path are exported
    only by a `make EXPERIMENTS=1` build, all of them or none."""
    from att_speech import _native
    names = _declared('asr_amd_experiments.h')
    assert set(names) == set(_native._EXPERIMENT_SIGNATURES)
    assert not set(names) & set(_declared())
    handle = ctypes.CDLL(_native.LIB_PATH)
    have = [hasattr(handle, n) for n in names]
    assert all(have) == _native.experiments_built() and (all(have) or not any(have))
    if os.environ.get('ASR_EXPECT_EXPERIMENTS', '0') != '1':
        assert not any(have), 'the default build must not export experiment entry points'


def test_argument_checks_need_no_gpu():
    from att_speech import _native
    L = _native.lib()
    # null pointers / bad shapes are rejected before anything is launched
    assert L.asr_lattice_fwbw_f32(None, 4, 2, 5, None, None, None, None, None,
                                  None, None, None, 3, 2, 2, 2, -1e20, None,
                                  None, None, None, 0, None) == _native.ASR_EINVAL
    assert L.asr_lattice_fwbw_f32(None, 4, 2, 5, None, None, None, None, None,
                                  None, None, None, 3, 2, 2, 3, -1e20, None,
                                  None, None, None, 0, None) == _native.ASR_EINVAL
    assert L.asr_log_softmax_fwd_f32(None, 4, 0, None, None) == _native.ASR_EINVAL
    assert L.asr_log_softmax_fwd_f32(None, 0, 7, None, None) == _native.ASR_OK
    assert L.asr_lattice_fwbw_workspace_bytes(10, 2, 5, 7) >= 10 * 2 * 7 * 4


def test_cpu_tensors_are_refused_loudly():
    import pytest
    import torch
    from att_speech import _native
    with pytest.raises(_native.NativeLibraryError):
        _native.log_softmax_fwd(torch.zeros(3, 4), 4)
