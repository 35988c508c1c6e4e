"""GPU: the allocation-free entry point (apr_prepare_batch_device_static) against the entry point it is a variant of
(apr_prepare_batch_device), through the C ABI: bit-identical output on two calls in a row, the second smaller, with the
shared static scratch left dirty by the first."""
import numpy as np
import pytest

from apr_amd import _capi, model
from apr_amd._capi import AprError, FLAVOUR_GO, FLAVOUR_LIBROSA

pytestmark = pytest.mark.gpu


def test_static_scratch_entry_equals_allocating_entry():
    import torch
    dev = torch.device("cuda", 0)
    fe = model.Engine(None, 0, 64)
    pcm = model.synth_pcm(21, 0, 66).view(np.uint8)                       # 65 windows at the 1-s hop
    d_pcm = torch.from_numpy(pcm.copy()).to(dev)

    def both(n, stride, flavour, total=None):
        total = d_pcm.numel() if total is None else total
        want = torch.full((n, 128, 188), -1.0, dtype=torch.float32, device=dev)
        got = torch.full((n, 128, 188), -2.0, dtype=torch.float32, device=dev)
        fe.prepare_batch_device(d_pcm, total, n, stride, want, flavour)
        fe.prepare_batch_device_static(d_pcm, total, n, stride, got, flavour)
        fe.sync()
        return want.cpu().numpy(), got.cpu().numpy()

    # first call: the maximum the static array holds; second call right after it: fewer windows, other flavour
    # (188 frames instead of 184), over scratch the first call left full
    w1, g1 = both(_capi.STATIC_MAX_WINDOWS, 96000, FLAVOUR_GO)
    w2, g2 = both(5, 192000, FLAVOUR_LIBROSA)
    assert np.array_equal(w1, g1) and np.array_equal(w2, g2)
    assert w1.min() >= 0.0 and w1.max() == 1.0 and not np.array_equal(w1[0], w1[1])     # real spectrograms, not fill values
    # a ragged tail (last window short -> zero-padded in the kernel) and a single clip (other launch shape)
    w3, g3 = both(3, 96000, FLAVOUR_GO, total=2 * 96000 + 100001)
    w4, g4 = both(1, 192000, FLAVOUR_GO)
    assert np.array_equal(w3, g3) and np.array_equal(w4, g4)
    # sizes beyond the static array are refused, nothing is launched, and the engine keeps working
    out = torch.empty((65, 128, 188), dtype=torch.float32, device=dev)
    with pytest.raises(AprError) as ei:
        fe.prepare_batch_device_static(d_pcm, d_pcm.numel(), 65, 96000, out)
    assert ei.value.code == _capi.ERR_INVALID and "APR_STATIC_MAX_WINDOWS" in str(ei.value)
    fe.prepare_batch_device_static(d_pcm, d_pcm.numel(), 0, 96000, out)                 # zero windows: a no-op
    w5, g5 = both(2, 96000, FLAVOUR_GO)
    assert np.array_equal(w5, g5) and np.array_equal(w5, w1[:2])
    fe.close()
