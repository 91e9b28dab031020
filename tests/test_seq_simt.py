"""Windows that stay on the device from frame to frame (vg_ba_seq_*, csrc/ba_seq.hip), kernel sources under the CPU fiber emulator:
the device-resident sequence must take the same decisions and hold the same window as the reference's bookkeeping restated on the
host (tests/seq_model.py) feeding the ordinary C-ABI frame by frame."""
import numpy as np
import pytest

import conftest
import seq_model as M


@pytest.fixture(scope="module")
def two_handles():
    a, b = conftest._simt_handle(), conftest._simt_handle()
    yield a, b
    a.close(); b.close()


@pytest.mark.parametrize("min_parallax,want_new,seeds,n_steps", [(10.0 / 460.0, False, [21], 3), (0.25, True, [21, 22], 4)])
def test_resident_sequence_equals_host_bookkeeping(two_handles, min_parallax, want_new, seeds, n_steps):
    h_seq, h_ref = two_handles
    flags = M.run_both(h_seq, h_ref, seeds=seeds, K=11, L=70, n_steps=n_steps, min_parallax=min_parallax, max_features=128, check=M.check_step)
    flat = [f for fr in flags for f in fr]
    assert (M.NEW in flat) == want_new and (M.OLD in flat or want_new)


def test_resident_sequence_against_the_reference_loop(two_handles):
    """The reference's own processIMU / processImage loop (oracle/_ref) against the device-resident sequence, emulated kernels,
    five frames with non-keyframes among them (the full-length runs are tests/test_seq_gpu.py)."""
    from oracle import ref as R
    if not R.available():
        pytest.skip("oracle/_ref is not built")
    flags, worst, flips = M.run_against_reference(two_handles[0], 0.1, n_frames=15)
    assert 1 in flags
