"""Round-5 GPU tests: page-locked packing buffers (``Engine.pinned_allocator``)."""

from __future__ import annotations

import numpy as np
import pytest

from conftest import load_case

pytestmark = pytest.mark.gpu


def test_batches_packed_into_pinned_buffers_give_the_same_results_and_survive_slot_reuse(hip_engine):
    """``pack_batch(graphs, alloc=engine.pinned_allocator(slot))``: the packed arrays live in one page-locked block per slot (uploads
    from it are DMA at the link rate).  Same arrays as the pageable packing, same E / F / S bit for bit; a block is reused by the
    next packing with the same slot while the offset tables of the earlier batch -- which outlive its upload -- stay intact."""
    from chgnet_amd.pack import pack_batch

    graphs_a = [load_case(n)[0] for n in ("limno2", "noangle", "s16tri")]
    graphs_b = [load_case(n)[0] for n in ("s40", "li9co7o16")]
    plain = pack_batch(graphs_a)
    pinned = pack_batch(graphs_a, alloc=hip_engine.pinned_allocator(0))
    for k, v in plain.arrays.items():
        assert np.array_equal(v, pinned.arrays[k]) and pinned.arrays[k].dtype == v.dtype, k
    off_before = pinned.atom_off.copy()

    def run(packed):
        batch = hip_engine.upload(packed)
        try:
            hip_engine.predict(batch, "efs")
            return hip_engine.download(batch, "efs")
        finally:
            batch.free()

    want, got = run(plain), run(pinned)
    for k in ("e", "f", "s"):
        assert np.array_equal(want[k], got[k]) or np.abs(want[k] - got[k]).max() < 1e-6, k     # (atomics: fp32 reassociation)
    other = pack_batch(graphs_b, alloc=hip_engine.pinned_allocator(0))       # same slot: the block is overwritten
    assert np.array_equal(pinned.atom_off, off_before)                         # ... the first batch's offsets are not
    second = pack_batch(graphs_a, alloc=hip_engine.pinned_allocator(1))       # another slot: independent block
    ref_b, got_b = run(pack_batch(graphs_b)), run(other)
    assert np.abs(ref_b["e"] - got_b["e"]).max() < 1e-6 and np.abs(ref_b["f"] - got_b["f"]).max() < 1e-6
    assert np.abs(run(second)["f"] - want["f"]).max() < 1e-6
