"""Placement logic of GraphedForward(zero_copy_memory=True) on the CPU (no graph is captured here: `_place_memory` decides where a
call reads its memory records and which ring buffer it writes; tests/test_gpu_graph_memory.py runs the real thing on the GPU).
Protocols: the Joint carry (one record, general_eval.py:52), the ESTM window memory (two records, eval_hybrid_seq.py:160-193), a
fixed record from elsewhere (bench.py), a new record at every call."""
import types

import torch

from estdepth_amd.graph import GraphedForward
from estdepth_amd.hybrid_depth_decoder import kv_views


def _gf():
    model = types.SimpleNamespace(ndepths=4, camera_algebra="host", _estd_weights_epoch=0)
    return GraphedForward(model, zero_copy_memory=True)


IMGS = torch.zeros(1, 3, 3, 16, 16)          # V = 3 -> one target, kv shape (1, 4, 4, 4, 32)


def _call(gf, records):
    """what __call__ does around a replay, without the replay: place, 'capture' (register the key), mark the slot written and hand
    the record of the last target out"""
    pc = {"keys": [r[0] for r in records], "values": [r[1] for r in records]} if records else None
    shape, ptrs, slot = gf._place_memory(IMGS, pc, "val", None)
    key = gf._signature(IMGS, pc, "val", None, (ptrs, slot))
    gf._graphs.setdefault(key, {})
    ring = gf._ring[shape]
    gf._mark_written(ring, slot)
    return kv_views(ring["bufs"][slot][-1]), ptrs, slot


def test_one_carried_record_alternates_between_two_buffers():
    gf = _gf()
    rec, slots = None, []
    for _ in range(8):
        rec, ptrs, slot = _call(gf, [rec] if rec else [])
        slots.append(slot)
        assert ptrs is None or len(ptrs) == 1
    assert slots == [0, 1, 0, 1, 0, 1, 0, 1]
    assert len(next(iter(gf._ring.values()))["bufs"]) == 2
    assert len(gf._graphs) == 3                      # no memory -> 0; (0) -> 1; (1) -> 0


def test_two_carried_records_rotate_through_three_buffers():
    gf = _gf()
    mem, slots = [], []
    for _ in range(10):
        rec, ptrs, slot = _call(gf, mem)
        # the buffer written is never one that is read, nor the one handed out last time
        assert all(r[1]._estd_kv.data_ptr() != rec[1]._estd_kv.data_ptr() for r in mem)
        mem = (mem + [rec])[-2:]
        slots.append(slot)
    assert slots == [0, 1, 2, 0, 1, 2, 0, 1, 2, 0]
    assert len(next(iter(gf._ring.values()))["bufs"]) == 3
    assert len(gf._graphs) == 5                      # (), (0), (0,1)->2, (1,2)->0, (2,0)->1


def test_fixed_record_from_elsewhere_is_read_in_place():
    gf = _gf()
    foreign = kv_views(torch.zeros(4, 4, 4, 32))
    prev = None
    for it in range(6):
        rec, ptrs, slot = _call(gf, [foreign])
        assert ptrs == (foreign[1]._estd_kv.data_ptr(),)
        if prev is not None:
            assert prev[1]._estd_kv.data_ptr() != rec[1]._estd_kv.data_ptr()        # the previous record is not overwritten (it may be in flight)
        prev = rec
    assert len(gf._graphs) == 2 and len(next(iter(gf._ring.values()))["bufs"]) == 2


def test_new_record_at_every_call_ends_on_the_copy_path():
    gf = _gf()
    hold, keyed = [], []
    for it in range(7):
        foreign = kv_views(torch.zeros(4, 4, 4, 32))
        hold.append(foreign)
        _, ptrs, _ = _call(gf, [foreign])
        keyed.append(ptrs is not None)
    assert keyed == [True] * GraphedForward.MAX_FOREIGN + [False] * (7 - GraphedForward.MAX_FOREIGN)


def test_records_that_are_not_kv_views_take_the_copy_path():
    gf = _gf()
    k, v = torch.zeros(1, 16, 4, 4, 4), torch.zeros(1, 16, 4, 4, 4)
    _, ptrs, _ = _call(gf, [(k, v)])
    assert ptrs is None
