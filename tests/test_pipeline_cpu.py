"""CPU: the bookkeeping of DescribePipeline (linetr_amd/engine.py) on a stub engine -- which slot a batch gets, which batch a submit hands
back, that every batch is joined exactly once and in submission order, that an empty batch is never joined (it queued nothing), and the
depth limits.  The device side of the pipeline is tests/test_gpu_pipeline.py."""
import types

import pytest

from linetr_amd.engine import DescribePipeline


class StubEngine:
    def __init__(self, max_slots=4):
        self._L = types.SimpleNamespace(linetr_pipeline_max_slots=lambda: max_slots)
        self.log = []

    def describe_lines(self, name, n, pipeline_slot=None, **kw):
        self.log.append(("submit", name, pipeline_slot))
        tb = types.SimpleNamespace(K=n, N=n, name=name)
        return tb, f"ld_{name}"

    def describe_join(self, slot):
        self.log.append(("join", slot))


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_batches_come_back_in_order_one_join_each(depth):
    e = StubEngine()
    p = DescribePipeline(e, depth)
    got = []
    for i in range(7):
        r = p.submit(f"b{i}", 5)
        assert (r is None) == (i < depth - 1)
        if r is not None:
            got.append(r[0].name)
    got += [tb.name for tb, _ld in p.drain()]
    assert got == [f"b{i}" for i in range(7)] and p.drain() == []
    submits = [x for x in e.log if x[0] == "submit"]
    assert [s[2] for s in submits] == [(i % depth, depth) for i in range(7)]          # slot i mod depth, depth announced to the library
    joins = [x[1] for x in e.log if x[0] == "join"]
    assert joins == [i % depth for i in range(7)]                                    # every batch joined once, oldest first
    # a batch is joined only after the depth - 1 batches behind it have been submitted (that is the overlap)
    join_pos = [k for k, x in enumerate(e.log) if x[0] == "join"]
    for i in range(7 - (depth - 1)):
        assert join_pos[i] > e.log.index(("submit", f"b{i + depth - 1}", ((i + depth - 1) % depth, depth)))


def test_an_empty_batch_is_handed_back_but_never_joined():
    e = StubEngine()
    p = DescribePipeline(e, 2)
    assert p.submit("a", 3) is None
    assert p.submit("empty", 0)[0].name == "a"
    out = p.submit("c", 4)
    assert out[0].name == "empty" and out[0].N == 0
    (last,) = p.drain()
    assert last[0].name == "c"
    assert [x[1] for x in e.log if x[0] == "join"] == [0, 0]       # "a" (slot 0) and "c" (slot 0); the empty batch of slot 1: no join


def test_depth_limits():
    e = StubEngine(max_slots=4)
    for bad in (0, 1, 5):
        with pytest.raises(ValueError):
            DescribePipeline(e, bad)
    DescribePipeline(e, 4)
