"""CPU: the StackChain's bookkeeping (wsscam._lib.StackChain) with recording stand-ins for contexts -- who waits for whom,
the marker after the phase, and that a failing phase still leaves the chain usable."""
import threading

import pytest

from wsscam import _lib


class _Ctx:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def wait_for_mark(self, other, slot):
        self.log.append(("wait", self.name, other.name, slot))

    def mark(self, slot):
        self.log.append(("mark", self.name, slot))


def test_stack_chain_waits_for_the_previous_context_only():
    log = []
    a, b = _Ctx("a", log), _Ctx("b", log)
    chain = _lib.StackChain()
    assert chain.run(a, lambda: log.append(("work", "a1")) or 1) == 1  # first phase: nobody to wait for
    chain.run(a, lambda: log.append(("work", "a2")))                   # same context again: its stream orders it already
    chain.run(b, lambda: log.append(("work", "b1")))
    chain.run(a, lambda: log.append(("work", "a3")))
    S = _lib.StackChain.SLOT
    assert log == [("work", "a1"), ("mark", "a", S), ("work", "a2"), ("mark", "a", S),
                   ("wait", "b", "a", S), ("work", "b1"), ("mark", "b", S),
                   ("wait", "a", "b", S), ("work", "a3"), ("mark", "a", S)]


def test_stack_chain_survives_a_failing_phase_and_serialises_threads():
    log = []
    a, b = _Ctx("a", log), _Ctx("b", log)
    chain = _lib.StackChain()

    def boom():
        raise RuntimeError("phase failed")

    with pytest.raises(RuntimeError):
        chain.run(a, boom)
    chain.run(b, lambda: None)  # the lock was released, the marker of the failed phase was still recorded
    assert ("mark", "a", _lib.StackChain.SLOT) in log and ("wait", "b", "a", _lib.StackChain.SLOT) in log
    # many threads: every phase is bracketed wait / work / mark without interleaving
    log.clear()
    ctxs = [_Ctx("c%d" % i, log) for i in range(4)]

    def lane(c):
        for k in range(50):
            chain.run(c, lambda: log.append(("work", c.name)))

    ts = [threading.Thread(target=lane, args=(c,)) for c in ctxs]
    [t.start() for t in ts]
    [t.join() for t in ts]
    works = [i for i, ev in enumerate(log) if ev[0] == "work"]
    assert len(works) == 200
    for i in works:
        assert log[i + 1] == ("mark", log[i][1], _lib.StackChain.SLOT)
