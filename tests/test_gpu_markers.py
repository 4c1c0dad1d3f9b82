"""GPU: stream markers of a context (wsc_ctx_mark / wsc_ctx_wait_mark / wsc_ctx_wait_for_mark) and the StackChain built on
them -- the primitives the drivers use to order work across the contexts of batches in flight (include/wsscam.h)."""
import numpy as np
import pytest

from wsscam import _lib

pytestmark = pytest.mark.gpu


def _fill(ctx, buf, value, nbytes):
    _lib.check(ctx._lib.wsc_memset(ctx.h, buf.ptr, value, nbytes))


def test_marker_orders_another_contexts_work():
    """B waits (on the device) for A's marker: what A enqueued before the record is visible to B's later work; the host wait
    returns after the same point; a second record moves the marker."""
    a, b = _lib.Context(0), _lib.Context(0)
    n = 64 << 20
    x = a.alloc(n)
    _fill(a, x, 0, n)
    a.sync()
    for value, slot in ((1, 3), (2, 3), (5, 0)):
        _fill(a, x, value, n)  # asynchronous on A's stream
        a.mark(slot)
        b.wait_for_mark(a, slot)
        got = b.to_host(x, (n,), np.uint8)  # B's stream: behind the marker
        assert got.min() == value and got.max() == value, (value, slot)
        a.wait_mark(slot)  # host side: returns with the fill done
    # never-recorded slots and a context's own marker are no-ops, not errors
    a.wait_mark(7)
    b.wait_for_mark(a, 6)
    a.wait_for_mark(a, 3)
    x.free()


def test_marker_argument_errors():
    a, b = _lib.Context(0), _lib.Context(0)
    for call in (lambda: a.mark(8), lambda: a.mark(-1), lambda: a.wait_mark(8), lambda: b.wait_for_mark(a, 8)):
        with pytest.raises(_lib.WscError) as ei:
            call()
        assert ei.value.status == _lib.WSC_ERR_INVALID


def test_stack_chain_orders_phases_across_contexts():
    """StackChain.run: a phase enqueued on one context starts after the phase the chain ran last on another one."""
    ctxs = [_lib.Context(0) for _ in range(3)]
    chain = _lib.StackChain()
    n = 32 << 20
    x = ctxs[0].alloc(n)
    _fill(ctxs[0], x, 0, n)
    ctxs[0].sync()
    for step in range(1, 7):  # the fills land in chain order although they sit on three streams
        c = ctxs[step % 3]
        assert chain.run(c, lambda c=c, step=step: (_fill(c, x, step, n), step)[1]) == step
    last = ctxs[6 % 3]
    got = last.to_host(x, (n,), np.uint8)
    assert got.min() == 6 and got.max() == 6
    x.free()
