"""Host-side layout of the trainers' flat parameter / gradient buffers (no GPU needed)."""
import numpy as np
import torch

import estts  # noqa: F401
from estts.forward_training import ParamStore


def test_every_view_starts_on_a_256_byte_boundary_and_round_trips():
    """The weight-gradient kernels add 64-byte row segments to the gradient views with float atomics; a view that starts off a
    line boundary makes every segment straddle two lines (measured: +30 % on those launches)."""
    rng = np.random.default_rng(0)
    shapes = {'a.kernel': (3, 17, 5), 'a.bias': (5,), 'scalar': (), 'b.kernel': (129, 64), 'b.bias': (64,), 'c': (1,)}
    store = ParamStore(torch.device('cpu'))
    arrays = {}
    for name, shape in shapes.items():
        arrays[name] = rng.normal(size=shape).astype(np.float32)
        store.add(name, arrays[name])
    store.add('bn.moving_mean', np.zeros(7, np.float32), trainable=False)
    store.add('bn.moving_variance', np.ones(7, np.float32), trainable=False)
    store.build()
    base_p, base_g = store.flat.data_ptr(), store.grad.data_ptr()
    for name, shape in shapes.items():
        off = store.table[name][0]
        assert off % 64 == 0, name
        assert (store.p[name].data_ptr() - base_p) == off * 4 and (store.g[name].data_ptr() - base_g) == off * 4
        np.testing.assert_array_equal(store.p[name].numpy().reshape(shape), arrays[name])
        assert tuple(store.p[name].shape) == (shape if shape else (1,))
    offs = sorted(o for o, _ in store.table.values())
    sizes = {o: int(np.prod(s)) if s else 1 for o, s in store.table.values()}
    for a, b in zip(offs, offs[1:]):
        assert a + sizes[a] <= b                                    # no overlap
    # padding between the views stays zero: Adam over the flat buffer leaves it zero, and a norm over the flat gradient ignores it
    mask = torch.ones_like(store.flat, dtype=torch.bool)
    for o, s in store.table.values():
        mask[o:o + (int(np.prod(s)) if s else 1)] = False
    assert float(store.flat[mask].abs().sum()) == 0.0
    for o, _ in store.state_table.values():
        assert o % 64 == 0
