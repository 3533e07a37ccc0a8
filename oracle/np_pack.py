"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy definition of the 2-bit packed spike tensor layout of include/ss_neuron.h
(ss_neuron_fwd_ex out_packed / skip_packed, ss_unpack_spikes, ss_im2col_cl_bf16_packed): the reference's spike tensors only take the
values 0..3 (/root/reference/network/SNN_models.py:171-192: 0/1 spikes, +1 per decoder skip add; blocks.py:171: SEW add, up to 3), so
16 neurons fit one 32-bit word: neuron n of a flattened [N] step -> word n // 16, bits 2*(n % 16) + {0, 1}.  The parity statement is
unpack(pack(x)) == x bit for bit, and every packed consumer == the same consumer on the dense tensor."""
import numpy as np


def pack(values):
    """values: [..., N] array of 0..3 (any numeric dtype), N % 16 == 0 -> uint32 [..., N // 16]."""
    v = np.asarray(values)
    assert v.shape[-1] % 16 == 0
    c = v.astype(np.uint32)
    assert np.array_equal(c.astype(v.dtype), v) and (c.max(initial=0) <= 3)
    c = c.reshape(v.shape[:-1] + (v.shape[-1] // 16, 16))
    return (c << (2 * np.arange(16, dtype=np.uint32))).sum(-1, dtype=np.uint32)


def unpack(words, dtype=np.float32):
    w = np.asarray(words, np.uint32)
    c = (w[..., None] >> (2 * np.arange(16, dtype=np.uint32))) & np.uint32(3)
    return c.reshape(w.shape[:-1] + (w.shape[-1] * 16,)).astype(dtype)
