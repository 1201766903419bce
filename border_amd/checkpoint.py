"""Named-tensor checkpoint files (host side; no GPU involved).

Mirror of what tch's `VarStore::save` / `VarStore::load` do for the reference's agents
(border-tch-agent/src/dqn/model/base.rs:134-148): the container follows the file name -
`*.safetensors` or, for anything else (the reference's default `*.pt.tch`), the libtorch named-tensor archive.
The bytes are written and parsed by the C++ library (`csrc/tch_archive.hpp`, `csrc/agent_api.hip`).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Sequence, Tuple

import numpy as np

from . import _lib

FORMATS = {"tch": 0, "safetensors": 1}
EXTENSIONS = {"tch": ".pt.tch", "safetensors": ".safetensors"}


def _meta(spec: Sequence[Tuple[str, Tuple[int, ...]]]):
    arr = (_lib.NamedTensorC * len(spec))()
    keep = []
    for i, (name, dims) in enumerate(spec):
        d = (C.c_uint64 * max(len(dims), 1))(*dims)
        b = name.encode()
        keep += [d, b]
        arr[i].name = b
        arr[i].dims = C.cast(d, C.POINTER(C.c_uint64))
        arr[i].ndim = len(dims)
    return arr, keep


def write(path: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write `tensors` (name -> f32 array, in dict order) to `path`."""
    spec = [(k, tuple(int(x) for x in np.shape(v))) for k, v in tensors.items()]
    flat = (np.concatenate([np.ascontiguousarray(v, np.float32).ravel() for v in tensors.values()])
            if tensors else np.zeros(0, np.float32))
    arr, keep = _meta(spec)
    _lib.check(_lib.lib().bdr_checkpoint_write(path.encode(), arr, len(spec), flat.ctypes.data_as(C.c_void_p), C.c_uint64(flat.size)))


def read(path: str, spec: Sequence[Tuple[str, Tuple[int, ...]]]) -> Dict[str, np.ndarray]:
    """Read the variables named in `spec` ((name, shape) pairs) from `path`; shapes must match, extras are ignored."""
    n = int(sum(int(np.prod(d, dtype=np.int64)) for _, d in spec))
    flat = np.zeros(n, np.float32)
    arr, keep = _meta(spec)
    _lib.check(_lib.lib().bdr_checkpoint_read(path.encode(), arr, len(spec), flat.ctypes.data_as(C.c_void_p), C.c_uint64(n)))
    out, o = {}, 0
    for name, dims in spec:
        k = int(np.prod(dims, dtype=np.int64))
        out[name] = flat[o:o + k].reshape(dims).copy()
        o += k
    return out
