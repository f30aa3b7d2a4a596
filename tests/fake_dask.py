"""A few dozen lines of `dask.array` -- TEST INFRASTRUCTURE for the dask slot of the public functions.

dask is not installable where this project is built and tested.  The slot (xrspatial_amd/utils.py: dask_overlap,
dask_blocks) only needs `Array.map_overlap(func, depth, boundary, meta)`, `Array.astype`, `map_blocks(func, *arrays,
meta)`, `stack`, for hotspots `nanmean` / `nanstd` / `compute`, for zonal.stats `numblocks` / `blocks[i, j]`; this module provides exactly those over numpy arrays cut into chunks, with dask's semantics:
a block is extended by `depth` cells of its neighbours (the `boundary` value beyond the array), the function runs on
the extended block, the overlap is trimmed from its result.  Evaluation is eager (`compute()` returns what is already
there); the chunk log lets a test check that the work really went block by block."""
import numpy as np


class Array:
    def __init__(self, values, chunks):
        self._v = np.asarray(values)
        self.chunks = tuple(tuple(int(c) for c in ch) for ch in chunks)
        assert tuple(sum(ch) for ch in self.chunks) == self._v.shape
        self.blocks_seen = []

    shape = property(lambda self: self._v.shape)
    dtype = property(lambda self: self._v.dtype)
    ndim = property(lambda self: self._v.ndim)

    def compute(self):
        return self._v

    @property
    def numblocks(self):
        return tuple(len(ch) for ch in self.chunks)

    @property
    def blocks(self):
        """`arr.blocks[i, j]`: that block as an array of one chunk (dask.array.Array.blocks)."""
        outer = self

        class _Blocks:
            def __getitem__(self, ij):
                (r0, r1), (c0, c1) = outer._spans()[0][ij[0]], outer._spans()[1][ij[1]]
                outer.blocks_seen.append((r1 - r0, c1 - c0))
                return Array(outer._v[r0:r1, c0:c1], ((r1 - r0,), (c1 - c0,)))
        return _Blocks()

    def __array__(self, dtype=None, copy=None):
        return self._v if dtype is None else self._v.astype(dtype)

    def astype(self, dtype):
        return Array(self._v.astype(dtype), self.chunks)

    def _spans(self):
        out = []
        for ch in self.chunks:
            edges = np.concatenate([[0], np.cumsum(ch)])
            out.append([(int(edges[i]), int(edges[i + 1])) for i in range(len(ch))])
        return out

    def map_overlap(self, func, depth, boundary=None, meta=None, **kwargs):
        depth = (depth,) * self.ndim if np.isscalar(depth) else tuple(depth)
        assert len(depth) == self.ndim == 2 and boundary is not None and boundary != boundary, "the slot asks for a NaN boundary"
        (d0, d1) = depth
        assert np.issubdtype(self._v.dtype, np.floating), "a NaN boundary on an integer array: dask would not give NaN there"
        padded = np.pad(self._v, ((d0, d0), (d1, d1)), constant_values=np.nan)
        rows, cols = self._spans()
        out = None
        for (r0, r1) in rows:
            for (c0, c1) in cols:
                block = padded[r0:r1 + 2 * d0, c0:c1 + 2 * d1]
                res = np.asarray(func(block, **kwargs))
                assert res.shape == block.shape, "a block function must keep the block's shape"
                self.blocks_seen.append(block.shape)
                if out is None:
                    out = np.empty(self._v.shape, res.dtype)
                out[r0:r1, c0:c1] = res[d0:res.shape[0] - d0, d1:res.shape[1] - d1]
        result = Array(out, self.chunks)
        result.blocks_seen = self.blocks_seen
        return result


def from_array(values, chunks):
    values = np.asarray(values)
    spans = []
    for n, c in zip(values.shape, chunks if isinstance(chunks, (tuple, list)) else (chunks,) * values.ndim):
        spans.append(tuple([c] * (n // c) + ([n % c] if n % c else [])))
    return Array(values, spans)


def map_blocks(func, *arrays, meta=None, **kwargs):
    first = arrays[0]
    assert all(a.chunks == first.chunks for a in arrays), "equally chunked inputs"
    rows, cols = first._spans()
    out = None
    for (r0, r1) in rows:
        for (c0, c1) in cols:
            res = np.asarray(func(*[a._v[r0:r1, c0:c1] for a in arrays], **kwargs))
            first.blocks_seen.append(res.shape)
            if out is None:
                out = np.empty(first.shape, res.dtype)
            out[r0:r1, c0:c1] = res
    result = Array(out, first.chunks)
    result.blocks_seen = first.blocks_seen
    return result


def nanmean(a):
    return np.nanmean(a._v)


def nanstd(a):
    return np.nanstd(a._v)


def compute(*values):
    return tuple(values)


def stack(arrays):
    a0 = arrays[0]
    out = Array(np.stack([a._v for a in arrays]), ((1,) * len(arrays),) + a0.chunks)
    for a in arrays:
        out.blocks_seen += a.blocks_seen
    return out
