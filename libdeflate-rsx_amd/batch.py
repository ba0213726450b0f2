"""Mirror of src/batch.rs: BatchCompressor / BatchDecompressor over the GPU engine."""
from .engine import FORMAT_DEFLATE, default_engine


class BatchCompressor:
    """src/batch.rs:5-59.  compress_batch(inputs) -> list of bytes, raw DEFLATE;
    an item that does not fit its bound yields b"" (batch.rs:52-54)."""

    def __init__(self, level, engine=None, fmt=FORMAT_DEFLATE):
        self.level = int(level)
        self.fmt = fmt
        self._engine = engine

    def compress_batch(self, inputs):
        if len(inputs) == 0:
            return []
        eng = self._engine or default_engine()
        # batch.rs:12 takes the level unchecked; levels above 12 behave as 12
        # (src/compress/mod.rs:597-600)
        res, _ = eng.batch_compress(inputs, self.fmt, min(self.level, 12))
        return [r if r is not None else b"" for r in res]


class BatchDecompressor:
    """src/batch.rs:61-101.  decompress_batch(inputs, max_out_sizes) -> list of
    bytes or None (any non-Success result, batch.rs:95-97)."""

    def __init__(self, engine=None, fmt=FORMAT_DEFLATE):
        self.fmt = fmt
        self._engine = engine

    def decompress_batch(self, inputs, max_out_sizes):
        if len(inputs) == 0:
            return []
        if len(inputs) != len(max_out_sizes):
            # the reference zips the two slices (batch.rs:79-81): extra entries are ignored
            k = min(len(inputs), len(max_out_sizes))
            inputs, max_out_sizes = inputs[:k], max_out_sizes[:k]
        eng = self._engine or default_engine()
        res, _ = eng.batch_decompress(inputs, max_out_sizes, self.fmt)
        return res
