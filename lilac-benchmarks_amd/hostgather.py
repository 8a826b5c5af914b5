"""Host gather for the row-partitioned multi-GPU path: every rank copies its
slice of y straight to its row offset of ONE host vector -- no collective.  With one process per
GPU that host vector lives in POSIX shared memory; inside one process (SPMV_HIP_DEVICES=N in the
library) it is simply the caller's `ov`."""
import os

import numpy as np


class SharedVector:
    """A float64/float32 vector in /dev/shm visible to every rank of the node."""

    def __init__(self, name, n, dtype=np.float64, create=False):
        self.path = os.path.join("/dev/shm", name)
        self.n = n
        self.dtype = np.dtype(dtype)
        if create:
            with open(self.path, "wb") as f:
                f.truncate(n * self.dtype.itemsize)
        self.arr = np.memmap(self.path, dtype=self.dtype, mode="r+", shape=(n,))

    def write_slice(self, row_begin, values):
        self.arr[row_begin:row_begin + values.size] = values
        self.arr.flush()

    def close(self, unlink=False):
        del self.arr
        if unlink and os.path.exists(self.path):
            os.unlink(self.path)
