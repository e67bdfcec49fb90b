"""src/data/confusion_matrix.py:7-63 -- host-side table the grid-update kernel consumes."""
import numpy as np


class ConfusionMatrix(object):
    """C[i, j] = observations of true class i predicted as j (sklearn convention)."""

    def __init__(self, load_path=None, matrix=None):
        self._cfn_mtx = np.load(load_path) if matrix is None else np.asarray(matrix, dtype=np.float64)
        height, width = self._cfn_mtx.shape
        assert height == width
        self.num_class = height

    def get_submatrix(self, indices, to_probability=False, use_log=False):
        """confusion_matrix.py:25-48: sub-select rows/cols, row-normalise, optionally log."""
        num_indices = len(indices)
        if num_indices == 0:
            return []
        if num_indices > self.num_class:
            raise ValueError("The number of indices is greater than the number of classes in the confusion matrix!")
        for i in indices:
            if i < 0 or i >= self.num_class:
                raise ValueError("Invalid index!", i)
        sub_mtx = self._cfn_mtx[np.ix_(indices, indices)]
        if to_probability:
            sub_mtx = sub_mtx / np.sum(sub_mtx, axis=1)[:, np.newaxis]
            if use_log:
                sub_mtx = np.log(sub_mtx)
        return sub_mtx

    def __len__(self):
        return self.num_class

    def __getitem__(self, item):
        return self._cfn_mtx[item]

    def __str__(self):
        return str(self._cfn_mtx)
