"""Host-side table the grid-update kernel consumes: the per-class log-likelihood columns.

Behaviour of the reference's src/data/confusion_matrix.py (sklearn convention: C[i, j] = observations of true
class i predicted as j; ``get_submatrix`` keeps the requested classes, optionally row-normalises to probabilities
and takes the log, :25-48 / :59-63), written independently.
"""
import numpy as np


class ConfusionMatrix(object):
    def __init__(self, load_path=None, matrix=None):
        table = np.load(load_path) if matrix is None else np.asarray(matrix, dtype=np.float64)
        if table.ndim != 2 or table.shape[0] != table.shape[1]:
            raise AssertionError("a confusion matrix is square, got %s" % (table.shape,))
        self._cfn_mtx = table
        self.num_class = table.shape[0]

    def get_submatrix(self, indices, to_probability=False, use_log=False):
        """Rows and columns `indices` of the matrix; with to_probability each row is divided by its sum, with use_log
        (only together with to_probability) the natural log of that."""
        indices = list(indices)
        if not indices:
            return []
        if len(indices) > self.num_class:
            raise ValueError("The number of indices is greater than the number of classes in the confusion matrix!")
        bad = [i for i in indices if not 0 <= i < self.num_class]
        if bad:
            raise ValueError("Invalid index!", bad[0])
        sub = self._cfn_mtx[np.ix_(indices, indices)]
        if not to_probability:
            return sub
        prob = sub / sub.sum(axis=1, keepdims=True)
        return np.log(prob) if use_log else prob

    def __len__(self):
        return self.num_class

    def __getitem__(self, item):
        return self._cfn_mtx[item]

    def __str__(self):
        return str(self._cfn_mtx)
