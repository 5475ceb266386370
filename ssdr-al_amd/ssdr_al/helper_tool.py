"""Mirror of the two native-op façade functions and the config constants of the reference's ``helper_tool.py``
(/root/reference/SSDR_AL_s3dis/helper_tool.py:46-117, 173-183, 215-235)."""
import numpy as np

import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling
import nearest_neighbors.lib.python.nearest_neighbors as nearest_neighbors


class ConfigS3DIS:                      # helper_tool.py:46-75 (the fields the hot path reads)
    k_n = 16
    num_layers = 5
    num_points = 40960
    num_classes = 13
    sub_grid_size = 0.04
    batch_size = 6
    val_batch_size = 20
    sub_sampling_ratio = [4, 4, 4, 4, 2]
    d_out = [16, 64, 128, 256, 512]
    noise_init = 3.5


class ConfigSemantic3D:                 # helper_tool.py:77-117
    k_n = 16
    num_layers = 5
    num_points = 65536
    num_classes = 8
    sub_grid_size = 0.06
    batch_size = 4
    val_batch_size = 16
    sub_sampling_ratio = [4, 4, 4, 4, 2]
    d_out = [16, 64, 128, 256, 512]
    noise_init = 3.5


class DataProcessing:
    @staticmethod
    def knn_search(support_pts, query_pts, k):
        """helper_tool.py:173-183 — support B*N1*3, query B*N2*3 -> int32 B*N2*k."""
        neighbor_idx = nearest_neighbors.knn_batch(support_pts, query_pts, k, omp=True)
        return neighbor_idx.astype(np.int32)

    @staticmethod
    def grid_sub_sampling(points, features=None, labels=None, grid_size=0.1, verbose=0):
        """helper_tool.py:215-235."""
        if (features is None) and (labels is None):
            return cpp_subsampling.compute(points, sampleDl=grid_size, verbose=verbose)
        elif labels is None:
            return cpp_subsampling.compute(points, features=features, sampleDl=grid_size, verbose=verbose)
        elif features is None:
            return cpp_subsampling.compute(points, classes=labels, sampleDl=grid_size, verbose=verbose)
        else:
            return cpp_subsampling.compute(points, features=features, classes=labels, sampleDl=grid_size,
                                           verbose=verbose)

    @staticmethod
    def shuffle_idx(x):                 # helper_tool.py:201-206
        idx = np.arange(len(x))
        np.random.shuffle(idx)
        return x[idx]
