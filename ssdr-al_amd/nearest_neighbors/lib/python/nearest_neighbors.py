"""Drop-in for the reference's Cython module: ``import nearest_neighbors.lib.python.nearest_neighbors``
(/root/reference/SSDR_AL_s3dis/helper_tool.py:15) keeps working with ``ssdr-al_amd/`` on ``sys.path``
in place of the reference's ``utils/``."""
from ssdr_al.knn import knn, knn_batch, knn_batch_distance_pick  # noqa: F401
