"""Drop-in for the reference's CPython extension: ``import cpp_wrappers.cpp_subsampling.grid_subsampling``
(/root/reference/SSDR_AL_s3dis/helper_tool.py:14) keeps working with ``ssdr-al_amd/`` on ``sys.path``."""
from ssdr_al.subsampling import compute  # noqa: F401
