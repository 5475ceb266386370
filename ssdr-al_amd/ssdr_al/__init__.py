"""ssdr_al — host-side mirror of the SSDR-AL hot-path interfaces over libssdr_al.so (gfx950 HIP)."""
from . import _lib  # noqa: F401
