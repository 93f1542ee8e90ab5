"""Writers whose numeric core runs on the MI355X: the SOG bundle and the compressed PLY."""
from .sog_writer import write_sog  # noqa: F401
from .compressed_ply_writer import write_compressed_ply  # noqa: F401
