"""Writers whose numeric core runs on the MI355X: currently the SOG bundle."""
from .sog_writer import write_sog  # noqa: F401
