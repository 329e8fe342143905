from dpft_amd.hip.lib import lib, LIB_PATH, HipLibraryError  # noqa: F401
