"""MI355X-native front-end hot path of ORB_SLAM2_E (ORB extract, Hamming match, FEM).

The compute path is hand-written HIP for gfx950 behind the C-ABI declared in
include/*.h (liborbslam_hip.so); this package is the thin host-side mirror of the
reference's ORBextractor / ORBmatcher / FEA2 interfaces used by tests and bench.
"""
from ._lib import OrbxError, build, lib  # noqa: F401
from .extractor import KP_DTYPE, ComputeStereoMatches, ORBextractor, extract_pair, stereo_download_batch, stereo_match_batch  # noqa: F401
from .matcher import Frame, ORBmatcher, Points, View  # noqa: F401
